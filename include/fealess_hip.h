/*
 * fealess_hip.h -- C ABI of libfealess_hip.so: the MI355X (gfx950) implementation of the
 * rlvc/FEALESS hot path (LINEMOD detection + depth back-projection + ICP refinement).
 *
 * The reference has no FFI of its own: its boundary is the C++ surface of
 * CadReco/obj_reco_temp.h, CadReco/obj_reco_lmicp.h, linemod/linemod.hpp, ICP/ICP.h,
 * ICP/detection.h and ICP/depth_to_3d.h.  Every entry point below names the reference
 * function it replaces (paths relative to the reference root).  The C++ adapter in
 * fealess_amd/cadreco/ re-exposes the reference's own class surface on top of this ABI
 * (see INTEGRATION.md).
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the ABI; every function
 * returns FL_OK (0) or a negative fl_status; fl_last_error() gives the text.  A context is
 * bound to one HIP device; calls on one context must be serialised by the caller (the
 * reference's CObjRecoLmICP is not thread-safe either, obj_reco_lmicp.h:37-51).  Work is queued
 * on the context's stream; functions that return results through host pointers synchronise
 * that stream before returning, functions that write device pointers do not.  The stream is
 * non-blocking (not ordered against the null stream): device buffers handed in (FL_MEM_DEVICE
 * frames, clouds, quantised images) must be complete when the call is made, or have been
 * produced on the stream given to fl_context_set_stream().
 * There is NO CPU fallback: without a HIP device fl_context_create fails.
 */
#ifndef FEALESS_HIP_H
#define FEALESS_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FL_ABI_VERSION 1

typedef enum {
  FL_OK = 0,
  FL_ERR_INVALID = -1,   /* bad argument (the reference returns ERROR_INVALID_PARAM / -1)       */
  FL_ERR_HIP = -2,       /* a HIP runtime call failed                                           */
  FL_ERR_ASSERT = -3,    /* the reference would hit a CV_Assert / cv::Exception here            */
  FL_ERR_OVERFLOW = -4,  /* a fixed-capacity device buffer overflowed (see fl_detector_limits)  */
  FL_ERR_NO_DEVICE = -5,
  FL_ERR_STATE = -6,     /* call order violated (e.g. matching before fl_detector_finalize)     */
  FL_ERR_NO_TEMPLATE = -7 /* fl_extract_template_pyramid: too few candidate features (addTemplate's -1) */
} fl_status;

typedef enum { FL_MEM_HOST = 0, FL_MEM_DEVICE = 1 } fl_mem;

/* ICP accumulation modes.
 *   FL_ICP_PARITY : the 15 centroid/covariance sums and the distance sum are accumulated as
 *                   sequential float32 chains in the reference's order (one lane per scalar),
 *                   so results are bit-identical to the reference's arithmetic (ICP.cpp:8-25,
 *                   731-735, 68-111).  Default.
 *   FL_ICP_FAST   : the same sums as parallel reductions (float32 per-thread partials of ~60
 *                   terms, fp64 tree across the workgroup, rounded once to float32): within 1e-4
 *                   of the EXACT sums' pose, not bit-identical to the reference's float32 chains.
 *                   Against the reference's float32 result the final pose differs by up to 1.1e-4
 *                   (R) / 1.1e-4 (T, relative to the object's distance) on 15 k-point clouds --
 *                   that is the reference's own summation noise (its float32 result is that far
 *                   from the exact sums); tests/test_gpu_icp.py asserts <= 2.5e-4 and "no farther
 *                   than the float32 result is from the exact sums".  Use FL_ICP_PARITY where the
 *                   1e-4 bar against the reference matters.  Timings of both: bench.py `icp_fast`.
 *   FL_ICP_POINT_TO_PLANE : opt-in extension with NO counterpart in the reference (SURVEY.md
 *                   section 8f rank 4): every iteration (the first included) pairs each model point
 *                   with its exact nearest reference point, gates the pair at distance
 *                   <= 3*dist_mean, and minimises sum (n_j . (R m_i + t - r_j))^2 linearised in
 *                   (omega, t): a 6x6 normal-equation system accumulated in fp64, solved by
 *                   Cholesky, omega -> R by Rodrigues.  Reference-cloud normals n_j come from the
 *                   organised scene depth (fl_detection / fl_recognize*) or from the caller
 *                   (fl_icp_point_to_plane).  Loop control, dist_mean / px_ratio and the pose
 *                   composition are those of icpCloudToCloud_Ex.  Validated against ground-truth
 *                   poses of synthetic scenes, not against the reference.                     */
typedef enum { FL_ICP_PARITY = 0, FL_ICP_FAST = 1, FL_ICP_POINT_TO_PLANE = 2 } fl_icp_mode;

typedef struct fl_context fl_context;
typedef struct fl_detector fl_detector;

/* linemod/linemod.hpp:32-43 (Feature), :47-58 (Template), :253-281 (Match) */
typedef struct { int32_t x, y, label; } fl_feature;
typedef struct {
  int32_t width, height, offset_x, offset_y, pyramid_level;
  int32_t feat_begin, feat_count;      /* range in the flat feature array handed over with it */
} fl_template;
typedef struct {
  int32_t x, y;
  float   similarity;
  int32_t class_idx;                   /* index of class_id in std::map (sorted) order */
  int32_t template_id;                 /* class-local id, as in the reference */
} fl_match;

/* CadReco/lotus_common.h:41-50 (TCamIntrinsicParam, without the unused distortion vector) */
typedef struct { int32_t width, height; double fx, fy, cx, cy; } fl_intrinsics;

/* result of icpCloudToCloud_Ex (ICP/ICP.cpp:617-809) */
typedef struct {
  float   R[9], T[3];
  float   dist_mean;                   /* the function's return value (-1 if < 3 points) */
  float   px_ratio;
  int32_t iters;                       /* value of `iter` on exit */
  int32_t n_corr_last;
} fl_icp_result;

/* result of detection() (ICP/detection.cpp:11-254) */
typedef struct {
  float   R_final[9], T_final[3];
  fl_icp_result icp;
  int32_t n_points;
  int32_t status;                      /* FL_OK or FL_ERR_ASSERT (rect outside the image, Q10) */
} fl_detection_result;

/* result of CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:86-204) for one frame */
typedef struct {
  int32_t status;                      /* FL_OK, or the error Recognition would return */
  int32_t found;                       /* 0: vtResult empty; 1: one TObjRecoResult */
  int32_t n_matches;                   /* matches.size() after sort/unique */
  fl_match best;                       /* matches[0] */
  float   pose[16];                    /* TObjRecoResult::tWorld2Cam, row-major 4x4 */
  fl_detection_result det;
} fl_recognition_result;

typedef struct {
  float   matching_threshold;          /* m_matching_threshold, default 75 (obj_reco_lmicp.cpp:52) */
  int32_t icp_it_thr;                  /* default 10 (:53) */
  float   dist_mean_thr;               /* default 0.5 (:54) */
  float   dist_diff_thr;               /* default 0.01 (:55) */
  int32_t icp_mode;                    /* fl_icp_mode */
} fl_recognition_params;

/* ---- context ---------------------------------------------------------------------------- */
int  fl_abi_version(void);
int  fl_context_create(int device, fl_context **out);
void fl_context_destroy(fl_context *ctx);   /* with detectors still alive: deferred until the last fl_detector_destroy */
const char *fl_last_error(const fl_context *ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL restores the context's own */
int  fl_context_set_stream(fl_context *ctx, void *hip_stream);
int  fl_context_synchronize(fl_context *ctx);
/* the stream the context queues its work on (a hipStream_t), its device, and a detector's context: what a host layer
 * that adds its own stream-ordered work (libfealess_mg.so: the RCCL collectives of the template-sharded path) needs */
void *fl_context_get_stream(fl_context *ctx);
int  fl_context_get_device(const fl_context *ctx);
fl_context *fl_detector_get_context(fl_detector *det);
/* Development / comparison switches.  They change how the work is scheduled, never a result: "scan_prune" (1; 0 = the scan adds
 * every feature everywhere, as the reference does), "scan_prune_mid" (bit mask of the 8-feature groups after which a modality
 * checks the pruning bound; -1 = built-in), "icp_wide" (-1 = by batch size; 0 / 1 force the 256- / 1024-thread ICP workgroup),
 * "icp_occ" (0 = by batch size; 4 / 5 force the 256-thread kernel built for that many workgroups per CU), "icp_order" (1 = ICP
 * jobs dealt longest first; 0 = frame order), "icp_wg_per_cu" (0 = as many 256-thread ICP workgroups per CU as fit; 1 .. 3 = at
 * most that many, the rest of the CU left to the kernels of other streams), and -- sampled by fl_detector_finalize -- "eager_frontend" (1 = finer pyramid
 * levels in full before the scan, the reference's order), "dev_poison" (1 = what the lazy path leaves uncomputed is filled
 * with 0xFF), "ws_pad" (extra bytes of frame-workspace stride).  Their INITIAL values are read once from the environment
 * when the context is created (FL_SCAN_PRUNE, FL_SCAN_PRUNE_MID (hex), FL_ICP_WIDE, FL_ICP_OCC, FL_ICP_ORDER, FL_ICP_WG_PER_CU,
 * FL_EAGER_FRONTEND, FL_DEV_POISON, FL_DEV_WS_PAD); nothing reads the environment after that, so a variable set in a host
 * process later on changes nothing.  Unknown names: FL_ERR_INVALID. */
int  fl_context_set_option(fl_context *ctx, const char *name, long value);
int  fl_context_get_option(const fl_context *ctx, const char *name, long *value);

/* ---- detector = cup_linemod::Detector state resident in HBM -------------------------------- */
/* Detector::Detector(modalities, T_pyramid) (linemod.cpp:1348-1354). modalities is 1 or 2:
 * index 0 = ColorGradient, 1 = DepthNormal (getDefaultLINEMOD, linemod.cpp:1829-1835). */
int  fl_detector_create(fl_context *ctx, int modalities, int levels, const int *T_at_level,
                        fl_detector **out);
void fl_detector_destroy(fl_detector *det);
/* Detector::addSyntheticTemplate x n (linemod.cpp:1636-1642) + addPoseInfo (:1617-1622):
 * append a class of n_pyramids template pyramids, each levels*modalities fl_template ordered
 * [l*modalities + m].  poses13 (n_pyramids*13 floats, may be NULL) is the row-major 3x4 pose +
 * distance side table.  Classes are kept in std::map order of class_id. */
int  fl_detector_add_class(fl_detector *det, const char *class_id, int n_pyramids,
                           const fl_template *templates, const fl_feature *features,
                           int n_features, const float *poses13);
/* The per-template depth renders CObjRecoLmICP reads from <dir>/depth/<template_id>.png on
 * every frame (obj_reco_lmicp.cpp:156-157): uploaded once, w*h u16 in 0.1 mm each, for the
 * pyramids [first, first+count) of class class_idx.  src may be host or device memory. */
int  fl_detector_set_model_depths(fl_detector *det, int class_idx, int first, int count,
                                  const uint16_t *depth_01mm, int w, int h, int mem);
/* Freeze the bank for frames of w0 x h0 and at most max_batch frames per call; uploads the
 * flattened feature tables and allocates every per-frame workspace in HBM.  max_candidates is
 * the per-frame capacity of the candidate / match buffers (0 = default 65536). */
int  fl_detector_finalize(fl_detector *det, int w0, int h0, int max_batch, int max_candidates);
/* max_candidates: per-frame capacity of the coarse-candidate / match buffers.  > 0 (or 0 = 65536): an INITIAL size -- a frame
 * that needs more makes the synchronous entry points (fl_match_quantized / _frame / _frame_masked, fl_recognize_batch /
 * _batch_zoom / _topk / _batch_topk) grow the buffers and run again, because the reference's vectors are unbounded
 * (linemod.cpp:1490-1504, 1575); the queued ones (fl_recognize_submit / fl_match_batch_submit) cannot replay a batch and
 * report FL_ERR_OVERFLOW in that frame's status (the other frames keep their results).  < 0: a hard cap of
 * -max_candidates, FL_ERR_OVERFLOW beyond it. */
/* For the QUEUED entry points (fl_recognize_submit, fl_match_batch_submit), which cannot replay a batch themselves: after
 * a batch in which a frame reported FL_ERR_OVERFLOW (fl_recognition_result.status, or FL_TOPK_OVERFLOW in the records of
 * fl_export_topk_batch), waits for the stream and grows the candidate buffers to what the fullest of the last batch's
 * first n_frames frames needs -- the std::vector growth of linemod.cpp:1490-1504 -- so that the caller can submit the
 * batch again.  *new_cap (may be NULL) = the capacity afterwards.  FL_ERR_OVERFLOW under a hard cap / without memory. */
int  fl_detector_grow_candidates(fl_detector *det, int n_frames, int *new_cap);
/* Detector::match's `class_ids` argument (linemod.hpp:319-327, linemod.cpp:1418-1434): n = 0 matches every
 * class (the default, and what Recognition passes); otherwise only the listed classes that exist.
 * Sticky until changed; may be called before or after fl_detector_finalize. */
int  fl_detector_set_class_filter(fl_detector *det, const char *const *class_ids, int n);
/* class_ids is treated as a SET.  The reference calls matchClass once per listed id, duplicates included
 * (linemod.cpp:1427-1433), and then relies on std::sort + adjacent std::unique (:1437-1439) to drop the repeated matches;
 * whether every repeat ends up adjacent is up to the (unstable) sort, so the duplicate-free list is one of its valid
 * outcomes -- the one returned here.  Unknown ids match nothing (:1429-1431). */
int  fl_detector_num_templates(const fl_detector *det);     /* Detector::numTemplates() :1652 */
int  fl_detector_num_classes(const fl_detector *det);

/* ---- stage entry points (each is the drop-in for one reference function) ------------------- */
/* quantizedOrientations + hysteresisGradient (linemod.cpp:230-385): bgr w*h*3 u8 -> w*h u8 */
int  fl_quantized_orientations(fl_context *ctx, const uint8_t *bgr, int w, int h,
                               float weak_threshold, uint8_t *dst, int mem);
/* quantizedNormals incl. medianBlur (linemod.cpp:595-685): depth w*h u16 (mm) -> w*h u8 */
int  fl_quantized_normals(fl_context *ctx, const uint16_t *depth, int w, int h,
                          int distance_threshold, int difference_threshold, uint8_t *dst, int mem);
/* cv::pyrDown on the colour image (linemod.cpp:443): w*h*3 -> (w/2)*(h/2)*3 */
int  fl_pyrdown_bgr(fl_context *ctx, const uint8_t *src, int w, int h, uint8_t *dst, int mem);
/* Detector::addTemplate (linemod.cpp:1579-1615) for the two default modalities ColorGradient(10, 63, 55) and
 * DepthNormal(2000, 50, 63, 2): quantized pyramids, extractTemplate per level and modality (:461-513, :747-825),
 * selectScatteredFeatures (:135-164), cropTemplates (:52-96).  mask: object mask (w0*h0 u8) or NULL.
 * Out (host): templates[levels * 2] ordered [l * 2 + m] with feat_begin = 63 * (l * 2 + m), features[levels * 2 * 63]
 * (template-relative after the crop), bb = {x, y, width, height} (may be NULL).  Returns FL_ERR_NO_TEMPLATE where
 * the reference returns -1.  The caller appends the pyramid to a class with fl_detector_add_class. */
int  fl_extract_template_pyramid(fl_context *ctx, const uint8_t *bgr, const uint16_t *depth, const uint8_t *mask, int w0,
                                 int h0, int levels, int mem, fl_template *templates, fl_feature *features, int bb[4]);
/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) as PrepareInputData applies it to frames
 * that are not 640 wide (obj_reco_lmicp.cpp:39-45, 229-249: TImage2Mat(..., true)); BGR8 and u16 */
int  fl_resize_linear_bgr8(fl_context *ctx, const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, int mem);
int  fl_resize_linear_u16(fl_context *ctx, const uint16_t *src, int sw, int sh, uint16_t *dst, int dw, int dh, int mem);
/* spread + computeResponseMaps + linearize x8 (linemod.cpp:950-1088) for one quantized image:
 * out = 8 * fl_lm_label_stride(w,h,T) bytes, layout [label][T*T grid][(w/T)*(h/T)] + zero pad */
size_t fl_lm_label_stride(int w, int h, int T);
int  fl_build_linear_memories(fl_context *ctx, const uint8_t *quantized, int w, int h, int T,
                              uint8_t *out, int mem);
/* cup_d2pc::depthTo3d, CV_16UC1 depth (ICP/depth_to_3d.cpp:190-221): out w*h*3 f32 (metres) */
int  fl_depth_to_3d(fl_context *ctx, const uint16_t *depth, int w, int h, double fx, double fy,
                    double cx, double cy, float *out, int mem);
/* icpCloudToCloud_Ex (ICP/ICP.cpp:617-809): clouds are n*3 f32 (mm). res is host memory. */
int  fl_icp(fl_context *ctx, const float *ref, int n_ref, const float *model, int n_model,
            int icp_it_thr, float dist_mean_thr, float dist_diff_thr, int icp_mode, int mem,
            fl_icp_result *res);
/* FL_ICP_POINT_TO_PLANE on caller-supplied clouds: ref_normals is n_ref*3 f32 (unit normals of the
 * reference cloud; an all-zero normal excludes that point from the 6x6 system).  fl_icp itself
 * refuses FL_ICP_POINT_TO_PLANE (FL_ERR_INVALID): it has no normals to work with.  No
 * counterpart in the reference. */
int  fl_icp_point_to_plane(fl_context *ctx, const float *ref, const float *ref_normals, int n_ref,
                           const float *model, int n_model, int icp_it_thr, float dist_mean_thr,
                           float dist_diff_thr, int mem, fl_icp_result *res);
/* detection() (ICP/detection.cpp:11-254): two w*h u16 depth images in mm. res is host memory. */
int  fl_detection(fl_context *ctx, const uint16_t *model_depth, const uint16_t *scene_depth,
                  int w, int h, const fl_intrinsics *K, const int rect_model[4],
                  const int rect_ref[4], int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
                  const float r_match[9], const float t_match[3], int icp_mode, int mem,
                  fl_detection_result *res);

/* ---- Detector::match (linemod.cpp:1356-1441) ------------------------------------------------ */
/* From caller-supplied quantized images (what Modality::process + quantize would return):
 * quantized[l*modalities + m] is (w0>>l)*(h0>>l) u8.  out/n_total are host memory; matches come
 * back sorted (similarity desc, template_id asc, then class, y, x asc -- one valid outcome of
 * the reference's unstable std::sort) and de-duplicated as std::unique with Match::operator==. */
int  fl_match_quantized(fl_detector *det, const uint8_t *const *quantized, int mem,
                        float threshold, fl_match *out, int cap, int *n_total);
/* From a BGR8 + depth16 frame with the default modality parameters (linemod.cpp:515-519,827-832) */
int  fl_match_frame(fl_detector *det, const uint8_t *bgr, const uint16_t *depth, int mem,
              float threshold, fl_match *out, int cap, int *n_total);
/* Detector::match's optional `masks` argument (linemod.hpp:319-327, linemod.cpp:1364-1379): masks is
 * NULL (empty vector) or an array of `modalities` pointers to w0*h0 u8 images in `mem`, each of
 * which may be NULL (empty Mat).  A level-l pixel is kept where the l-times NN-halved mask is
 * non-zero (linemod.cpp:445-459, 733-745). */
int  fl_match_frame_masked(fl_detector *det, const uint8_t *bgr, const uint16_t *depth,
                           const uint8_t *const *masks, int mem, float threshold, fl_match *out,
                           int cap, int *n_total);
/* Detector::match for a batch of frames (the reference is called once per camera frame, linemod.cpp:1356-1441;
 * test/linemod_recon.cpp:43-80): front-end and match of n_frames frames are queued on the context's stream, the sorted
 * match lists stay in HBM.  fl_match_batch_collect waits and copies frame `frame`'s list out (same order and
 * de-duplication as fl_match_frame); fl_export_topk reads them on the device. */
int  fl_match_batch_submit(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                           int mem, float threshold);
int  fl_match_batch_collect(fl_detector *det, int frame, fl_match *out, int cap, int *n_total);
/* similarity + addSimilarities (linemod.cpp:1130-1214,1322-1338) for pyramids [first,first+count)
 * at the coarsest level of the frame last passed to fl_match_*: out = count * (W_T*H_T) u16 (host) */
int  fl_similarity_maps(fl_detector *det, int first, int count, uint16_t *out);
/* the quantized images of the frame last passed to fl_match (Detector::match's optional
 * quantized_images output, linemod.cpp:1411-1412): levels*modalities images back to back (host) */
int  fl_last_quantized(fl_detector *det, uint8_t *out);

/* ---- CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:86-204), batched ---------------- */
/* n_frames frames of w0 x h0 (already 640 wide), all stages on the GPU with
 * no host round trip between LINEMOD and ICP.  bgr[i], depth[i] are host or device pointers
 * (all the same kind).  results is host memory.  Frames are independent; results[i] equals what
 * Recognition() returns for frame i. */
int  fl_recognize_batch(fl_detector *det, int n_frames, const uint8_t *const *bgr,
                        const uint16_t *const *depth, int mem, const fl_intrinsics *K,
                        const fl_recognition_params *params, fl_recognition_result *results);
/* PrepareInputData's zoom + Recognition (obj_reco_lmicp.cpp:229-249, 39-45): n_frames frames of src_w x src_h are
 * resized (cv::resize INTER_LINEAR semantics, as fl_resize_linear_*) to the detector's w0 x h0 on the device and
 * recognised from HBM without a host round trip.  K is passed through as the reference passes it. */
int  fl_recognize_batch_zoom(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                             int src_w, int src_h, int mem, const fl_intrinsics *K, const fl_recognition_params *params,
                             fl_recognition_result *results);
/* same, but only queues the work on the context's stream; fl_recognize_collect() waits and
 * copies the results out.  Lets the caller overlap two contexts / batches. */
int  fl_recognize_submit(fl_detector *det, int n_frames, const uint8_t *const *bgr,
                         const uint16_t *const *depth, int mem, const fl_intrinsics *K,
                         const fl_recognition_params *params);
int  fl_recognize_collect(fl_detector *det, int n_frames, fl_recognition_result *results);

/* ---- multi-GPU support: top-k records for the all-gather ------------------------------------ */
/* Multi-hypothesis recognition (the pipeline ICP/NMS.cpp + obj_data.h sketch): the refinement of
 * Recognition() (obj_reco_lmicp.cpp:111-197) for the first k matches of ONE frame, k ICP workgroups in one launch.
 * results[r] (host, r < *n_results = min(k, matches)) has the same content fl_recognize_batch gives for matches[0];
 * a hypothesis whose crop leaves the image has found = 0 and status = FL_ERR_ASSERT. */
int  fl_recognize_topk(fl_detector *det, const uint8_t *bgr, const uint16_t *depth, int mem, const fl_intrinsics *K,
                       const fl_recognition_params *params, int k, fl_recognition_result *results, int *n_results);
/* The same for a batch (n_frames * k ICP workgroups in one launch): results[f * k + r] (host) is hypothesis r of
 * frame f, n_results[f] = min(k, matches of frame f).  Frame pointers as for fl_recognize_batch. */
int  fl_recognize_batch_topk(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                             int mem, const fl_intrinsics *K, const fl_recognition_params *params, int k,
                             fl_recognition_result *results, int *n_results);
/* nonMaximumSuppression (ICP/NMS.cpp:6-40) over refined hypotheses in list order: winners[g] = index of the
 * hypothesis representing group g (translation closer than th_obj_dist to the group's current best). Host only. */
int  fl_nms(const fl_recognition_result *objs, int n, float th_obj_dist, int *winners, int *n_winners);
/* After fl_match_frame / fl_recognize_submit: copy frame `frame`'s first k sorted matches into a
 * device buffer (k * sizeof(fl_match) bytes, padded with template_id = -1) for an RCCL
 * all-gather by the caller; template ids are offset by template_id_base (the shard's first
 * global id).  Queued on the context's stream, no synchronisation. */
int  fl_export_topk(fl_detector *det, int frame, int k, int template_id_base, void *dev_out);
/* the same for every frame of the last batch in one launch: dev_out[frame * k + i].
 * A frame whose candidate buffers overflowed (its list is truncated in an order that depends on atomics) is exported as
 * record 0 = {template_id = FL_TOPK_OVERFLOW, class_idx = -1} and padding: the flag travels with the all-gather, so every
 * rank sees it (fl_select_best_batch turns it into that frame's status; the host merge ignores records with ids < 0). */
#define FL_TOPK_OVERFLOW (-2)
int  fl_export_topk_batch(fl_detector *det, int n_frames, int k, int template_id_base, void *dev_out);
/* merge n_ranks*k gathered records (host) exactly as one Detector::match over the union would
 * order them; returns the number written to out (<= cap). */
int  fl_merge_topk(const fl_match *gathered, int n_records, fl_match *out, int cap);

/* fl_merge_topk per frame of a batch: gathered[(rank * n_frames + frame) * k + i] is what an all-gather of the ranks'
 * fl_export_topk_batch buffers yields; out[frame * cap + j], n_out[frame] (all host memory) */
int  fl_merge_topk_batch(const fl_match *gathered, int n_ranks, int n_frames, int k, fl_match *out, int cap, int *n_out);
/* Template-sharded recognition (BASELINE configs[3]): after the merge, the rank that owns frame f's winning template
 * refines it -- the second half of Recognition() (obj_reco_lmicp.cpp:111-197: crop rectangles from the template, the
 * depth render, detection(), pose) for a match chosen by the caller instead of this detector's own matches[0].
 * frames[j] indexes the batch last queued with fl_match_batch_submit (whose depth frames must still be where they were),
 * matches[j].template_id is class-local on THIS detector; results[j] (host) as fl_recognize_batch fills them. */
int  fl_refine_matches(fl_detector *det, int n_jobs, const int32_t *frames, const fl_match *matches, const fl_intrinsics *K,
                       const fl_recognition_params *params, fl_recognition_result *results);
/* The same hand-over without leaving the device (no host round trip between Detector::match and the ICP):
 * fl_select_best_batch reads the all-gathered records IN DEVICE MEMORY (dev_gathered[(rank * n_frames + frame) * k + i],
 * global template ids) and writes, per frame, matches[0] of one global std::sort + std::unique over all ranks' lists
 * (linemod.cpp:1437-1439, Match::operator< linemod.hpp:262-267; Recognition() uses nothing else, obj_reco_lmicp.cpp:111)
 * to dev_best[frame] (global ids; template_id = -1: no rank matched; FL_TOPK_OVERFLOW: some rank's list overflowed) --
 * every list is sorted, so it is the best of the ranks' first records.  Frames whose winner lies in this rank's slice
 * [tid_first, tid_first + tid_count) become the detector's pending refinement jobs.
 * fl_refine_selected then runs the second half of Recognition() for those jobs (one launch over n_frames workgroups,
 * the ones without a job exit at once) and writes dev_rows[frame][17] = {found, row-major 4x4 pose} as float32, zeros
 * for the frames this rank does not own (a sum or a gather over the ranks gives every frame's pose).  depth_base /
 * depth_stride (bytes) locate the batch's depth frames on the device; NULL / 0 = where fl_match_batch_submit left them.
 * Both are queued on the context's stream without synchronisation. */
int  fl_select_best_batch(fl_detector *det, const void *dev_gathered, int n_ranks, int n_frames, int k, int tid_first,
                          int tid_count, void *dev_best);
int  fl_refine_selected(fl_detector *det, int n_frames, const fl_intrinsics *K, const fl_recognition_params *params,
                        const uint16_t *depth_base, size_t depth_stride, float *dev_rows);

/* diagnostics: per-frame counters of the last match {coarse candidates, matches after sort/unique, overflow flag,
 * level-0 tiles (60 x 60 pixels) whose colour quantisation the lazy fine levels computed; -1 when the batch ran eagerly}
 * (host memory) */
int  fl_frame_counters(fl_detector *det, int frame, int32_t out[4]);

/* per-stage device time (ms) of the last fl_recognize_* call, by stage index; for bench.py */
typedef struct {
  float frontend_ms, linmem_ms, scan_ms, refine_ms, sort_ms, backproject_ms, icp_ms, total_ms;
  int32_t icp_iters_total, icp_launches;
  double scan_algorithmic_bytes;
  float lazy_frontend_ms;   /* fl_recognize_*: colour quantisation + spread of the finer levels, computed after the scan
                               and only in the tiles the candidates touch (0 when FL_EAGER_FRONTEND=1: then part of
                               frontend_ms / linmem_ms).  refine_ms excludes it. */
  float reserved0;
} fl_stage_times;
int  fl_last_stage_times(fl_detector *det, fl_stage_times *out);

#ifdef __cplusplus
}
#endif
#endif /* FEALESS_HIP_H */
