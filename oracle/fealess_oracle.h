/*
 * fealess_oracle.h -- CPU restatement (the ORACLE) of the rlvc/FEALESS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and there
 * only as the checker / the timed CPU baseline.  The product path (fealess_amd/csrc) never
 * links, loads or calls it and has no CPU fallback.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4), and it cannot be built here (every hot-path TU needs OpenCV 3.x,
 * which is absent from the image and un-vendored).  This oracle is therefore a careful
 * restatement from the reference's sources, each function citing the file:line it follows,
 * plus the published OpenCV 3.x algorithms for the un-vendored calls (GaussianBlur, Sobel,
 * fastAtan2, pyrDown, medianBlur, JacobiSVD, FLANN exact NN) -- stated explicitly where used.
 *
 * All paths are relative to /root/reference.
 */
#ifndef FEALESS_ORACLE_H
#define FEALESS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data model (linemod/linemod.hpp:32-58) -------------------------------------------- */
typedef struct { int32_t x, y, label; } orc_feature;           /* Feature, linemod.hpp:32-43 */
typedef struct {                                               /* Template, linemod.hpp:47-58 */
  int32_t width, height, offset_x, offset_y, pyramid_level;
  int32_t feat_begin, feat_count;                              /* range in a flat feature array */
} orc_template;
typedef struct {                                               /* Match, linemod.hpp:253-281 */
  int32_t x, y;
  float   similarity;
  int32_t class_idx;      /* index of class_id in sorted (std::map) order */
  int32_t template_id;    /* class-local id */
} orc_match;

/* A template bank of one class: n_pyramids TemplatePyramids, each levels*modalities templates
 * ordered [l*M + m] (linemod.hpp:372-373, linemod.cpp:1602). */
typedef struct {
  int32_t n_pyramids, levels, modalities;
  const orc_template *templates;   /* n_pyramids*levels*modalities */
  const orc_feature  *features;    /* flat */
} orc_bank;

/* ---- LINEMOD scan stages --------------------------------------------------------------- */
void orc_similarity_lut(uint8_t lut[256]);                                  /* linemod.cpp:970  */
void orc_spread(const uint8_t *src, int w, int h, int T, uint8_t *dst);      /* linemod.cpp:950  */
void orc_response_maps(const uint8_t *spread, int w, int h, uint8_t *maps8); /* linemod.cpp:979  */
/* linearize (linemod.cpp:1060): out is T*T rows of (w/T)*(h/T) bytes. returns 0 / -1 on assert */
int  orc_linearize(const uint8_t *map, int w, int h, int T, uint8_t *out);
size_t orc_lm_label_stride(int w, int h, int T);  /* bytes per label incl. zero pad (see .c) */
/* build the 8 padded linear memories of one (level, modality): out has 8*orc_lm_label_stride */
int  orc_build_linear_memories(const uint8_t *quantized, int w, int h, int T, uint8_t *out);
/* similarity (linemod.cpp:1130): dst is (h/T)*(w/T) u8 */
int  orc_similarity(const uint8_t *lm8, const orc_template *t, const orc_feature *feats,
                    int w, int h, int T, uint8_t *dst);
/* similarityLocal (linemod.cpp:1226): dst is 16*16 u8 */
int  orc_similarity_local(const uint8_t *lm8, const orc_template *t, const orc_feature *feats,
                          int w, int h, int T, int cx, int cy, uint8_t *dst);
/* raw u16 total similarity map of one pyramid at the coarsest level (linemod.cpp:1467-1481) */
int  orc_total_similarity(const uint8_t *const *lm_level /*[M]*/, const orc_bank *bank, int pyramid,
                          int w, int h, int T, uint16_t *dst);

/* Detector::match on caller-supplied quantized images (pass-through modality):
 * quantized[l*M+m] is the w_l x h_l u8 one-hot image of level l, modality m
 * (w_l = w0 >> l, h_l = h0 >> l).  banks[c] are the classes in std::map order.
 * Returns number of matches written (<= cap), or -1 on a reference assert; *n_total gets the
 * count before truncation.  Output order: see orc_sort_unique. (linemod.cpp:1356-1577) */
int  orc_match_quantized(const uint8_t *const *quantized, int w0, int h0,
                         int levels, int modalities, const int *T_at_level,
                         const orc_bank *banks, int n_classes, float threshold,
                         orc_match *out, int cap, int *n_total);
/* canonical std::sort + std::unique (linemod.cpp:1437-1439, linemod.hpp:262-274) */
int  orc_sort_unique(orc_match *m, int n);

/* ---- quantisation front-end ------------------------------------------------------------ */
void orc_normal_lut(uint8_t lut[8000]);                                     /* normal_lut.i     */
/* quantizedNormals (linemod.cpp:595-685) incl. medianBlur 5 */
void orc_quantized_normals(const uint16_t *depth, int w, int h, int distance_threshold,
                           int difference_threshold, uint8_t *dst);
/* quantizedOrientations + hysteresisGradient (linemod.cpp:230-385); magnitude may be NULL */
void orc_quantized_orientations(const uint8_t *bgr, int w, int h, float weak_threshold,
                                uint8_t *dst, float *magnitude);
void orc_pyrdown_bgr(const uint8_t *src, int w, int h, uint8_t *dst);       /* cv::pyrDown 8UC3 */
void orc_resize_nn_half(const uint8_t *src, int w, int h, uint8_t *dst);    /* linemod.cpp:731  */
void orc_gaussian7_bgr(const uint8_t *src, int w, int h, uint8_t *dst);
void orc_median5(const uint8_t *src, int w, int h, uint8_t *dst);
float orc_fast_atan2(float y, float x);
/* Detector::match from BGR + depth16 with the two default modalities
 * (ColorGradient(10,63,55), DepthNormal(2000,50,63,2); linemod.cpp:515-519,827-832).
 * quantized_out (optional) receives levels*2 images back to back. */
int  orc_match_images(const uint8_t *bgr, const uint16_t *depth, int w0, int h0,
                      int levels, const int *T_at_level,
                      const orc_bank *banks, int n_classes, float threshold,
                      orc_match *out, int cap, int *n_total, uint8_t *quantized_out);

/* cv::resize(INTER_LINEAR) of PrepareInputData (obj_reco_lmicp.cpp:39-45, 248-249); cn = channels */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int cn, uint8_t *dst, int dw, int dh);
void orc_resize_linear_u16(const uint16_t *src, int sw, int sh, uint16_t *dst, int dw, int dh);

/* same with the optional per-modality masks of Detector::match (linemod.hpp:319-327); NULL = empty */
int  orc_match_images_masked(const uint8_t *bgr, const uint16_t *depth, int w0, int h0,
                             int levels, const int *T_at_level,
                             const orc_bank *banks, int n_classes, float threshold,
                             const uint8_t *mask_color, const uint8_t *mask_depth,
                             orc_match *out, int cap, int *n_total, uint8_t *quantized_out);

/* ---- template extraction (extract_oracle.c) ------------------------------ */
void orc_erode_rect(const uint8_t *src, int w, int h, int iterations, uint8_t *dst);            /* cv::erode 3x3 rect */
void orc_distance_transform_c3(const uint8_t *src, int w, int h, float *dst);                   /* DIST_C, 3x3 */
int  orc_extract_template_color(const uint8_t *quantized, const float *magnitude, const uint8_t *mask, int w, int h,
                                float strong_threshold, int num_features, orc_feature *out);    /* linemod.cpp:461-513 */
int  orc_extract_template_depth(const uint8_t *normal, const uint8_t *mask, int w, int h, int extract_threshold,
                                int num_features, orc_feature *out);                            /* linemod.cpp:747-825 */
void orc_crop_templates(orc_template *t, int n, orc_feature *feats, int bb[4]);                 /* linemod.cpp:52-96 */
/* Detector::addTemplate (linemod.cpp:1579-1615), default modalities; templates[levels*2] ordered [l*2+m], feature
 * slot of template k = feats + 63*k; returns 0 / -1 */
int  orc_add_template(const uint8_t *bgr, const uint16_t *depth, const uint8_t *mask, int w0, int h0, int levels,
                      orc_template *templates, orc_feature *feats, int bb[4]);

/* ---- back-projection + ICP ------------------------------------------------------------- */
/* cup_d2pc::depthTo3d u16 path (depth_to_3d.cpp:99-137,190-221,244-269): out is w*h*3 f32, metres */
void orc_depth_to_3d(const uint16_t *depth, int w, int h, double fx, double fy, double cx,
                     double cy, float *out);
typedef struct {
  float R[9], T[3];
  float dist_mean;      /* return value of icpCloudToCloud_Ex */
  float px_ratio;
  int32_t iters;        /* value of `iter` on exit */
  int32_t n_corr_last;  /* correspondences in the last executed iteration */
} orc_icp_result;
/* icpCloudToCloud_Ex (ICP/ICP.cpp:617-809).  accum64 != 0: sums in double (the "exact64" yardstick,
 * NOT the reference's arithmetic); accum64 == 0 mimics the reference's sequential float32 sums.
 * use_kdtree != 0 uses the kd-tree NN (the timed baseline), else brute force; both are exact
 * with ties to the lowest index. trace (optional) gets per-iteration {n_corr, dist_mean,
 * C[9], R_opt[9], T_opt[3]} = 23 floats per iteration. */
int  orc_icp(const float *ref, int n_ref, const float *model, int n_model,
             int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
             int accum64, int use_kdtree, orc_icp_result *res, float *trace, int trace_cap);
/* cv::SVD::compute on a 3x3 float matrix (OpenCV JacobiSVDImpl_<float> restated). A = U diag(W) Vt */
void orc_svd3(const float A[9], float W[3], float U[9], float Vt[9]);
/* detection() (ICP/detection.cpp:11-254, live lines) on two full-frame u16 depth images (mm). */
typedef struct {
  float R_final[9], T_final[3];
  orc_icp_result icp;
  int32_t n_points;
} orc_detection_result;
int  orc_detection(const uint16_t *model_depth, const uint16_t *scene_depth, int w, int h,
                   double fx, double fy, double cx, double cy,
                   const int rect_model[4], const int rect_ref[4],
                   int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
                   const float r_match[9], const float t_match[3],
                   int accum64, int use_kdtree, orc_detection_result *res);
/* CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:86-204) for one class, width-640 input.
 * model_depths_01mm: per-pyramid full-frame u16 depth renders in 0.1 mm (the depth/<id>.png files).
 * Returns 0 with *found = 0/1. */
typedef struct {
  int32_t found;
  orc_match best;
  float pose[16];
  orc_detection_result det;
  int32_t n_matches;
} orc_recognition_result;
int  orc_recognition(const uint8_t *bgr, const uint16_t *depth, int w, int h,
                     double fx, double fy, double cx, double cy,
                     int levels, const int *T_at_level, const orc_bank *bank,
                     const float *poses13, const uint16_t *const *model_depths_01mm,
                     float threshold, int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
                     int accum64, int use_kdtree, orc_recognition_result *res);
/* stage times (ms) of the calling thread's last orc_recognition at the reference's own timer points:
 * [0] "Time of linemod" (CadReco/obj_reco_lmicp.cpp:88,124-125), [1] "Time of ICP" (:126,201-202) */
void orc_last_stage_ms(double out[2]);

/* multi-hypothesis refinement + nonMaximumSuppression (ICP/NMS.cpp:6-40, obj_data.h) */
int orc_recognition_topk(const uint8_t *bgr, const uint16_t *depth, int w, int h, double fx, double fy, double cx, double cy,
                         int levels, const int *T_at_level, const orc_bank *bank, const float *poses13,
                         const uint16_t *const *model_depths_01mm, float threshold, int icp_it_thr, float dist_mean_thr,
                         float dist_diff_thr, int accum64, int use_kdtree, int k, orc_recognition_result *results);
int orc_nms(const orc_recognition_result *objs, int n, float th_obj_dist, int *winners);

#ifdef __cplusplus
}
#endif
#endif
