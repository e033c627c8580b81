// fl_internal.h -- shared host-side state of libfealess_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/fealess_hip.h"

#define FL_WAVE 64
#define FL_CU_TABLE 4096              // entries of fl_context::d_cu_chain: (XCC_ID << 8) | HW_ID[15:8]

struct fl_context {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  // scratch for the single-shot stage entry points (grown on demand)
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  void *pinned = nullptr;       // small pinned host buffer for result read-back
  size_t pinned_bytes = 0;
  int cus = 256;                // multiProcessorCount of the device (launch heuristics), read once
  // per-CU bookings of the ICP kernel's chain waves (IcpSharedT::cw): 4 x 8-bit counts per compute unit, zeroed once
  unsigned *d_cu_chain = nullptr;
  // Development / comparison switches (fl_context_set_option).  Speed only: results are identical whatever they hold.
  // Initial values come from the environment ONCE, when the context is created; no launch path reads the environment.
  struct Options {
    long scan_prune = 1;        // FL_SCAN_PRUNE: k_scan's exact pruning
    long scan_prune_mid = -1;   // FL_SCAN_PRUNE_MID: bit mask of the 8-feature groups after which a modality checks the bound (-1: built-in)
    long icp_wide = -1;         // FL_ICP_WIDE: -1 by batch size, 0 / 1 force the 256- / 1024-thread ICP kernel
    long icp_occ = 0;           // FL_ICP_OCC: 0 by batch size, 4 / 5 force the 256-thread parity kernel built for 4 / 5 workgroups per CU
    long icp_order = 1;         // FL_ICP_ORDER: ICP jobs dealt longest first
    long icp_wg_per_cu = 0;     // FL_ICP_WG_PER_CU: 0 = as many 256-thread ICP workgroups per CU as fit (4); 1 .. 3 = at most that many, the
                                // rest of the CU left to kernels of other streams (two pipelines on one GPU)
    long eager_frontend = 0;    // FL_EAGER_FRONTEND: finer pyramid levels in full before the scan (read by fl_detector_finalize)
    long dev_poison = 0;        // FL_DEV_POISON: fill what the lazy path leaves uncomputed with 0xFF (read by fl_detector_finalize)
    long ws_pad = 0;            // FL_DEV_WS_PAD: extra bytes of frame workspace stride (read by fl_detector_finalize)
  } opt;
  int detectors = 0;            // live fl_detector objects on this context
  bool destroy_pending = false; // fl_context_destroy was called while detectors were alive: the last one releases the context
};

int fl_set_error(fl_context *ctx, int code, const char *fmt, ...);
int fl_scratch(fl_context *ctx, size_t bytes, void **out);
int fl_pinned(fl_context *ctx, size_t bytes, void **out);

#define FL_HIP(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fl_set_error((ctx), FL_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
                          hipGetErrorString(e_));                                          \
  } while (0)

static inline size_t fl_align(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device-side table entries ------------------------------------------------------------
// Offset of a feature's linear memory relative to the (level, modality) LM base of a frame.
#define FL_MAX_LEVELS 4
#define FL_MAX_MODALITIES 2
#define FL_MAX_FEATURES 63

struct FlScanHdr {       // one per (pyramid g, modality m) at the coarsest level
  int32_t P;             // template_positions (linemod.cpp:1155); <= 0: nothing to add
  int32_t off_begin;     // first entry in scan_offsets
  int32_t n_pad;         // entries, padded to a multiple of 8 with the zero offset
  int32_t nf;            // templ.features.size() (counts skipped features too)
};
struct FlFineFeat {      // one per feature at the finer levels (12 bytes)
  int16_t x, y;          // template-relative position
  int16_t qx, qy;        // floor(x / T), floor(y / T)
  uint8_t gx, gy;        // x mod T, y mod T (non-negative): the grid cell, invariant under offsets that are multiples of T
  uint8_t label, pad;    // quantized orientation 0..7
};
struct FlFineHdr {       // one per (pyramid g, level l < L-1, modality m)
  int32_t feat_begin, feat_count;
  int32_t width, height; // of this template (tp[start].width/height are taken from m == 0)
};
struct FlPyrInfo {       // one per pyramid g
  int32_t class_idx, template_id;
  int32_t off_x0, off_y0, width0, height0;   // template[0]: rect_model_raw (obj_reco_lmicp.cpp:129)
  int32_t depth_slot;                        // index into the model-depth bank or -1
  int32_t pad;
};
struct FlCand {          // a candidate / match in flight
  int32_t x, y;
  int32_t g;             // global pyramid index; -1 once filtered out
  float sim;
};

struct FlRefineJob {     // fl_refine_matches: refine `match` (class-local template id) on frame `frame` of the last batch
  int32_t frame;
  fl_match match;
};

#define FL_TILE 60                 // tile edge in pixels (= CQ_COLS = CQ_CH of k_color_quantize)
#define FL_TILE_WORDS 16           // bitmap words per frame, level and kind: up to 512 tiles (1280x960: 352)
// One block of uint32 per frame and fine level: bm[2][FL_TILE_WORDS] (kind 0: tiles whose spread bytes are read,
// kind 1: tiles whose quantised pixels those spreads are made of), then rows[2][32 * FL_TILE_WORDS]: for a marked
// tile (first needed image row << 16) | last needed image row.
#define FL_TILE_BLOCK_WORDS (66 * FL_TILE_WORDS)
#define FL_NUM_EVENTS 24

struct FlLevelGeom {
  int32_t w, h, T, W, H, WH;
  uint32_t stride;       // bytes per label block
  uint32_t zero_off;     // offset (within a modality's LM) of >= WH+16W+64 zero bytes
  size_t quant_off[FL_MAX_MODALITIES];   // offsets inside a frame workspace
  size_t lm_off[FL_MAX_MODALITIES];      // coarsest level only: the 8 linear memories
  size_t spread_off[FL_MAX_MODALITIES];  // finer levels: the spread image (w*h bytes)
  size_t bgr_off;                        // colour image of this level
};

struct FlClass {
  std::string id;
  int n_pyramids = 0;
  std::vector<fl_template> templates;
  std::vector<fl_feature> features;
  std::vector<float> poses;              // n_pyramids*13 or empty
  int first_g = 0;                       // set at finalize
  uint16_t *d_depths = nullptr;          // n_pyramids * dw*dh u16 (0.1 mm) or null
  int dw = 0, dh = 0;
};

struct fl_detector {
  fl_context *ctx = nullptr;
  int M = 0, L = 0;
  int T[FL_MAX_LEVELS] = {0};
  std::vector<FlClass> classes;          // kept sorted by id
  bool finalized = false;
  int w0 = 0, h0 = 0, max_batch = 0, cap = 0;
  bool cap_hard = false;                 // cap is a hard limit (FL_ERR_OVERFLOW) instead of an initial size that grows
  int n_pyr = 0;
  FlLevelGeom geom[FL_MAX_LEVELS];

  // device tables
  FlScanHdr *d_scan_hdr = nullptr;       // n_pyr * M
  int2 *d_scan_items = nullptr;          // (pyramid g, 1024-position chunk) pairs that have positions to scan: k_scan's work list
  int n_scan_items = 0;
  uint32_t *d_scan_off = nullptr;
  FlFineHdr *d_fine_hdr = nullptr;       // n_pyr * (L-1) * M, index (g*(L-1)+l)*M+m
  FlFineFeat *d_fine_feat = nullptr;
  FlPyrInfo *d_pyr = nullptr;            // n_pyr
  int *d_class_first = nullptr;          // first global pyramid index of each class
  uint8_t *d_pyr_enabled = nullptr;      // n_pyr: 0 = its class is excluded by the class filter of match()
  std::vector<std::string> class_filter; // empty = all classes (Detector::match's class_ids)
  float *d_poses = nullptr;              // n_pyr * 13 (zeros when absent)
  const uint16_t **d_depth_ptrs = nullptr; // n_pyr pointers to model depth (0.1mm) or null
  int depth_w = 0, depth_h = 0;
  int max_tw = 0, max_th = 0;            // largest template[0] width/height: bounds the ICP clouds

  // per-frame workspace: one allocation, frame stride `ws_stride`
  uint8_t *d_ws = nullptr;
  size_t ws_stride = 0;
  size_t off_bgr = 0, off_depth = 0, off_cand = 0, off_count = 0, off_keys = 0, off_match = 0;
  size_t off_icp = 0, off_tmp = 0;
  // Lazy fine levels (fl_recognize_*): the finer levels are only ever read by k_refine, around the candidates the
  // coarse scan produced.  Their colour quantisation and spread images are therefore computed after the scan and
  // only in the 60x60-pixel tiles the candidates' 16x16 patches can touch (k_mark_tiles).  Two bitmaps per frame
  // and fine level: tiles whose spread bytes are read, tiles whose quantised pixels those spreads read.
  size_t off_tiles = 0;                  // [L-1][FL_TILE_BLOCK_WORDS] uint32 per frame
  bool lazy_capable = false;             // tile grid fits the bitmaps
  bool lazy = false;                     // this batch runs lazily (set by fl_launch_frontend)
  bool eager_env = false, poison_env = false;   // FL_EAGER_FRONTEND / FL_DEV_POISON (dev knobs, read at finalize)
  const uint8_t *lazy_bgr = nullptr;     // level-0 colour frames of the batch in flight
  // Host frames (fl_recognize_submit with FL_MEM_HOST): uploaded on a copy stream into one of two input buffers, so
  // the upload of batch i+1 overlaps the compute of batch i (the compute stream waits on the upload's event, the
  // copy stream on the event of the compute that last read the buffer it is about to overwrite).
  hipStream_t copy_stream = nullptr;
  uint8_t *d_in[2] = {nullptr, nullptr}; // max_batch * (bgr + depth) bytes each, allocated on first use
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_read[2] = {nullptr, nullptr};
  bool read_pending[2] = {false, false};
  int in_flip = 0;
  size_t lazy_bgr_stride = 0;
  uint8_t *d_zoom = nullptr;             // fl_recognize_batch_zoom: the batch's zoomed frames (max_batch * w0*h0*5 bytes), on first use
  uint8_t *d_zoom_src = nullptr;         // and the un-zoomed host frames uploaded for it
  size_t zoom_src_bytes = 0;
  int n_pts_max = 0;
  int last_batch = 0;
  bool last_from_images = false;
  const uint16_t *last_depth_base = nullptr;   // the last batch's depth frames on the device (fl_refine_matches reads them)
  size_t last_depth_stride = 0;
  bool last_match_only = false;          // the last batch was fl_match_batch_submit (no ICP stage to time)
  bool last_refinable = false;           // the last batch came from a batch submit (stage_and_match) and its depth frames are still where
                                         // last_depth_base says: only then may fl_refine_matches / fl_export_topk_batch follow

  int *d_icp_order = nullptr;            // ICP launch: job order (longest first) + the jobs' size estimates, 2 * max_batch ints, on first use
  // template-sharded recognition on the device: the jobs fl_select_best_batch chose (frame = -1: not this rank's)
  FlRefineJob *d_jobs = nullptr;         // max_batch, allocated on first use
  int selected_frames = 0;               // frames of the last fl_select_best_batch (0: none pending)

  // results
  fl_recognition_result *d_results = nullptr;   // max_batch
  fl_recognition_result *h_results = nullptr;   // pinned
  // timing
  hipEvent_t ev[FL_NUM_EVENTS] = {nullptr};   // 0..6 stage boundaries, 8 + 2l / 9 + 2l around the lazy work of fine level l
  fl_stage_times times;
  bool have_times = false;
  double scan_bytes_per_frame = 0;       // SURVEY 8(d) B_tmpl summed over the bank
};

int fl_apply_class_filter(fl_detector *det);
int fl_grow_candidates(fl_detector *det, int needed);      // re-lays the frame workspaces out for >= needed candidates per frame
int fl_overflow_needed(fl_detector *det, int n_frames, int *needed);   // after a sync: largest candidate count of an overflowed frame, 0 if none
void fl_update_stage_times(fl_detector *det, int n_frames, const fl_recognition_result *results);

// ---- stage launchers (defined in the per-domain .hip files) ---------------------------------
// linemod
int fl_launch_build_lm(fl_context *ctx, const uint8_t *quant, size_t quant_stride, uint8_t *lm,
                       size_t lm_stride, int n_frames, int w, int h, int T);
int fl_launch_spread(fl_context *ctx, const uint8_t *quant, size_t quant_stride, uint8_t *spread,
                     size_t spread_stride, int n_frames, int w, int h, int T);
int fl_launch_match_core(fl_detector *det, int n_frames, float threshold);
int fl_launch_lazy_level(fl_detector *det, int n_frames, int level);   // frontend: colour quantisation + spreads of the marked tiles
// frontend
int fl_launch_quantized_orientations(fl_context *ctx, const uint8_t *bgr, size_t in_stride,
                                     uint8_t *dst, size_t out_stride, int n_frames, int w, int h,
                                     float weak_threshold);
int fl_launch_quantized_normals(fl_context *ctx, const uint16_t *depth, size_t in_stride,
                                uint8_t *dst, size_t out_stride, uint8_t *tmp, size_t tmp_stride,
                                int n_frames, int w, int h, int distance_threshold,
                                int difference_threshold);
int fl_launch_pyrdown_bgr(fl_context *ctx, const uint8_t *src, size_t in_stride, uint8_t *dst,
                          size_t out_stride, int n_frames, int w, int h);
int fl_launch_resize_nn_half(fl_context *ctx, const uint8_t *src, size_t in_stride, uint8_t *dst,
                             size_t out_stride, int n_frames, int w, int h);
int fl_launch_quantized_orientations_mag(fl_context *ctx, const uint8_t *bgr, size_t in_stride, uint8_t *dst, size_t out_stride,
                                         int n_frames, int w, int h, float weak_threshold, float *mag_out);
int fl_launch_frontend(fl_detector *det, int n_frames, const uint8_t *bgr, size_t bgr_stride,
                       const uint16_t *depth, size_t depth_stride, bool allow_lazy);
int fl_launch_spread_tiles(fl_context *ctx, const uint8_t *quant, size_t quant_stride, uint8_t *spread, size_t spread_stride,
                           int n_frames, int w, int h, int T, const uint32_t *tiles, size_t tiles_stride);
int fl_launch_resize_linear_bgr8(fl_context *ctx, const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);
int fl_launch_resize_linear_u16(fl_context *ctx, const uint16_t *src, int sw, int sh, uint16_t *dst, int dw, int dh);
// icp
size_t fl_icp_ws_bytes(int n_pts_max);
int fl_icp_prepare(fl_detector *det);      // fl_detector_finalize: job-order buffer, function attributes
int fl_launch_detection_topk(fl_detector *det, int n_frames, int k, const fl_intrinsics *K, const fl_recognition_params *p,
                             const uint16_t *depth, size_t depth_stride, uint8_t *ws, fl_recognition_result *d_results);
size_t fl_icp_ws_bytes(int n_pts_max);
int fl_launch_detection_jobs(fl_detector *det, int n_jobs, const FlRefineJob *d_jobs, const fl_intrinsics *K, const fl_recognition_params *p,
                             const uint16_t *depth, size_t depth_stride);
int fl_launch_detection_batch(fl_detector *det, int n_frames, const fl_intrinsics *K,
                              const fl_recognition_params *p, const uint16_t *depth,
                              size_t depth_stride);
