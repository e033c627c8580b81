// fl_context.hip -- context, detector construction and HBM layout of libfealess_hip.so.
//
// Replaces (reference paths): cup_linemod::Detector construction / addSyntheticTemplate /
// addPoseInfo (linemod/linemod.cpp:1348-1354, 1617-1642) and the per-frame disk read of the
// template depth render (CadReco/obj_reco_lmicp.cpp:156-157, uploaded once here instead).
#include "fl_internal.h"
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <algorithm>

int fl_set_error(fl_context *ctx, int code, const char *fmt, ...)
{
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

int fl_scratch(fl_context *ctx, size_t bytes, void **out)
{
  if (bytes > ctx->scratch_bytes) {
    if (ctx->scratch) {
      FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
      FL_HIP(ctx, hipFree(ctx->scratch));
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
    }
    size_t nb = fl_align(bytes + bytes / 4, 1 << 20);
    FL_HIP(ctx, hipMalloc(&ctx->scratch, nb));
    ctx->scratch_bytes = nb;
  }
  *out = ctx->scratch;
  return FL_OK;
}

int fl_pinned(fl_context *ctx, size_t bytes, void **out)
{
  if (bytes > ctx->pinned_bytes) {
    if (ctx->pinned) {
      FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
      FL_HIP(ctx, hipHostFree(ctx->pinned));
      ctx->pinned = nullptr;
      ctx->pinned_bytes = 0;
    }
    size_t nb = fl_align(bytes, 1 << 16);
    FL_HIP(ctx, hipHostMalloc(&ctx->pinned, nb, hipHostMallocDefault));
    ctx->pinned_bytes = nb;
  }
  *out = ctx->pinned;
  return FL_OK;
}

extern "C" int fl_abi_version(void) { return FL_ABI_VERSION; }

struct FlOptionName { const char *name, *env; long fl_context::Options::*field; bool hex; };
static const FlOptionName fl_option_names[] = {
  {"scan_prune", "FL_SCAN_PRUNE", &fl_context::Options::scan_prune, false},
  {"scan_prune_mid", "FL_SCAN_PRUNE_MID", &fl_context::Options::scan_prune_mid, true},
  {"icp_wide", "FL_ICP_WIDE", &fl_context::Options::icp_wide, false},
  {"icp_occ", "FL_ICP_OCC", &fl_context::Options::icp_occ, false},
  {"icp_order", "FL_ICP_ORDER", &fl_context::Options::icp_order, false},
  {"icp_wg_per_cu", "FL_ICP_WG_PER_CU", &fl_context::Options::icp_wg_per_cu, false},
  {"eager_frontend", "FL_EAGER_FRONTEND", &fl_context::Options::eager_frontend, false},
  {"dev_poison", "FL_DEV_POISON", &fl_context::Options::dev_poison, false},
  {"ws_pad", "FL_DEV_WS_PAD", &fl_context::Options::ws_pad, false},
};

extern "C" int fl_context_create(int device, fl_context **out)
{
  if (!out) return FL_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FL_ERR_NO_DEVICE;   // no CPU fallback
  if (device < 0 || device >= n) return FL_ERR_INVALID;
  fl_context *ctx = new fl_context();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return FL_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->cus = prop.multiProcessorCount;
  }
  if (hipMalloc((void **)&ctx->d_cu_chain, sizeof(unsigned) * FL_CU_TABLE) != hipSuccess ||
      hipMemset(ctx->d_cu_chain, 0, sizeof(unsigned) * FL_CU_TABLE) != hipSuccess) {
    (void)hipGetLastError();
    if (ctx->d_cu_chain) (void)hipFree(ctx->d_cu_chain);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return FL_ERR_HIP;
  }
  // the development switches' initial values: the environment is read here and nowhere else
  for (const FlOptionName &o : fl_option_names)
    if (const char *e = getenv(o.env)) ctx->opt.*(o.field) = o.hex ? (long)strtoul(e, nullptr, 16) : atol(e);
  *out = ctx;
  return FL_OK;
}

extern "C" void *fl_context_get_stream(fl_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int fl_context_get_device(const fl_context *ctx) { return ctx ? ctx->device : -1; }
extern "C" fl_context *fl_detector_get_context(fl_detector *det) { return det ? det->ctx : nullptr; }

extern "C" int fl_context_set_option(fl_context *ctx, const char *name, long value)
{
  if (!ctx || !name) return FL_ERR_INVALID;
  for (const FlOptionName &o : fl_option_names)
    if (!strcmp(name, o.name)) { ctx->opt.*(o.field) = value; return FL_OK; }
  return fl_set_error(ctx, FL_ERR_INVALID, "unknown option '%s'", name);
}

extern "C" int fl_context_get_option(const fl_context *ctx, const char *name, long *value)
{
  if (!ctx || !name || !value) return FL_ERR_INVALID;
  for (const FlOptionName &o : fl_option_names)
    if (!strcmp(name, o.name)) { *value = ctx->opt.*(o.field); return FL_OK; }
  return FL_ERR_INVALID;
}

// A context outlives the detectors created on it whatever order the caller destroys them in: fl_context_destroy on a
// context that still has detectors only marks it; the last fl_detector_destroy releases it.
static void context_release(fl_context *ctx)
{
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->d_cu_chain) (void)hipFree(ctx->d_cu_chain);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

extern "C" void fl_context_destroy(fl_context *ctx)
{
  if (!ctx) return;
  if (ctx->detectors > 0) { ctx->destroy_pending = true; return; }
  context_release(ctx);
}

extern "C" const char *fl_last_error(const fl_context *ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int fl_context_set_stream(fl_context *ctx, void *hip_stream)
{
  if (!ctx) return FL_ERR_INVALID;
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return FL_OK;
}

extern "C" int fl_context_synchronize(fl_context *ctx)
{
  if (!ctx) return FL_ERR_INVALID;
  FL_HIP(ctx, hipSetDevice(ctx->device));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FL_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" size_t fl_lm_label_stride(int w, int h, int T)
{
  // [T*T grids][W*H] + zero pad covering every over-read the reference can make inside one
  // continuous linear-memory Mat (Q2 of SURVEY.md section 8; oracle/linemod_oracle.c has the
  // same formula)
  size_t W = (size_t)(w / T), H = (size_t)(h / T);
  size_t pad = W * H + 16 * W + 64;
  size_t s = (size_t)T * T * W * H + pad;
  return (s + 63) & ~(size_t)63;
}

extern "C" int fl_detector_create(fl_context *ctx, int modalities, int levels, const int *T_at_level,
                                  fl_detector **out)
{
  if (!ctx || !out || !T_at_level) return FL_ERR_INVALID;
  if (modalities < 1 || modalities > FL_MAX_MODALITIES || levels < 1 || levels > FL_MAX_LEVELS)
    return fl_set_error(ctx, FL_ERR_INVALID, "modalities must be 1..%d and levels 1..%d", FL_MAX_MODALITIES,
                        FL_MAX_LEVELS);
  for (int l = 0; l < levels; ++l)
    if (T_at_level[l] < 1 || T_at_level[l] > 16) return fl_set_error(ctx, FL_ERR_INVALID, "T out of range");
  fl_detector *det = new fl_detector();
  det->ctx = ctx;
  ++ctx->detectors;
  det->M = modalities;
  det->L = levels;
  for (int l = 0; l < levels; ++l) det->T[l] = T_at_level[l];
  memset(&det->times, 0, sizeof(det->times));
  *out = det;
  return FL_OK;
}

static void free_device_tables(fl_detector *det)
{
  (void)hipFree(det->d_scan_hdr);
  (void)hipFree(det->d_scan_items);
  (void)hipFree(det->d_scan_off);
  (void)hipFree(det->d_fine_hdr);
  (void)hipFree(det->d_fine_feat);
  (void)hipFree(det->d_pyr);
  (void)hipFree(det->d_poses);
  (void)hipFree(det->d_class_first);
  (void)hipFree(det->d_pyr_enabled);
  (void)hipFree((void *)det->d_depth_ptrs);
  (void)hipFree(det->d_ws);
  (void)hipFree(det->d_results);
  if (det->h_results) (void)hipHostFree(det->h_results);
  det->d_scan_hdr = nullptr;
  det->d_scan_items = nullptr;
  det->n_scan_items = 0;
  det->d_scan_off = nullptr;
  det->d_fine_hdr = nullptr;
  det->d_fine_feat = nullptr;
  det->d_pyr = nullptr;
  det->d_poses = nullptr;
  det->d_class_first = nullptr;
  det->d_pyr_enabled = nullptr;
  det->d_depth_ptrs = nullptr;
  det->d_ws = nullptr;
  det->d_results = nullptr;
  det->h_results = nullptr;
  for (auto &e : det->ev)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (det->d_jobs) { (void)hipFree(det->d_jobs); det->d_jobs = nullptr; }
  if (det->d_icp_order) { (void)hipFree(det->d_icp_order); det->d_icp_order = nullptr; }
  det->selected_frames = 0;
  if (det->d_zoom) { (void)hipFree(det->d_zoom); det->d_zoom = nullptr; }
  if (det->d_zoom_src) { (void)hipFree(det->d_zoom_src); det->d_zoom_src = nullptr; det->zoom_src_bytes = 0; }
  for (int b = 0; b < 2; ++b) {
    if (det->d_in[b]) { (void)hipFree(det->d_in[b]); det->d_in[b] = nullptr; }
    if (det->ev_up[b]) { (void)hipEventDestroy(det->ev_up[b]); det->ev_up[b] = nullptr; }
    if (det->ev_read[b]) { (void)hipEventDestroy(det->ev_read[b]); det->ev_read[b] = nullptr; }
    det->read_pending[b] = false;
  }
  if (det->copy_stream) { (void)hipStreamSynchronize(det->copy_stream); (void)hipStreamDestroy(det->copy_stream); det->copy_stream = nullptr; }
}

extern "C" void fl_detector_destroy(fl_detector *det)
{
  if (!det) return;
  (void)hipSetDevice(det->ctx->device);
  (void)hipStreamSynchronize(det->ctx->stream);
  free_device_tables(det);
  for (auto &c : det->classes)
    if (c.d_depths) (void)hipFree(c.d_depths);
  fl_context *ctx = det->ctx;
  delete det;
  if (--ctx->detectors == 0 && ctx->destroy_pending) context_release(ctx);
}

extern "C" int fl_detector_add_class(fl_detector *det, const char *class_id, int n_pyramids,
                                     const fl_template *templates, const fl_feature *features,
                                     int n_features, const float *poses13)
{
  if (!det || !class_id || n_pyramids < 0 || (n_pyramids && !templates)) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "detector already finalized");
  for (auto &c : det->classes)
    if (c.id == class_id) return fl_set_error(ctx, FL_ERR_ASSERT, "class '%s' already present (linemod.cpp:1727)", class_id);
  size_t nt = (size_t)n_pyramids * det->L * det->M;
  for (size_t i = 0; i < nt; ++i) {
    const fl_template &t = templates[i];
    if (t.feat_count < 0 || t.feat_begin < 0 || (long)t.feat_begin + t.feat_count > n_features)
      return fl_set_error(ctx, FL_ERR_INVALID, "template %zu: feature range outside the feature array", i);
    if (t.feat_count > FL_MAX_FEATURES)
      return fl_set_error(ctx, FL_ERR_ASSERT, "template %zu has %d features > 63 (CV_Assert linemod.cpp:1137)", i,
                          t.feat_count);
    for (int k = 0; k < t.feat_count; ++k) {
      const fl_feature &f = features[t.feat_begin + k];
      if (f.label < 0 || f.label > 7) return fl_set_error(ctx, FL_ERR_INVALID, "feature label outside 0..7");
      if (f.x < -32768 || f.x > 32767 || f.y < -32768 || f.y > 32767)
        return fl_set_error(ctx, FL_ERR_INVALID, "feature coordinate outside int16");
    }
  }
  FlClass c;
  c.id = class_id;
  c.n_pyramids = n_pyramids;
  c.templates.assign(templates, templates + nt);
  c.features.assign(features, features + n_features);
  if (poses13) c.poses.assign(poses13, poses13 + (size_t)n_pyramids * 13);
  auto it = std::lower_bound(det->classes.begin(), det->classes.end(), c,
                             [](const FlClass &a, const FlClass &b) { return a.id < b.id; });
  det->classes.insert(it, std::move(c));
  return FL_OK;
}

extern "C" int fl_detector_set_model_depths(fl_detector *det, int class_idx, int first, int count,
                                            const uint16_t *depth_01mm, int w, int h, int mem)
{
  if (!det || !depth_01mm || w <= 0 || h <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (class_idx < 0 || class_idx >= (int)det->classes.size()) return fl_set_error(ctx, FL_ERR_INVALID, "class_idx");
  FlClass &c = det->classes[class_idx];
  if (first < 0 || count < 0 || first + count > c.n_pyramids) return fl_set_error(ctx, FL_ERR_INVALID, "pyramid range");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  size_t per = (size_t)w * h * sizeof(uint16_t);
  if (!c.d_depths) {
    FL_HIP(ctx, hipMalloc((void **)&c.d_depths, per * (size_t)c.n_pyramids));
    FL_HIP(ctx, hipMemsetAsync(c.d_depths, 0, per * (size_t)c.n_pyramids, ctx->stream));
    c.dw = w;
    c.dh = h;
  } else if (c.dw != w || c.dh != h)
    return fl_set_error(ctx, FL_ERR_INVALID, "model depth size changed");
  FL_HIP(ctx, hipMemcpyAsync((uint8_t *)c.d_depths + per * (size_t)first, depth_01mm, per * (size_t)count,
                             mem == FL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FL_OK;
}

extern "C" int fl_detector_num_templates(const fl_detector *det)
{
  if (!det) return FL_ERR_INVALID;
  int n = 0;
  for (auto &c : det->classes) n += c.n_pyramids;
  return n;
}
extern "C" int fl_detector_num_classes(const fl_detector *det) { return det ? (int)det->classes.size() : FL_ERR_INVALID; }

template <typename T>
static int upload(fl_context *ctx, const std::vector<T> &v, T **out)
{
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  FL_HIP(ctx, hipMalloc((void **)out, bytes));
  if (!v.empty()) FL_HIP(ctx, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return FL_OK;
}

// Detector::match's class_ids (linemod.cpp:1418-1434): empty = every class; otherwise only the listed classes
// that exist are matched (unknown ids are ignored, repeats change nothing after sort + unique).
int fl_apply_class_filter(fl_detector *det)
{
  fl_context *ctx = det->ctx;
  std::vector<uint8_t> en;
  for (size_t ci = 0; ci < det->classes.size(); ++ci) {
    bool on = det->class_filter.empty();
    for (size_t k = 0; k < det->class_filter.size(); ++k) on = on || det->class_filter[k] == det->classes[ci].id;
    en.insert(en.end(), (size_t)det->classes[ci].n_pyramids, on ? 1 : 0);
  }
  if (en.empty()) return FL_OK;
  if (!det->d_pyr_enabled) FL_HIP(ctx, hipMalloc((void **)&det->d_pyr_enabled, en.size()));
  FL_HIP(ctx, hipMemcpyAsync(det->d_pyr_enabled, en.data(), en.size(), hipMemcpyHostToDevice, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FL_OK;
}

extern "C" int fl_detector_set_class_filter(fl_detector *det, const char *const *class_ids, int n)
{
  if (!det || n < 0 || (n > 0 && !class_ids)) return FL_ERR_INVALID;
  det->class_filter.clear();
  for (int i = 0; i < n; ++i) det->class_filter.push_back(class_ids[i] ? class_ids[i] : "");
  if (!det->finalized) return FL_OK;                     // applied by fl_detector_finalize
  FL_HIP(det->ctx, hipSetDevice(det->ctx->device));
  return fl_apply_class_filter(det);
}

// The per-frame workspace (one allocation, frame stride ws_stride) for candidate capacity `cap`; allocates d_ws.
static int layout_workspace(fl_detector *det, int cap)
{
  fl_context *ctx = det->ctx;
  const int L = det->L, M = det->M, w0 = det->w0, h0 = det->h0;
  // per-frame workspace layout
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = fl_align(off + bytes, 256); return o; };
  det->off_bgr = take((size_t)w0 * h0 * 3);
  det->off_depth = take((size_t)w0 * h0 * 2);
  for (int l = 0; l < L; ++l) {
    FlLevelGeom &g = det->geom[l];
    g.bgr_off = l == 0 ? det->off_bgr : take((size_t)g.w * g.h * 3);
    for (int m = 0; m < M; ++m) g.quant_off[m] = take((size_t)g.w * g.h);
    // only the coarsest level is scanned by every template and gets linear memories; the finer
    // levels are touched by a handful of 16x16 patches per frame and keep just the spread image
    for (int m = 0; m < M; ++m) {
      g.lm_off[m] = l == L - 1 ? take((size_t)8 * g.stride + 64) : 0;
      g.spread_off[m] = l == L - 1 ? 0 : take((size_t)g.w * g.h + 64);
    }
  }
  det->off_tmp = take((size_t)w0 * h0);
  det->off_count = take(256);
  det->off_tiles = take((size_t)(L > 1 ? L - 1 : 1) * FL_TILE_BLOCK_WORDS * sizeof(uint32_t));
  det->lazy_capable = L > 1;
  for (int l = 0; l + 1 < L; ++l) {
    const FlLevelGeom &g = det->geom[l];
    if (((g.w + FL_TILE - 1) / FL_TILE) * ((g.h + FL_TILE - 1) / FL_TILE) > FL_TILE_WORDS * 32) det->lazy_capable = false;
  }
  det->off_cand = take((size_t)cap * sizeof(FlCand));
  det->off_keys = take((size_t)cap * 16);
  det->off_match = take((size_t)cap * sizeof(fl_match));
  det->n_pts_max = det->max_tw * det->max_th;
  det->off_icp = take(fl_icp_ws_bytes(det->n_pts_max));
  det->ws_stride = fl_align(off, 4096);
  det->cap = cap;
  if (ctx->opt.ws_pad > 0) det->ws_stride += fl_align((size_t)ctx->opt.ws_pad, 4096);   // dev aid: stride sensitivity
  if (hipMalloc((void **)&det->d_ws, det->ws_stride * (size_t)det->max_batch) != hipSuccess) {
    det->d_ws = nullptr;
    (void)hipGetLastError();
    return fl_set_error(ctx, FL_ERR_HIP, "frame workspaces: %zu MB for %d frames with %d candidates each", (det->ws_stride * (size_t)det->max_batch) >> 20,
                        det->max_batch, cap);
  }
  // on the context's stream (it is non-blocking: a null-stream hipMemset is not ordered against it and would still be
  // zeroing the workspace while the replay after fl_grow_candidates fills it), and finished before anything else starts
  FL_HIP(ctx, hipMemsetAsync(det->d_ws, 0, det->ws_stride * (size_t)det->max_batch, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FL_OK;
}

// The reference's candidate / match vectors grow without limit (linemod.cpp:1490-1504, 1575); here they are fixed-size
// slices of the frame workspace, so a frame that overflows them makes the caller-facing entry points grow the slices
// (workspace re-laid out for the next power of two >= `needed`) and run the batch again.  Nothing but scratch lives
// in the workspace between calls.
int fl_grow_candidates(fl_detector *det, int needed)
{
  fl_context *ctx = det->ctx;
  if (det->cap_hard) return fl_set_error(ctx, FL_ERR_OVERFLOW, "more than %d candidates in one frame (hard cap asked for at fl_detector_finalize)", det->cap);
  long long want = det->cap;
  while (want < needed || want <= det->cap) want <<= 1;
  if (want > (1ll << 28)) return fl_set_error(ctx, FL_ERR_OVERFLOW, "%d candidates in one frame", needed);
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int old_cap = det->cap;
  // the frame workspaces (and with them any depth frames gathered into them) are about to be freed: nothing queued
  // before this point may be refined afterwards
  det->last_refinable = false;
  det->last_depth_base = nullptr;
  det->last_depth_stride = 0;
  det->last_batch = 0;
  (void)hipFree(det->d_ws);
  det->d_ws = nullptr;
  int rc = layout_workspace(det, (int)want);
  if (rc) {                                              // not enough memory: back to the old size, report the overflow
    if (layout_workspace(det, old_cap)) det->finalized = false;
    return fl_set_error(ctx, FL_ERR_OVERFLOW, "%d candidates in one frame and no memory for %d frames of that capacity", needed, det->max_batch);
  }
  return FL_OK;
}

extern "C" int fl_detector_finalize(fl_detector *det, int w0, int h0, int max_batch, int max_candidates)
{
  if (!det || w0 <= 0 || h0 <= 0 || max_batch <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  FL_HIP(ctx, hipSetDevice(ctx->device));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_device_tables(det);
  det->finalized = false;
  const int L = det->L, M = det->M;
  // max_candidates > 0: initial per-frame capacity of the candidate / match buffers, grown on demand (the reference's
  // vectors are unbounded, linemod.cpp:1490-1504); < 0: a hard cap of -max_candidates (FL_ERR_OVERFLOW beyond it);
  // 0: 65536, grown on demand
  int cap = max_candidates > 0 ? max_candidates : (max_candidates < 0 ? -max_candidates : 65536);
  det->cap_hard = max_candidates < 0;
  int cap2 = 1;
  while (cap2 < cap) cap2 <<= 1;
  cap = cap2;                                        // power of two (bitonic sort)

  // geometry per level; reference asserts (linemod.cpp:981, 1062-1063)
  for (int l = 0; l < L; ++l) {
    FlLevelGeom &g = det->geom[l];
    g.w = w0 >> l;
    g.h = h0 >> l;
    g.T = det->T[l];
    if (g.w <= 0 || g.h <= 0) return fl_set_error(ctx, FL_ERR_INVALID, "level %d is empty", l);
    if (g.w % g.T || g.h % g.T)
      return fl_set_error(ctx, FL_ERR_ASSERT, "level %d: %dx%d not divisible by T=%d (CV_Assert linemod.cpp:1062)", l, g.w,
                          g.h, g.T);
    if ((g.w * g.h) % 16) return fl_set_error(ctx, FL_ERR_ASSERT, "level %d: rows*cols %% 16 != 0 (linemod.cpp:981)", l);
    g.W = g.w / g.T;
    g.H = g.h / g.T;
    g.WH = g.W * g.H;
    size_t s = fl_lm_label_stride(g.w, g.h, g.T);
    if (8 * s > 0xFFFFFFFFull) return fl_set_error(ctx, FL_ERR_INVALID, "linear memory too large");
    g.stride = (uint32_t)s;
    g.zero_off = (uint32_t)(7 * s + (size_t)g.T * g.T * g.WH);   // pad of label 7
  }

  // flatten the bank
  std::vector<FlScanHdr> scan_hdr;
  std::vector<uint32_t> scan_off;
  std::vector<FlFineHdr> fine_hdr;
  std::vector<FlFineFeat> fine_feat;
  std::vector<FlPyrInfo> pyr;
  std::vector<float> poses;
  std::vector<const uint16_t *> depth_ptrs;
  std::vector<int> class_first;
  int g_idx = 0;
  double scan_bytes = 0.0;
  det->max_tw = det->max_th = 1;
  det->depth_w = det->depth_h = 0;
  for (size_t ci = 0; ci < det->classes.size(); ++ci) {
    FlClass &c = det->classes[ci];
    c.first_g = g_idx;
    class_first.push_back(g_idx);
    if (c.d_depths) {
      if (c.dw != w0 || c.dh != h0)
        return fl_set_error(ctx, FL_ERR_INVALID, "model depth renders must be %dx%d like the frames", w0, h0);
      det->depth_w = c.dw;
      det->depth_h = c.dh;
    }
    for (int p = 0; p < c.n_pyramids; ++p, ++g_idx) {
      const fl_template *tp = &c.templates[(size_t)p * L * M];
      FlPyrInfo pi;
      pi.class_idx = (int)ci;
      pi.template_id = p;
      pi.off_x0 = tp[0].offset_x;
      pi.off_y0 = tp[0].offset_y;
      pi.width0 = tp[0].width;
      pi.height0 = tp[0].height;
      pi.depth_slot = c.d_depths ? p : -1;
      pi.pad = 0;
      pyr.push_back(pi);
      det->max_tw = std::max(det->max_tw, std::min(tp[0].width, w0));
      det->max_th = std::max(det->max_th, std::min(tp[0].height, h0));
      for (int k = 0; k < 13; ++k) poses.push_back(c.poses.empty() ? 0.f : c.poses[(size_t)p * 13 + k]);
      depth_ptrs.push_back(c.d_depths ? c.d_depths + (size_t)p * c.dw * c.dh : nullptr);
      // coarsest level: scan tables (similarity(), linemod.cpp:1130-1214)
      {
        const FlLevelGeom &g = det->geom[L - 1];
        for (int m = 0; m < M; ++m) {
          const fl_template &t = tp[(L - 1) * M + m];
          FlScanHdr h;
          int wf = (t.width - 1) / g.T + 1, hf = (t.height - 1) / g.T + 1;
          int span_x = g.W - wf, span_y = g.H - hf;
          h.P = span_y * g.W + span_x + 1;
          if (h.P > g.WH) h.P = g.WH;              // dst has only W*H cells (width/height <= 0 input)
          h.off_begin = (int)scan_off.size();
          h.nf = t.feat_count;
          int n = 0;
          for (int k = 0; k < t.feat_count; ++k) {
            const fl_feature &f = c.features[t.feat_begin + k];
            if (f.x < 0 || f.x >= g.w || f.y < 0 || f.y >= g.h) continue;        // :1179
            uint32_t off = (uint32_t)f.label * g.stride + (uint32_t)((f.y % g.T) * g.T + (f.x % g.T)) * g.WH +
                           (uint32_t)((f.y / g.T) * g.W + f.x / g.T);             // accessLinearMemory :1094
            scan_off.push_back(off);
            ++n;
          }
          while (n % 8) { scan_off.push_back(g.zero_off); ++n; }
          h.n_pad = n;
          scan_hdr.push_back(h);
          // B_tmpl = sum_m nf_m*P + M*P + 2*P (bytes the reference touches)
          const double Pd = h.P > 0 ? (double)h.P : 0.0;
          scan_bytes += (double)t.feat_count * Pd + Pd + (m == 0 ? 2.0 * Pd : 0.0);
        }
      }
      // finer levels: refinement tables (similarityLocal(), linemod.cpp:1226-1300)
      for (int l = 0; l < L - 1; ++l) {
        for (int m = 0; m < M; ++m) {
          const fl_template &t = tp[l * M + m];
          FlFineHdr h;
          h.feat_begin = (int)fine_feat.size();
          h.feat_count = t.feat_count;
          h.width = t.width;
          h.height = t.height;
          for (int k = 0; k < t.feat_count; ++k) {
            const fl_feature &f = c.features[t.feat_begin + k];
            FlFineFeat ff;
            ff.x = (int16_t)f.x;
            ff.y = (int16_t)f.y;
            {
              const int Tl = det->geom[l].T;
              const int rx = ((f.x % Tl) + Tl) % Tl, ry = ((f.y % Tl) + Tl) % Tl;
              ff.gx = (uint8_t)rx;
              ff.gy = (uint8_t)ry;
              ff.qx = (int16_t)((f.x - rx) / Tl);
              ff.qy = (int16_t)((f.y - ry) / Tl);
              ff.label = (uint8_t)f.label;
              ff.pad = 0;
            }
            fine_feat.push_back(ff);
          }
          fine_hdr.push_back(h);
        }
      }
    }
  }
  det->n_pyr = g_idx;
  det->scan_bytes_per_frame = scan_bytes;

  int rc;
  if ((rc = upload(ctx, scan_hdr, &det->d_scan_hdr))) return rc;
  {
    // k_scan's work list: one wave per (pyramid, chunk of 1024 coarse positions) -- only the chunks some modality's
    // template_positions reach (a 160-pixel template at VGA level 1 has P = 831 of 1200 cells: its second chunk is empty)
    std::vector<int2> items;
    const int nchunks = (det->geom[L - 1].WH + 1023) / 1024;
    for (int g = 0; g < g_idx; ++g) {
      int pmax = 0;
      for (int m = 0; m < M; ++m) pmax = std::max(pmax, (int)scan_hdr[(size_t)g * M + m].P);
      for (int c = 0; c < nchunks && (c == 0 || c * 1024 < pmax); ++c) items.push_back(make_int2(g, c));   // chunk 0 always: the debug tap zero-fills through it
    }
    det->n_scan_items = (int)items.size();
    if ((rc = upload(ctx, items, &det->d_scan_items))) return rc;
  }
  if ((rc = upload(ctx, scan_off, &det->d_scan_off))) return rc;
  if ((rc = upload(ctx, fine_hdr, &det->d_fine_hdr))) return rc;
  if ((rc = upload(ctx, fine_feat, &det->d_fine_feat))) return rc;
  if ((rc = upload(ctx, pyr, &det->d_pyr))) return rc;
  if ((rc = upload(ctx, poses, &det->d_poses))) return rc;
  if ((rc = upload(ctx, class_first, &det->d_class_first))) return rc;
  if ((rc = fl_apply_class_filter(det))) return rc;
  {
    const uint16_t **tmp = nullptr;
    if ((rc = upload(ctx, depth_ptrs, (const uint16_t ***)&tmp))) return rc;
    det->d_depth_ptrs = tmp;
  }

  det->w0 = w0;
  det->h0 = h0;
  det->max_batch = max_batch;
  det->eager_env = ctx->opt.eager_frontend != 0;          // development switches (fl_context_set_option), sampled here
  det->poison_env = ctx->opt.dev_poison != 0;
  if ((rc = layout_workspace(det, cap))) return rc;
  if ((rc = fl_icp_prepare(det))) return rc;
  FL_HIP(ctx, hipMalloc((void **)&det->d_results, sizeof(fl_recognition_result) * (size_t)max_batch));
  FL_HIP(ctx, hipHostMalloc((void **)&det->h_results, sizeof(fl_recognition_result) * (size_t)max_batch,
                            hipHostMallocDefault));
  for (auto &e : det->ev) FL_HIP(ctx, hipEventCreate(&e));
  det->finalized = true;
  det->last_batch = 0;
  det->last_refinable = false;
  det->last_depth_base = nullptr;
  det->last_depth_stride = 0;
  return FL_OK;
}
