// fl_pipeline.hip -- CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:86-204) for a batch
// of frames: quantise -> linear memories -> scan -> refine -> sort/unique -> (device-side
// hand-over of matches[0]) -> crop back-projection -> ICP -> 4x4 pose, queued back to back on
// one HIP stream with no host synchronisation in between.  Every kernel is launched over the
// whole batch (grid.z / grid.y / one workgroup per frame), so the launch count per batch is
// constant (about a dozen) whatever the number of frames.
#include "fl_internal.h"
#include <cmath>
#include <vector>
#include <string.h>

static bool uniform_stride(const void *const *ptrs, int n, size_t min_bytes, size_t *stride)
{
  if (n == 1) { *stride = 0; return true; }
  const uint8_t *p0 = (const uint8_t *)ptrs[0], *p1 = (const uint8_t *)ptrs[1];
  if (p1 < p0 + min_bytes) return false;
  const size_t s = (size_t)(p1 - p0);
  for (int i = 2; i < n; ++i)
    if ((const uint8_t *)ptrs[i] != p0 + s * (size_t)i) return false;
  *stride = s;
  return true;
}

// argument checks, frame staging, front-end and Detector::match of a batch; the depth frames' device location comes back
// (K == nullptr: Detector::match only, no intrinsics to check)
static int stage_and_match(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth, int mem,
                           const fl_intrinsics *K, float threshold, const uint16_t **depth_base_out,
                           size_t *depth_stride_out, int *host_buf_out)
{
  int host_buf = -1;     // the input buffer the batch was uploaded to (host frames), released by input_done()
  if (!det || !bgr || n_frames <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "fl_detector_finalize first");
  if (n_frames > det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "n_frames %d > max_batch %d", n_frames, det->max_batch);
  if (det->M == 2 && !depth) return fl_set_error(ctx, FL_ERR_INVALID, "depth frames required (2 modalities)");
  // PrepareInputData (obj_reco_lmicp.cpp:216-259): image size must equal the intrinsics' size
  if (K && (K->width != det->w0 || K->height != det->h0))
    return fl_set_error(ctx, FL_ERR_INVALID, "intrinsics are %dx%d, detector finalized for %dx%d", K->width, K->height,
                        det->w0, det->h0);
  for (int i = 0; i < n_frames; ++i)
    if (!bgr[i] || (det->M == 2 && !depth[i])) return fl_set_error(ctx, FL_ERR_INVALID, "frame %d: null pData (CheckTImage)", i);
  FL_HIP(ctx, hipSetDevice(ctx->device));

  const size_t bgr_bytes = (size_t)det->w0 * det->h0 * 3, depth_bytes = (size_t)det->w0 * det->h0 * 2;
  const uint8_t *bgr_base = det->d_ws + det->off_bgr;
  const uint16_t *depth_base = (const uint16_t *)(det->d_ws + det->off_depth);
  size_t bgr_stride = det->ws_stride, depth_stride = det->ws_stride;
  size_t s1 = 0, s2 = 0;
  if (mem == FL_MEM_DEVICE && uniform_stride((const void *const *)bgr, n_frames, bgr_bytes, &s1) &&
      (det->M < 2 || (uniform_stride((const void *const *)depth, n_frames, depth_bytes, &s2) && s2 % 2 == 0))) {
    // frames already in HBM at a regular pitch: read them in place
    bgr_base = bgr[0];
    bgr_stride = s1;
    if (det->M == 2) { depth_base = depth[0]; depth_stride = s2; }
  } else if (mem == FL_MEM_HOST) {
    // host frames: upload on the copy stream into the input buffer that is not being read (see fl_internal.h)
    const int b = det->in_flip;
    det->in_flip ^= 1;
    const size_t depth_off = fl_align(bgr_bytes, 256), frame_in = depth_off + fl_align(depth_bytes, 256);
    if (!det->copy_stream) FL_HIP(ctx, hipStreamCreateWithFlags(&det->copy_stream, hipStreamNonBlocking));
    if (!det->d_in[b]) {
      FL_HIP(ctx, hipMalloc((void **)&det->d_in[b], frame_in * (size_t)det->max_batch));
      FL_HIP(ctx, hipEventCreateWithFlags(&det->ev_up[b], hipEventDisableTiming));
      FL_HIP(ctx, hipEventCreateWithFlags(&det->ev_read[b], hipEventDisableTiming));
    }
    if (det->read_pending[b]) FL_HIP(ctx, hipStreamWaitEvent(det->copy_stream, det->ev_read[b], 0));
    size_t hs1 = 0, hs2 = 0;
    if (n_frames > 1 && uniform_stride((const void *const *)bgr, n_frames, bgr_bytes, &hs1) &&
        (det->M < 2 || uniform_stride((const void *const *)depth, n_frames, depth_bytes, &hs2))) {
      // frames at a regular pitch (one host array): one strided copy per modality instead of one per frame
      FL_HIP(ctx, hipMemcpy2DAsync(det->d_in[b], frame_in, bgr[0], hs1, bgr_bytes, n_frames, hipMemcpyHostToDevice, det->copy_stream));
      if (det->M == 2)
        FL_HIP(ctx, hipMemcpy2DAsync(det->d_in[b] + depth_off, frame_in, depth[0], hs2, depth_bytes, n_frames, hipMemcpyHostToDevice,
                                     det->copy_stream));
    } else {
      for (int i = 0; i < n_frames; ++i) {
        uint8_t *dst = det->d_in[b] + (size_t)i * frame_in;
        FL_HIP(ctx, hipMemcpyAsync(dst, bgr[i], bgr_bytes, hipMemcpyHostToDevice, det->copy_stream));
        if (det->M == 2) FL_HIP(ctx, hipMemcpyAsync(dst + depth_off, depth[i], depth_bytes, hipMemcpyHostToDevice, det->copy_stream));
      }
    }
    FL_HIP(ctx, hipEventRecord(det->ev_up[b], det->copy_stream));
    FL_HIP(ctx, hipStreamWaitEvent(ctx->stream, det->ev_up[b], 0));
    bgr_base = det->d_in[b];
    bgr_stride = frame_in;
    depth_base = (const uint16_t *)(det->d_in[b] + depth_off);
    depth_stride = frame_in;
    host_buf = b;
  } else {
    for (int i = 0; i < n_frames; ++i) {     // device frames at irregular addresses: gather them into the frame workspaces
      uint8_t *ws = det->d_ws + (size_t)i * det->ws_stride;
      FL_HIP(ctx, hipMemcpyAsync(ws + det->off_bgr, bgr[i], bgr_bytes, hipMemcpyDeviceToDevice, ctx->stream));
      if (det->M == 2) FL_HIP(ctx, hipMemcpyAsync(ws + det->off_depth, depth[i], depth_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  det->have_times = true;
  FL_HIP(ctx, hipEventRecord(det->ev[0], ctx->stream));
  int rc = fl_launch_frontend(det, n_frames, bgr_base, bgr_stride, depth_base, depth_stride, true);
  if (rc) return rc;
  rc = fl_launch_match_core(det, n_frames, threshold);
  if (rc) return rc;
  *depth_base_out = depth_base;
  *depth_stride_out = depth_stride;
  det->last_depth_base = depth_base;
  det->last_depth_stride = depth_stride;
  det->last_refinable = depth != nullptr;  // fl_refine_matches runs the ICP half: it needs the batch's depth frames
  *host_buf_out = host_buf;
  return FL_OK;
}

// per-stage device times of the batch that was just synchronised (HIP events recorded on the launch stream)
void fl_update_stage_times(fl_detector *det, int n_frames, const fl_recognition_result *results)
{
  if (!det->have_times) return;
  fl_stage_times &t = det->times;
  memset(&t, 0, sizeof(t));
  float ms = 0;
  auto el = [&](int a, int b) { ms = 0; (void)hipEventElapsedTime(&ms, det->ev[a], det->ev[b]); return ms; };
  t.frontend_ms = el(0, 1);
  t.linmem_ms = el(1, 2);
  t.scan_ms = el(2, 3);
  t.refine_ms = el(3, 4);
  if (det->lazy) {
    for (int l = 0; l + 1 < det->L; ++l) t.lazy_frontend_ms += el(8 + 2 * l, 9 + 2 * l);
    t.refine_ms -= t.lazy_frontend_ms;
  }
  t.sort_ms = el(4, 5);
  t.icp_ms = det->last_match_only ? 0.f : el(5, 6);
  t.total_ms = el(0, 6);
  t.backproject_ms = 0;                 // fused into the per-frame ICP workgroup
  t.icp_launches = det->last_match_only ? 0 : 1;
  if (results)
    for (int i = 0; i < n_frames; ++i) t.icp_iters_total += results[i].found ? results[i].det.icp.iters : 0;
  t.scan_algorithmic_bytes = det->scan_bytes_per_frame * n_frames;
  det->have_times = false;
}

// after the last kernel that reads the batch's frames has been queued: the copy stream may reuse the buffer after it
static int input_done(fl_detector *det, int host_buf)
{
  if (host_buf < 0) return FL_OK;
  fl_context *ctx = det->ctx;
  FL_HIP(ctx, hipEventRecord(det->ev_read[host_buf], ctx->stream));
  det->read_pending[host_buf] = true;
  return FL_OK;
}

extern "C" int fl_recognize_submit(fl_detector *det, int n_frames, const uint8_t *const *bgr,
                                   const uint16_t *const *depth, int mem, const fl_intrinsics *K,
                                   const fl_recognition_params *params)
{
  const uint16_t *depth_base = nullptr;
  size_t depth_stride = 0;
  int host_buf = -1;
  if (!K || !params) return FL_ERR_INVALID;
  int rc = stage_and_match(det, n_frames, bgr, depth, mem, K, params->matching_threshold, &depth_base, &depth_stride, &host_buf);
  if (rc) return rc;
  fl_context *ctx = det->ctx;
  FL_HIP(ctx, hipMemsetAsync(det->d_results, 0, sizeof(fl_recognition_result) * (size_t)n_frames, ctx->stream));
  rc = fl_launch_detection_batch(det, n_frames, K, params, depth_base, depth_stride);
  if (rc) return rc;
  if ((rc = input_done(det, host_buf))) return rc;
  FL_HIP(ctx, hipEventRecord(det->ev[6], ctx->stream));
  FL_HIP(ctx, hipMemcpyAsync(det->h_results, det->d_results, sizeof(fl_recognition_result) * (size_t)n_frames,
                             hipMemcpyDeviceToHost, ctx->stream));
  det->last_batch = n_frames;
  det->last_from_images = true;
  det->last_match_only = false;
  return FL_OK;
}

// Detector::match (linemod.cpp:1356-1441) for a batch of frames: the reference is called once per camera frame; here
// front-end and match of n_frames frames are queued together and the sorted match lists stay in HBM until
// fl_match_batch_collect / fl_export_topk reads them.
extern "C" int fl_match_batch_submit(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                                     int mem, float threshold)
{
  const uint16_t *depth_base = nullptr;
  size_t depth_stride = 0;
  int host_buf = -1;
  int rc = stage_and_match(det, n_frames, bgr, depth, mem, nullptr, threshold, &depth_base, &depth_stride, &host_buf);
  if (rc) return rc;
  fl_context *ctx = det->ctx;
  if ((rc = input_done(det, host_buf))) return rc;
  FL_HIP(ctx, hipEventRecord(det->ev[6], ctx->stream));
  det->last_batch = n_frames;
  det->last_from_images = true;
  det->last_match_only = true;
  return FL_OK;
}

extern "C" int fl_recognize_collect(fl_detector *det, int n_frames, fl_recognition_result *results)
{
  if (!det || !results || n_frames <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || n_frames > det->last_batch) return fl_set_error(ctx, FL_ERR_STATE, "nothing submitted");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(results, det->h_results, sizeof(fl_recognition_result) * (size_t)n_frames);
  fl_update_stage_times(det, n_frames, results);
  return FL_OK;
}

extern "C" int fl_recognize_batch(fl_detector *det, int n_frames, const uint8_t *const *bgr,
                                  const uint16_t *const *depth, int mem, const fl_intrinsics *K,
                                  const fl_recognition_params *params, fl_recognition_result *results)
{
  // a frame with more coarse candidates than the buffers hold reports FL_ERR_OVERFLOW in its status: grow the buffers to
  // what it needs and run the batch again, so that no valid input of the reference turns into an error here
  for (int attempt = 0;; ++attempt) {
    int rc = fl_recognize_submit(det, n_frames, bgr, depth, mem, K, params);
    if (rc) return rc;
    if ((rc = fl_recognize_collect(det, n_frames, results))) return rc;
    bool over = false;
    for (int i = 0; i < n_frames; ++i) over = over || results[i].status == FL_ERR_OVERFLOW;
    if (!over || attempt >= 6) return FL_OK;
    int needed = 0;
    if (fl_overflow_needed(det, n_frames, &needed) != FL_OK || needed <= 0) return FL_OK;
    if (fl_grow_candidates(det, needed) != FL_OK) return FL_OK;      // hard cap / no memory: the per-frame statuses say so
  }
}

// PrepareInputData's zoom (obj_reco_lmicp.cpp:229-249: TImage2Mat(..., true) = cv::resize INTER_LINEAR of both images to
// width 640) followed by Recognition, for a batch, with the zoomed frames staying in HBM: the sw x sh sources (host or
// device) are resized on the device into a detector-owned buffer and recognised from there.
extern "C" int fl_recognize_batch_zoom(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                                       int src_w, int src_h, int mem, const fl_intrinsics *K, const fl_recognition_params *params,
                                       fl_recognition_result *results)
{
  if (!det || !bgr || !depth || !K || !params || !results || n_frames <= 0 || src_w <= 0 || src_h <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "fl_detector_finalize first");
  if (n_frames > det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "n_frames %d > max_batch %d", n_frames, det->max_batch);
  for (int i = 0; i < n_frames; ++i)
    if (!bgr[i] || !depth[i]) return fl_set_error(ctx, FL_ERR_INVALID, "frame %d: null pData (CheckTImage)", i);
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const int w = det->w0, h = det->h0;
  const size_t zb = fl_align((size_t)w * h * 3, 256), zf = zb + fl_align((size_t)w * h * 2, 256);
  if (!det->d_zoom) FL_HIP(ctx, hipMalloc((void **)&det->d_zoom, zf * (size_t)det->max_batch));
  const size_t sb = fl_align((size_t)src_w * src_h * 3, 256), sf = sb + fl_align((size_t)src_w * src_h * 2, 256);
  if (mem == FL_MEM_HOST && det->zoom_src_bytes < sf * (size_t)n_frames) {
    FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (det->d_zoom_src) (void)hipFree(det->d_zoom_src);
    det->d_zoom_src = nullptr;
    det->zoom_src_bytes = 0;
    FL_HIP(ctx, hipMalloc((void **)&det->d_zoom_src, sf * (size_t)n_frames));
    det->zoom_src_bytes = sf * (size_t)n_frames;
  }
  std::vector<const uint8_t *> zbp(n_frames);
  std::vector<const uint16_t *> zdp(n_frames);
  for (int i = 0; i < n_frames; ++i) {
    const uint8_t *sbgr = bgr[i];
    const uint16_t *sdep = depth[i];
    if (mem == FL_MEM_HOST) {
      uint8_t *d = det->d_zoom_src + sf * (size_t)i;
      FL_HIP(ctx, hipMemcpyAsync(d, bgr[i], (size_t)src_w * src_h * 3, hipMemcpyHostToDevice, ctx->stream));
      FL_HIP(ctx, hipMemcpyAsync(d + sb, depth[i], (size_t)src_w * src_h * 2, hipMemcpyHostToDevice, ctx->stream));
      sbgr = d;
      sdep = (const uint16_t *)(d + sb);
    }
    uint8_t *z = det->d_zoom + zf * (size_t)i;
    int rc = fl_launch_resize_linear_bgr8(ctx, sbgr, src_w, src_h, z, w, h);
    if (rc == FL_OK) rc = fl_launch_resize_linear_u16(ctx, sdep, src_w, src_h, (uint16_t *)(z + zb), w, h);
    if (rc) return rc;
    zbp[i] = z;
    zdp[i] = (const uint16_t *)(z + zb);
  }
  return fl_recognize_batch(det, n_frames, zbp.data(), zdp.data(), FL_MEM_DEVICE, K, params, results);
}

// The refinement half of Recognition() (obj_reco_lmicp.cpp:111-197) for matches the CALLER chose, on frames of the batch
// last queued with fl_match_batch_submit: template-sharded recognition merges the ranks' top-k lists and then asks the rank
// that owns the winning template to refine it.  matches[j].template_id is class-local on this detector.
extern "C" int fl_refine_matches(fl_detector *det, int n_jobs, const int32_t *frames, const fl_match *matches, const fl_intrinsics *K,
                                 const fl_recognition_params *params, fl_recognition_result *results)
{
  if (!det || !frames || !matches || !K || !params || !results || n_jobs <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  // only a batch submit leaves the depth frames of the batch where last_depth_base says; fl_match_frame*, fl_match_quantized,
  // fl_recognize_topk and a candidate-buffer growth (which frees the frame workspaces) all clear last_refinable
  if (!det->finalized || det->last_batch < 1 || !det->last_from_images || !det->last_refinable || !det->last_depth_base)
    return fl_set_error(ctx, FL_ERR_STATE, "fl_match_batch_submit first");
  if (n_jobs > det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "n_jobs %d > max_batch %d", n_jobs, det->max_batch);
  if (K->width != det->w0 || K->height != det->h0) return fl_set_error(ctx, FL_ERR_INVALID, "intrinsics size");
  std::vector<FlRefineJob> jobs((size_t)n_jobs);
  for (int j = 0; j < n_jobs; ++j) {
    if (frames[j] < 0 || frames[j] >= det->last_batch) return fl_set_error(ctx, FL_ERR_INVALID, "job %d: frame %d is not in the last batch", j, frames[j]);
    const fl_match &m = matches[j];
    if (m.class_idx < 0 || m.class_idx >= (int)det->classes.size() || m.template_id < 0 ||
        m.template_id >= det->classes[m.class_idx].n_pyramids)
      return fl_set_error(ctx, FL_ERR_INVALID, "job %d: template %d of class %d is not on this detector", j, m.template_id, m.class_idx);
    jobs[j].frame = frames[j];
    jobs[j].match = m;
  }
  FL_HIP(ctx, hipSetDevice(ctx->device));
  void *sv = nullptr;
  int rc = fl_scratch(ctx, sizeof(FlRefineJob) * (size_t)n_jobs, &sv);
  if (rc) return rc;
  FL_HIP(ctx, hipMemcpyAsync(sv, jobs.data(), sizeof(FlRefineJob) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
  FL_HIP(ctx, hipMemsetAsync(det->d_results, 0, sizeof(fl_recognition_result) * (size_t)n_jobs, ctx->stream));
  det->have_times = false;
  if ((rc = fl_launch_detection_jobs(det, n_jobs, (const FlRefineJob *)sv, K, params, det->last_depth_base, det->last_depth_stride))) return rc;
  FL_HIP(ctx, hipMemcpyAsync(det->h_results, det->d_results, sizeof(fl_recognition_result) * (size_t)n_jobs, hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));          // also covers the pageable `jobs` upload
  memcpy(results, det->h_results, sizeof(fl_recognition_result) * (size_t)n_jobs);
  return FL_OK;
}

// {found, 4x4 pose} rows of a batch of results, for the ranks' exchange (template-sharded recognition)
__global__ __launch_bounds__(64) void k_pack_pose_rows(const fl_recognition_result *__restrict__ res, int n, float *__restrict__ rows)
{
  const int f = blockIdx.x;
  if (f >= n) return;
  const int t = threadIdx.x;
  if (t == 0) rows[(size_t)f * 17] = res[f].found ? 1.0f : 0.0f;
  else if (t <= 16) rows[(size_t)f * 17 + t] = res[f].found ? res[f].pose[t - 1] : 0.0f;
}

// The device-side twin of fl_refine_matches: the jobs fl_select_best_batch left in det->d_jobs (frame = -1: not ours, the
// workgroup exits), results packed as rows for the exchange; nothing here touches the host.
extern "C" int fl_refine_selected(fl_detector *det, int n_frames, const fl_intrinsics *K, const fl_recognition_params *params,
                                  const uint16_t *depth_base, size_t depth_stride, float *dev_rows)
{
  if (!det || !K || !params || !dev_rows || n_frames <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || !det->d_jobs || det->selected_frames != n_frames)
    return fl_set_error(ctx, FL_ERR_STATE, "fl_select_best_batch for these %d frames first", n_frames);
  if (K->width != det->w0 || K->height != det->h0) return fl_set_error(ctx, FL_ERR_INVALID, "intrinsics size");
  if (!depth_base) {
    if (!det->last_refinable || !det->last_depth_base || n_frames > det->last_batch)
      return fl_set_error(ctx, FL_ERR_STATE, "no depth frames given and fl_match_batch_submit did not leave any");
    depth_base = det->last_depth_base;
    depth_stride = det->last_depth_stride;
  }
  FL_HIP(ctx, hipSetDevice(ctx->device));
  FL_HIP(ctx, hipMemsetAsync(det->d_results, 0, sizeof(fl_recognition_result) * (size_t)n_frames, ctx->stream));
  det->have_times = false;
  int rc = fl_launch_detection_jobs(det, n_frames, det->d_jobs, K, params, depth_base, depth_stride);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pack_pose_rows, dim3(n_frames), dim3(64), 0, ctx->stream, det->d_results, n_frames, dev_rows);
  FL_HIP(ctx, hipGetLastError());
  det->selected_frames = 0;
  return FL_OK;
}

// Multi-hypothesis recognition of one frame + nonMaximumSuppression (ICP/NMS.cpp, obj_data.h)
static int recognize_topk_once(fl_detector *det, const uint8_t *bgr, const uint16_t *depth, int mem, const fl_intrinsics *K,
                               const fl_recognition_params *params, int k, fl_recognition_result *results, int *n_results)
{
  if (!det || !bgr || !depth || !K || !params || !results || !n_results || k < 1 || k > 1024) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "fl_detector_finalize first");
  if (det->M != 2) return fl_set_error(ctx, FL_ERR_INVALID, "needs the colour + depth modalities");
  if (K->width != det->w0 || K->height != det->h0)
    return fl_set_error(ctx, FL_ERR_INVALID, "intrinsics are %dx%d, detector finalized for %dx%d", K->width, K->height, det->w0, det->h0);
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const hipMemcpyKind kind = mem == FL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  FL_HIP(ctx, hipMemcpyAsync(det->d_ws + det->off_bgr, bgr, (size_t)det->w0 * det->h0 * 3, kind, ctx->stream));
  FL_HIP(ctx, hipMemcpyAsync(det->d_ws + det->off_depth, depth, (size_t)det->w0 * det->h0 * 2, kind, ctx->stream));
  det->have_times = false;
  const uint16_t *d_depth = (const uint16_t *)(det->d_ws + det->off_depth);
  int rc = fl_launch_frontend(det, 1, det->d_ws + det->off_bgr, det->ws_stride, d_depth, det->ws_stride, true);
  if (rc) return rc;
  rc = fl_launch_match_core(det, 1, params->matching_threshold);
  if (rc) return rc;
  const size_t icp_bytes = fl_align(fl_icp_ws_bytes(det->n_pts_max), 256) * (size_t)k, res_bytes = sizeof(fl_recognition_result) * (size_t)k;
  void *sv = nullptr;
  if ((rc = fl_scratch(ctx, icp_bytes + fl_align(res_bytes, 256), &sv))) return rc;
  fl_recognition_result *d_res = (fl_recognition_result *)((uint8_t *)sv + icp_bytes);
  FL_HIP(ctx, hipMemsetAsync(d_res, 0, res_bytes, ctx->stream));
  if ((rc = fl_launch_detection_topk(det, 1, k, K, params, d_depth, 0, (uint8_t *)sv, d_res))) return rc;
  FL_HIP(ctx, hipMemcpyAsync(results, d_res, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  det->last_batch = 1;
  det->last_from_images = true;
  det->last_refinable = false;             // the frame lives in workspace 0 only until the next call: not a batch to refine later
  det->last_depth_base = nullptr;
  if (results[0].status == FL_ERR_OVERFLOW) return fl_set_error(ctx, FL_ERR_OVERFLOW, "more than %d candidates in the frame", det->cap);
  const int n = results[0].n_matches < k ? results[0].n_matches : k;
  *n_results = n < 0 ? 0 : n;
  return FL_OK;
}

static bool grow_and_retry(fl_detector *det, int n_frames, int attempt, int *rc)
{
  if (!det || *rc != FL_ERR_OVERFLOW || attempt >= 6) return false;
  int needed = 0;
  if (fl_overflow_needed(det, n_frames, &needed) != FL_OK || needed <= 0) return false;
  const int g = fl_grow_candidates(det, needed);
  if (g != FL_OK) { *rc = g; return false; }
  return true;
}

extern "C" int fl_recognize_topk(fl_detector *det, const uint8_t *bgr, const uint16_t *depth, int mem, const fl_intrinsics *K,
                                 const fl_recognition_params *params, int k, fl_recognition_result *results, int *n_results)
{
  for (int attempt = 0;; ++attempt) {
    int rc = recognize_topk_once(det, bgr, depth, mem, K, params, k, results, n_results);
    if (!grow_and_retry(det, 1, attempt, &rc)) return rc;
  }
}

// The same for a whole batch: n_frames * k ICP workgroups in one launch.  results[f * k + r] is hypothesis r of frame f,
// n_results[f] = min(k, matches of frame f).
static int recognize_batch_topk_once(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                                     int mem, const fl_intrinsics *K, const fl_recognition_params *params, int k,
                                     fl_recognition_result *results, int *n_results)
{
  if (!results || !n_results || k < 1 || k > 1024) return FL_ERR_INVALID;
  if (det && det->M != 2) return fl_set_error(det->ctx, FL_ERR_INVALID, "needs the colour + depth modalities");
  const uint16_t *depth_base = nullptr;
  size_t depth_stride = 0;
  int host_buf = -1;
  if (!K || !params) return FL_ERR_INVALID;
  int rc = stage_and_match(det, n_frames, bgr, depth, mem, K, params->matching_threshold, &depth_base, &depth_stride, &host_buf);
  if (rc) return rc;
  fl_context *ctx = det->ctx;
  det->have_times = false;
  const size_t jobs = (size_t)n_frames * k, ws_one = fl_align(fl_icp_ws_bytes(det->n_pts_max), 256);
  const size_t icp_bytes = ws_one * jobs, res_bytes = sizeof(fl_recognition_result) * jobs;
  if (jobs > (1u << 20) || icp_bytes > ((size_t)96 << 30))
    return fl_set_error(ctx, FL_ERR_INVALID, "%d frames x %d hypotheses need %zu MB of ICP workspaces", n_frames, k, icp_bytes >> 20);
  void *sv = nullptr;
  if ((rc = fl_scratch(ctx, icp_bytes + fl_align(res_bytes, 256), &sv))) return rc;
  fl_recognition_result *d_res = (fl_recognition_result *)((uint8_t *)sv + icp_bytes);
  FL_HIP(ctx, hipMemsetAsync(d_res, 0, res_bytes, ctx->stream));
  if ((rc = fl_launch_detection_topk(det, n_frames, k, K, params, depth_base, depth_stride, (uint8_t *)sv, d_res))) return rc;
  if ((rc = input_done(det, host_buf))) return rc;
  FL_HIP(ctx, hipMemcpyAsync(results, d_res, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  det->last_batch = n_frames;
  det->last_from_images = true;
  for (int f = 0; f < n_frames; ++f) {
    const fl_recognition_result &r0 = results[(size_t)f * k];
    if (r0.status == FL_ERR_OVERFLOW) return fl_set_error(ctx, FL_ERR_OVERFLOW, "frame %d: more than %d candidates", f, det->cap);
    const int n = r0.n_matches < k ? r0.n_matches : k;
    n_results[f] = n < 0 ? 0 : n;
  }
  return FL_OK;
}

extern "C" int fl_recognize_batch_topk(fl_detector *det, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth,
                                       int mem, const fl_intrinsics *K, const fl_recognition_params *params, int k,
                                       fl_recognition_result *results, int *n_results)
{
  for (int attempt = 0;; ++attempt) {
    int rc = recognize_batch_topk_once(det, n_frames, bgr, depth, mem, K, params, k, results, n_results);
    if (!grow_and_retry(det, n_frames, attempt, &rc)) return rc;
  }
}

// nonMaximumSuppression (ICP/NMS.cpp:6-40) over refined hypotheses, in list order.  winners[g] = index of the
// hypothesis that represents group g; returns the number of groups in *n_winners.  Host-only arithmetic.
extern "C" int fl_nms(const fl_recognition_result *objs, int n, float th_obj_dist, int *winners, int *n_winners)
{
  if ((n > 0 && !objs) || !winners || !n_winners || n < 0) return FL_ERR_INVALID;
  std::vector<char> done((size_t)(n > 0 ? n : 1), 0);
  int n_out = 0;
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    int o = i;
    const int size_th = static_cast<int>((float)objs[i].det.n_points * 0.85);
    for (int j = i + 1; j < n; ++j) {
      if (done[j]) continue;
      double s = 0;
      for (int c = 0; c < 3; ++c) { const double d = (double)objs[o].det.T_final[c] - (double)objs[j].det.T_final[c]; s += d * d; }
      if (std::sqrt(s) < th_obj_dist) {
        done[j] = 1;
        if (objs[j].det.n_points > size_th && objs[j].det.icp.dist_mean < objs[o].det.icp.dist_mean) o = j;
      }
    }
    winners[n_out++] = o;
  }
  *n_winners = n_out;
  return FL_OK;
}

extern "C" int fl_last_stage_times(fl_detector *det, fl_stage_times *out)
{
  if (!det || !out) return FL_ERR_INVALID;
  *out = det->times;
  return FL_OK;
}
