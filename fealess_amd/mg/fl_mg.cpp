// fl_mg.cpp -- libfealess_mg.so: template-sharded recognition across GPUs, host side in C++ on RCCL (include/fealess_mg.h).
// Everything between Detector::match and the pose is device work on the context's stream; the two collectives are
// queued on that same stream (they are kilobytes: latency-bound, nothing to overlap inside one batch).
#include "../../include/fealess_mg.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

struct fl_mg {
  fl_detector *det = nullptr;
  fl_context *ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int n_ranks = 0, rank = 0, tid_first = 0, tid_count = 0, k = 0, device = 0;
  // device buffers, grown on demand to the largest n_frames seen
  int cap_frames = 0;
  void *d_local = nullptr, *d_gathered = nullptr, *d_best = nullptr;
  float *d_rows = nullptr;
  int *d_flag = nullptr;
  fl_match *h_best = nullptr;     // pinned
  float *h_rows = nullptr;
  int *h_flag = nullptr;
  int attempts = 0;
  size_t ag_bytes = 0, ar_bytes = 0;
  char err[512] = {0};
};

static int mg_fail(fl_mg *mg, int code, const char *fmt, ...)
{
  if (mg) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mg->err, sizeof(mg->err), fmt, ap);
    va_end(ap);
  }
  return code;
}
#define MG_HIP(mg, call)                                                                                           \
  do {                                                                                                             \
    hipError_t e_ = (call);                                                                                        \
    if (e_ != hipSuccess) return mg_fail((mg), FL_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)
#define MG_NCCL(mg, call)                                                                                          \
  do {                                                                                                             \
    ncclResult_t r_ = (call);                                                                                      \
    if (r_ != ncclSuccess) return mg_fail((mg), FL_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
  } while (0)
#define MG_FL(mg, call)                                                                                            \
  do {                                                                                                             \
    int rc_ = (call);                                                                                              \
    if (rc_ != FL_OK) return mg_fail((mg), rc_, "%s -> %d: %s", #call, rc_, fl_last_error((mg)->ctx));             \
  } while (0)

extern "C" int fl_mg_unique_id(void *id_out, size_t bytes)
{
  static_assert(sizeof(ncclUniqueId) == FL_MG_ID_BYTES, "FL_MG_ID_BYTES");
  if (!id_out || bytes < sizeof(ncclUniqueId)) return FL_ERR_INVALID;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return FL_ERR_HIP;
  memcpy(id_out, &id, sizeof(id));
  return FL_OK;
}

extern "C" const char *fl_mg_last_error(const fl_mg *mg) { return mg ? mg->err : "null group"; }

static void mg_free_buffers(fl_mg *mg)
{
  if (mg->d_local) (void)hipFree(mg->d_local);
  if (mg->d_gathered) (void)hipFree(mg->d_gathered);
  if (mg->d_best) (void)hipFree(mg->d_best);
  if (mg->d_rows) (void)hipFree(mg->d_rows);
  if (mg->h_best) (void)hipHostFree(mg->h_best);
  if (mg->h_rows) (void)hipHostFree(mg->h_rows);
  mg->d_local = mg->d_gathered = mg->d_best = nullptr;
  mg->d_rows = nullptr;
  mg->h_best = nullptr;
  mg->h_rows = nullptr;
  mg->cap_frames = 0;
}

extern "C" void fl_mg_destroy(fl_mg *mg)
{
  if (!mg) return;
  (void)hipSetDevice(mg->device);
  if (mg->stream) (void)hipStreamSynchronize(mg->stream);
  mg_free_buffers(mg);
  if (mg->d_flag) (void)hipFree(mg->d_flag);
  if (mg->h_flag) (void)hipHostFree(mg->h_flag);
  if (mg->comm) (void)ncclCommDestroy(mg->comm);
  delete mg;
}

extern "C" int fl_mg_create(fl_detector *det, const void *id, int n_ranks, int rank, int tid_first, int tid_count, int k, fl_mg **out)
{
  if (!out) return FL_ERR_INVALID;
  *out = nullptr;
  if (!det || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks || tid_first < 0 || tid_count < 0 || k < 1) return FL_ERR_INVALID;
  fl_context *ctx = fl_detector_get_context(det);
  if (!ctx) return FL_ERR_INVALID;
  if (fl_detector_num_templates(det) != tid_count) return FL_ERR_INVALID;      // the slice IS the detector's bank
  fl_mg *mg = new fl_mg();
  mg->det = det;
  mg->ctx = ctx;
  mg->n_ranks = n_ranks;
  mg->rank = rank;
  mg->tid_first = tid_first;
  mg->tid_count = tid_count;
  mg->k = k;
  mg->device = fl_context_get_device(ctx);
  mg->stream = (hipStream_t)fl_context_get_stream(ctx);
  ncclUniqueId nid;
  memcpy(&nid, id, sizeof(nid));
  int rc = FL_OK;
  if (hipSetDevice(mg->device) != hipSuccess) rc = FL_ERR_HIP;
  if (rc == FL_OK && ncclCommInitRank(&mg->comm, n_ranks, nid, rank) != ncclSuccess) { mg->comm = nullptr; rc = FL_ERR_HIP; }
  if (rc == FL_OK && (hipMalloc((void **)&mg->d_flag, sizeof(int)) != hipSuccess ||
                      hipHostMalloc((void **)&mg->h_flag, sizeof(int), hipHostMallocDefault) != hipSuccess))
    rc = FL_ERR_HIP;
  if (rc != FL_OK) { fl_mg_destroy(mg); return rc; }
  *out = mg;
  return FL_OK;
}

static int mg_reserve(fl_mg *mg, int n_frames)
{
  if (n_frames <= mg->cap_frames) return FL_OK;
  MG_HIP(mg, hipStreamSynchronize(mg->stream));
  mg_free_buffers(mg);
  const size_t rec = sizeof(fl_match) * (size_t)n_frames * (size_t)mg->k;
  MG_HIP(mg, hipMalloc(&mg->d_local, rec));
  MG_HIP(mg, hipMalloc(&mg->d_gathered, rec * (size_t)mg->n_ranks));
  MG_HIP(mg, hipMalloc(&mg->d_best, sizeof(fl_match) * (size_t)n_frames));
  MG_HIP(mg, hipMalloc((void **)&mg->d_rows, sizeof(float) * 17 * (size_t)n_frames));
  MG_HIP(mg, hipHostMalloc((void **)&mg->h_best, sizeof(fl_match) * (size_t)n_frames, hipHostMallocDefault));
  MG_HIP(mg, hipHostMalloc((void **)&mg->h_rows, sizeof(float) * 17 * (size_t)n_frames, hipHostMallocDefault));
  mg->cap_frames = n_frames;
  return FL_OK;
}

extern "C" int fl_mg_recognize_batch(fl_mg *mg, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth, int mem,
                                     const fl_intrinsics *K, const fl_recognition_params *params, fl_mg_result *results)
{
  if (!mg || !bgr || !depth || !K || !params || !results || n_frames <= 0) return FL_ERR_INVALID;
  MG_HIP(mg, hipSetDevice(mg->device));
  mg->stream = (hipStream_t)fl_context_get_stream(mg->ctx);               // the caller may have set another one since
  int rc = mg_reserve(mg, n_frames);
  if (rc) return rc;
  const size_t rec = sizeof(fl_match) * (size_t)n_frames * (size_t)mg->k;
  mg->ag_bytes = rec * (size_t)mg->n_ranks;
  mg->ar_bytes = sizeof(float) * 17 * (size_t)n_frames;
  bool overflow = false;
  for (mg->attempts = 1;; ++mg->attempts) {
    // Detector::match of this rank's slice, k records per frame with global template ids (linemod.cpp:1356-1441 per slice)
    MG_FL(mg, fl_match_batch_submit(mg->det, n_frames, bgr, depth, mem, params->matching_threshold));
    MG_FL(mg, fl_export_topk_batch(mg->det, n_frames, mg->k, mg->tid_first, mg->d_local));
    // [rank][frame][k] records on every rank
    MG_NCCL(mg, ncclAllGather(mg->d_local, mg->d_gathered, rec, ncclUint8, mg->comm, mg->stream));
    // matches[0] of one global std::sort + std::unique (linemod.cpp:1437-1439), the owner refines it (obj_reco_lmicp.cpp:111-197)
    MG_FL(mg, fl_select_best_batch(mg->det, mg->d_gathered, mg->n_ranks, n_frames, mg->k, mg->tid_first, mg->tid_count, mg->d_best));
    MG_FL(mg, fl_refine_selected(mg->det, n_frames, K, params, nullptr, 0, mg->d_rows));
    // one owner per frame, zero rows elsewhere: the int32 sum of the bit patterns is the owner's row, bit for bit
    MG_NCCL(mg, ncclAllReduce(mg->d_rows, mg->d_rows, (size_t)17 * (size_t)n_frames, ncclInt32, ncclSum, mg->comm, mg->stream));
    MG_HIP(mg, hipMemcpyAsync(mg->h_best, mg->d_best, sizeof(fl_match) * (size_t)n_frames, hipMemcpyDeviceToHost, mg->stream));
    MG_HIP(mg, hipMemcpyAsync(mg->h_rows, mg->d_rows, sizeof(float) * 17 * (size_t)n_frames, hipMemcpyDeviceToHost, mg->stream));
    MG_HIP(mg, hipStreamSynchronize(mg->stream));
    overflow = false;                                                      // the flag is in the gathered records: every rank sees the same
    for (int f = 0; f < n_frames; ++f) overflow = overflow || mg->h_best[f].template_id == FL_TOPK_OVERFLOW;
    if (!overflow || mg->attempts > 6) break;
    // every rank grows; whether that worked is made collective, so that a rank under a hard cap / out of memory does not
    // leave the others waiting in the next all-gather
    const int grown = fl_detector_grow_candidates(mg->det, n_frames, nullptr);
    *mg->h_flag = grown == FL_OK ? 0 : 1;
    MG_HIP(mg, hipMemcpyAsync(mg->d_flag, mg->h_flag, sizeof(int), hipMemcpyHostToDevice, mg->stream));
    MG_NCCL(mg, ncclAllReduce(mg->d_flag, mg->d_flag, 1, ncclInt32, ncclSum, mg->comm, mg->stream));
    MG_HIP(mg, hipMemcpyAsync(mg->h_flag, mg->d_flag, sizeof(int), hipMemcpyDeviceToHost, mg->stream));
    MG_HIP(mg, hipStreamSynchronize(mg->stream));
    if (*mg->h_flag != 0) break;                                           // some rank could not grow: the frames stay flagged
  }
  for (int f = 0; f < n_frames; ++f) {
    fl_mg_result &r = results[f];
    memset(&r, 0, sizeof(r));
    r.best = mg->h_best[f];
    const float *row = mg->h_rows + (size_t)17 * f;
    if (r.best.template_id == FL_TOPK_OVERFLOW) {
      r.status = FL_ERR_OVERFLOW;
      r.best.template_id = -1;
      continue;
    }
    r.status = FL_OK;
    r.found = row[0] == 1.0f ? 1 : 0;
    if (r.found) memcpy(r.pose, row + 1, sizeof(float) * 16);
  }
  return FL_OK;
}

extern "C" int fl_mg_last_stats(const fl_mg *mg, int32_t *attempts, size_t *allgather_bytes, size_t *allreduce_bytes)
{
  if (!mg) return FL_ERR_INVALID;
  if (attempts) *attempts = mg->attempts;
  if (allgather_bytes) *allgather_bytes = mg->ag_bytes;
  if (allreduce_bytes) *allreduce_bytes = mg->ar_bytes;
  return FL_OK;
}
