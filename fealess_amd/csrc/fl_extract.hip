// fl_extract.hip -- template extraction on the GPU:
//   Detector::addTemplate (linemod/linemod.cpp:1579-1615) for the two default modalities,
//   ColorGradientPyramid::extractTemplate (:461-513), DepthNormalPyramid::extractTemplate (:747-825),
//   QuantizedPyramid::selectScatteredFeatures (:135-164), cropTemplates (:52-96), pyrDown (:434-453, :721-739).
//
// Offline path (one call per training view), so the kernels are simple; what matters is that the selected
// features equal the reference's bit for bit (oracle/extract_oracle.c):
//   k_color_candidates   border mask (mask - erode3x3(mask)), strong-gradient test, 64-bit sort keys
//   k_depth_candidates   eroded (5x5) mask, exact chessboard distance to the nearest pixel of another label
//                        (= cv::distanceTransform(DIST_C, 3) of the per-label images), per-label counts
//   k_depth_keys         score / label_count -> sort keys
//   k_bitonic_step/_local  bitonic sort of the keys (2048-key blocks in LDS, the wider steps in global memory):
//                        (score desc, raster order asc) is exactly what std::stable_sort with Candidate::operator< yields
//   k_select_scattered   the greedy selection, sequential in its result but not in its work: every candidate keeps its
//                        squared distance to the nearest chosen feature, the walk to the next passing candidate is a
//                        workgroup-wide min (1024 threads)
#include "fl_internal.h"
#include <limits.h>
#include <utility>
#include <vector>

namespace {

struct ExtractCounters {
  int n_cand;            // number of candidates
  int label_counts[8];   // depth modality
  int area;              // countNonZero(local_mask)
  int n_out;             // features written (= num_features on success)
};

__device__ __forceinline__ int ex_label(int q)           // getLabel (linemod.cpp:15-30); -1 where the reference throws
{
  return (q != 0 && (q & (q - 1)) == 0 && q < 256) ? (31 - __clz(q)) : -1;
}

__device__ __forceinline__ int ex_min_rect(const uint8_t *m, int w, int h, int x, int y, int r)   // cv::erode, BORDER_REPLICATE
{
  int v = 255;
  for (int dy = -r; dy <= r; ++dy) {
    const int yy = min(max(y + dy, 0), h - 1);
    for (int dx = -r; dx <= r; ++dx) v = min(v, (int)m[(size_t)yy * w + min(max(x + dx, 0), w - 1)]);
  }
  return v;
}

// local_mask of the two extractTemplate()s: iterations = 1 -> border of the mask (mask - erode(mask), :467-468),
// iterations = 2 -> the mask eroded by a 5x5 rectangle (:752-755)
__global__ __launch_bounds__(256) void k_local_mask(const uint8_t *__restrict__ mask, int w, int h, int iterations, int border,
                                                    uint8_t *__restrict__ local)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  const int e = ex_min_rect(mask, w, h, x, y, iterations);
  const int m = mask[i];
  local[i] = (uint8_t)(border ? (m > e ? m - e : 0) : e);
}

__global__ __launch_bounds__(256) void k_color_candidates(const uint8_t *__restrict__ quantized, const float *__restrict__ mag,
                                                          const uint8_t *__restrict__ mask, int w, int h, float thr_sq,
                                                          unsigned long long *__restrict__ keys, ExtractCounters *cnt)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  if (mask && !mask[i]) return;                            // `mask` is the precomputed local_mask here
  const int q = quantized[i];
  const float score = mag[i];
  if (q > 0 && score > thr_sq) {
    const int pos = atomicAdd(&cnt->n_cand, 1);
    // descending score, then raster order: positive floats order like their bit patterns
    keys[pos] = ((unsigned long long)(~__float_as_uint(score)) << 32) | (unsigned)i;
  }
}

// exact L-infinity distance from (x, y) to the nearest pixel whose per-label image is zero, i.e. a pixel outside
// the eroded mask or whose normal does not carry `bit`; 8192 when there is none (OpenCV's capped INIT_DIST0)
__device__ float ex_chessboard(const uint8_t *normal, const uint8_t *mask, int w, int h, int x, int y, int bit)
{
  const int rmax = max(max(x, w - 1 - x), max(y, h - 1 - y));
  for (int r = 1; r <= rmax; ++r) {
    bool zero = false;
    for (int k = -r; k <= r && !zero; ++k) {
      // the four sides of the ring at radius r
      const int xs[4] = {x + k, x + k, x - r, x + r}, ys[4] = {y - r, y + r, y + k, y + k};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int xx = xs[s4], yy = ys[s4];
        if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;          // outside the image: not a source
        const size_t j = (size_t)yy * w + xx;
        if ((mask && !mask[j]) || !(normal[j] & bit)) zero = true;
      }
    }
    if (zero) return (float)r;
  }
  return 8192.0f;
}

__global__ __launch_bounds__(256) void k_depth_candidates(const uint8_t *__restrict__ normal, const uint8_t *__restrict__ mask,
                                                          int w, int h, int extract_threshold, int *__restrict__ raster,
                                                          float *__restrict__ score, ExtractCounters *cnt)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  const bool in_mask = !mask || mask[i] != 0;              // `mask` is the precomputed local_mask (5x5-eroded)
  if (mask && in_mask) atomicAdd(&cnt->area, 1);
  if (!in_mask) return;
  const int q = normal[i];
  if (q == 0 || q == 255) return;                          // background and shadow (:782)
  const int label = ex_label(q);
  if (label < 0) return;
  const float d = ex_chessboard(normal, mask, w, h, x, y, 1 << label);
  if (d >= (float)extract_threshold) {
    const int pos = atomicAdd(&cnt->n_cand, 1);
    raster[pos] = (int)i;
    score[pos] = d;
    atomicAdd(&cnt->label_counts[label], 1);
  }
}

__global__ __launch_bounds__(256) void k_depth_keys(const uint8_t *__restrict__ normal, const int *__restrict__ raster,
                                                    const float *__restrict__ score, const ExtractCounters *cnt,
                                                    unsigned long long *__restrict__ keys)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= cnt->n_cand) return;
  const int i = raster[k];
  const float s = score[k] / (float)cnt->label_counts[ex_label(normal[i])];   // :806-810
  keys[k] = ((unsigned long long)(~__float_as_uint(s)) << 32) | (unsigned)i;
}

__global__ __launch_bounds__(256) void k_pad_keys(unsigned long long *keys, const ExtractCounters *cnt, int n_pow2)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= cnt->n_cand && k < n_pow2) keys[k] = ~0ull;
}

__global__ __launch_bounds__(256) void k_bitonic_step(unsigned long long *keys, int n_pow2, int kk, int j)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pow2) return;
  const int l = i ^ j;
  if (l > i) {
    const unsigned long long a = keys[i], b = keys[l];
    const bool up = (i & kk) == 0;
    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
  }
}

// The bitonic network's steps with partner distance j < SORT_CHUNK stay inside a SORT_CHUNK-key block: one workgroup
// runs them back to back in LDS instead of one launch per step.  first_kk == 2: every stage up to SORT_CHUNK (a full
// sort of each block, direction by the global index); otherwise the tail (j = SORT_CHUNK/2 .. 1) of stage kk.
#define SORT_CHUNK 2048
__global__ __launch_bounds__(SORT_CHUNK / 2) void k_bitonic_local(unsigned long long *keys, int n_pow2, int first_kk, int last_kk)
{
  __shared__ unsigned long long sk[SORT_CHUNK];
  const int chunk = min(n_pow2, SORT_CHUNK), base = blockIdx.x * chunk, t = threadIdx.x;
  for (int e = t; e < chunk; e += SORT_CHUNK / 2) sk[e] = keys[base + e];
  __syncthreads();
  for (int kk = first_kk; kk <= last_kk; kk <<= 1) {
    for (int j = min(kk >> 1, chunk >> 1); j > 0; j >>= 1) {
      if (t < (chunk >> 1)) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const unsigned long long a = sk[i], b = sk[l];
        const bool up = ((base + i) & kk) == 0;
        if ((a > b) == up) { sk[i] = b; sk[l] = a; }
      }
      __syncthreads();
    }
  }
  for (int e = t; e < chunk; e += SORT_CHUNK / 2) keys[base + e] = sk[e];
}

// selectScatteredFeatures (:135-164).  The reference re-tests every candidate against every chosen feature on every
// pass (~candidates x features x passes distance tests, with the distance dropping by one per pass).  Here each
// candidate carries mind2 = its squared distance to the nearest chosen feature so far, so `keep` is one compare,
// an accepted feature is folded into mind2 with one test per candidate, and the sequential walk "next candidate
// at or after i that passes" is a workgroup-wide min.  Same features in the same order.  Thread t owns candidates
// t, t + 1024, ...: xy / mind2 (global scratch) are only ever touched by their owner, so no fences are needed.
#define SEL_BS 1024
__global__ __launch_bounds__(SEL_BS) void k_select_scattered(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ labels_img,
                                                             int w, int num_features, int depth_mode, int total_px,
                                                             ExtractCounters *cnt, fl_feature *__restrict__ out,
                                                             uint32_t *__restrict__ xy, int *__restrict__ mind2)
{
  __shared__ int s_min[SEL_BS / 64];
  const int tid = threadIdx.x;
  const int n = cnt->n_cand;
  if (n < num_features || num_features > 1024) { if (tid == 0) cnt->n_out = -1; return; }   // "We require a certain number of features"
  float distance;
  if (depth_mode) {
    const float area = cnt->area > 0 || depth_mode == 2 ? (float)cnt->area : (float)total_px;
    distance = sqrtf(area) / sqrtf((float)num_features) + 1.5f;                  // :815-817
  } else {
    distance = (float)(n / num_features + 1);                                    // :503-505
  }
  float distance_sq = distance * distance;
  for (int c = tid; c < n; c += SEL_BS) {
    const unsigned raster = (unsigned)(keys[c] & 0xFFFFFFFFull);
    xy[c] = (raster % (unsigned)w) | ((raster / (unsigned)w) << 16);
    mind2[c] = INT_MAX;                                    // no feature chosen yet: every test passes
  }
  int nf = 0, i = 0, fx = 0, fy = 0;
  bool fold = false;
  while (nf < num_features) {
    int mine = INT_MAX;                                    // my first candidate >= i that is far enough from all features
    for (int c = tid; c < n; c += SEL_BS) {
      int m = mind2[c];
      if (fold) {
        const uint32_t u = xy[c];
        const int dx = (int)(u & 0xFFFFu) - fx, dy = (int)(u >> 16) - fy;
        m = min(m, dx * dx + dy * dy);
        mind2[c] = m;
      }
      if (c >= i && (float)m >= distance_sq) mine = min(mine, c);
    }
    fold = false;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) mine = min(mine, __shfl_xor(mine, sft, 64));
    __syncthreads();                                       // s_min of the previous round has been read by everyone
    if ((tid & 63) == 0) s_min[tid >> 6] = mine;
    __syncthreads();
    int p = INT_MAX;
#pragma unroll
    for (int k = 0; k < SEL_BS / 64; ++k) p = min(p, s_min[k]);
    if (p == INT_MAX) {                                    // nobody up to the end of the list: relax and start over
      i = 0;
      distance -= 1.0f;
      distance_sq = distance * distance;
      continue;
    }
    const int raster = (int)(keys[p] & 0xFFFFFFFFull);
    fx = raster % w;
    fy = raster / w;
    if (tid == 0) {
      out[nf].x = fx;
      out[nf].y = fy;
      out[nf].label = ex_label(labels_img[raster]);
    }
    fold = true;
    ++nf;
    i = p + 1;
    if (i == n) {                                          // start over with a relaxed distance
      i = 0;
      distance -= 1.0f;
      distance_sq = distance * distance;
    }
  }
  if (tid == 0) cnt->n_out = nf;
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

// One (level, modality): candidates -> sort -> select.  `feats` (device) receives num_features entries.
static int extract_level(fl_context *ctx, int modality, const uint8_t *q_img, const float *mag, const uint8_t *mask, int w, int h,
                         int num_features, float strong_threshold, int extract_threshold, unsigned long long *keys, int *raster,
                         float *score, uint8_t *local, ExtractCounters *d_cnt, fl_feature *d_feats, int *ok)
{
  FL_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(ExtractCounters), ctx->stream));
  const dim3 grid((w + 63) / 64, (h + 3) / 4), blk(256);
  if (mask) {
    hipLaunchKernelGGL(k_local_mask, grid, blk, 0, ctx->stream, mask, w, h, modality == 0 ? 1 : 2, modality == 0 ? 1 : 0, local);
    mask = local;
  }
  if (modality == 0) {
    hipLaunchKernelGGL(k_color_candidates, grid, blk, 0, ctx->stream, q_img, mag, mask, w, h, strong_threshold * strong_threshold,
                       keys, d_cnt);
  } else {
    hipLaunchKernelGGL(k_depth_candidates, grid, blk, 0, ctx->stream, q_img, mask, w, h, extract_threshold, raster, score, d_cnt);
  }
  FL_HIP(ctx, hipGetLastError());
  ExtractCounters hc;
  FL_HIP(ctx, hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *ok = hc.n_cand >= num_features;
  if (!*ok) return FL_OK;
  if (modality == 1) {
    hipLaunchKernelGGL(k_depth_keys, dim3((hc.n_cand + 255) / 256), blk, 0, ctx->stream, q_img, raster, score, d_cnt, keys);
    FL_HIP(ctx, hipGetLastError());
  }
  const int np2 = next_pow2(hc.n_cand);
  hipLaunchKernelGGL(k_pad_keys, dim3((np2 + 255) / 256), blk, 0, ctx->stream, keys, d_cnt, np2);
  {
    const int chunk = np2 < SORT_CHUNK ? np2 : SORT_CHUNK, nblk = np2 / chunk;
    hipLaunchKernelGGL(k_bitonic_local, dim3(nblk), dim3(SORT_CHUNK / 2), 0, ctx->stream, keys, np2, 2, chunk);
    for (int kk = 2 * SORT_CHUNK; kk <= np2; kk <<= 1) {
      for (int j = kk >> 1; j >= SORT_CHUNK; j >>= 1)
        hipLaunchKernelGGL(k_bitonic_step, dim3((np2 + 255) / 256), blk, 0, ctx->stream, keys, np2, kk, j);
      hipLaunchKernelGGL(k_bitonic_local, dim3(nblk), dim3(SORT_CHUNK / 2), 0, ctx->stream, keys, np2, kk, kk);
    }
  }
  FL_HIP(ctx, hipGetLastError());
  // raster / score are dead once the keys exist: reused as the selection's xy / mind2 scratch
  hipLaunchKernelGGL(k_select_scattered, dim3(1), dim3(SEL_BS), 0, ctx->stream, keys, q_img, w, num_features,
                     modality == 0 ? 0 : (mask ? 2 : 1), w * h, d_cnt, d_feats, (uint32_t *)raster, (int *)score);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

// cropTemplates (:52-96), host side
static void crop_templates(fl_template *t, int n, fl_feature *f, int bb[4])
{
  int min_x = INT_MAX, min_y = INT_MAX, max_x = INT_MIN, max_y = INT_MIN;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < t[i].feat_count; ++j) {
      const int x = f[t[i].feat_begin + j].x << t[i].pyramid_level, y = f[t[i].feat_begin + j].y << t[i].pyramid_level;
      min_x = x < min_x ? x : min_x;
      min_y = y < min_y ? y : min_y;
      max_x = x > max_x ? x : max_x;
      max_y = y > max_y ? y : max_y;
    }
  if (min_x % 2 == 1) --min_x;
  if (min_y % 2 == 1) --min_y;
  for (int i = 0; i < n; ++i) {
    t[i].width = (max_x - min_x) >> t[i].pyramid_level;
    t[i].height = (max_y - min_y) >> t[i].pyramid_level;
    t[i].offset_x = min_x >> t[i].pyramid_level;
    t[i].offset_y = min_y >> t[i].pyramid_level;
    for (int j = 0; j < t[i].feat_count; ++j) {
      f[t[i].feat_begin + j].x -= t[i].offset_x;
      f[t[i].feat_begin + j].y -= t[i].offset_y;
    }
  }
  bb[0] = min_x; bb[1] = min_y; bb[2] = max_x - min_x; bb[3] = max_y - min_y;
}

extern "C" int fl_extract_template_pyramid(fl_context *ctx, const uint8_t *bgr, const uint16_t *depth, const uint8_t *mask, int w0,
                                           int h0, int levels, int mem, fl_template *templates, fl_feature *features, int bb[4])
{
  if (!ctx || !bgr || !depth || !templates || !features || w0 < 16 || h0 < 16 || levels < 1 || levels > FL_MAX_LEVELS)
    return FL_ERR_INVALID;
  if ((w0 >> (levels - 1)) < 8 || (h0 >> (levels - 1)) < 8) return fl_set_error(ctx, FL_ERR_INVALID, "image too small for %d levels", levels);
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t px = (size_t)w0 * h0;
  // scratch layout (bytes): bgr A | bgr B | depth | mask A | mask B | quant | normal A | normal B | mag | keys | raster | score | cnt | feats
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += fl_align(b, 256); return o; };
  const size_t o_bgr0 = take(px * 3), o_bgr1 = take(px * 3), o_depth = take(px * 2), o_m0 = take(px), o_m1 = take(px), o_q = take(px),
               o_n0 = take(px), o_n1 = take(px), o_mag = take(px * 4), o_keys = take((size_t)next_pow2((int)px) * 8),
               o_raster = take(px * 4), o_score = take(px * 4), o_local = take(px), o_cnt = take(sizeof(ExtractCounters)),
               o_feats = take(sizeof(fl_feature) * 64 * 2 * FL_MAX_LEVELS);
  void *sv = nullptr;
  int rc = fl_scratch(ctx, off, &sv);
  if (rc) return rc;
  uint8_t *s = (uint8_t *)sv;
  const hipMemcpyKind kind = mem == FL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  FL_HIP(ctx, hipMemcpyAsync(s + o_bgr0, bgr, px * 3, kind, ctx->stream));
  FL_HIP(ctx, hipMemcpyAsync(s + o_depth, depth, px * 2, kind, ctx->stream));
  if (mask) FL_HIP(ctx, hipMemcpyAsync(s + o_m0, mask, px, kind, ctx->stream));
  ExtractCounters *d_cnt = (ExtractCounters *)(s + o_cnt);
  fl_feature *d_feats = (fl_feature *)(s + o_feats);
  const int M = 2;
  int all_ok = 1;
  // modality 0: ColorGradient(10, 63, 55) (:515-519); pyrDown :434-453
  {
    int w = w0, h = h0, nf = 63;
    uint8_t *src = s + o_bgr0, *nxt = s + o_bgr1, *mk = mask ? s + o_m0 : nullptr, *mk2 = s + o_m1;
    for (int l = 0; l < levels && all_ok; ++l) {
      if (l > 0) {
        nf /= 2;
        if ((rc = fl_launch_pyrdown_bgr(ctx, src, 0, nxt, 0, 1, w, h))) return rc;
        std::swap(src, nxt);
        if (mk) { if ((rc = fl_launch_resize_nn_half(ctx, mk, 0, mk2, 0, 1, w, h))) return rc; std::swap(mk, mk2); }
        w /= 2;
        h /= 2;
      }
      if ((rc = fl_launch_quantized_orientations_mag(ctx, src, 0, s + o_q, 0, 1, w, h, 10.0f, (float *)(s + o_mag)))) return rc;
      int ok = 0;
      if ((rc = extract_level(ctx, 0, s + o_q, (const float *)(s + o_mag), mk, w, h, nf, 55.0f, 0, (unsigned long long *)(s + o_keys),
                              (int *)(s + o_raster), (float *)(s + o_score), s + o_local, d_cnt, d_feats + 64 * (l * M + 0), &ok)))
        return rc;
      all_ok = all_ok && ok;
      fl_template &t = templates[l * M + 0];
      t.width = t.height = -1; t.offset_x = t.offset_y = 0; t.pyramid_level = l; t.feat_begin = 63 * (l * M + 0); t.feat_count = ok ? nf : 0;
    }
  }
  // modality 1: DepthNormal(2000, 50, 63, 2) (:827-832); pyrDown :721-739
  if (all_ok) {
    int w = w0, h = h0, nf = 63, ext = 2;
    if (mask) FL_HIP(ctx, hipMemcpyAsync(s + o_m0, mask, px, kind, ctx->stream));       // level-0 mask again
    uint8_t *nrm = s + o_n0, *nrm2 = s + o_n1, *mk = mask ? s + o_m0 : nullptr, *mk2 = s + o_m1;
    if ((rc = fl_launch_quantized_normals(ctx, (const uint16_t *)(s + o_depth), 0, nrm, 0, s + o_q, 0, 1, w, h, 2000, 50))) return rc;
    for (int l = 0; l < levels && all_ok; ++l) {
      if (l > 0) {
        nf /= 2;
        ext /= 2;
        if ((rc = fl_launch_resize_nn_half(ctx, nrm, 0, nrm2, 0, 1, w, h))) return rc;
        std::swap(nrm, nrm2);
        if (mk) { if ((rc = fl_launch_resize_nn_half(ctx, mk, 0, mk2, 0, 1, w, h))) return rc; std::swap(mk, mk2); }
        w /= 2;
        h /= 2;
      }
      int ok = 0;
      if ((rc = extract_level(ctx, 1, nrm, nullptr, mk, w, h, nf, 0.f, ext, (unsigned long long *)(s + o_keys), (int *)(s + o_raster),
                              (float *)(s + o_score), s + o_local, d_cnt, d_feats + 64 * (l * M + 1), &ok)))
        return rc;
      all_ok = all_ok && ok;
      fl_template &t = templates[l * M + 1];
      t.width = t.height = -1; t.offset_x = t.offset_y = 0; t.pyramid_level = l; t.feat_begin = 63 * (l * M + 1); t.feat_count = ok ? nf : 0;
    }
  }
  if (!all_ok) {
    FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fl_set_error(ctx, FL_ERR_NO_TEMPLATE, "too few candidate features at some pyramid level (addTemplate returns -1)");
  }
  std::vector<fl_feature> hf((size_t)64 * M * levels);
  FL_HIP(ctx, hipMemcpyAsync(hf.data(), d_feats, sizeof(fl_feature) * hf.size(), hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < levels * M; ++k)
    for (int j = 0; j < templates[k].feat_count; ++j) features[63 * k + j] = hf[(size_t)64 * k + j];
  int box[4];
  crop_templates(templates, levels * M, features, box);
  if (bb) { bb[0] = box[0]; bb[1] = box[1]; bb[2] = box[2]; bb[3] = box[3]; }
  return FL_OK;
}
