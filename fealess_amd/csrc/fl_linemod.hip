// fl_linemod.hip -- hand-written gfx950 kernels for the online matching half of
// cup_linemod::Detector (reference: linemod/linemod.cpp:882-1577).
//
//   k_build_lm     spread + computeResponseMaps + linearize x8 fused        (:950-1088)
//   k_scan         similarity + addSimilarities + coarse threshold          (:1130-1214, 1322-1338, 1483-1506)
//   k_refine       similarityLocal + 16x16 argmax + filter, one level       (:1226-1300, 1509-1573)
//   k_sort_unique  std::sort + std::unique of the surviving matches         (:1437-1439)
//
// Data layout in HBM (per frame, per level, per modality): the 8 linear memories
//   LM[label 0..7][grid (y%T)*T + x%T][(y/T)*W + x/T]   + per-label zero pad
// i.e. exactly the reference's "linearized" Mats laid back to back, so that the reference's
// 1-D scan over template_positions (including its row wrap-around, quirk Q1, and its over-read
// into the next grid row, Q2) is reproduced by plain linear addressing.
// All integer work; the only float expressions are the reference's own score formulas, built
// with -ffp-contract=off so each operator is one IEEE binary32 operation.
#include "fl_internal.h"
#include <limits.h>
#include <string.h>

// ------------------------------------------------------------------------------------------
// k_build_lm: one thread per (grid cell, linear position p); the 8 label bytes it produces go
// to 8 coalesced byte streams.  Response of orientation `ori` to a spread byte b
// (SIMILARITY_LUT of this fork, linemod.cpp:970): 4 if bit ori is set, else 2 if a
// neighbouring orientation (+-1 mod 8) is set, else 1 if +-2 is set, else 0 -- evaluated with
// an 8-bit rotate instead of the two 16-entry table look-ups (identical integers).
__global__ __launch_bounds__(256) void k_build_lm_generic(const uint8_t *__restrict__ quant, size_t quant_stride,
                                                          uint8_t *__restrict__ lm, size_t lm_stride, int w, int h,
                                                          int T, int W, int WH, uint32_t stride)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int gi = blockIdx.y;
  const uint8_t *q = quant + (size_t)blockIdx.z * quant_stride;
  uint8_t *out = lm + (size_t)blockIdx.z * lm_stride;
  if (p >= WH) return;
  const int gy = gi / T, gx = gi - gy * T;
  const int yt = p / W, xt = p - yt * W;
  const int y = yt * T + gy, x = xt * T + gx;
  // spread (linemod.cpp:950-965): OR over the T x T window anchored at (y, x), clipped
  unsigned b = 0;
  const int rmax = min(T, h - y), cmax = min(T, w - x);
  for (int r = 0; r < rmax; ++r) {
    const uint8_t *row = q + (size_t)(y + r) * w + x;
    for (int c = 0; c < cmax; ++c) b |= row[c];
  }
  const unsigned bb = b | (b << 8);           // rotate helper
#pragma unroll
  for (int ori = 0; ori < 8; ++ori) {
    unsigned rot = (bb >> ori) & 0xFFu;
    unsigned r = (rot & 1u) ? 4u : ((rot & 0x82u) ? 2u : ((rot & 0x44u) ? 1u : 0u));
    out[(size_t)ori * stride + (size_t)gi * WH + p] = (uint8_t)r;
  }
}

// Fast path (w % 4 == 0, (w/T) % 4 == 0, T <= 8): one workgroup owns a full-width strip of
// RS = k*T image rows.  The quantized strip (+T-1 rows below) is staged in LDS once, OR-ed
// horizontally and vertically there (separable spread: 2T instead of T*T reads per pixel, 4 pixels
// per 32-bit operation), then every (label, grid cell) stream of the strip -- k*W contiguous bytes
// of linear memory -- is written as coalesced dwords.  The 8 responses of a spread byte come from
// one 8-byte LDS table entry (the SIMILARITY_LUT evaluated for all 8 orientations at once).
// i / d for the small loop indices of the strip kernels: (i + 0.5) * (1/d) in float is at least 0.5/d away from an
// integer while its rounding error is below i/d * 2^-22, so the truncation is exact for i < 2^20 (here i < 2^13).
// 4 VALU instructions instead of the ~35 of an integer division by a run-time divisor.
__device__ __forceinline__ int div_small(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

// OR of the T four-byte windows at byte offsets 0 .. T - 1 (T <= 8) of the 12 bytes s0 s1 s2: four pixels of the horizontal
// spread.  One v_alignbyte_b32 extracts a window (full rate); the 64-bit shifts this replaces compiled to a quarter-rate
// 64-bit multiply-add per offset.
__device__ __forceinline__ uint32_t or_windows(uint32_t s0, uint32_t s1, uint32_t s2, int T)
{
  uint32_t acc = s0;
  for (int c = 1; c < T; ++c)
    acc |= c < 4 ? __builtin_amdgcn_alignbyte(s1, s0, (unsigned)c) : __builtin_amdgcn_alignbyte(s2, s1, (unsigned)(c - 4));
  return acc;
}

__global__ __launch_bounds__(256) void k_build_lm(const uint8_t *__restrict__ quant, size_t quant_stride,
                                                  uint8_t *__restrict__ lm, size_t lm_stride, int w, int h, int T,
                                                  int W, int H, int WH, uint32_t stride, int RS)
{
  extern __shared__ __align__(16) uint8_t smem[];
  const int ws = w + 16;                       // padded LDS row (zero pad: windows may run past x = w-1)
  const int rows_in = RS + T - 1;
  uint8_t *A = smem, *B = smem + (size_t)rows_in * ws;
  unsigned long long *tab = (unsigned long long *)(smem + (size_t)2 * rows_in * ws);
  const uint8_t *q = quant + (size_t)blockIdx.z * quant_stride;
  uint8_t *out = lm + (size_t)blockIdx.z * lm_stride;
  const int y0 = blockIdx.x * RS;
  const int tid = threadIdx.x;
  {
    const unsigned b = (unsigned)tid, bb = b | (b << 8);
    unsigned long long v = 0;
#pragma unroll
    for (int ori = 0; ori < 8; ++ori) {
      const unsigned rot = (bb >> ori) & 0xFFu;
      const unsigned long long r = (rot & 1u) ? 4u : ((rot & 0x82u) ? 2u : ((rot & 0x44u) ? 1u : 0u));
      v |= r << (8 * ori);
    }
    tab[tid] = v;
  }
  const int ws4 = ws >> 2, w4 = w >> 2;
  const float inv_ws4 = 1.0f / (float)ws4, inv_w4 = 1.0f / (float)w4;
  for (int i = tid; i < rows_in * ws4; i += 256) {
    const int r = div_small(i, inv_ws4), c4 = i - __mul24(r, ws4);
    const int y = y0 + r;
    uint32_t v = 0;
    if (y < h && c4 < w4) v = *(const uint32_t *)(q + (size_t)y * w + 4 * c4);
    ((uint32_t *)A)[i] = v;
  }
  __syncthreads();
  // horizontal OR over T columns, 4 pixels per thread
  for (int i = tid; i < rows_in * w4; i += 256) {
    const int r = div_small(i, inv_w4), c4 = i - __mul24(r, w4);
    const int o = __mul24(r, ws4) + c4;                  // dword offset in the padded LDS images (32-bit: a (size_t) row offset
    const uint32_t *src = (const uint32_t *)A + o;       // costs a quarter-rate 64-bit multiply-add per row)
    ((uint32_t *)B)[o] = or_windows(src[0], src[1], src[2], T);                                    // T <= 8: offsets <= 7
  }
  __syncthreads();
  // vertical OR over T rows -> spread image of the strip, back into A
  for (int i = tid; i < RS * w4; i += 256) {
    const int r = div_small(i, inv_w4), c4 = i - __mul24(r, w4);
    const int o = __mul24(r, ws4) + c4;
    const uint32_t *bp = (const uint32_t *)B + o;
    uint32_t acc = 0;
    for (int rr = 0; rr < T; ++rr, bp += ws4) acc |= *bp;
    ((uint32_t *)A)[o] = acc;
  }
  __syncthreads();
  const int yt0 = y0 / T, nyt = min(RS / T, H - yt0), W4 = W >> 2;
  const int items = T * T * nyt * W4;
  const float inv_W4 = 1.0f / (float)W4, inv_nyt = 1.0f / (float)max(nyt, 1), inv_T = 1.0f / (float)T;
  for (int it = tid; it < items; it += 256) {
    const int t2 = div_small(it, inv_W4), xt4 = it - __mul24(t2, W4);
    const int gi = div_small(t2, inv_nyt), yt = t2 - __mul24(gi, nyt);
    const int gy = div_small(gi, inv_T), gx = gi - __mul24(gy, T);
    const uint8_t *srow = A + (size_t)(yt * T + gy) * ws + gx + (size_t)xt4 * 4 * T;
    const unsigned long long r0 = tab[srow[0]], r1 = tab[srow[T]], r2 = tab[srow[2 * T]], r3 = tab[srow[3 * T]];
    uint8_t *dst = out + (size_t)gi * WH + (size_t)(yt0 + yt) * W + 4 * xt4;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const uint32_t v = (uint32_t)((r0 >> (8 * l)) & 0xFF) | ((uint32_t)((r1 >> (8 * l)) & 0xFF) << 8) |
                         ((uint32_t)((r2 >> (8 * l)) & 0xFF) << 16) | ((uint32_t)((r3 >> (8 * l)) & 0xFF) << 24);
      *(uint32_t *)(dst + (size_t)l * stride) = v;
    }
  }
}

// k_spread: the spread image alone (linemod.cpp:950-965) for the finer pyramid levels, same LDS
// separable OR as k_build_lm.  Those levels are only read by k_refine (a few 16x16 patches per
// frame), which evaluates the response LUT on the fly instead of reading 8 linear memories.
__global__ __launch_bounds__(256) void k_spread(const uint8_t *__restrict__ quant, size_t quant_stride,
                                                uint8_t *__restrict__ spread, size_t spread_stride, int w, int h, int T, int RS,
                                                const uint32_t *__restrict__ tiles, size_t tiles_stride)
{
  extern __shared__ __align__(16) uint8_t smem[];
  const int ws = w + 16, rows_in = RS + T - 1;
  uint8_t *A = smem, *B = smem + (size_t)rows_in * ws;
  const uint8_t *q = quant + (size_t)blockIdx.z * quant_stride;
  uint8_t *out = spread + (size_t)blockIdx.z * spread_stride;
  const int y0 = blockIdx.x * RS, tid = threadIdx.x;
  const int ws4 = ws >> 2, w4 = w >> 2;
  // column window [lo4, lo4 + n_out) in dwords: everything, or (lazy fine level) the span of the marked tiles
  // that these rows cross -- nothing marked, nothing to do
  int lo4 = 0, n_out = w4;
  if (tiles) {
    const uint32_t *tf = tiles + (size_t)blockIdx.z * tiles_stride;
    const int nstrips = (w + FL_TILE - 1) / FL_TILE;
    const int c0 = y0 / FL_TILE, c1 = min(y0 + RS - 1, h - 1) / FL_TILE;
    int smin = nstrips, smax = -1;
    for (int c = c0; c <= c1; ++c)
      for (int st = 0; st < nstrips; ++st) {
        const int t = c * nstrips + st;
        if (!((tf[t >> 5] >> (t & 31)) & 1u)) continue;
        const uint32_t rw = tf[2 * FL_TILE_WORDS + t];     // rows of this tile that are read: does the strip touch them?
        if ((int)(rw & 0xFFFFu) < y0 || (int)(rw >> 16) > y0 + RS - 1) continue;
        smin = min(smin, st);
        smax = max(smax, st);
      }
    if (smax < 0) return;                                  // workgroup-uniform
    lo4 = (smin * FL_TILE) >> 2;
    n_out = (min(w, (smax + 1) * FL_TILE) + 3) / 4 - lo4;
  }
  const int n_in = min(ws4, lo4 + n_out + 3) - lo4;        // the horizontal OR of dword c reads dwords c .. c + 3
  const float inv_in = 1.0f / (float)n_in, inv_out = 1.0f / (float)n_out;
  for (int i = tid; i < rows_in * n_in; i += 256) {
    const int r = div_small(i, inv_in), c4 = lo4 + (i - __mul24(r, n_in));
    const int y = y0 + r;
    uint32_t v = 0;
    if (y < h && c4 < w4) v = *(const uint32_t *)(q + (size_t)y * w + 4 * c4);
    ((uint32_t *)A)[__mul24(r, ws4) + c4] = v;
  }
  __syncthreads();
  for (int i = tid; i < rows_in * n_out; i += 256) {
    const int r = div_small(i, inv_out), c4 = lo4 + (i - __mul24(r, n_out));
    const int o = __mul24(r, ws4) + c4;
    const uint32_t *src = (const uint32_t *)A + o;
    ((uint32_t *)B)[o] = or_windows(src[0], src[1], src[2], T);
  }
  __syncthreads();
  for (int i = tid; i < RS * n_out; i += 256) {
    const int r = div_small(i, inv_out), c4 = lo4 + (i - __mul24(r, n_out));
    if (y0 + r >= h) continue;
    const uint32_t *bp = (const uint32_t *)B + (__mul24(r, ws4) + c4);
    uint32_t acc = 0;
    for (int rr = 0; rr < T; ++rr, bp += ws4) acc |= *bp;
    *(uint32_t *)(out + (size_t)(y0 + r) * w + 4 * c4) = acc;
  }
}

__global__ __launch_bounds__(256) void k_spread_generic(const uint8_t *__restrict__ quant, size_t quant_stride,
                                                        uint8_t *__restrict__ spread, size_t spread_stride, int w, int h, int T)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const uint8_t *q = quant + (size_t)blockIdx.z * quant_stride;
  unsigned b = 0;
  const int rmax = min(T, h - y), cmax = min(T, w - x);
  for (int r = 0; r < rmax; ++r)
    for (int c = 0; c < cmax; ++c) b |= q[(size_t)(y + r) * w + x + c];
  spread[(size_t)blockIdx.z * spread_stride + (size_t)y * w + x] = (uint8_t)b;
}

// tiles == nullptr: the whole image.  Otherwise (lazy fine level) only what the marked tiles need; the generic
// fallback kernel ignores the marks and spreads everything, which is always sufficient.
int fl_launch_spread_tiles(fl_context *ctx, const uint8_t *quant, size_t quant_stride, uint8_t *spread, size_t spread_stride,
                           int n_frames, int w, int h, int T, const uint32_t *tiles, size_t tiles_stride)
{
  const bool aligned = ((uintptr_t)quant % 4 == 0) && (quant_stride % 4 == 0) && ((uintptr_t)spread % 4 == 0) && (spread_stride % 4 == 0);
  if (w % 4 == 0 && T <= 8 && T >= 2 && aligned) {
    int k = 4;                                   // strips of k x T rows (taller strips cost the lazy fine level more than they save:
                                                 // k = 6 took 5.0 against 3.95 ms per 4096 frames, a strip works for every marked tile it touches)
    size_t lds;
    for (;;) {
      lds = (size_t)2 * (k * T + T - 1) * (w + 16);
      if (lds <= 60 * 1024 || k == 1) break;
      --k;
    }
    if (lds <= 64 * 1024) {
      const int RS = k * T;
      dim3 grid((h + RS - 1) / RS, 1, n_frames);
      hipLaunchKernelGGL(k_spread, grid, dim3(256), lds, ctx->stream, quant, quant_stride, spread, spread_stride, w, h, T, RS,
                         tiles, tiles_stride);
      FL_HIP(ctx, hipGetLastError());
      return FL_OK;
    }
  }
  dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames);
  hipLaunchKernelGGL(k_spread_generic, grid, dim3(256), 0, ctx->stream, quant, quant_stride, spread, spread_stride, w, h, T);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

int fl_launch_spread(fl_context *ctx, const uint8_t *quant, size_t quant_stride, uint8_t *spread, size_t spread_stride,
                     int n_frames, int w, int h, int T)
{
  return fl_launch_spread_tiles(ctx, quant, quant_stride, spread, spread_stride, n_frames, w, h, T, nullptr, 0);
}

int fl_launch_build_lm(fl_context *ctx, const uint8_t *quant, size_t quant_stride, uint8_t *lm,
                       size_t lm_stride, int n_frames, int w, int h, int T)
{
  const int W = w / T, H = h / T, WH = W * H;
  const uint32_t stride = (uint32_t)fl_lm_label_stride(w, h, T);
  const bool aligned = ((uintptr_t)quant % 4 == 0) && (quant_stride % 4 == 0) && ((uintptr_t)lm % 4 == 0) && (lm_stride % 4 == 0);
  if (w % 4 == 0 && W % 4 == 0 && T <= 8 && T >= 2 && aligned) {
    int k = 6;                                   // strips of k x T rows: the taller, the longer the runs a strip writes into each linear
                                                 // memory (k * W bytes); measured at VGA level 1, ms per 2048 frames, k = 2 / 3 / 4 / 6 / 8:
                                                 // 1.49 / 1.37 / 1.30 - 1.35 / 1.18 / 1.16 - 1.20
    size_t lds;
    for (;;) {
      lds = (size_t)2 * (k * T + T - 1) * (w + 16) + 2048;
      if (lds <= 60 * 1024 || k == 1) break;
      --k;
    }
    if (lds <= 64 * 1024) {
      const int RS = k * T;
      dim3 grid((h + RS - 1) / RS, 1, n_frames);
      hipLaunchKernelGGL(k_build_lm, grid, dim3(256), lds, ctx->stream, quant, quant_stride, lm, lm_stride, w, h, T, W, H, WH,
                         stride, RS);
      FL_HIP(ctx, hipGetLastError());
      return FL_OK;
    }
  }
  dim3 grid((WH + 255) / 256, T * T, n_frames);
  hipLaunchKernelGGL(k_build_lm_generic, grid, dim3(256), 0, ctx->stream, quant, quant_stride, lm, lm_stride, w, h, T, W, WH,
                     stride);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

// ------------------------------------------------------------------------------------------
// k_scan: one wavefront per (pyramid, 1024-position chunk, frame).  Lane i owns 16 consecutive
// positions and keeps their u8 partial sums packed in four 32-bit registers per modality:
// with <= 63 features of response <= 4 a byte never carries (63*4 = 252, linemod.cpp:1133-1137),
// so one 32-bit add sums four positions.  Feature offsets are wave-uniform (scalar loads), padded
// to a multiple of 8 with the offset of the zero pad so that 8 independent 16-byte loads are in
// flight per step.  The linear memories (1.2 MB per frame at VGA/T=8) are shared by all
// templates and are served from L2; the feature tables are the only per-template HBM stream.
// Between the modalities and after every 8 features a wave whose positions can no longer reach the coarse threshold stops
// (exact: see `prune` in the kernel); the reference adds every feature of every template everywhere (linemod.cpp:1471-1481).
struct ScanArgs {
  const FlScanHdr *hdr;
  const int2 *items;         // work list: (pyramid, chunk)
  int n_items;
  const uint32_t *offs;
  const uint8_t *enabled;    // per pyramid: class selected by match()'s class_ids (linemod.cpp:1418-1434)
  uint8_t *ws;               // frame workspace base
  size_t ws_stride;
  size_t lm_off[FL_MAX_MODALITIES];
  size_t off_count, off_cand;
  int n_pyr, M, nchunks, W, WH, T, cap;
  int bpf, n_frames;         // blocks per frame, frames in this launch
  float threshold;
  int prune;                 // stop a (template, chunk) between modalities once no position can reach the threshold (exact)
  unsigned prune_mid;        // ... and inside a modality after the 8-feature groups whose bit is set (same bound, same exactness)
  uint16_t *dbg;             // optional raw u16 maps of pyramids [dbg_first, dbg_first+dbg_count)
  int dbg_first, dbg_count;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// the larger even byte and the larger odd byte of a word, as two 16-bit values (v_pk_max_u16)
__device__ __forceinline__ u16x2 pk_bytes_max(uint32_t a)
{
  const uint32_t lo = a & 0x00FF00FFu, hi = (a >> 8) & 0x00FF00FFu;
  return __builtin_elementwise_max(__builtin_bit_cast(u16x2, lo), __builtin_bit_cast(u16x2, hi));
}
__device__ __forceinline__ uint4 ld16(const uint8_t *p)
{
  uint4 v;
  __builtin_memcpy(&v, p, 16);     // global_load_dwordx4, any byte alignment (unaligned access mode)
  return v;
}

#ifndef FL_SCAN_PRUNE_MID
#define FL_SCAN_PRUNE_MID 0x7Fu   // 8-feature groups after which a modality checks the bound (every one but the last: the check
                                  // after 16 features is the one that pays -- chunks without colour edges stop there)
#endif
#ifndef FL_SCAN_WPE
#define FL_SCAN_WPE 5             // waves per SIMD; measured after the DPP change (ms per 1280 frames x 360 templates): 3: 2.32, 4: 1.95, 5: 1.71, 6: 1.77, 8: 2.17
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FL_SCAN_WPE, FL_SCAN_WPE))) void k_scan(ScanArgs a)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs, each with its own
  // 4 MB L2.  All blocks of one frame are given the same residue mod 8, so a frame's linear
  // memories (1.2 MB at VGA level 1) are fetched into ONE L2 and reused by every template there,
  // instead of being streamed into all eight.  Placement only affects speed, never results.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int frame = xcd + 8 * (q / a.bpf);
  if (frame >= a.n_frames) return;
  const int item = __builtin_amdgcn_readfirstlane((q % a.bpf) * 4 + wave);
  if (item >= a.n_items) return;
  const int2 gc = a.items[item];
  const int g = __builtin_amdgcn_readfirstlane(gc.x), chunk = __builtin_amdgcn_readfirstlane(gc.y);
  if (!a.enabled[g]) return;                             // wave-uniform: matchClass is not called for this class
  uint8_t *ws = a.ws + (size_t)frame * a.ws_stride;
  const int j0 = chunk * 1024 + lane * 16;

  uint32_t tot[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) tot[i] = 0;
  // Exact pruning between modalities: a feature adds at most 4 to a position (SIMILARITY_LUT's largest entry), so once the
  // largest total of this wave's 1024 positions plus 4 x (the features of the modalities still to come) cannot exceed the
  // coarse threshold, no position of the chunk can become a candidate (`raw > raw_threshold`, linemod.cpp:1490-1492) and the
  // rest of the template's additions have no observable effect: the wave stops.  The colour modality comes first and is
  // sparse (gradients only on edges), so on most (template, chunk) pairs the depth modality -- half of the scan's loads -- is
  // never read.  The same bound is checked inside a modality after the 8-feature groups a.prune_mid names (chunks without
  // colour edges stop after 16 features).  Not when the raw maps are tapped (fl_similarity_maps), and FL_SCAN_PRUNE=0
  // switches it off (a.prune).
  int nf_all = 0;
  for (int m = 0; m < a.M; ++m) nf_all += a.hdr[g * a.M + m].nf;
  const int prune_threshold = (int)(2 * nf_all + (a.threshold / 100.f) * (2 * nf_all) + 0.5f);   // = raw_threshold below
  const bool prune = a.prune && !(a.dbg && g >= a.dbg_first && g < a.dbg_first + a.dbg_count);
  int nf = 0;
  for (int m = 0; m < a.M; ++m) {
    const FlScanHdr h = a.hdr[g * a.M + m];
    uint32_t mx_tot = 0;                                 // the lane's largest total of the modalities done
    if (prune && m > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) mx_tot = max(mx_tot, tot[i]);
      if (__ballot((int)mx_tot + 4 * (nf_all - nf) > prune_threshold) == 0ull) return;    // wave-uniform
    }
    const unsigned mid = prune ? a.prune_mid : 0u;
    nf += h.nf;
    if (chunk * 1024 >= h.P) continue;                  // wave-uniform; lanes past P load along (masked below): the
                                                       // next lane's first dword is this lane's bytes 16..19
    const uint8_t *lm_u = ws + a.lm_off[m] + chunk * 1024;   // lane 0's bytes: wave-uniform
    const unsigned lane_off = (unsigned)lane * 16u;
    // descriptor over the frame's linear memories from 4 bytes below this chunk on (scalar offsets stay positive)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(lm_u - 4), 0, 0x7FFFFFFF, 0x00020000);
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const uint32_t *offs = a.offs + h.off_begin;
    // A feature's 16 bytes start at an arbitrary byte.  A byte-aligned 16-byte load costs ~2.4x a 4-byte aligned one
    // on gfx950 (tools/probes/ldwidth.hip) and this loop is bound by exactly those loads, so each feature is fetched as
    // an aligned 16 + 4 bytes and shifted into place with four v_alignbyte (the misalignment is wave-uniform).
    // The extra 4 bytes are the next lane's first dword: one DPP wave shift instead of a second vector load; lane 63's
    // come from a wave-uniform address, i.e. a scalar load.
    const unsigned lm_mis = (unsigned)(size_t)lm_u & 3u;  // same for every lane (lane * 16): keep it a scalar
    // Only the lanes whose 16 positions start below template_positions have anything to add (P = 831 of the 1024 positions
    // of a chunk for a 160-pixel template at VGA level 1: 52 of 64 lanes); one more lane stays on because its first dword is
    // the previous lane's bytes 16..19.  The condition is a contiguous lane range, i.e. one EXEC mask around the whole loop --
    // not a select per load (measured in round 1: that serialised the eight loads in flight) -- and the masked lanes' loads
    // are simply not issued: 19 % less L2 traffic.
    const int lanes_on = min(64, ((h.P - chunk * 1024 + 15) >> 4) + 1);
    const bool on = lane < lanes_on;
    for (int k = 0; k < h.n_pad; k += 8) {
      if (on) {
      uint4 v[8];
      uint32_t e[8];
      unsigned mis[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned off = offs[k + u];
        mis[u] = (lm_mis + off) & 3u;
        // buffer_load ... v_lane_off, s[rsrc], s_feature_off offen: the lane's byte offset is one loop-invariant VGPR and the
        // feature's offset a scalar operand -- no per-lane 64-bit address arithmetic (it was 2 of the 10 VALU instructions
        // per feature)
        const u32x4 ld = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, off - mis[u] + 4u, 0);   // 4-byte aligned
        v[u] = make_uint4(ld.x, ld.y, ld.z, ld.w);
        const uint32_t tail = *(const uint32_t *)__builtin_assume_aligned(lm_u + off - mis[u] + 1024, 4);
        e[u] = (uint32_t)__builtin_amdgcn_update_dpp((int)tail, (int)v[u].x, 0x130, 0xF, 0xF, false);   // wave_shl:1
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 += __builtin_amdgcn_alignbyte(v[u].y, v[u].x, mis[u]);
        a1 += __builtin_amdgcn_alignbyte(v[u].z, v[u].y, mis[u]);
        a2 += __builtin_amdgcn_alignbyte(v[u].w, v[u].z, mis[u]);
        a3 += __builtin_amdgcn_alignbyte(e[u], v[u].w, mis[u]);
      }
      }
      if ((mid >> (k >> 3)) & 1u) {                      // wave-uniform
        // the same bound inside the modality: the lane's largest byte so far on top of its largest finished total
        // (an over-estimate of its largest partial total, so nothing reachable is ever dropped)
        const u16x2 b0 = pk_bytes_max(a0), b1 = pk_bytes_max(a1), b2 = pk_bytes_max(a2), b3 = pk_bytes_max(a3);
        const u16x2 bm = __builtin_elementwise_max(__builtin_elementwise_max(b0, b1), __builtin_elementwise_max(b2, b3));
        const int part = (int)max(bm.x, bm.y);
        const int left = nf_all - nf + max(0, h.nf - (k + 8));
        // (every lane votes: one past this modality's template_positions adds nothing here, its bound is simply generous --
        // it may still collect from a modality whose template_positions reach further)
        if (__ballot((int)mx_tot + part + 4 * left > prune_threshold) == 0ull) return;          // wave-uniform
      }
    }
    const uint32_t acc[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint32_t byte = (acc[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
      tot[i] += (j0 + i < h.P) ? byte : 0u;        // cells >= template_positions stay 0 (:1160)
    }
  }
  if (a.dbg && g >= a.dbg_first && g < a.dbg_first + a.dbg_count) {
    uint16_t *d = a.dbg + ((size_t)frame * a.dbg_count + (g - a.dbg_first)) * a.WH;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (j0 + i < a.WH) d[j0 + i] = (uint16_t)tot[i];
  }
  // coarse threshold (linemod.cpp:1483-1506), float expression evaluated as written
  const int raw_threshold = (int)(2 * nf + (a.threshold / 100.f) * (2 * nf) + 0.5f);
  uint32_t mx = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) mx = max(mx, tot[i]);
  if ((int)mx <= raw_threshold) return;
  int *count = (int *)(ws + a.off_count);
  FlCand *cand = (FlCand *)(ws + a.off_cand);
  const int offset = a.T / 2 + (a.T % 2 - 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int raw = (int)tot[i];
    const int j = j0 + i;
    if (raw > raw_threshold && j < a.WH) {
      const int r = j / a.W, c = j - r * a.W;
      const int idx = atomicAdd(count, 1);
      if (idx < a.cap) {
        FlCand cd;
        cd.x = c * a.T + offset;
        cd.y = r * a.T + offset;
        cd.g = g;
        cd.sim = (raw * 100.f) / (4 * nf) + 0.5f;                               // :1502
        cand[idx] = cd;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_refine: one wavefront per candidate.  The 16x16 patch is held as 64 lanes x 4 packed bytes
// (lane = row*4 + column group); each in-bounds feature costs one 4-byte load per lane.
// Slow path of k_refine: 4 consecutive linear-memory positions that run past a row end (xt >= W) or
// past the memory (yt >= H -> next grid cell; past the last one: 0), mapped back to spread pixels.
// Returns the 4 spread bytes packed.
__device__ __noinline__ uint32_t refine_slow_path(const uint8_t *sp, bool in, int xt0, int yt0, int gx, int gy, int T,
                                                  int W, int H, int w)
{
  uint32_t out = 0;
  const int g0 = gy * T + gx, TT = T * T;
  for (int c = 0; c < 4; ++c) {
    int xx = xt0 + c, yy = yt0, gg = g0;
    const int wx = xx >= W ? xx / W : 0;
    xx -= wx * W;
    yy += wx;
    const int wy = yy >= H ? yy / H : 0;
    yy -= wy * H;
    gg += wy;
    if (in && gg < TT) {
      const int ggy = gg / T, ggx = gg - ggy * T;
      out |= (uint32_t)sp[(size_t)(yy * T + ggy) * w + (size_t)(xx * T + ggx)] << (8 * c);
    }
  }
  return out;
}

struct RefineArgs {
  const FlFineHdr *hdr;
  const FlFineFeat *feat;
  uint8_t *ws;
  size_t ws_stride;
  size_t spread_off[FL_MAX_MODALITIES];
  size_t off_count, off_cand;
  int M, Lm1, level, w, h, T, W, H, cap;
  float threshold;
};

#ifndef FL_REFINE_WPE
#define FL_REFINE_WPE 6           // waves per SIMD; measured (ms per 1280 frames): 4: 0.92, 5 (the compiler's choice): 0.80, 6: 0.76, 8: 1.03
#endif
__attribute__((amdgpu_waves_per_eu(FL_REFINE_WPE, FL_REFINE_WPE)))
__global__ __launch_bounds__(256) void k_refine(RefineArgs a)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  uint8_t *ws = a.ws + (size_t)frame * a.ws_stride;
  const int n = min(*(const int *)(ws + a.off_count), a.cap);
  FlCand *cand = (FlCand *)(ws + a.off_cand);
  const int row = lane >> 2, col4 = (lane & 3) * 4;
  const int T = a.T, W = a.W;
  const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
  for (int ci = blockIdx.x * 4 + wave; ci < n; ci += gridDim.x * 4) {
    FlCand cd = cand[ci];
    const int g = __builtin_amdgcn_readfirstlane(cd.g);
    if (g < 0) continue;
    const FlFineHdr *hdr = a.hdr + ((size_t)g * a.Lm1 + a.level) * a.M;
    const int max_x = a.w - hdr[0].width - border;                      // :1517-1518 (tp[start])
    const int max_y = a.h - hdr[0].height - border;
    int x = __builtin_amdgcn_readfirstlane(cd.x) * 2 + 1, y = __builtin_amdgcn_readfirstlane(cd.y) * 2 + 1;
    x = max(x, border);
    y = max(y, border);
    x = min(x, max_x);
    y = min(y, max_y);
    const int offset_x = (x / T - 8) * T, offset_y = (y / T - 8) * T;   // C division (trunc), :1240
    uint32_t tot0 = 0, tot1 = 0, tot2 = 0, tot3 = 0;                    // u16 totals of 4 positions
    int numFeatures = 0;
    const int H = a.H;
    const int offx_t = offset_x / T, offy_t = offset_y / T;
    const size_t lane_off = (size_t)row * T * a.w + (size_t)col4 * T;
    for (int m = 0; m < a.M; ++m) {
      const FlFineHdr h = hdr[m];
      numFeatures += h.feat_count;
      const uint8_t *sp = ws + a.spread_off[m];
      const FlFineFeat *ff = a.feat + h.feat_begin;
      uint32_t acc = 0;
      // The reference reads 16 rows x 16 bytes of the feature's linear memory at stride W starting at
      // lm_index (:1260-1297).  Here the same linear index is mapped back to the pixel it was
      // linearised from -- grid cell g, position k = yt*W + xt -> (yt*T + g/T, xt*T + g%T) -- and the
      // response LUT is evaluated on the spread byte.  Running past a row (xt >= W) or past the
      // memory (yt >= H -> next grid cell, past the last one: 0) follows the linear layout exactly.
      // lane l holds feature l of this template (<= 63 features <= 64 lanes): one coalesced 12-byte
      // load per lane, after which every feature is broadcast from registers -- no per-feature
      // memory round trip.  Then 8 features x 4 positions = 32 independent byte loads per step.
      int f_xy = 0, f_q = 0, f_g = 0;
      if (lane < h.feat_count) {
        const FlFineFeat f = ff[lane];
        f_xy = (int)(uint16_t)f.x | ((int)(uint16_t)f.y << 16);
        f_q = (int)(uint16_t)f.qx | ((int)(uint16_t)f.qy << 16);
        f_g = (int)f.gx | ((int)f.gy << 8) | ((int)f.label << 16);
      }
      for (int k = 0; k < h.feat_count; k += 8) {
        uint32_t b[8][4];
        int lab[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = min(k + u, h.feat_count - 1);                    // wave-uniform
          const int xy = __shfl(f_xy, idx, 64), qq = __shfl(f_q, idx, 64), gl = __shfl(f_g, idx, 64);
          const int fx = (int)(int16_t)(xy & 0xFFFF) + offset_x, fy = (int)(int16_t)(xy >> 16) + offset_y;
          const bool in = (k + u < h.feat_count) && fx >= 0 && fy >= 0 && fx < a.w && fy < a.h;   // :1257
          lab[u] = (gl >> 16) & 0xFF;
          // offsets are multiples of T, so (f + offset) mod T and div T follow from the stored residues
          const int gx = gl & 0xFF, gy = (gl >> 8) & 0xFF;
          const int lm_x = (int)(int16_t)(qq & 0xFFFF) + offx_t, lm_y = (int)(int16_t)(qq >> 16) + offy_t;
          if (in && lm_x + 15 < W && lm_y + 15 < H) {                      // wave-uniform: the usual case
            // the 16x16 patch stays inside its linear memory: position (row, col4 + c) is simply the
            // pixel (fy + row*T, fx + (col4 + c)*T)
            // The lane's four positions are T bytes apart: for the usual T one (or two) 16-byte loads replace four
            // byte loads -- the stage is bound by vector-memory instructions, 8 features x 4 in flight per step.
            const uint8_t *p0 = sp + (size_t)(fy * a.w + fx) + lane_off;
            if (T == 5) {
              const uint4 v = ld16(p0);                                     // bytes 0, 5, 10, 15
              b[u][0] = v.x & 0xFFu; b[u][1] = (v.y >> 8) & 0xFFu; b[u][2] = (v.z >> 16) & 0xFFu; b[u][3] = v.w >> 24;
            } else if (T == 4) {
              const uint4 v = ld16(p0);                                     // bytes 0, 4, 8, 12
              b[u][0] = v.x & 0xFFu; b[u][1] = v.y & 0xFFu; b[u][2] = v.z & 0xFFu; b[u][3] = v.w & 0xFFu;
            } else if (T == 8) {
              const uint4 v0 = ld16(p0), v1 = ld16(p0 + 16);                // bytes 0, 8 | 16, 24
              b[u][0] = v0.x & 0xFFu; b[u][1] = v0.z & 0xFFu; b[u][2] = v1.x & 0xFFu; b[u][3] = v1.z & 0xFFu;
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) b[u][c] = p0[c * T];
            }
          } else {
            // rare: the patch leaves its linear memory (or the feature is out of the image); kept out
            // of line so the unrolled step stays small (instruction-cache footprint)
            uint32_t r4 = refine_slow_path(sp, in, lm_x + col4, lm_y + row, gx, gy, T, W, H, a.w);
#pragma unroll
            for (int c = 0; c < 4; ++c) b[u][c] = (r4 >> (8 * c)) & 0xFFu;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          uint32_t packed = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned bb = b[u][c] | (b[u][c] << 8);
            const unsigned rot = (bb >> lab[u]) & 0xFFu;
            const unsigned r = (rot & 1u) ? 4u : ((rot & 0x82u) ? 2u : ((rot & 0x44u) ? 1u : 0u));
            packed |= r << (8 * c);
          }
          acc += packed;                                                   // 4 packed u8 adds
        }
      }
      tot0 += acc & 0xFFu;
      tot1 += (acc >> 8) & 0xFFu;
      tot2 += (acc >> 16) & 0xFFu;
      tot3 += acc >> 24;
    }
    // first strict maximum in row-major order (:1547-1562): key = score << 8 | (255 - pos)
    const int pos = row * 16 + col4;
    uint32_t key = (tot0 << 8) | (uint32_t)(255 - pos);
    key = max(key, (tot1 << 8) | (uint32_t)(254 - pos));
    key = max(key, (tot2 << 8) | (uint32_t)(253 - pos));
    key = max(key, (tot3 << 8) | (uint32_t)(252 - pos));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, s, 64));
    const int best_score = (int)(key >> 8);
    int best_r = -1, best_c = -1;                                       // Q4: all-zero patch
    if (best_score > 0) {
      const int bp = 255 - (int)(key & 0xFFu);
      best_r = bp >> 4;
      best_c = bp & 15;
    }
    if (lane == 0) {
      cd.x = (x / T - 8 + best_c) * T + offset;                         // :1564-1566
      cd.y = (y / T - 8 + best_r) * T + offset;
      cd.sim = (best_score * 100.f) / (4 * numFeatures);
      if (cd.sim < a.threshold) cd.g = -1;                              // MatchPredicate :1447
      cand[ci] = cd;
    }
  }
}

// k_mark_tiles (lazy fine levels): which 60x60 tiles of level a.level will k_refine read?  One workgroup per
// frame, one wave per candidate, lane = feature, with k_refine's own arithmetic: a feature whose 16x16 patch stays
// inside its linear memory reads the spread bytes (fy + r*T, fx + c*T), r, c < 16; the candidate's features span a
// rectangle, its tiles go into bitmap 0 ("spread bytes read") and the tiles of the rectangle grown by T-1 to the
// right and below into bitmap 1 ("quantised pixels those spreads are made of").  Anything unusual -- a feature
// outside the image, a patch that leaves its linear memory and wraps (Q1/Q2) -- marks the whole frame, which is
// the eager computation.
__global__ __launch_bounds__(256) void k_mark_tiles(RefineArgs a, size_t off_tiles)
{
  __shared__ uint32_t bm[2][FL_TILE_WORDS];
  __shared__ int rlo[2][32 * FL_TILE_WORDS], rhi[2][32 * FL_TILE_WORDS];
  __shared__ int s_all;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.x;
  uint8_t *ws = a.ws + (size_t)frame * a.ws_stride;
  for (int i = threadIdx.x; i < 2 * FL_TILE_WORDS; i += 256) bm[0][i] = 0;
  for (int i = threadIdx.x; i < 2 * 32 * FL_TILE_WORDS; i += 256) { rlo[0][i] = 0xFFFF; rhi[0][i] = 0; }
  if (threadIdx.x == 0) s_all = 0;
  __syncthreads();
  const int n = min(*(const int *)(ws + a.off_count), a.cap);
  const FlCand *cand = (const FlCand *)(ws + a.off_cand);
  const int T = a.T, W = a.W, H = a.H;
  const int border = 8 * T;
  const int nstrips = (a.w + FL_TILE - 1) / FL_TILE, nchunks = (a.h + FL_TILE - 1) / FL_TILE;
  for (int ci = wave; ci < n; ci += 4) {
    const FlCand cd = cand[ci];
    const int g = __builtin_amdgcn_readfirstlane(cd.g);
    if (g < 0) continue;
    const FlFineHdr *hdr = a.hdr + ((size_t)g * a.Lm1 + a.level) * a.M;
    const int max_x = a.w - hdr[0].width - border, max_y = a.h - hdr[0].height - border;
    int x = __builtin_amdgcn_readfirstlane(cd.x) * 2 + 1, y = __builtin_amdgcn_readfirstlane(cd.y) * 2 + 1;
    x = max(x, border);
    y = max(y, border);
    x = min(x, max_x);
    y = min(y, max_y);
    const int offset_x = (x / T - 8) * T, offset_y = (y / T - 8) * T;
    const int offx_t = offset_x / T, offy_t = offset_y / T;
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
    bool odd = false;
    for (int m = 0; m < a.M; ++m) {
      const FlFineHdr h = hdr[m];
      if (lane < h.feat_count) {
        const FlFineFeat f = a.feat[h.feat_begin + lane];
        const int fx = (int)f.x + offset_x, fy = (int)f.y + offset_y;
        const bool in = fx >= 0 && fy >= 0 && fx < a.w && fy < a.h;
        const int lm_x = (int)f.qx + offx_t, lm_y = (int)f.qy + offy_t;
        if (in && lm_x + 15 < W && lm_y + 15 < H) {
          x0 = min(x0, fx); y0 = min(y0, fy);
          x1 = max(x1, fx + 15 * T); y1 = max(y1, fy + 15 * T);
        } else {
          odd = true;
        }
      }
      if (h.feat_count > 64) odd = true;                    // k_refine keeps one feature per lane
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      x0 = min(x0, __shfl_xor(x0, sft, 64)); y0 = min(y0, __shfl_xor(y0, sft, 64));
      x1 = max(x1, __shfl_xor(x1, sft, 64)); y1 = max(y1, __shfl_xor(y1, sft, 64));
    }
    if (__any(odd) || x1 >= a.w || y1 >= a.h) { if (lane == 0) s_all = 1; continue; }
    if (x1 < 0) continue;                                  // no feature at all
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      const int grow = kind ? T - 1 : 0;
      const int yb = min(a.h - 1, y1 + grow);
      const int s0 = x0 / FL_TILE, s1 = min(a.w - 1, x1 + grow) / FL_TILE, c0 = y0 / FL_TILE, c1 = yb / FL_TILE;
      const int ns = s1 - s0 + 1, nt = ns * (c1 - c0 + 1);
      for (int t = lane; t < nt; t += 64) {
        const int c = t / ns, tile = (c0 + c) * nstrips + s0 + (t - c * ns);
        atomicOr(&bm[kind][tile >> 5], 1u << (tile & 31));
        atomicMin(&rlo[kind][tile], max(y0, (c0 + c) * FL_TILE));
        atomicMax(&rhi[kind][tile], min(yb, (c0 + c) * FL_TILE + FL_TILE - 1));
      }
    }
  }
  __syncthreads();
  uint32_t *out = (uint32_t *)(ws + off_tiles) + (size_t)a.level * FL_TILE_BLOCK_WORDS;
  const int ntiles = nstrips * nchunks;
  const bool all = s_all != 0;
  for (int i = threadIdx.x; i < 2 * FL_TILE_WORDS; i += 256) {
    const int wd = i % FL_TILE_WORDS;
    uint32_t v = bm[0][i];
    if (all) v = wd * 32 + 32 <= ntiles ? 0xFFFFFFFFu : (wd * 32 < ntiles ? (1u << (ntiles - wd * 32)) - 1u : 0u);
    out[i] = v;
  }
  for (int i = threadIdx.x; i < 2 * 32 * FL_TILE_WORDS; i += 256) {
    const int tile = i % (32 * FL_TILE_WORDS);
    uint32_t v = ((uint32_t)rlo[0][i] << 16) | (uint32_t)rhi[0][i];
    if (all) v = (uint32_t)(a.h - 1);                       // rows 0 .. h-1: the consumers clip to their own tile
    (void)tile;
    out[2 * FL_TILE_WORDS + i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// k_sort_unique: one 1024-thread workgroup per frame.  128-bit keys, sorted descending:
//   hi = similarity bits << 32 | (0x7FFFFFFF - template_id)     (Match::operator<, linemod.hpp:262)
//   lo = (0xFFFF - class) << 32 | (0xFFFF - (y+32768)) << 16 | (0xFFFF - (x+32768))
// lo only fixes the order std::sort leaves unspecified (Q5).  Bitonic network in LDS when the
// padded count fits, otherwise in the frame's key scratch in HBM (same code, block-level sync).
struct Key128 { unsigned long long hi, lo; };
__device__ __forceinline__ bool key_gt(const Key128 &a, const Key128 &b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }

struct SortArgs {
  const FlPyrInfo *pyr;
  uint8_t *ws;
  size_t ws_stride;
  size_t off_count, off_cand, off_keys, off_match;
  int cap;
};

#define FL_SORT_LDS_KEYS 2048

template <bool IN_LDS>
__device__ void bitonic_desc(Key128 *k, int n2)
{
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const bool desc = ((i & size) == 0);
        Key128 a = k[i], b = k[j];
        if (key_gt(b, a) == desc) { k[i] = b; k[j] = a; }
      }
    }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sort_unique(SortArgs a)
{
  __shared__ Key128 lds_keys[FL_SORT_LDS_KEYS];
  __shared__ int s_n, s_scan[1024], s_base;
  const int frame = blockIdx.x;
  uint8_t *ws = a.ws + (size_t)frame * a.ws_stride;
  int *counters = (int *)(ws + a.off_count);      // [0] candidates, [1] matches out, [2] overflow flag
  const int n_raw = *counters;
  const int n = min(n_raw, a.cap);
  const FlCand *cand = (const FlCand *)(ws + a.off_cand);
  Key128 *gkeys = (Key128 *)(ws + a.off_keys);
  fl_match *out = (fl_match *)(ws + a.off_match);
  if (threadIdx.x == 0) { s_n = 0; s_base = 0; }
  __syncthreads();
  // gather live candidates (order irrelevant before the sort)
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const FlCand c = cand[i];
    if (c.g >= 0) {
      const FlPyrInfo pi = a.pyr[c.g];
      Key128 k;
      k.hi = ((unsigned long long)__float_as_uint(c.sim) << 32) | (unsigned)(0x7FFFFFFF - pi.template_id);
      k.lo = ((unsigned long long)(0xFFFF - pi.class_idx) << 32) | ((unsigned long long)(0xFFFF - (c.y + 32768)) << 16) |
             (unsigned long long)(0xFFFF - (c.x + 32768));
      const int slot = atomicAdd(&s_n, 1);
      gkeys[slot] = k;
    }
  }
  __syncthreads();
  const int nl = s_n;
  int n2 = 1;
  while (n2 < nl) n2 <<= 1;
  const Key128 zero = {0ull, 0ull};
  Key128 *keys;
  if (n2 <= FL_SORT_LDS_KEYS) {
    for (int i = threadIdx.x; i < n2; i += blockDim.x) lds_keys[i] = i < nl ? gkeys[i] : zero;
    keys = lds_keys;
    bitonic_desc<true>(keys, n2);
  } else {
    for (int i = nl + threadIdx.x; i < n2; i += blockDim.x) gkeys[i] = zero;
    keys = gkeys;
    bitonic_desc<false>(keys, n2);
  }
  // std::unique with Match::operator== (x, y, similarity, class; template_id ignored) + compaction
  for (int base = 0; base < nl; base += blockDim.x) {
    const int i = base + threadIdx.x;
    int flag = 0;
    Key128 k = zero;
    if (i < nl) {
      k = keys[i];
      if (i == 0) flag = 1;
      else {
        const Key128 p = keys[i - 1];
        flag = !((k.hi >> 32) == (p.hi >> 32) && k.lo == p.lo);
      }
    }
    s_scan[threadIdx.x] = flag;
    __syncthreads();
    for (int d = 1; d < (int)blockDim.x; d <<= 1) {      // Hillis-Steele inclusive scan
      int v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    const int pos = s_base + s_scan[threadIdx.x] - flag;
    if (flag) {
      fl_match m;
      m.similarity = __uint_as_float((unsigned)(k.hi >> 32));
      m.template_id = 0x7FFFFFFF - (int)(k.hi & 0xFFFFFFFFu);
      m.class_idx = 0xFFFF - (int)(k.lo >> 32);
      m.y = (0xFFFF - (int)((k.lo >> 16) & 0xFFFF)) - 32768;
      m.x = (0xFFFF - (int)(k.lo & 0xFFFF)) - 32768;
      out[pos] = m;
    }
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_base += s_scan[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counters[1] = s_base;
    counters[2] = n_raw > a.cap ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------
static int launch_scan_refine_sort(fl_detector *det, int n_frames, float threshold, uint16_t *dbg, int dbg_first,
                                   int dbg_count)
{
  fl_context *ctx = det->ctx;
  const int L = det->L, M = det->M;
  {
    const FlLevelGeom &g = det->geom[L - 1];
    ScanArgs a;
    a.hdr = det->d_scan_hdr;
    a.offs = det->d_scan_off;
    a.enabled = det->d_pyr_enabled;
    a.ws = det->d_ws;
    a.ws_stride = det->ws_stride;
    for (int m = 0; m < FL_MAX_MODALITIES; ++m) a.lm_off[m] = g.lm_off[m < M ? m : 0];
    a.off_count = det->off_count;
    a.off_cand = det->off_cand;
    a.n_pyr = det->n_pyr;
    a.M = M;
    a.nchunks = (g.WH + 1023) / 1024;
    a.W = g.W;
    a.WH = g.WH;
    a.T = g.T;
    a.cap = det->cap;
    a.threshold = threshold;
    a.prune = ctx->opt.scan_prune != 0;                   // development switches (fl_context_set_option)
    a.prune_mid = ctx->opt.scan_prune_mid >= 0 ? (unsigned)ctx->opt.scan_prune_mid : FL_SCAN_PRUNE_MID;
    a.dbg = dbg;
    a.dbg_first = dbg_first;
    a.dbg_count = dbg_count;
    a.items = det->d_scan_items;
    a.n_items = det->n_scan_items;
    const int items = det->n_scan_items;
    if (items > 0) {
      a.bpf = (items + 3) / 4;
      a.n_frames = n_frames;
      dim3 grid((unsigned)(a.bpf * ((n_frames + 7) / 8) * 8));
      hipLaunchKernelGGL(k_scan, grid, dim3(256), 0, ctx->stream, a);
      FL_HIP(ctx, hipGetLastError());
    }
  }
  if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[3], ctx->stream));
  for (int l = L - 2; l >= 0; --l) {
    const FlLevelGeom &g = det->geom[l];
    RefineArgs a;
    const bool lazy_level = det->lazy;
    a.hdr = det->d_fine_hdr;
    a.feat = det->d_fine_feat;
    a.ws = det->d_ws;
    a.ws_stride = det->ws_stride;
    for (int m = 0; m < FL_MAX_MODALITIES; ++m) a.spread_off[m] = g.spread_off[m < M ? m : 0];
    a.off_count = det->off_count;
    a.off_cand = det->off_cand;
    a.M = M;
    a.Lm1 = L - 1;
    a.level = l;
    a.w = g.w;
    a.h = g.h;
    a.T = g.T;
    a.W = g.W;
    a.H = g.H;
    a.cap = det->cap;
    a.threshold = threshold;
    if (lazy_level) {
      // the level's colour quantisation and spread images, only where these candidates will look
      if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[8 + 2 * l], ctx->stream));
      hipLaunchKernelGGL(k_mark_tiles, dim3(n_frames), dim3(256), 0, ctx->stream, a, det->off_tiles);
      FL_HIP(ctx, hipGetLastError());
      int rc = fl_launch_lazy_level(det, n_frames, l);
      if (rc) return rc;
      if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[9 + 2 * l], ctx->stream));
    }
    // few workgroups per frame: candidates are typically tens per frame and idle workgroups are not
    // free (every wave still fetches the frame's counter); heavy frames just loop longer
    dim3 grid(n_frames >= 64 ? 4 : (n_frames >= 8 ? 32 : 256), n_frames);
    hipLaunchKernelGGL(k_refine, grid, dim3(256), 0, ctx->stream, a);
    FL_HIP(ctx, hipGetLastError());
  }
  if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[4], ctx->stream));
  {
    SortArgs a;
    a.pyr = det->d_pyr;
    a.ws = det->d_ws;
    a.ws_stride = det->ws_stride;
    a.off_count = det->off_count;
    a.off_cand = det->off_cand;
    a.off_keys = det->off_keys;
    a.off_match = det->off_match;
    a.cap = det->cap;
    hipLaunchKernelGGL(k_sort_unique, dim3(n_frames), dim3(1024), 0, ctx->stream, a);
    FL_HIP(ctx, hipGetLastError());
  }
  if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[5], ctx->stream));
  return FL_OK;
}

// spread/response/linearize for all levels and modalities of n_frames frames, then the match core
int fl_launch_match_core(fl_detector *det, int n_frames, float threshold)
{
  fl_context *ctx = det->ctx;
  // zero the per-frame counters
  FL_HIP(ctx, hipMemset2DAsync(det->d_ws + det->off_count, det->ws_stride, 0, 16, n_frames, ctx->stream));
  for (int l = 0; l < det->L; ++l) {
    const FlLevelGeom &g = det->geom[l];
    for (int m = 0; m < det->M; ++m) {
      if (det->lazy && l < det->L - 1) continue;           // spread of a fine level: after the scan, marked tiles only
      int rc = l == det->L - 1
                   ? fl_launch_build_lm(ctx, det->d_ws + g.quant_off[m], det->ws_stride, det->d_ws + g.lm_off[m],
                                        det->ws_stride, n_frames, g.w, g.h, g.T)
                   : fl_launch_spread(ctx, det->d_ws + g.quant_off[m], det->ws_stride, det->d_ws + g.spread_off[m],
                                      det->ws_stride, n_frames, g.w, g.h, g.T);
      if (rc) return rc;
    }
  }
  if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[2], ctx->stream));
  return launch_scan_refine_sort(det, n_frames, threshold, nullptr, 0, 0);
}

static int read_matches(fl_detector *det, int frame, fl_match *out, int cap, int *n_total)
{
  fl_context *ctx = det->ctx;
  int counters[4];
  uint8_t *ws = det->d_ws + (size_t)frame * det->ws_stride;
  FL_HIP(ctx, hipMemcpyAsync(counters, ws + det->off_count, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (counters[2]) return fl_set_error(ctx, FL_ERR_OVERFLOW, "more than %d candidates in one frame", det->cap);
  if (n_total) *n_total = counters[1];
  int n = counters[1] < cap ? counters[1] : cap;
  if (n > 0 && out) {
    FL_HIP(ctx, hipMemcpyAsync(out, ws + det->off_match, sizeof(fl_match) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return FL_OK;
}

int fl_overflow_needed(fl_detector *det, int n_frames, int *needed)
{
  fl_context *ctx = det->ctx;
  std::vector<int> c((size_t)4 * n_frames);
  FL_HIP(ctx, hipMemcpy2DAsync(c.data(), 16, det->d_ws + det->off_count, det->ws_stride, 16, n_frames, hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *needed = 0;
  for (int f = 0; f < n_frames; ++f)
    if (c[4 * f + 2] && c[4 * f] > *needed) *needed = c[4 * f];
  return FL_OK;
}

// a synchronous entry point's retry policy: on FL_ERR_OVERFLOW grow the candidate buffers to what the frame needs and
// run the call again (a valid Detector::match input never becomes an error unless the caller asked for a hard cap)
static bool grow_after_overflow(fl_detector *det, int n_frames, int attempt, int *rc)
{
  if (*rc != FL_ERR_OVERFLOW || attempt >= 6) return false;
  int needed = 0;
  if (fl_overflow_needed(det, n_frames, &needed) != FL_OK || needed <= 0) return false;
  const int g = fl_grow_candidates(det, needed);
  if (g != FL_OK) { *rc = g; return false; }
  return true;
}

// The queued entry points cannot replay a batch themselves; their caller can: after a batch in which a frame reported
// FL_ERR_OVERFLOW (fl_recognition_result.status, or FL_TOPK_OVERFLOW in the exported records), this waits for the stream,
// grows the candidate buffers to what the fullest of the last batch's first n_frames frames needs, and the caller submits
// the batch again.  *new_cap = the capacity afterwards (unchanged when no frame had overflowed).
extern "C" int fl_detector_grow_candidates(fl_detector *det, int n_frames, int *new_cap)
{
  if (!det || n_frames <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || n_frames > det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "n_frames");
  if (det->last_batch == 0) {                            // nothing matched yet: no counters to read, just report the capacity
    if (new_cap) *new_cap = det->cap;
    return FL_OK;
  }
  // only the last batch's frames have counters that mean anything: a flag left by an earlier, larger batch must not trigger a
  // stream sync, a free and a re-layout of every frame workspace
  if (n_frames > det->last_batch)
    return fl_set_error(ctx, FL_ERR_STATE, "n_frames %d > the %d frames of the last batch", n_frames, det->last_batch);
  FL_HIP(ctx, hipSetDevice(ctx->device));
  int needed = 0;
  int rc = fl_overflow_needed(det, n_frames, &needed);
  if (rc == FL_OK && needed > 0) rc = fl_grow_candidates(det, needed);
  if (new_cap) *new_cap = det->cap;
  return rc;
}

extern "C" int fl_match_quantized(fl_detector *det, const uint8_t *const *quantized, int mem, float threshold,
                                  fl_match *out, int cap, int *n_total)
{
  if (!det || !quantized || cap < 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "fl_detector_finalize first");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  for (int attempt = 0;; ++attempt) {
    for (int l = 0; l < det->L; ++l)
      for (int m = 0; m < det->M; ++m) {
        const FlLevelGeom &g = det->geom[l];
        FL_HIP(ctx, hipMemcpyAsync(det->d_ws + g.quant_off[m], quantized[l * det->M + m], (size_t)g.w * g.h,
                                   mem == FL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
      }
    det->have_times = false;
    det->lazy = false;                     // every level's quantised image was just supplied
    int rc = fl_launch_match_core(det, 1, threshold);
    if (rc) return rc;
    det->last_batch = 1;
    det->last_from_images = false;
    det->last_refinable = false;           // no depth frame belongs to this batch
    det->last_depth_base = nullptr;
    rc = read_matches(det, 0, out, cap, n_total);
    if (!grow_after_overflow(det, 1, attempt, &rc)) return rc;
  }
}

// QuantizedPyramid::quantize with a mask (`angle.copyTo(dst, mask)` linemod.cpp:455-459, `normal.copyTo(dst, mask)`
// :741-745): level l keeps a pixel only where the l-times NN-halved mask (:445-450, :733-738) is non-zero.  The chain of
// cv::resize(INTER_NEAREST) source coordinates is composed here so the mask pyramid is never materialised.
__global__ __launch_bounds__(256) void k_apply_mask(uint8_t *__restrict__ quant, const uint8_t *__restrict__ mask, int w0,
                                                    int h0, int level)
{
  const int w = w0 >> level, h = h0 >> level;
  int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const size_t o = (size_t)y * w + x;
  for (int k = level; k >= 1; --k) {
    const int sw = w0 >> (k - 1), sh = h0 >> (k - 1), dw = sw / 2, dh = sh / 2;
    x = min((int)floor(x * (1.0 / ((double)dw / sw))), sw - 1);
    y = min((int)floor(y * (1.0 / ((double)dh / sh))), sh - 1);
  }
  if (!mask[(size_t)y * w0 + x]) quant[o] = 0;
}

static int match_frame_masked_once(fl_detector *det, const uint8_t *bgr, const uint16_t *depth,
                                   const uint8_t *const *masks, int mem, float threshold, fl_match *out, int cap,
                                   int *n_total)
{
  if (!det || !bgr || cap < 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "fl_detector_finalize first");
  if (det->M == 2 && !depth) return fl_set_error(ctx, FL_ERR_INVALID, "sources.size() != modalities.size() (linemod.cpp:1364)");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const hipMemcpyKind kind = mem == FL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const size_t px = (size_t)det->w0 * det->h0;
  FL_HIP(ctx, hipMemcpyAsync(det->d_ws + det->off_bgr, bgr, px * 3, kind, ctx->stream));
  if (det->M == 2) FL_HIP(ctx, hipMemcpyAsync(det->d_ws + det->off_depth, depth, px * 2, kind, ctx->stream));
  det->have_times = false;
  // whole quantised pyramids (masks apply to them, fl_last_quantized returns them): no lazy levels here
  int rc = fl_launch_frontend(det, 1, det->d_ws + det->off_bgr, det->ws_stride,
                              (const uint16_t *)(det->d_ws + det->off_depth), det->ws_stride, false);
  if (rc) return rc;
  uint8_t *d_mask = nullptr;
  if (masks) {
    if (mem != FL_MEM_DEVICE) FL_HIP(ctx, hipMalloc(&d_mask, px * det->M));
    for (int m = 0; m < det->M && rc == FL_OK; ++m) {
      if (!masks[m]) continue;
      const uint8_t *dm = masks[m];
      if (mem != FL_MEM_DEVICE) {
        dm = d_mask + px * m;
        if (hipMemcpyAsync((void *)dm, masks[m], px, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = FL_ERR_HIP;
      }
      for (int l = 0; l < det->L && rc == FL_OK; ++l) {
        const FlLevelGeom &g = det->geom[l];
        hipLaunchKernelGGL(k_apply_mask, dim3((g.w + 63) / 64, (g.h + 3) / 4), dim3(256), 0, ctx->stream,
                           det->d_ws + g.quant_off[m], dm, det->w0, det->h0, l);
        if (hipGetLastError() != hipSuccess) rc = FL_ERR_HIP;
      }
    }
  }
  if (rc == FL_OK) rc = fl_launch_match_core(det, 1, threshold);
  if (rc == FL_OK) {
    det->last_batch = 1;
    det->last_from_images = true;
    det->last_refinable = false;           // single-frame Detector::match: fl_refine_matches belongs to fl_match_batch_submit
    det->last_depth_base = nullptr;
    rc = read_matches(det, 0, out, cap, n_total);
  } else if (rc == FL_ERR_HIP) {
    fl_set_error(ctx, rc, "mask upload / k_apply_mask launch failed");
  }
  if (d_mask) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_mask);
  }
  return rc;
}

extern "C" int fl_match_frame_masked(fl_detector *det, const uint8_t *bgr, const uint16_t *depth,
                                     const uint8_t *const *masks, int mem, float threshold, fl_match *out, int cap,
                                     int *n_total)
{
  for (int attempt = 0;; ++attempt) {
    int rc = match_frame_masked_once(det, bgr, depth, masks, mem, threshold, out, cap, n_total);
    if (!det || !grow_after_overflow(det, 1, attempt, &rc)) return rc;
  }
}

extern "C" int fl_match_frame(fl_detector *det, const uint8_t *bgr, const uint16_t *depth, int mem, float threshold,
                        fl_match *out, int cap, int *n_total)
{
  return fl_match_frame_masked(det, bgr, depth, nullptr, mem, threshold, out, cap, n_total);
}

// the sorted, de-duplicated matches of frame `frame` of the batch last queued with fl_match_batch_submit (or any other
// fl_match_* / fl_recognize_* call: the lists stay in HBM until the next batch)
extern "C" int fl_match_batch_collect(fl_detector *det, int frame, fl_match *out, int cap, int *n_total)
{
  if (!det || cap < 0 || (cap > 0 && !out)) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || frame < 0 || frame >= det->last_batch) return fl_set_error(ctx, FL_ERR_STATE, "no such frame in the last batch");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fl_update_stage_times(det, det->last_batch, nullptr);
  return read_matches(det, frame, out, cap, n_total);
}

extern "C" int fl_similarity_maps(fl_detector *det, int first, int count, uint16_t *out)
{
  if (!det || !out || first < 0 || count <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || det->last_batch < 1) return fl_set_error(ctx, FL_ERR_STATE, "no frame matched yet");
  if (first + count > det->n_pyr) return fl_set_error(ctx, FL_ERR_INVALID, "pyramid range");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const FlLevelGeom &g = det->geom[det->L - 1];
  size_t bytes = (size_t)count * g.WH * sizeof(uint16_t);
  void *d = nullptr;
  int rc = fl_scratch(ctx, bytes, &d);
  if (rc) return rc;
  FL_HIP(ctx, hipMemsetAsync(d, 0, bytes, ctx->stream));
  // re-run the scan on frame 0's resident linear memories with the debug tap on; threshold 200%
  // keeps the candidate buffer untouched in practice (counters are reset by the launcher)
  det->have_times = false;
  // frame 0's counters are saved and restored around the debug scan: its match list (fl_export_topk, fl_frame_counters,
  // fl_match_batch_collect) stays what the last match left (no candidate survives 200 %, so the list itself is untouched)
  uint8_t *saved = det->d_ws + det->off_count + 64;
  FL_HIP(ctx, hipMemcpyAsync(saved, det->d_ws + det->off_count, 16, hipMemcpyDeviceToDevice, ctx->stream));
  FL_HIP(ctx, hipMemsetAsync(det->d_ws + det->off_count, 0, 16, ctx->stream));
  const bool was_lazy = det->lazy;       // no candidate survives 200 %: nothing of the finer levels is needed, and the
  det->lazy = false;                     // batch's colour frames (lazy_bgr) may be gone by now
  rc = launch_scan_refine_sort(det, 1, 200.0f, (uint16_t *)d, first, count);
  det->lazy = was_lazy;
  if (rc) return rc;
  FL_HIP(ctx, hipMemcpyAsync(det->d_ws + det->off_count, saved, 16, hipMemcpyDeviceToDevice, ctx->stream));
  FL_HIP(ctx, hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FL_OK;
}

extern "C" int fl_frame_counters(fl_detector *det, int frame, int32_t out[4])
{
  if (!det || !out) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || frame < 0 || frame >= det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "frame");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  FL_HIP(ctx, hipMemcpyAsync(out, det->d_ws + (size_t)frame * det->ws_stride + det->off_count, 16, hipMemcpyDeviceToHost, ctx->stream));
  uint32_t bm[2 * FL_TILE_WORDS];
  const bool lazy = det->lazy && det->L > 1 && frame < det->last_batch;
  if (lazy)        // level 0's two tile bitmaps: [0] tiles whose spread bytes are read, [1] tiles whose pixels are quantised
    FL_HIP(ctx, hipMemcpyAsync(bm, det->d_ws + (size_t)frame * det->ws_stride + det->off_tiles, sizeof(bm), hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out[3] = -1;
  if (lazy) {
    int n = 0;
    for (int k = 0; k < FL_TILE_WORDS; ++k) n += __builtin_popcount(bm[FL_TILE_WORDS + k]);
    out[3] = n;
  }
  return FL_OK;
}

extern "C" int fl_last_quantized(fl_detector *det, uint8_t *out)
{
  if (!det || !out) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || det->last_batch < 1) return fl_set_error(ctx, FL_ERR_STATE, "no frame matched yet");
  if (det->lazy)
    return fl_set_error(ctx, FL_ERR_STATE, "the last batch (fl_recognize_*) quantised its finer levels only around the candidates; "
                                           "use fl_match_frame, or FL_EAGER_FRONTEND=1, for whole quantised pyramids");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  for (int l = 0; l < det->L; ++l)
    for (int m = 0; m < det->M; ++m) {
      const FlLevelGeom &g = det->geom[l];
      FL_HIP(ctx, hipMemcpyAsync(out, det->d_ws + g.quant_off[m], (size_t)g.w * g.h, hipMemcpyDeviceToHost, ctx->stream));
      out += (size_t)g.w * g.h;
    }
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FL_OK;
}

extern "C" int fl_build_linear_memories(fl_context *ctx, const uint8_t *quantized, int w, int h, int T, uint8_t *out,
                                        int mem)
{
  if (!ctx || !quantized || !out || w <= 0 || h <= 0 || T < 1 || T > 16) return FL_ERR_INVALID;
  if (w % T || h % T) return fl_set_error(ctx, FL_ERR_ASSERT, "rows/cols %% T != 0 (CV_Assert linemod.cpp:1062-1063)");
  if ((w * h) % 16) return fl_set_error(ctx, FL_ERR_ASSERT, "rows*cols %% 16 != 0 (CV_Assert linemod.cpp:981)");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nq = (size_t)w * h, nlm = 8 * fl_lm_label_stride(w, h, T);
  const uint8_t *dq = quantized;
  uint8_t *dlm = out;
  if (mem == FL_MEM_HOST) {
    void *s = nullptr;
    int rc = fl_scratch(ctx, fl_align(nq, 256) + nlm, &s);
    if (rc) return rc;
    FL_HIP(ctx, hipMemcpyAsync(s, quantized, nq, hipMemcpyHostToDevice, ctx->stream));
    dq = (const uint8_t *)s;
    dlm = (uint8_t *)s + fl_align(nq, 256);
  }
  FL_HIP(ctx, hipMemsetAsync(dlm, 0, nlm, ctx->stream));       // the zero pads
  int rc = fl_launch_build_lm(ctx, dq, 0, dlm, 0, 1, w, h, T);
  if (rc) return rc;
  if (mem == FL_MEM_HOST) {
    FL_HIP(ctx, hipMemcpyAsync(out, dlm, nlm, hipMemcpyDeviceToHost, ctx->stream));
    FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return FL_OK;
}

// ------------------------------------------------------------------------------------------
// multi-GPU helpers: fixed-size top-k export for the RCCL all-gather, and the host merge
__global__ void k_export_topk(const uint8_t *ws, size_t off_count, size_t off_match, int k, int tid_base, fl_match *out)
{
  const int n = ((const int *)(ws + off_count))[1];
  const fl_match *m = (const fl_match *)(ws + off_match);
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    fl_match r;
    if (i < n) { r = m[i]; r.template_id += tid_base; }
    else { r.x = r.y = 0; r.similarity = 0.f; r.class_idx = -1; r.template_id = -1; }
    out[i] = r;
  }
}

extern "C" int fl_export_topk(fl_detector *det, int frame, int k, int template_id_base, void *dev_out)
{
  if (!det || !dev_out || k <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || frame < 0 || frame >= det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "frame");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_export_topk, dim3(1), dim3(256), 0, ctx->stream, det->d_ws + (size_t)frame * det->ws_stride,
                     det->off_count, det->off_match, k, template_id_base, (fl_match *)dev_out);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

__global__ void k_export_topk_batch(const uint8_t *ws, size_t ws_stride, size_t off_count, size_t off_match, int k, int tid_base,
                                    fl_match *out)
{
  const uint8_t *w = ws + (size_t)blockIdx.x * ws_stride;
  const int *counters = (const int *)(w + off_count);
  const bool over = counters[2] != 0;                      // the candidate buffers overflowed: the list is not the frame's list
  const int n = over ? 0 : counters[1];
  const fl_match *m = (const fl_match *)(w + off_match);
  fl_match *o = out + (size_t)blockIdx.x * k;
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    fl_match r;
    if (i < n) { r = m[i]; r.template_id += tid_base; }
    else { r.x = r.y = 0; r.similarity = 0.f; r.class_idx = -1; r.template_id = (over && i == 0) ? FL_TOPK_OVERFLOW : -1; }
    o[i] = r;
  }
}

// the first k matches of each of the last batch's n_frames frames: dev_out[frame * k + i]; one launch
extern "C" int fl_export_topk_batch(fl_detector *det, int n_frames, int k, int template_id_base, void *dev_out)
{
  if (!det || !dev_out || k <= 0 || n_frames <= 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized || n_frames > det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "n_frames");
  if (n_frames > det->last_batch) return fl_set_error(ctx, FL_ERR_STATE, "%d frames asked for, the last batch had %d", n_frames, det->last_batch);
  FL_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_export_topk_batch, dim3(n_frames), dim3(64), 0, ctx->stream, det->d_ws, det->ws_stride, det->off_count,
                     det->off_match, k, template_id_base, (fl_match *)dev_out);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

// Match::operator< (linemod.hpp:262-267: similarity descending, then template id ascending) extended to a total order
// by (class, y, x) -- the comparator of fl_merge_topk
__device__ __forceinline__ bool match_before(const fl_match &a, const fl_match &b)
{
  if (a.similarity != b.similarity) return a.similarity > b.similarity;
  if (a.template_id != b.template_id) return a.template_id < b.template_id;
  if (a.class_idx != b.class_idx) return a.class_idx < b.class_idx;
  if (a.y != b.y) return a.y < b.y;
  return a.x < b.x;
}

// one thread per frame: the best of the ranks' first records = matches[0] of the global sort; its refinement job if the
// winner's template lies in this rank's slice
__global__ __launch_bounds__(64) void k_select_best(const fl_match *__restrict__ gathered, int n_ranks, int n_frames, int k,
                                                    int tid_first, int tid_count, fl_match *__restrict__ best, FlRefineJob *__restrict__ jobs)
{
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  fl_match b;
  b.x = b.y = 0; b.similarity = 0.f; b.class_idx = -1; b.template_id = -1;
  bool over = false;
  for (int r = 0; r < n_ranks; ++r) {
    const fl_match m = gathered[((size_t)r * n_frames + f) * k];
    if (m.template_id == FL_TOPK_OVERFLOW) over = true;
    else if (m.template_id >= 0 && (b.template_id < 0 || match_before(m, b))) b = m;
  }
  if (over) { b.x = b.y = 0; b.similarity = 0.f; b.class_idx = -1; b.template_id = FL_TOPK_OVERFLOW; }
  best[f] = b;
  FlRefineJob j;
  j.frame = -1;
  j.match = b;
  if (b.template_id >= tid_first && b.template_id < tid_first + tid_count) {
    j.frame = f;
    j.match.template_id = b.template_id - tid_first;      // class-local on this rank's detector
  }
  jobs[f] = j;
}

extern "C" int fl_select_best_batch(fl_detector *det, const void *dev_gathered, int n_ranks, int n_frames, int k, int tid_first,
                                    int tid_count, void *dev_best)
{
  if (!det || !dev_gathered || !dev_best || n_ranks <= 0 || n_frames <= 0 || k <= 0 || tid_first < 0 || tid_count < 0) return FL_ERR_INVALID;
  fl_context *ctx = det->ctx;
  if (!det->finalized) return fl_set_error(ctx, FL_ERR_STATE, "fl_detector_finalize first");
  if (n_frames > det->max_batch) return fl_set_error(ctx, FL_ERR_INVALID, "n_frames %d > max_batch %d", n_frames, det->max_batch);
  if (det->classes.size() != 1 || tid_count != det->classes[0].n_pyramids)
    return fl_set_error(ctx, FL_ERR_INVALID, "template-sharded refinement: one class per detector, tid_count = its %d pyramids",
                        det->classes.empty() ? 0 : det->classes[0].n_pyramids);
  FL_HIP(ctx, hipSetDevice(ctx->device));
  if (!det->d_jobs) FL_HIP(ctx, hipMalloc((void **)&det->d_jobs, sizeof(FlRefineJob) * (size_t)det->max_batch));
  hipLaunchKernelGGL(k_select_best, dim3((n_frames + 63) / 64), dim3(64), 0, ctx->stream, (const fl_match *)dev_gathered, n_ranks,
                     n_frames, k, tid_first, tid_count, (fl_match *)dev_best, det->d_jobs);
  FL_HIP(ctx, hipGetLastError());
  det->selected_frames = n_frames;
  return FL_OK;
}

#include <algorithm>
extern "C" int fl_merge_topk(const fl_match *gathered, int n_records, fl_match *out, int cap)
{
  if (!gathered || !out || n_records < 0 || cap < 0) return FL_ERR_INVALID;
  std::vector<fl_match> v;
  for (int i = 0; i < n_records; ++i)
    if (gathered[i].template_id >= 0) v.push_back(gathered[i]);
  std::sort(v.begin(), v.end(), [](const fl_match &a, const fl_match &b) {
    if (a.similarity != b.similarity) return a.similarity > b.similarity;
    if (a.template_id != b.template_id) return a.template_id < b.template_id;
    if (a.class_idx != b.class_idx) return a.class_idx < b.class_idx;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
  });
  auto eq = [](const fl_match &a, const fl_match &b) {
    return a.x == b.x && a.y == b.y && a.similarity == b.similarity && a.class_idx == b.class_idx;
  };
  v.erase(std::unique(v.begin(), v.end(), eq), v.end());
  int n = (int)std::min<size_t>(v.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) out[i] = v[i];
  return n;
}

// fl_merge_topk for a whole batch: gathered = what an all-gather of fl_export_topk_batch buffers gives, i.e.
// gathered[(rank * n_frames + frame) * k + i]; out[frame * cap + j], n_out[frame]
extern "C" int fl_merge_topk_batch(const fl_match *gathered, int n_ranks, int n_frames, int k, fl_match *out, int cap, int *n_out)
{
  if (!gathered || !out || !n_out || n_ranks <= 0 || n_frames <= 0 || k <= 0 || cap <= 0) return FL_ERR_INVALID;
  std::vector<fl_match> one((size_t)n_ranks * k);
  for (int f = 0; f < n_frames; ++f) {
    for (int r = 0; r < n_ranks; ++r)
      memcpy(&one[(size_t)r * k], gathered + ((size_t)r * n_frames + f) * k, sizeof(fl_match) * (size_t)k);
    const int n = fl_merge_topk(one.data(), n_ranks * k, out + (size_t)f * cap, cap);
    if (n < 0) return n;
    n_out[f] = n;
  }
  return FL_OK;
}
