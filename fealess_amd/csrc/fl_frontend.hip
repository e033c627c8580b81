// fl_frontend.hip -- gfx950 kernels for the quantisation front-end of cup_linemod
// (reference: linemod/linemod.cpp:230-385 quantizedOrientations + hysteresisGradient,
//  :434-453 ColorGradientPyramid::pyrDown, :567-685 quantizedNormals, :721-739 NN downsample).
//
//   k_color_quantize   GaussianBlur 7x7 -> Sobel x/y -> max-magnitude channel -> fastAtan2 ->
//                      16-bin quantise -> 3x3 majority vote, fused in registers (sliding window per
//                      column, wave shuffles for x+-1): the BGR image is read from HBM once and only
//                      the one-hot byte image is written.
//   k_pyrdown_bgr      cv::pyrDown [1 4 6 4 1]^2, (acc+128)>>8, REFLECT_101
//   k_depth_quantize   bilateral 8-neighbour LSQ normal (int64) + NORMAL_LUT + exact 5x5 median of the
//                      one-hot codes (replicated border), fused in registers like k_color_quantize
//   k_resize_nn_half   src(2y, 2x)
//
// The integer stages are exact; the float stages restate OpenCV 3.x's arithmetic operator by
// operator (built with -ffp-contract=off; IEEE divide/sqrt are hipcc's default), so they equal
// oracle/frontend_oracle.c bit for bit.  OpenCV itself is not available to compare with:
// "parity unpinned" at this boundary.
#include "fl_internal.h"
#include <float.h>

#ifndef FL_CQ_DOT4
#define FL_CQ_DOT4 1               // horizontal blur taps by v_dot4_u32_u8 against constant weight words
#endif

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int p, int len)
{
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// cv::fastAtan2 in degrees (OpenCV 3.x hal::fastAtan32f), used by cv::phase (linemod.cpp:303)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ------------------------------------------------------------------------------------------
// k_color_quantize: one wavefront owns a strip of 64 adjacent image columns (60 outputs plus two
// halo columns on each side) and walks CQ_CH output rows downwards with the whole filter chain in
// registers: a 7-row ring of horizontal-blur sums per channel, 3-row rings of the smoothed pixel
// (packed B|G<<8|R<<16) and of the 16-bin code for x-1, x, x+1 -- horizontal neighbours come from
// wave shuffles, vertical ones from the rings.  No LDS, no barriers; every source byte is fetched
// once per strip (plus the 4-column / 10-row halo) through two wide loads per row and lane.
#define CQ_COLS 60
#define CQ_CH 60

// Neighbour lanes by whole-wave DPP shifts (gfx9 wave_shr:1 / wave_shl:1): one v_mov_dpp instead of the
// ds_bpermute + index arithmetic of __shfl_up / __shfl_down; lane 0 / lane 63 keep their own value, as those do.
__device__ __forceinline__ unsigned wave_from_prev(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned wave_from_next(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xF, 0xF, false); }

__device__ __forceinline__ void cq_hrow(const uint8_t *__restrict__ src, int w, int h, int row, int xc, bool interior,
                                        int *h3)
{
  const int kk[7] = {8, 28, 56, 72, 56, 28, 8};
  const uint8_t *p = src + (size_t)clampi(row, 0, h - 1) * w * 3;
  int a0 = 0, a1 = 0, a2 = 0;
  if (interior) {          // wave-uniform: no tap of any lane is clamped and 24 bytes are readable
    uint32_t wd[6];
    __builtin_memcpy(wd, p + 3 * (xc - 3), 16);
    __builtin_memcpy(wd + 4, p + 3 * (xc - 3) + 16, 8);
    // the 7 taps of a channel sit at bytes 3 t + ch of the 24-byte window: per dword one v_dot4_u32_u8 against a constant
    // word that holds the tap weights at this channel's byte positions (0 elsewhere) -- 6 dot products per channel instead of
    // 7 byte extractions + 7 multiply-adds; integer arithmetic, the same sums (<= 255 * 256)
#if FL_CQ_DOT4
    unsigned acc[3] = {0u, 0u, 0u};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        unsigned wk = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t)
          if (((3 * t + ch) >> 2) == k) wk |= (unsigned)kk[t] << (8 * ((3 * t + ch) & 3));
        acc[ch] = __builtin_amdgcn_udot4(wd[k], wk, acc[ch], false);
      }
    a0 = (int)acc[0]; a1 = (int)acc[1]; a2 = (int)acc[2];
#else
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int b0 = 3 * t, b1 = 3 * t + 1, b2 = 3 * t + 2;
      a0 += kk[t] * (int)((wd[b0 >> 2] >> (8 * (b0 & 3))) & 0xFFu);
      a1 += kk[t] * (int)((wd[b1 >> 2] >> (8 * (b1 & 3))) & 0xFFu);
      a2 += kk[t] * (int)((wd[b2 >> 2] >> (8 * (b2 & 3))) & 0xFFu);
    }
#endif
  } else {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const uint8_t *q = p + 3 * clampi(xc + t - 3, 0, w - 1);       // BORDER_REPLICATE
      a0 += kk[t] * q[0];
      a1 += kk[t] * q[1];
      a2 += kk[t] * q[2];
    }
  }
  h3[0] = a0;
  h3[1] = a1;
  h3[2] = a2;
}

#ifndef FL_CQ_WPE
#define FL_CQ_WPE 6              // 80 VGPRs, no spills: 6 waves per SIMD (measured: 5 -> 6 gives -2 % front-end time, 7/8 spill)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FL_CQ_WPE, FL_CQ_WPE))) void k_color_quantize(const uint8_t *__restrict__ bgr, size_t in_stride,
                                                        uint8_t *__restrict__ dst, size_t out_stride, int w, int h,
                                                        float threshold_sq, int nstrips, int nchunks,
                                                        float *__restrict__ mag_out, const uint32_t *__restrict__ tiles,
                                                        size_t tiles_stride)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;
  if (item >= nstrips * nchunks) return;
  const int strip = item % nstrips, chunk = item / nstrips;
  int row_lo = 0, row_hi = h - 1;
  if (tiles) {             // lazy fine level: only the tiles (and, inside a tile, the rows) some candidate's spreads read
    const uint32_t *tb = tiles + (size_t)blockIdx.z * tiles_stride;
    const int t = chunk * nstrips + strip;
    if (!((tb[FL_TILE_WORDS + (t >> 5)] >> (t & 31)) & 1u)) return;                          // wave-uniform
    const uint32_t rw = tb[2 * FL_TILE_WORDS + 32 * FL_TILE_WORDS + t];
    row_lo = (int)(rw >> 16);
    row_hi = (int)(rw & 0xFFFFu);
  }
  const uint8_t *src = bgr + (size_t)blockIdx.z * in_stride;
  uint8_t *out = dst + (size_t)blockIdx.z * out_stride;
  const int x = strip * CQ_COLS + lane - 2;
  // a sample requested outside the image is the sample at the clamped position -- for the blur's own
  // taps and for Sobel's BORDER_REPLICATE on the *smoothed* image alike (linemod.cpp:247-249)
  const int xc = clampi(x, 0, w - 1);
  const int y0 = max(chunk * CQ_CH, row_lo), y1 = min(min(h, chunk * CQ_CH + CQ_CH), row_hi + 1);
  const bool interior = __all(xc >= 3 && xc <= w - 5);

  int H[7][3];
  int center = clampi(y0 - 2, 0, h - 1);
#pragma unroll
  for (int i = 0; i < 7; ++i) cq_hrow(src, w, h, center + i - 3, xc, interior, H[i]);
  uint32_t Sl[3] = {0, 0, 0}, Sc[3] = {0, 0, 0}, Sr[3] = {0, 0, 0};
  uint32_t Qp[3] = {0, 0, 0};
  float Mg[3] = {0.f, 0.f, 0.f};
  for (int yv = y0 - 2; yv <= y1 + 1; ++yv) {
    const int c = clampi(yv, 0, h - 1);
    if (c != center) {                                 // advance the 7-row window by one source row
#pragma unroll
      for (int i = 0; i < 6; ++i) { H[i][0] = H[i + 1][0]; H[i][1] = H[i + 1][1]; H[i][2] = H[i + 1][2]; }
      cq_hrow(src, w, h, c + 3, xc, interior, H[6]);
      center = c;
    }
    // vertical pass + the single rounding of the 8-bit GaussianBlur: (acc + 2^15) >> 16
    // H <= 255 * 256, so every product fits 24 bits: __umul24 is a full-rate multiply where the generic 32-bit one
    // runs at quarter rate; the kernel is symmetric, so 4 multiplies per channel instead of 7
    int v0, v1, v2;
    {
      int vv[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        vv[ch] = (int)(__umul24(8u, (unsigned)(H[0][ch] + H[6][ch])) + __umul24(28u, (unsigned)(H[1][ch] + H[5][ch])) +
                       __umul24(56u, (unsigned)(H[2][ch] + H[4][ch])) + __umul24(72u, (unsigned)H[3][ch]));
      v0 = vv[0]; v1 = vv[1]; v2 = vv[2];
    }
    const uint32_t S = (uint32_t)((v0 + (1 << 15)) >> 16) | ((uint32_t)((v1 + (1 << 15)) >> 16) << 8) |
                       ((uint32_t)((v2 + (1 << 15)) >> 16) << 16);
    const uint32_t L = wave_from_prev(S), R = wave_from_next(S);
    Sl[0] = Sl[1]; Sl[1] = Sl[2]; Sl[2] = L;
    Sc[0] = Sc[1]; Sc[1] = Sc[2]; Sc[2] = S;
    Sr[0] = Sr[1]; Sr[1] = Sr[2]; Sr[2] = R;
    if (yv < y0) continue;                             // wave-uniform
    // Sobel 3x3 at row ys = yv - 1 (rings hold virtual rows ys-1, ys, ys+1), strongest channel,
    // fastAtan2, 16 bins (:248-303, :314)
    const int ys = yv - 1;
    int bdx = 0, bdy = 0, bmag = 0;
    {
      int dxs[3], dys[3], mags[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int sh = 8 * ch;
#define PX(v) ((int)(((v) >> sh) & 0xFFu))
        const int dx = (PX(Sr[0]) - PX(Sl[0])) + 2 * (PX(Sr[1]) - PX(Sl[1])) + (PX(Sr[2]) - PX(Sl[2]));
        const int dy = (PX(Sl[2]) - PX(Sl[0])) + 2 * (PX(Sc[2]) - PX(Sc[0])) + (PX(Sr[2]) - PX(Sr[0]));
#undef PX
        dxs[ch] = dx;
        dys[ch] = dy;
        mags[ch] = __mul24(dx, dx) + __mul24(dy, dy);      // |dx|, |dy| <= 1020: the full-rate 24-bit multiply
      }
      if (mags[0] >= mags[1] && mags[0] >= mags[2]) { bdx = dxs[0]; bdy = dys[0]; bmag = mags[0]; }
      else if (mags[1] >= mags[0] && mags[1] >= mags[2]) { bdx = dxs[1]; bdy = dys[1]; bmag = mags[1]; }
      else { bdx = dxs[2]; bdy = dys[2]; bmag = mags[2]; }
    }
    const float ang = fast_atan2_deg((float)bdy, (float)bdx);
    int qi = __float2int_rn(ang * (float)(16.0 / 360.0));        // cvRound: round half to even
    qi = qi < 0 ? 0 : (qi > 255 ? 255 : qi);
    // hysteresisGradient :316-335: border rows/cols zeroed, interior folded to 8 bins
    const bool inside = ys > 0 && ys < h - 1 && x > 0 && x < w - 1;
    // the vote counts the labels of the 3x3 neighbourhood in eight 4-bit counters; each pixel contributes the
    // counter word of its own label, the row sum x-1, x, x+1 is formed once and kept for three rows
    const uint32_t q = inside ? (uint32_t)(qi & 7) : 0u;
    const uint32_t oh = 1u << (4 * q);
    const uint32_t ohl = wave_from_prev(oh), ohr = wave_from_next(oh);
    Qp[0] = Qp[1]; Qp[1] = Qp[2]; Qp[2] = oh + ohl + ohr;
    Mg[0] = Mg[1]; Mg[1] = Mg[2]; Mg[2] = (float)bmag;
    if (yv < y0 + 2) continue;
    // 3x3 majority vote at row yo = yv - 2 (:337-384)
    const int yo = yv - 2;
    if (lane >= 2 && lane < 2 + CQ_COLS && x < w && yo < y1) {
      uint8_t res = 0;
      if (yo >= 1 && yo < h - 1 && x >= 1 && x < w - 1 && Mg[1] > threshold_sq) {
        // :337-384: the first label with the most votes wins if it has >= 5 of the 9 -- a strict majority, so it
        // is the only counter >= 5: adding 3 to every counter (max 9 + 3, no carry) sets bit 3 exactly there
        const unsigned hist = Qp[0] + Qp[1] + Qp[2];       // 8 x 4-bit counters
        const unsigned maj = (hist + 0x33333333u) & 0x88888888u;
        if (maj) res = (uint8_t)(1u << ((__ffs((int)maj) - 4) >> 2));
      }
      out[(size_t)yo * w + x] = res;
      // template extraction also wants the squared magnitude image (linemod.cpp:461-513); one frame only
      if (mag_out) mag_out[(size_t)yo * w + x] = Mg[1];
    }
  }
}

int fl_launch_quantized_orientations(fl_context *ctx, const uint8_t *bgr, size_t in_stride, uint8_t *dst,
                                     size_t out_stride, int n_frames, int w, int h, float weak_threshold)
{
  return fl_launch_quantized_orientations_mag(ctx, bgr, in_stride, dst, out_stride, n_frames, w, h, weak_threshold, nullptr);
}

static_assert(CQ_COLS == FL_TILE && CQ_CH == FL_TILE, "the lazy tile grid is k_color_quantize's work decomposition");
static int launch_color_quantize(fl_context *ctx, const uint8_t *bgr, size_t in_stride, uint8_t *dst, size_t out_stride, int n_frames,
                                 int w, int h, float weak_threshold, float *mag_out, const uint32_t *tiles, size_t tiles_stride)
{
  const int nstrips = (w + CQ_COLS - 1) / CQ_COLS, nchunks = (h + CQ_CH - 1) / CQ_CH;
  dim3 grid((nstrips * nchunks + 3) / 4, 1, n_frames);
  hipLaunchKernelGGL(k_color_quantize, grid, dim3(256), 0, ctx->stream, bgr, in_stride, dst, out_stride, w, h,
                     weak_threshold * weak_threshold, nstrips, nchunks, mag_out, tiles, tiles_stride);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

int fl_launch_quantized_orientations_mag(fl_context *ctx, const uint8_t *bgr, size_t in_stride, uint8_t *dst,
                                         size_t out_stride, int n_frames, int w, int h, float weak_threshold, float *mag_out)
{
  return launch_color_quantize(ctx, bgr, in_stride, dst, out_stride, n_frames, w, h, weak_threshold, mag_out, nullptr, 0);
}

// ------------------------------------------------------------------------------------------
// cv::pyrDown.  The stage is bound by vector-memory instructions, whose cost on gfx950 depends on alignment and lane
// stride much more than on width (tools/probes/ldwidth.hip: a 16-byte load costs ~30 cycles per wave when 4-byte
// aligned, ~70 at byte alignment or a 6-byte lane stride; a byte load at stride 5 ~25).  So:
//   k_pyrdown_pairs   interior: one thread makes TWO adjacent output pixels; their 7 source pixels are 21 bytes
//                     inside the 4-byte aligned 24-byte window starting at 12 x' - 8 -> one 16-byte + one 8-byte
//                     aligned load per source row, three 2-byte stores per thread
//   k_pyrdown_general any pixel, REFLECT_101 taps fetched one by one: the two border pixel pairs of every row
//                     (second, tiny grid) and whole images that are too small for the pair kernel
__device__ __forceinline__ void pyrdown_general_px(const uint8_t *__restrict__ src, int w, int h, int x, int y, uint8_t *o)
{
  const int k[5] = {1, 4, 6, 4, 1};
  int acc0 = 0, acc1 = 0, acc2 = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int yy = reflect101(2 * y + j - 2, h);
    int r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int xx = reflect101(2 * x + i - 2, w);
      const uint8_t *p = src + ((size_t)yy * w + xx) * 3;
      r0 += k[i] * p[0];
      r1 += k[i] * p[1];
      r2 += k[i] * p[2];
    }
    acc0 += k[j] * r0;
    acc1 += k[j] * r1;
    acc2 += k[j] * r2;
  }
  o[0] = (uint8_t)((acc0 + 128) >> 8);
  o[1] = (uint8_t)((acc1 + 128) >> 8);
  o[2] = (uint8_t)((acc2 + 128) >> 8);
}

// border_only = 0: every pixel; 1: the pixels of the first and of the last pair of each row (x in {0, 1, dw-2, dw-1})
__global__ __launch_bounds__(256) void k_pyrdown_general(const uint8_t *__restrict__ src_, size_t in_stride,
                                                         uint8_t *__restrict__ dst_, size_t out_stride, int w, int h, int border_only)
{
  const int dw = w / 2, dh = h / 2;
  int x, y;
  if (border_only) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    y = t >> 2;
    const int c = t & 3;
    x = c < 2 ? c : dw - 4 + c;
    if (y >= dh) return;
  } else {
    x = blockIdx.x * 32 + (threadIdx.x & 31);
    y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= dw || y >= dh) return;
  }
  const uint8_t *src = src_ + (size_t)blockIdx.z * in_stride;
  uint8_t *dst = dst_ + (size_t)blockIdx.z * out_stride;
  pyrdown_general_px(src, w, h, x, y, dst + ((size_t)y * dw + x) * 3);
}

// A wave owns 64 pairs of adjacent output pixels and walks PD_ROWS output rows down: output row y reads source rows
// 2y - 2 .. 2y + 2 and shares three of them with row y + 1, so the horizontal sums of a source row (six per lane: two pixels x
// three channels) are formed once and kept in a five-row register ring that trades roles by unrolling (two new source rows
// per output row instead of five, 140 vector instructions per output row instead of 312; measured: front-end 4.96 -> 4.87 ms
// per 2048 VGA frames).
#define PD_ROWS 8
__global__ __launch_bounds__(256) void k_pyrdown_pairs(const uint8_t *__restrict__ src_, size_t in_stride,
                                                       uint8_t *__restrict__ dst_, size_t out_stride, int w, int h)
{
  const int dw = w / 2, dh = h / 2;
  const int xp = blockIdx.x * 64 + (threadIdx.x & 63) + 1;                    // pair index >= 1
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PD_ROWS;
  if (xp > dw / 2 - 2 || y0 >= dh) return;                // the last pair (and an odd last pixel) belong to the border grid
  const uint8_t *src = src_ + (size_t)blockIdx.z * in_stride;
  uint8_t *dst = dst_ + (size_t)blockIdx.z * out_stride;
  // horizontal [1 4 6 4 1] sums of one source row for this lane's two output pixels (taps 0..4 and 2..6 of the seven source
  // pixels 4 xp - 2 .. 4 xp + 4; bytes 12 xp - 8 .. 12 xp + 15 hold them from byte 2 on)
  auto hrow = [&](int r, int (&o)[6]) {
    const int yy = reflect101(r, h);
    uint32_t wd[6];
    const uint8_t *p = src + (size_t)yy * w * 3 + 12 * xp - 8;
    __builtin_memcpy(wd, p, 16);
    __builtin_memcpy(wd + 4, p + 16, 8);
#define SB(b) ((int)((wd[(b) >> 2] >> (8 * ((b) & 3))) & 0xFFu))
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      o[ch] = SB(2 + ch) + 4 * SB(5 + ch) + 6 * SB(8 + ch) + 4 * SB(11 + ch) + SB(14 + ch);
      o[3 + ch] = SB(8 + ch) + 4 * SB(11 + ch) + 6 * SB(14 + ch) + 4 * SB(17 + ch) + SB(20 + ch);
    }
#undef SB
  };
  int H[5][6];                                            // ring: source rows 2y - 2 .. 2y + 2 of the current output row
  hrow(2 * y0 - 2, H[0]);
  hrow(2 * y0 - 1, H[1]);
  hrow(2 * y0, H[2]);
#pragma unroll
  for (int k = 0; k < PD_ROWS; ++k) {
    const int y = y0 + k;
    if (y < dh) {                                          // wave-uniform
      // slots (2k + j) % 5 hold source row 2y - 2 + j
      hrow(2 * y + 1, H[(2 * k + 3) % 5]);
      hrow(2 * y + 2, H[(2 * k + 4) % 5]);
      uint32_t v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int acc = H[(2 * k) % 5][c] + 4 * H[(2 * k + 1) % 5][c] + 6 * H[(2 * k + 2) % 5][c] + 4 * H[(2 * k + 3) % 5][c] +
                        H[(2 * k + 4) % 5][c];
        v[c] = (uint32_t)((acc + 128) >> 8);
      }
      uint16_t *o = (uint16_t *)(dst + ((size_t)y * dw + 2 * xp) * 3);         // 6 bytes at an even offset
      o[0] = (uint16_t)(v[0] | (v[1] << 8));
      o[1] = (uint16_t)(v[2] | (v[3] << 8));
      o[2] = (uint16_t)(v[4] | (v[5] << 8));
    }
  }
}

int fl_launch_pyrdown_bgr(fl_context *ctx, const uint8_t *src, size_t in_stride, uint8_t *dst, size_t out_stride,
                          int n_frames, int w, int h)
{
  const int dw = w / 2, dh = h / 2;
  // the pair kernel stores 16-bit words: rows and frames of the output must start at even addresses
  const bool pairs_ok = dw >= 8 && (dw & 1) == 0 && ((3 * dw) & 1) == 0 && (out_stride & 1) == 0 && (((size_t)dst) & 1) == 0;
  if (!pairs_ok) {
    dim3 grid((dw + 31) / 32, (dh + 7) / 8, n_frames);
    hipLaunchKernelGGL(k_pyrdown_general, grid, dim3(256), 0, ctx->stream, src, in_stride, dst, out_stride, w, h, 0);
  } else {
    dim3 grid((dw / 2 - 2 + 63) / 64, (dh + 4 * PD_ROWS - 1) / (4 * PD_ROWS), n_frames);             // pairs 1 .. dw/2 - 2
    hipLaunchKernelGGL(k_pyrdown_pairs, grid, dim3(256), 0, ctx->stream, src, in_stride, dst, out_stride, w, h);
    dim3 gridb((4 * dh + 255) / 256, 1, n_frames);
    hipLaunchKernelGGL(k_pyrdown_general, gridb, dim3(256), 0, ctx->stream, src, in_stride, dst, out_stride, w, h, 1);
  }
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

// ------------------------------------------------------------------------------------------
// NORMAL_LUT[20][20][20] (linemod/normal_lut.i): independent of its first index; entry [y][x]
// is 1 << k with k the first of the 8 directions k*45deg maximising (x-10)cos + (y-10)sin.
// The 400 distinct bytes are generated on the host by the same rule as oracle/frontend_oracle.c
// and uploaded once.
__constant__ uint8_t c_normal_lut[400];
static bool g_normal_lut_ready[64] = {false};

static int ensure_normal_lut(fl_context *ctx)
{
  if (ctx->device < 64 && g_normal_lut_ready[ctx->device]) return FL_OK;
  uint8_t lut[400];
  for (int y = 0; y < 20; ++y)
    for (int x = 0; x < 20; ++x) {
      double best = -1e300;
      int bk = 0;
      for (int k = 0; k < 8; ++k) {
        double a = k * (3.14159265358979323846 / 4);
        double d = (x - 10) * cos(a) + (y - 10) * sin(a);
        if (d > best + 1e-9) { best = d; bk = k; }
      }
      lut[y * 20 + x] = (uint8_t)(1 << bk);
    }
  FL_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_normal_lut), lut, sizeof(lut)));
  if (ctx->device < 64) g_normal_lut_ready[ctx->device] = true;
  return FL_OK;
}

// accumBilateral (linemod.cpp:567-579) over the 8 ring neighbours, restated on what it adds up to.  With
// f_k = |delta_k| < threshold and offsets (i, j) in {-5, 0, 5}:  A[0] = sum f*i*i = 25 * #{f_k : i != 0},
// A[3] likewise for j, A[1] = sum f*i*j = 25 * (f(-,-) + f(+,+) - f(+,-) - f(-,+)), b[0] = sum f*i*delta =
// 5 * (sum over i = +5 of f*delta  -  sum over i = -5), b[1] likewise for j.  The same integers as the reference's
// `long` accumulation (|A| <= 150, |b| <= 30 * (threshold - 1)), in 32-bit registers.
//
// k_depth_quantize: quantizedNormals (linemod.cpp:595-683) fused with the medianBlur(5) that ends
// it (:684).  Same execution shape as k_color_quantize: a wavefront owns 64 adjacent columns (60
// outputs + 2 halo columns per side), walks DQ_CH output rows and keeps the last five rows in registers.
// The codes are 0 or one-hot, i.e. 9 classes ordered like their byte values, and the exact 5x5 median is
// the first class whose CUMULATIVE count reaches 13.  A pixel of class c therefore contributes the
// pattern "1 in every 4-bit field i >= c" (fields 0..7; class 8's own field would always hold 25 and is
// implied): five rows add up to a column word (fields <= 5), the x-1, x+1 / x-2, x+2 columns come by
// whole-wave DPP shifts, 3 + 2 columns are added in 4-bit fields (<= 15, <= 10), and the ">= 13" test runs on
// the even / odd fields widened to bytes (sum + 115 sets bit 7).  The flags are monotone in the class, so
// median class = 8 - popcount.  The un-medianed normal image never exists in memory.
#define DQ_COLS 60
#define DQ_CH 60

__device__ __forceinline__ unsigned dq_pattern(unsigned cls) { return cls >= 8u ? 0u : 0x11111111u << (4u * cls); }

__device__ __forceinline__ unsigned dq_normal_pattern(const uint16_t *__restrict__ depth, int w, int h, int y, int x,
                                                      int distance_threshold, int difference_threshold, const uint8_t *lut)
{
  const int r = 5;
  if (!(y >= r && y < h - r - 1 && x >= r && x < w - r - 1)) return dq_pattern(0);   // loop bounds :619, :624
  const uint16_t *p = depth + (__umul24((unsigned)y, (unsigned)w) + (unsigned)x);   // (rows, columns < 2^24: the full-rate multiply)
  const int d = p[0];
  if (!(d < distance_threshold)) return dq_pattern(0);
  const int t = difference_threshold;
  const unsigned span = t > 0 ? (unsigned)(2 * t - 1) : 0u;   // t <= 0 admits nothing
  // neighbours 0..7: (-,-) (0,-) (+,-) (-,0) (+,0) (-,+) (0,+) (+,+)
  const int e0 = (int)p[-r - r * w] - d, e1 = (int)p[0 - r * w] - d, e2 = (int)p[+r - r * w] - d, e3 = (int)p[-r] - d,
            e4 = (int)p[+r] - d, e5 = (int)p[-r + r * w] - d, e6 = (int)p[0 + r * w] - d, e7 = (int)p[+r + r * w] - d;
  // |e| < t  <=>  (unsigned)(e + t - 1) < 2t - 1                                                    :569
  const bool g0 = (unsigned)(e0 + t - 1) < span, g1 = (unsigned)(e1 + t - 1) < span,
             g2 = (unsigned)(e2 + t - 1) < span, g3 = (unsigned)(e3 + t - 1) < span,
             g4 = (unsigned)(e4 + t - 1) < span, g5 = (unsigned)(e5 + t - 1) < span,
             g6 = (unsigned)(e6 + t - 1) < span, g7 = (unsigned)(e7 + t - 1) < span;
  const int f0 = g0, f1 = g1, f2 = g2, f3 = g3, f4 = g4, f5 = g5, f6 = g6, f7 = g7;
  const int m0 = g0 ? e0 : 0, m1 = g1 ? e1 : 0, m2 = g2 ? e2 : 0, m3 = g3 ? e3 : 0, m4 = g4 ? e4 : 0, m5 = g5 ? e5 : 0,
            m6 = g6 ? e6 : 0, m7 = g7 ? e7 : 0;
  const int corners = f0 + f2 + f5 + f7;
  // every factor below fits 24 bits, so the products are v_mul_i32_i24 (full rate) instead of the quarter-rate 32-bit multiply
  // the compiler has to assume: |A| <= 150, det <= 22500, and with t <= 400 |b| <= 30 * 399, |dd| < 3.0e6, d < 65536
  const int A0 = __mul24(25, corners + f3 + f4), A3 = __mul24(25, corners + f1 + f6), A1 = __mul24(25, (f0 + f7) - (f2 + f5));
  const int det = __mul24(A0, A3) - __mul24(A1, A1);       // <= 22500
  float nx, ny, nz;
  if (t <= 400) {
    // |b| <= 30 * 399, |dd| <= 250 * |b| < 3.0e6: 617 * dd and det * d (<= 22500 * 65535) fit in int32
    const int b0 = __mul24(5, (m2 + m4 + m7) - (m0 + m3 + m5)), b1 = __mul24(5, (m5 + m6 + m7) - (m0 + m1 + m2));
    const int ddx = __mul24(A3, b0) - __mul24(A1, b1), ddy = __mul24(A0, b1) - __mul24(A1, b0);
    nx = (float)__mul24(617, ddx);
    ny = (float)__mul24(617, ddy);
    nz = (float)(-__mul24(det, d));
  } else {
    const int b0 = 5 * ((m2 + m4 + m7) - (m0 + m3 + m5)), b1 = 5 * ((m5 + m6 + m7) - (m0 + m1 + m2));
    const long long ddx = (long long)A3 * b0 - (long long)A1 * b1, ddy = -(long long)A1 * b0 + (long long)A0 * b1;
    nx = (float)(617LL * ddx);
    ny = (float)(617LL * ddy);
    nz = (float)(-(long long)det * d);
  }
  const float s = sqrtf(nx * nx + ny * ny + nz * nz);
  if (!(s > 0)) return dq_pattern(0);                       // shadows of the depth sensor
  // (Round 4 measured the LUT indices from a v_rsq_f32 ESTIMATE with the correctly rounded sqrtf + division only for lanes
  // within 4e-5 of a bin edge -- exact by construction, bit-equal in the tests -- and it was 2 % SLOWER (front-end 18.04
  // against 17.70 ms per 4096 eager frames): the test for "near an edge" costs what the two correctly rounded operations
  // cost, and a wavefront takes the exact branch whenever one of its 64 lanes is near an edge.  profiles/README.md.)
  const float inv = 1.0f / s;
  nx *= inv;
  ny *= inv;
  nz *= inv;
  const int v1 = (int)(nx * 10 + 10);
  const int v2 = (int)(ny * 10 + 10);
  const int v3 = (int)(nz * 20 + 20);
  // Q7: v3 == 20 is an out-of-bounds read in the reference; defined as 0 here
  if (!(v1 >= 0 && v1 <= 19 && v2 >= 0 && v2 <= 19 && v3 >= 0 && v3 <= 19)) return dq_pattern(0);
  const unsigned v = lut[v2 * 20 + v1];
  return dq_pattern(v ? (unsigned)(32 - __clz(v)) : 0u);   // 0 -> class 0, 1<<k -> class k+1 (ascending in value)
}

__global__ __launch_bounds__(256) void k_depth_quantize(const uint16_t *__restrict__ depth_, size_t in_stride,
                                                        uint8_t *__restrict__ dst_, size_t out_stride, int w, int h,
                                                        int distance_threshold, int difference_threshold, int nstrips,
                                                        int nchunks)
{
  __shared__ uint8_t s_lut[400];                           // NORMAL_LUT's 20x20 face: a per-lane lookup every row
  for (int k = threadIdx.x; k < 400; k += 256) s_lut[k] = c_normal_lut[k];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;
  if (item >= nstrips * nchunks) return;
  const int strip = item % nstrips, chunk = item / nstrips;
  const uint16_t *depth = depth_ + (size_t)blockIdx.z * in_stride;
  uint8_t *dst = dst_ + (size_t)blockIdx.z * out_stride;
  const int x = strip * DQ_COLS + lane - 2;
  const int xc = clampi(x, 0, w - 1);                      // medianBlur replicates the border
  const int y0 = chunk * DQ_CH, y1 = min(h, y0 + DQ_CH);
  unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;         // cumulative-count patterns of the last five rows, p0 newest
  int center = -1;
  unsigned pat = 0;
  for (int yv = y0 - 2; yv <= y1 + 1; ++yv) {
    const int c = clampi(yv, 0, h - 1);
    if (c != center) {                                     // wave-uniform
      pat = dq_normal_pattern(depth, w, h, c, xc, distance_threshold, difference_threshold, s_lut);
      center = c;
    }
    p4 = p3; p3 = p2; p2 = p1; p1 = p0; p0 = pat;
    if (yv < y0 + 2) continue;
    const unsigned col = p0 + p1 + p2 + p3 + p4;           // fields <= 5
    const unsigned a1 = wave_from_next(col), b1 = wave_from_prev(col), a2 = wave_from_next(a1), b2 = wave_from_prev(b1);
    const unsigned s3 = col + a1 + b1, s2 = a2 + b2;       // fields <= 15, <= 10
    const unsigned te = (s3 & 0x0F0F0F0Fu) + (s2 & 0x0F0F0F0Fu) + 0x73737373u;
    const unsigned to = ((s3 >> 4) & 0x0F0F0F0Fu) + ((s2 >> 4) & 0x0F0F0F0Fu) + 0x73737373u;
    const int idx = 8 - __popc((te & 0x80808080u) | ((to & 0x80808080u) >> 1));
    const int yo = yv - 2;
    if (lane >= 2 && lane < 2 + DQ_COLS && x < w) dst[__umul24((unsigned)yo, (unsigned)w) + (unsigned)x] = idx ? (uint8_t)(1u << (idx - 1)) : 0;
  }
}

int fl_launch_quantized_normals(fl_context *ctx, const uint16_t *depth, size_t in_stride, uint8_t *dst,
                                size_t out_stride, uint8_t *tmp, size_t tmp_stride, int n_frames, int w, int h,
                                int distance_threshold, int difference_threshold)
{
  int rc = ensure_normal_lut(ctx);
  if (rc) return rc;
  (void)tmp;
  (void)tmp_stride;
  const int nstrips = (w + DQ_COLS - 1) / DQ_COLS, nchunks = (h + DQ_CH - 1) / DQ_CH;
  dim3 grid((nstrips * nchunks + 3) / 4, 1, n_frames);
  hipLaunchKernelGGL(k_depth_quantize, grid, dim3(256), 0, ctx->stream, depth, in_stride / sizeof(uint16_t), dst, out_stride,
                     w, h, distance_threshold, difference_threshold, nstrips, nchunks);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

// cv::resize INTER_NEAREST to half size (linemod.cpp:731): sx = min(floor(x * (1 / (dw / w))), w - 1).  For even w and h
// that is src(2y, 2x); then one thread makes 4 output pixels from one aligned 8-byte load and stores one dword (the
// one-pixel-per-thread version spent its time on byte loads and byte stores).  Other sizes take the general kernel.
__global__ __launch_bounds__(256) void k_resize_nn_half4(const uint8_t *__restrict__ src_, size_t in_stride,
                                                         uint8_t *__restrict__ dst_, size_t out_stride, int w, int h)
{
  const int dw = w / 2, dh = h / 2;
  const int x4 = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (4 * x4 >= dw || y >= dh) return;
  const uint8_t *src = src_ + (size_t)blockIdx.z * in_stride;
  uint8_t *dst = dst_ + (size_t)blockIdx.z * out_stride;
  uint2 v;
  __builtin_memcpy(&v, __builtin_assume_aligned(src + (size_t)(2 * y) * w + 8 * x4, 8), 8);
  const uint32_t o = (v.x & 0xFFu) | ((v.x >> 8) & 0xFF00u) | ((v.y & 0xFFu) << 16) | ((v.y << 8) & 0xFF000000u);
  *(uint32_t *)(dst + (size_t)y * dw + 4 * x4) = o;
}

__global__ __launch_bounds__(256) void k_resize_nn_half(const uint8_t *__restrict__ src_, size_t in_stride,
                                                        uint8_t *__restrict__ dst_, size_t out_stride, int w, int h)
{
  const int dw = w / 2, dh = h / 2;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const uint8_t *src = src_ + (size_t)blockIdx.z * in_stride;
  uint8_t *dst = dst_ + (size_t)blockIdx.z * out_stride;
  int sx = (int)floor(x * (1.0 / ((double)dw / w))), sy = (int)floor(y * (1.0 / ((double)dh / h)));
  sx = min(sx, w - 1);
  sy = min(sy, h - 1);
  dst[(size_t)y * dw + x] = src[(size_t)sy * w + sx];
}

int fl_launch_resize_nn_half(fl_context *ctx, const uint8_t *src, size_t in_stride, uint8_t *dst, size_t out_stride,
                             int n_frames, int w, int h)
{
  // even sizes: 1 / ((w/2) / w) == 2 exactly, so the source pixel is (2y, 2x); w % 8 == 0 keeps the 8-byte loads aligned
  const bool fast = (w % 8) == 0 && (h % 2) == 0 && ((in_stride | out_stride) & 7) == 0 && ((((size_t)src) | ((size_t)dst)) & 7) == 0;
  if (fast) {
    dim3 grid((w / 8 + 63) / 64, (h / 2 + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_resize_nn_half4, grid, dim3(256), 0, ctx->stream, src, in_stride, dst, out_stride, w, h);
  } else {
    dim3 grid((w / 2 + 63) / 64, (h / 2 + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_resize_nn_half, grid, dim3(256), 0, ctx->stream, src, in_stride, dst, out_stride, w, h);
  }
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}

// ------------------------------------------------------------------------------------------
// Modality::process + pyrDown + quantize for every level (linemod.cpp:1369-1416) of n_frames frames
// resident in the detector workspace.  Default modality parameters (linemod.cpp:515-519, 827-832).
int fl_launch_frontend(fl_detector *det, int n_frames, const uint8_t *bgr, size_t bgr_stride, const uint16_t *depth,
                       size_t depth_stride, bool allow_lazy)
{
  fl_context *ctx = det->ctx;
  int rc;
  // Lazy fine levels: everything that feeds the coarsest level (colour pyramid, depth quantisation and its NN
  // pyramid, the coarsest colour quantisation) runs here; the colour quantisation of the finer levels waits for the
  // scan's candidates (fl_launch_lazy_level).
  det->lazy = allow_lazy && det->lazy_capable && !det->eager_env;
  det->lazy_bgr = bgr;
  det->lazy_bgr_stride = bgr_stride;
  for (int l = 0; l < det->L; ++l) {
    const FlLevelGeom &g = det->geom[l];
    if (l > 0) {
      const FlLevelGeom &p = det->geom[l - 1];
      rc = fl_launch_pyrdown_bgr(ctx, l == 1 ? bgr : det->d_ws + p.bgr_off, l == 1 ? bgr_stride : det->ws_stride,
                                 det->d_ws + g.bgr_off, det->ws_stride, n_frames, p.w, p.h);
      if (rc) return rc;
      if (det->M > 1) {
        rc = fl_launch_resize_nn_half(ctx, det->d_ws + p.quant_off[1], det->ws_stride, det->d_ws + g.quant_off[1],
                                      det->ws_stride, n_frames, p.w, p.h);
        if (rc) return rc;
      }
    } else if (det->M > 1) {
      rc = fl_launch_quantized_normals(ctx, depth, depth_stride,
                                       det->d_ws + g.quant_off[1], det->ws_stride, det->d_ws + det->off_tmp,
                                       det->ws_stride, n_frames, g.w, g.h, 2000, 50);
      if (rc) return rc;
    }
    if (det->lazy && l < det->L - 1) continue;
    rc = fl_launch_quantized_orientations(ctx, l == 0 ? bgr : det->d_ws + g.bgr_off, l == 0 ? bgr_stride : det->ws_stride,
                                          det->d_ws + g.quant_off[0], det->ws_stride, n_frames, g.w, g.h, 10.0f);
    if (rc) return rc;
  }
  if (det->have_times) FL_HIP(ctx, hipEventRecord(det->ev[1], ctx->stream));
  return FL_OK;
}

// Fine level `level` of a lazy batch, after k_mark_tiles: colour quantisation of the tiles in the "quantised pixels
// needed" bitmap, then both modalities' spread images in the rows/columns of the "spread bytes needed" bitmap.
int fl_launch_lazy_level(fl_detector *det, int n_frames, int level)
{
  fl_context *ctx = det->ctx;
  const FlLevelGeom &g = det->geom[level];
  const uint32_t *tiles = (const uint32_t *)(det->d_ws + det->off_tiles) + (size_t)level * FL_TILE_BLOCK_WORDS;
  const size_t tstride = det->ws_stride / sizeof(uint32_t);
  if (det->poison_env) {   // dev aid: a byte read outside the marked tiles must not look plausible
    FL_HIP(ctx, hipMemset2DAsync(det->d_ws + g.quant_off[0], det->ws_stride, 0xFF, (size_t)g.w * g.h, n_frames, ctx->stream));
    for (int m = 0; m < det->M; ++m)
      FL_HIP(ctx, hipMemset2DAsync(det->d_ws + g.spread_off[m], det->ws_stride, 0xFF, (size_t)g.w * g.h, n_frames, ctx->stream));
  }
  int rc = launch_color_quantize(ctx, level == 0 ? det->lazy_bgr : det->d_ws + g.bgr_off, level == 0 ? det->lazy_bgr_stride : det->ws_stride,
                                 det->d_ws + g.quant_off[0], det->ws_stride, n_frames, g.w, g.h, 10.0f, nullptr,
                                 tiles, tstride);
  if (rc) return rc;
  for (int m = 0; m < det->M; ++m) {
    rc = fl_launch_spread_tiles(ctx, det->d_ws + g.quant_off[m], det->ws_stride, det->d_ws + g.spread_off[m], det->ws_stride, n_frames,
                                g.w, g.h, g.T, tiles, tstride);
    if (rc) return rc;
  }
  return FL_OK;
}

// ---- single-image stage entry points ----------------------------------------------------------
template <typename F>
static int run_stage(fl_context *ctx, const void *in, size_t in_bytes, void *out, size_t out_bytes, size_t tmp_bytes,
                     int mem, F body)
{
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const uint8_t *din = (const uint8_t *)in;
  uint8_t *dout = (uint8_t *)out, *dtmp = nullptr;
  void *s = nullptr;
  size_t need = fl_align(tmp_bytes, 256) + (mem == FL_MEM_HOST ? fl_align(in_bytes, 256) + fl_align(out_bytes, 256) : 0);
  if (need) {
    int rc = fl_scratch(ctx, need, &s);
    if (rc) return rc;
  }
  dtmp = (uint8_t *)s;
  if (mem == FL_MEM_HOST) {
    uint8_t *b = (uint8_t *)s + fl_align(tmp_bytes, 256);
    FL_HIP(ctx, hipMemcpyAsync(b, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    din = b;
    dout = b + fl_align(in_bytes, 256);
  }
  int rc = body(din, dout, dtmp);
  if (rc) return rc;
  if (mem == FL_MEM_HOST) {
    FL_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return FL_OK;
}

extern "C" int fl_quantized_orientations(fl_context *ctx, const uint8_t *bgr, int w, int h, float weak_threshold,
                                         uint8_t *dst, int mem)
{
  if (!ctx || !bgr || !dst || w < 3 || h < 3) return FL_ERR_INVALID;
  return run_stage(ctx, bgr, (size_t)w * h * 3, dst, (size_t)w * h, 0, mem,
                   [&](const uint8_t *i, uint8_t *o, uint8_t *) {
                     return fl_launch_quantized_orientations(ctx, i, 0, o, 0, 1, w, h, weak_threshold);
                   });
}

extern "C" int fl_quantized_normals(fl_context *ctx, const uint16_t *depth, int w, int h, int distance_threshold,
                                    int difference_threshold, uint8_t *dst, int mem)
{
  if (!ctx || !depth || !dst || w <= 0 || h <= 0) return FL_ERR_INVALID;
  return run_stage(ctx, depth, (size_t)w * h * 2, dst, (size_t)w * h, (size_t)w * h, mem,
                   [&](const uint8_t *i, uint8_t *o, uint8_t *t) {
                     return fl_launch_quantized_normals(ctx, (const uint16_t *)i, 0, o, 0, t, 0, 1, w, h,
                                                        distance_threshold, difference_threshold);
                   });
}

extern "C" int fl_pyrdown_bgr(fl_context *ctx, const uint8_t *src, int w, int h, uint8_t *dst, int mem)
{
  if (!ctx || !src || !dst || w < 2 || h < 2) return FL_ERR_INVALID;
  return run_stage(ctx, src, (size_t)w * h * 3, dst, (size_t)(w / 2) * (h / 2) * 3, 0, mem,
                   [&](const uint8_t *i, uint8_t *o, uint8_t *) { return fl_launch_pyrdown_bgr(ctx, i, 0, o, 0, 1, w, h); });
}

// ------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) as CObjRecoLmICP::PrepareInputData applies it to frames that are not 640
// wide (obj_reco_lmicp.cpp:39-45, 229-249).  OpenCV 3.x semantics restated in oracle/frontend_oracle.c
// (orc_resize_linear_*): exact 2x2 decimation takes the INTER_AREA fast path, everything else two taps
// per axis -- 11-bit fixed point for 8-bit data, float for 16-bit data.  One thread per output pixel;
// this stage runs once per frame ahead of the batch and is HBM-bound (reads 4 source pixels).
struct FlResizeTap { int ofs; float w0, w1; };
__device__ __forceinline__ FlResizeTap fl_resize_tap(int d, int ssize, double scale)
{
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  return {s, 1.f - f, f};
}
__device__ __forceinline__ int fl_sat_short(float v)
{
  const int r = (int)rintf(v);                           // saturate_cast<short>(float): round half even
  return r < -32768 ? -32768 : r > 32767 ? 32767 : r;
}

template <typename T, int CN>
__global__ __launch_bounds__(256) void k_resize_linear(const T *__restrict__ src, int sw, int sh, T *__restrict__ dst, int dw,
                                                       int dh, double scale_x, double scale_y, int area2)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  if (area2) {                                           // (a + b + c + d + 2) >> 2
    const T *s0 = src + ((size_t)(2 * y) * sw + 2 * x) * CN, *s1 = s0 + (size_t)sw * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) dst[((size_t)y * dw + x) * CN + c] = (T)(((int)s0[c] + s0[CN + c] + s1[c] + s1[CN + c] + 2) >> 2);
    return;
  }
  const FlResizeTap ty = fl_resize_tap(y, sh, scale_y), tx = fl_resize_tap(x, sw, scale_x);
  const int y0 = ty.ofs, y1 = min(ty.ofs + 1, sh - 1), x0 = tx.ofs, x1 = min(tx.ofs + 1, sw - 1);
  const T *r0 = src + (size_t)y0 * sw * CN, *r1 = src + (size_t)y1 * sw * CN;
  if (sizeof(T) == 1) {
    const int a0 = fl_sat_short(tx.w0 * 2048.f), a1 = fl_sat_short(tx.w1 * 2048.f);
    const int b0 = fl_sat_short(ty.w0 * 2048.f), b1 = fl_sat_short(ty.w1 * 2048.f);
#pragma unroll
    for (int c = 0; c < CN; ++c) {
      const int S0 = (int)r0[x0 * CN + c] * a0 + (int)r0[x1 * CN + c] * a1;
      const int S1 = (int)r1[x0 * CN + c] * a0 + (int)r1[x1 * CN + c] * a1;
      dst[((size_t)y * dw + x) * CN + c] = (T)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CN; ++c) {
      float S0 = (float)r0[x0 * CN + c] * tx.w0;
      S0 = S0 + (float)r0[x1 * CN + c] * tx.w1;
      float S1 = (float)r1[x0 * CN + c] * tx.w0;
      S1 = S1 + (float)r1[x1 * CN + c] * tx.w1;
      float v = S0 * ty.w0;
      v = v + S1 * ty.w1;
      const int r = (int)rintf(v);
      dst[((size_t)y * dw + x) * CN + c] = (T)(r < 0 ? 0 : r > 65535 ? 65535 : r);
    }
  }
}

template <typename T, int CN>
static int fl_resize_linear(fl_context *ctx, const T *src, int sw, int sh, T *dst, int dw, int dh, int mem)
{
  if (!ctx || !src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return FL_ERR_INVALID;
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  const int area2 = fabs(scale_x - 2.0) < 2.220446049250313e-16 && fabs(scale_y - 2.0) < 2.220446049250313e-16;
  return run_stage(ctx, src, (size_t)sw * sh * CN * sizeof(T), dst, (size_t)dw * dh * CN * sizeof(T), 0, mem,
                   [&](const uint8_t *i, uint8_t *o, uint8_t *) {
                     hipLaunchKernelGGL((k_resize_linear<T, CN>), dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, ctx->stream,
                                        (const T *)i, sw, sh, (T *)o, dw, dh, scale_x, scale_y, area2);
                     FL_HIP(ctx, hipGetLastError());
                     return (int)FL_OK;
                   });
}

// device-to-device launchers for the pipeline (fl_recognize_batch_zoom)
int fl_launch_resize_linear_bgr8(fl_context *ctx, const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
  return fl_resize_linear<uint8_t, 3>(ctx, src, sw, sh, dst, dw, dh, FL_MEM_DEVICE);
}
int fl_launch_resize_linear_u16(fl_context *ctx, const uint16_t *src, int sw, int sh, uint16_t *dst, int dw, int dh)
{
  return fl_resize_linear<uint16_t, 1>(ctx, src, sw, sh, dst, dw, dh, FL_MEM_DEVICE);
}

extern "C" int fl_resize_linear_bgr8(fl_context *ctx, const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, int mem)
{
  return fl_resize_linear<uint8_t, 3>(ctx, src, sw, sh, dst, dw, dh, mem);
}

extern "C" int fl_resize_linear_u16(fl_context *ctx, const uint16_t *src, int sw, int sh, uint16_t *dst, int dw, int dh, int mem)
{
  return fl_resize_linear<uint16_t, 1>(ctx, src, sw, sh, dst, dw, dh, mem);
}
