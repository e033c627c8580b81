/*
 * frontend_oracle.c -- ORACLE (test infrastructure, not product): CPU restatement of the
 * quantisation front-end of cup_linemod (linemod/linemod.cpp:230-385, 434-453, 567-745 of
 * /root/reference) and of Detector::match driven from BGR + depth.
 *
 * PARITY UNPINNED.  The arithmetic of GaussianBlur / Sobel / phase / convertTo / pyrDown /
 * medianBlur / resize lives in OpenCV 3.x, which is not vendored in the reference and is
 * absent from this image; their published 8-bit algorithms are restated below and cannot be
 * checked against the real library here.  Build with -ffp-contract=off: every float operation
 * below is meant as one IEEE-754 binary32 operation, in the order written.
 */
#include "fealess_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflect101(int p, int len)
{
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

/* cv::GaussianBlur(src, dst, Size(7,7), 0, 0, BORDER_REPLICATE) on CV_8UC3 (linemod.cpp:247).
 * OpenCV 3.x: sigma <= 0 and ksize 7 selects the fixed table {.03125,.109375,.21875,.28125,...}
 * = {8,28,56,72,56,28,8}/256; the 8-bit separable path runs both passes in integers with 8
 * fractional bits each and rounds once at the end: (acc + 2^15) >> 16. */
void orc_gaussian7_bgr(const uint8_t *src, int w, int h, uint8_t *dst)
{
  static const int k[7] = {8, 28, 56, 72, 56, 28, 8};
  int *row = (int *)malloc(sizeof(int) * (size_t)w * h * 3);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int i = 0; i < 7; ++i) acc += k[i] * src[((size_t)y * w + clampi(x + i - 3, 0, w - 1)) * 3 + c];
        row[((size_t)y * w + x) * 3 + c] = acc;
      }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int i = 0; i < 7; ++i) acc += k[i] * row[((size_t)clampi(y + i - 3, 0, h - 1) * w + x) * 3 + c];
        dst[((size_t)y * w + x) * 3 + c] = (uint8_t)((acc + (1 << 15)) >> 16);
      }
  free(row);
}

/* cv::fastAtan2 / hal::fastAtan32f in degrees (OpenCV 3.x mathfuncs_core): 7th-order odd
 * polynomial on min/max, then quadrant fix-up.  Used by cv::phase(..., true) (linemod.cpp:303). */
float orc_fast_atan2(float y, float x)
{
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
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

/* quantizedOrientations (linemod.cpp:230-305) + hysteresisGradient (:307-385) */
void orc_quantized_orientations(const uint8_t *bgr, int w, int h, float weak_threshold,
                                uint8_t *dst, float *magnitude_out)
{
  size_t n = (size_t)w * h;
  uint8_t *sm = (uint8_t *)malloc(n * 3);
  float *mag = (float *)malloc(sizeof(float) * n);
  uint8_t *qu = (uint8_t *)malloc(n);
  orc_gaussian7_bgr(bgr, w, h, sm);

  for (int y = 0; y < h; ++y) {
    int ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1);
    for (int x = 0; x < w; ++x) {
      int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
      int best_dx = 0, best_dy = 0, best_mag = 0;
      int dxs[3], dys[3], mags[3];
      for (int c = 0; c < 3; ++c) {
        /* cv::Sobel(.., CV_16S, 1,0,3) / (0,1,3), BORDER_REPLICATE (linemod.cpp:248-249) */
#define S(yy, xx) ((int)sm[((size_t)(yy) * w + (xx)) * 3 + c])
        int dx = (S(ym, xp) - S(ym, xm)) + 2 * (S(y, xp) - S(y, xm)) + (S(yp, xp) - S(yp, xm));
        int dy = (S(yp, xm) - S(ym, xm)) + 2 * (S(yp, x) - S(ym, x)) + (S(yp, xp) - S(ym, xp));
#undef S
        dxs[c] = dx;
        dys[c] = dy;
        mags[c] = dx * dx + dy * dy;
      }
      /* channel of largest magnitude, ties to the first (linemod.cpp:275-292) */
      if (mags[0] >= mags[1] && mags[0] >= mags[2]) { best_dx = dxs[0]; best_dy = dys[0]; best_mag = mags[0]; }
      else if (mags[1] >= mags[0] && mags[1] >= mags[2]) { best_dx = dxs[1]; best_dy = dys[1]; best_mag = mags[1]; }
      else { best_dx = dxs[2]; best_dy = dys[2]; best_mag = mags[2]; }
      mag[(size_t)y * w + x] = (float)best_mag;
      float ang = orc_fast_atan2((float)best_dy, (float)best_dx);       /* phase(dx, dy, ag, true) */
      /* angle.convertTo(CV_8U, 16.0/360.0) (:314): float multiply, round half to even, saturate */
      float v = ang * (float)(16.0 / 360.0);
      long q = lrintf(v);
      qu[(size_t)y * w + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
  }
  /* zero the border, fold 16 -> 8 bins in the interior (:316-335) */
  for (int x = 0; x < w; ++x) { qu[x] = 0; qu[(size_t)(h - 1) * w + x] = 0; }
  for (int y = 0; y < h; ++y) { qu[(size_t)y * w] = 0; qu[(size_t)y * w + w - 1] = 0; }
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) qu[(size_t)y * w + x] &= 7;

  float threshold = weak_threshold * weak_threshold;                   /* :304 */
  memset(dst, 0, n);
  for (int r = 1; r < h - 1; ++r)
    for (int c = 1; c < w - 1; ++c) {
      if (mag[(size_t)r * w + c] > threshold) {
        int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) hist[qu[(size_t)(r + dy) * w + c + dx] & 7]++;
        /* NB: border cells of qu are 0 and interior cells are < 8, so "& 7" above is a no-op guard */
        int max_votes = 0, index = -1;
        for (int i = 0; i < 8; ++i)
          if (max_votes < hist[i]) { index = i; max_votes = hist[i]; }
        if (max_votes >= 5) dst[(size_t)r * w + c] = (uint8_t)(1 << index);
      }
    }
  if (magnitude_out) memcpy(magnitude_out, mag, sizeof(float) * n);
  free(sm);
  free(mag);
  free(qu);
}

/* cv::pyrDown(src, dst, Size(w/2, h/2)) on CV_8UC3 (linemod.cpp:441-444): separable
 * [1 4 6 4 1] in integers, BORDER_REFLECT_101, (acc + 128) >> 8. */
void orc_pyrdown_bgr(const uint8_t *src, int w, int h, uint8_t *dst)
{
  static const int k[5] = {1, 4, 6, 4, 1};
  int dw = w / 2, dh = h / 2;
  int *row = (int *)malloc(sizeof(int) * (size_t)dw * h * 3);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < dw; ++x)
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int i = 0; i < 5; ++i) acc += k[i] * src[((size_t)y * w + reflect101(2 * x + i - 2, w)) * 3 + c];
        row[((size_t)y * dw + x) * 3 + c] = acc;
      }
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x)
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int i = 0; i < 5; ++i) acc += k[i] * row[((size_t)reflect101(2 * y + i - 2, h) * dw + x) * 3 + c];
        dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((acc + 128) >> 8);
      }
  free(row);
}

/* cv::resize(src, dst, Size(w/2,h/2), 0, 0, INTER_NEAREST) (linemod.cpp:731): picks src(2y, 2x) */
void orc_resize_nn_half(const uint8_t *src, int w, int h, uint8_t *dst)
{
  int dw = w / 2, dh = h / 2;
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      /* OpenCV: ifx = 1. / ((double)dsize.width / ssize.width); sx = min(cvFloor(x * ifx), w - 1) */
      int sx = (int)floor(x * (1.0 / ((double)dw / w))), sy = (int)floor(y * (1.0 / ((double)dh / h)));
      if (sx > w - 1) sx = w - 1;
      if (sy > h - 1) sy = h - 1;
      dst[(size_t)y * dw + x] = src[(size_t)sy * w + sx];
    }
}

/* cv::medianBlur(src, dst, 5) on CV_8UC1: exact 5x5 median, replicated border (linemod.cpp:684) */
void orc_median5(const uint8_t *src, int w, int h, uint8_t *dst)
{
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int hist[256];
      memset(hist, 0, sizeof(hist));
      for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx)
          hist[src[(size_t)clampi(y + dy, 0, h - 1) * w + clampi(x + dx, 0, w - 1)]]++;
      int acc = 0, v = 0;
      for (; v < 256; ++v) { acc += hist[v]; if (acc >= 13) break; }
      dst[(size_t)y * w + x] = (uint8_t)v;
    }
}

/* NORMAL_LUT[20][20][20] (linemod/normal_lut.i, "create_depth_normal_lut.py 20").  The table
 * does not depend on its first (z) index; entry [z][y][x] = 1 << k where k is the first of the 8
 * directions (cos, sin)(k*45 deg) maximising (x-10)*cos + (y-10)*sin.  Regenerated from that
 * rule (tan 22.5 deg is irrational, so integer offsets never tie except at the centre);
 * tests/test_oracle_cpu.py::test_tables_equal_reference_text checks all 8000 bytes against the reference text when present. */
void orc_normal_lut(uint8_t lut[8000])
{
  for (int y = 0; y < 20; ++y)
    for (int x = 0; x < 20; ++x) {
      double best = -1e300;
      int bk = 0;
      for (int k = 0; k < 8; ++k) {
        double a = k * (3.14159265358979323846 / 4);
        double d = (x - 10) * cos(a) + (y - 10) * sin(a);
        if (d > best + 1e-9) { best = d; bk = k; }
      }
      for (int z = 0; z < 20; ++z) lut[(z * 20 + y) * 20 + x] = (uint8_t)(1 << bk);
    }
}

/* accumBilateral (linemod.cpp:567-579) */
static inline void accum_bilateral(long delta, long i, long j, long *A, long *b, int threshold)
{
  long f = labs(delta) < threshold ? 1 : 0;
  const long fi = f * i, fj = f * j;
  A[0] += fi * i;
  A[1] += fi * j;
  A[3] += fj * j;
  b[0] += fi * delta;
  b[1] += fj * delta;
}

/* quantizedNormals (linemod.cpp:595-685) */
void orc_quantized_normals(const uint16_t *depth, int w, int h, int distance_threshold,
                           int difference_threshold, uint8_t *dst)
{
  static __thread uint8_t lut[8000];
  static __thread int lut_ready = 0;
  if (!lut_ready) { orc_normal_lut(lut); lut_ready = 1; }
  size_t n = (size_t)w * h;
  uint8_t *tmp = (uint8_t *)calloc(n, 1);
  const int r = 5, G = 20;
  const int offx = G / 2, offy = G / 2;
  for (int y = r; y < h - r - 1; ++y)
    for (int x = r; x < w - r - 1; ++x) {
      const uint16_t *p = depth + (size_t)y * w + x;
      long d = p[0];
      uint8_t out = 0;
      if (d < distance_threshold) {
        long A[4] = {0, 0, 0, 0}, b[2] = {0, 0};
        accum_bilateral(p[-r - r * w] - d, -r, -r, A, b, difference_threshold);
        accum_bilateral(p[0 - r * w] - d, 0, -r, A, b, difference_threshold);
        accum_bilateral(p[+r - r * w] - d, +r, -r, A, b, difference_threshold);
        accum_bilateral(p[-r] - d, -r, 0, A, b, difference_threshold);
        accum_bilateral(p[+r] - d, +r, 0, A, b, difference_threshold);
        accum_bilateral(p[-r + r * w] - d, -r, +r, A, b, difference_threshold);
        accum_bilateral(p[0 + r * w] - d, 0, +r, A, b, difference_threshold);
        accum_bilateral(p[+r + r * w] - d, +r, +r, A, b, difference_threshold);
        long det = A[0] * A[3] - A[1] * A[1];
        long ddx = A[3] * b[0] - A[1] * b[1];
        long ddy = -A[1] * b[0] + A[0] * b[1];
        float nx = (float)(617 * ddx);
        float ny = (float)(617 * ddy);
        float nz = (float)(-det * d);
        float s = sqrtf(nx * nx + ny * ny + nz * nz);
        if (s > 0) {
          float inv = 1.0f / s;
          nx *= inv;
          ny *= inv;
          nz *= inv;
          int v1 = (int)(nx * offx + offx);
          int v2 = (int)(ny * offy + offy);
          int v3 = (int)(nz * G + G);
          /* Q7: v3 == 20 (nz == 0: det == 0 or d == 0 with a non-zero gradient) indexes past the
           * table in the reference (UB).  Defined here (and in the HIP path) as 0. */
          if (v1 < 0 || v1 > 19 || v2 < 0 || v2 > 19 || v3 < 0 || v3 > 19) out = 0;
          else out = lut[(v3 * 20 + v2) * 20 + v1];
        }
      }
      tmp[(size_t)y * w + x] = out;
    }
  orc_median5(tmp, w, h, dst);
  free(tmp);
}

/* Detector::match from images: ColorGradientPyramid + DepthNormalPyramid quantizers
 * (linemod.cpp:416-459, 710-745) feeding orc_match_quantized.  masks (optional, one per modality,
 * each may be NULL = empty Mat): quantize() keeps a pixel only where the mask is non-zero
 * (`angle.copyTo(dst, mask)`, :455-459, :741-745); the mask pyramid is NN-downsampled by 2 per level
 * (:445-450, :733-738); the un-masked normal image is what gets downsampled (:731). */
int orc_match_images_masked(const uint8_t *bgr, const uint16_t *depth, int w0, int h0,
                            int levels, const int *T_at_level,
                            const orc_bank *banks, int n_classes, float threshold,
                            const uint8_t *mask_color, const uint8_t *mask_depth,
                            orc_match *out, int cap, int *n_total, uint8_t *quantized_out)
{
  const int M = 2;
  uint8_t *q[32] = {0};
  uint8_t *src = (uint8_t *)malloc((size_t)w0 * h0 * 3);
  memcpy(src, bgr, (size_t)w0 * h0 * 3);
  uint8_t *normal = (uint8_t *)malloc((size_t)w0 * h0);
  uint8_t *mk[2] = {NULL, NULL};
  const uint8_t *mk_in[2] = {mask_color, mask_depth};
  for (int m = 0; m < 2; ++m)
    if (mk_in[m]) { mk[m] = (uint8_t *)malloc((size_t)w0 * h0); memcpy(mk[m], mk_in[m], (size_t)w0 * h0); }
  orc_quantized_normals(depth, w0, h0, 2000, 50, normal);             /* DepthNormal() :827-832 */
  int w = w0, h = h0;
  for (int l = 0; l < levels; ++l) {
    if (l > 0) {
      uint8_t *next = (uint8_t *)malloc((size_t)(w / 2) * (h / 2) * 3);
      orc_pyrdown_bgr(src, w, h, next);
      free(src);
      src = next;
      uint8_t *nn = (uint8_t *)malloc((size_t)(w / 2) * (h / 2));
      orc_resize_nn_half(normal, w, h, nn);
      free(normal);
      normal = nn;
      for (int m = 0; m < 2; ++m)
        if (mk[m]) {
          uint8_t *nm = (uint8_t *)malloc((size_t)(w / 2) * (h / 2));
          orc_resize_nn_half(mk[m], w, h, nm);
          free(mk[m]);
          mk[m] = nm;
        }
      w /= 2;
      h /= 2;
    }
    q[l * M] = (uint8_t *)malloc((size_t)w * h);
    orc_quantized_orientations(src, w, h, 10.0f, q[l * M], NULL);     /* ColorGradient() :515-519 */
    q[l * M + 1] = (uint8_t *)malloc((size_t)w * h);
    memcpy(q[l * M + 1], normal, (size_t)w * h);
    for (int m = 0; m < 2; ++m)
      if (mk[m])
        for (size_t i = 0; i < (size_t)w * h; ++i)
          if (!mk[m][i]) q[l * M + m][i] = 0;
  }
  free(src);
  free(normal);
  free(mk[0]);
  free(mk[1]);
  if (quantized_out) {
    uint8_t *o = quantized_out;
    for (int l = 0; l < levels; ++l)
      for (int m = 0; m < M; ++m) {
        size_t n = (size_t)(w0 >> l) * (h0 >> l);
        memcpy(o, q[l * M + m], n);
        o += n;
      }
  }
  int rc = orc_match_quantized((const uint8_t *const *)q, w0, h0, levels, M, T_at_level, banks,
                               n_classes, threshold, out, cap, n_total);
  for (int i = 0; i < levels * M; ++i) free(q[i]);
  return rc;
}

int orc_match_images(const uint8_t *bgr, const uint16_t *depth, int w0, int h0,
                     int levels, const int *T_at_level,
                     const orc_bank *banks, int n_classes, float threshold,
                     orc_match *out, int cap, int *n_total, uint8_t *quantized_out)
{
  return orc_match_images_masked(bgr, depth, w0, h0, levels, T_at_level, banks, n_classes, threshold, NULL, NULL, out, cap,
                                 n_total, quantized_out);
}

/* ---- cv::resize(..., INTER_LINEAR) as PrepareInputData uses it (obj_reco_lmicp.cpp:39-45, 248-249:
 * `TImage2Mat(img, w, h, type, true)` -> interpolation flag 1) -------------------------------------
 * OpenCV 3.x imgproc/imgwarp.cpp restated (un-vendored; not checkable here):
 *  - scale = src/dst (double); an exact 2x2 decimation is redirected to the INTER_AREA fast path
 *    `(a + b + c + d + 2) >> 2`;
 *  - otherwise resizeGeneric_: fx = (float)((dx + 0.5) * scale_x - 0.5), sx = floor(fx), clamped at both
 *    borders with a zero fraction; two taps per axis;
 *  - 8-bit: taps = saturate_cast<short>(w * 2048) (round half even), horizontal sums in int,
 *    vertical `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`;
 *  - 16-bit: float taps, horizontal `S[sx] * a0 + S[sx + cn] * a1`, vertical `S0 * b0 + S1 * b1`,
 *    saturate_cast<ushort> (round half even). */
static int resize_is_area2(int sw, int sh, int dw, int dh)
{
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  const int ix = (int)lrint(scale_x), iy = (int)lrint(scale_y);
  return fabs(scale_x - ix) < 2.220446049250313e-16 && fabs(scale_y - iy) < 2.220446049250313e-16 && ix == 2 && iy == 2;
}

static void resize_taps(int d, int ssize, int dsize, int *ofs, float *w0, float *w1)
{
  const double scale = (double)ssize / dsize;
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  *ofs = s;
  *w0 = 1.f - f;
  *w1 = f;
}

static int round_half_even_f(float v) { return (int)lrintf(v); }
static short sat_short(float v)
{
  int r = round_half_even_f(v);
  return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int cn, uint8_t *dst, int dw, int dh)
{
  if (resize_is_area2(sw, sh, dw, dh)) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int c = 0; c < cn; ++c) {
          const uint8_t *s0 = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c, *s1 = s0 + (size_t)sw * cn;
          dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
        }
    return;
  }
  for (int y = 0; y < dh; ++y) {
    int sy;
    float fy0, fy1;
    resize_taps(y, sh, dh, &sy, &fy0, &fy1);
    /* rows sy and sy + 1, clipped (resizeGeneric_Invoker: clip(sy0 - ksize2 + 1 + k, 0, ssize.height)) */
    const int y0 = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy, y1 = sy + 1 >= sh ? sh - 1 : sy + 1;
    const short b0 = sat_short(fy0 * 2048.f), b1 = sat_short(fy1 * 2048.f);
    for (int x = 0; x < dw; ++x) {
      int sx;
      float fx0, fx1;
      resize_taps(x, sw, dw, &sx, &fx0, &fx1);
      const short a0 = sat_short(fx0 * 2048.f), a1 = sat_short(fx1 * 2048.f);
      const int sx1 = sx + 1 < sw ? sx + 1 : sx;       /* a1 == 0 there (dx >= xmax: D = S[sx] * ONE) */
      for (int c = 0; c < cn; ++c) {
        const int S0 = src[((size_t)y0 * sw + sx) * cn + c] * a0 + src[((size_t)y0 * sw + sx1) * cn + c] * a1;
        const int S1 = src[((size_t)y1 * sw + sx) * cn + c] * a0 + src[((size_t)y1 * sw + sx1) * cn + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        dst[((size_t)y * dw + x) * cn + c] = (uint8_t)v;     /* `uchar(...)`: plain truncation, value is in range */
      }
    }
  }
}

void orc_resize_linear_u16(const uint16_t *src, int sw, int sh, uint16_t *dst, int dw, int dh)
{
  if (resize_is_area2(sw, sh, dw, dh)) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x) {
        const uint16_t *s0 = src + (size_t)(2 * y) * sw + 2 * x, *s1 = s0 + sw;
        dst[(size_t)y * dw + x] = (uint16_t)((s0[0] + s0[1] + s1[0] + s1[1] + 2) >> 2);
      }
    return;
  }
  for (int y = 0; y < dh; ++y) {
    int sy;
    float b0, b1;
    resize_taps(y, sh, dh, &sy, &b0, &b1);
    const int y0 = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy, y1 = sy + 1 >= sh ? sh - 1 : sy + 1;
    for (int x = 0; x < dw; ++x) {
      int sx;
      float a0, a1;
      resize_taps(x, sw, dw, &sx, &a0, &a1);
      const int sx1 = sx + 1 < sw ? sx + 1 : sx;
      float S0 = (float)src[(size_t)y0 * sw + sx] * a0;
      S0 = S0 + (float)src[(size_t)y0 * sw + sx1] * a1;
      float S1 = (float)src[(size_t)y1 * sw + sx] * a0;
      S1 = S1 + (float)src[(size_t)y1 * sw + sx1] * a1;
      float v = S0 * b0;
      v = v + S1 * b1;
      int r = round_half_even_f(v);
      dst[(size_t)y * dw + x] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : r);
    }
  }
}
