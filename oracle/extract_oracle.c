/*
 * extract_oracle.c -- ORACLE (test infrastructure only, see fealess_oracle.h) for template extraction,
 * Detector::addTemplate (linemod/linemod.cpp:1579-1615),
 * ColorGradientPyramid::extractTemplate (:461-513), DepthNormalPyramid::extractTemplate (:747-825),
 * QuantizedPyramid::selectScatteredFeatures (:135-164), cropTemplates (:52-96), the pyrDown of both
 * quantized pyramids (:434-453, :721-739).
 *
 * PARITY UNPINNED, as everywhere: the reference has no fixtures.  The two OpenCV calls on this path are
 * restated from OpenCV 3.x: cv::erode with the default 3x3 rectangle (n iterations of a rectangle = one
 * (2n+1)^2 rectangle, BORDER_REPLICATE as the caller asks) and cv::distanceTransform(DIST_C, 3) (two-pass
 * chamfer with a = b = 1 in 16.16 fixed point, INIT_DIST0 = INT_MAX >> 2 outside the image, result capped and
 * scaled to float).  tests/test_oracle_cpu.py pins both against scipy.ndimage.
 */
#include "fealess_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cv::erode(src, dst, Mat(), Point(-1,-1), iterations, BORDER_REPLICATE): min over a (2*it+1)^2 window */
void orc_erode_rect(const uint8_t *src, int w, int h, int iterations, uint8_t *dst)
{
  const int r = iterations;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int m = 255;
      for (int dy = -r; dy <= r; ++dy) {
        int yy = y + dy < 0 ? 0 : y + dy >= h ? h - 1 : y + dy;
        for (int dx = -r; dx <= r; ++dx) {
          int xx = x + dx < 0 ? 0 : x + dx >= w ? w - 1 : x + dx;
          int v = src[(size_t)yy * w + xx];
          if (v < m) m = v;
        }
      }
      dst[(size_t)y * w + x] = (uint8_t)m;
    }
}

/* cv::distanceTransform(src, dst, DIST_C, 3): OpenCV 3.x distanceTransform_3x3 with HV = DIAG = 1 << 16 */
void orc_distance_transform_c3(const uint8_t *src, int w, int h, float *dst)
{
  const int INIT = INT_MAX >> 2, HV = 1 << 16, DIAG = 1 << 16, step = w + 2;
  const float scale = 1.f / (1 << 16);
  int *buf = (int *)malloc(sizeof(int) * (size_t)step * (h + 2));
  for (size_t i = 0; i < (size_t)step * (h + 2); ++i) buf[i] = INIT;
  for (int i = 0; i < h; ++i) {                              /* forward pass */
    int *tmp = buf + (size_t)(i + 1) * step + 1;
    const uint8_t *s = src + (size_t)i * w;
    for (int j = 0; j < w; ++j) {
      if (!s[j]) {
        tmp[j] = 0;
      } else {
        int t0 = tmp[j - step - 1] + DIAG, t = tmp[j - step] + HV;
        if (t0 > t) t0 = t;
        t = tmp[j - step + 1] + DIAG;
        if (t0 > t) t0 = t;
        t = tmp[j - 1] + HV;
        if (t0 > t) t0 = t;
        tmp[j] = t0;
      }
    }
  }
  for (int i = h - 1; i >= 0; --i) {                         /* backward pass */
    int *tmp = buf + (size_t)(i + 1) * step + 1;
    float *d = dst + (size_t)i * w;
    for (int j = w - 1; j >= 0; --j) {
      int t0 = tmp[j];
      if (t0 > HV) {
        int t = tmp[j + step + 1] + DIAG;
        if (t0 > t) t0 = t;
        t = tmp[j + step] + HV;
        if (t0 > t) t0 = t;
        t = tmp[j + step - 1] + DIAG;
        if (t0 > t) t0 = t;
        t = tmp[j + 1] + HV;
        if (t0 > t) t0 = t;
        tmp[j] = t0;
      }
      t0 = t0 > INIT ? INIT : t0;
      d[j] = (float)t0 * scale;
    }
  }
  free(buf);
}

static int get_label(int quantized)                          /* getLabel, linemod.cpp:15-30 */
{
  switch (quantized) {
    case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3;
    case 16: return 4; case 32: return 5; case 64: return 6; case 128: return 7;
    default: return -1;                                      /* CV_Error in the reference */
  }
}

typedef struct { int x, y, label; float score; int order; } cand_t;
static int cmp_cand(const void *a, const void *b)            /* std::stable_sort with Candidate::operator< (linemod.hpp:98-101) */
{
  const cand_t *p = (const cand_t *)a, *q = (const cand_t *)b;
  if (p->score > q->score) return -1;
  if (p->score < q->score) return 1;
  return p->order - q->order;
}

/* selectScatteredFeatures (:135-164); candidates sorted, n >= num_features guaranteed by the callers */
static void select_scattered(const cand_t *c, int n, int num_features, float distance, orc_feature *out)
{
  int nf = 0, i = 0;
  float distance_sq = distance * distance;
  while (nf < num_features) {
    int keep = 1;
    for (int j = 0; j < nf && keep; ++j) {
      const int dx = c[i].x - out[j].x, dy = c[i].y - out[j].y;
      keep = (float)(dx * dx + dy * dy) >= distance_sq;      /* int expression compared with a float */
    }
    if (keep) { out[nf].x = c[i].x; out[nf].y = c[i].y; out[nf].label = c[i].label; ++nf; }
    if (++i == n) {                                          /* start over with a relaxed distance */
      i = 0;
      distance -= 1.0f;
      distance_sq = distance * distance;
    }
  }
}

/* ColorGradientPyramid::extractTemplate (:461-513).  quantized = the 3x3-voted one-hot image, magnitude = the
 * squared gradient magnitude (both from quantizedOrientations), mask optional.  returns 1 / 0 (too few candidates) */
int orc_extract_template_color(const uint8_t *quantized, const float *magnitude, const uint8_t *mask, int w, int h,
                               float strong_threshold, int num_features, orc_feature *out)
{
  uint8_t *local = NULL;
  if (mask) {                                                /* border of the mask: mask - erode(mask) */
    local = (uint8_t *)malloc((size_t)w * h);
    orc_erode_rect(mask, w, h, 1, local);
    for (size_t i = 0; i < (size_t)w * h; ++i) { int v = mask[i] - local[i]; local[i] = (uint8_t)(v < 0 ? 0 : v); }
  }
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)w * h);
  int n = 0;
  const float thr = strong_threshold * strong_threshold;
  for (int r = 0; r < h; ++r)
    for (int x = 0; x < w; ++x) {
      const size_t i = (size_t)r * w + x;
      if ((!local || local[i]) && quantized[i] > 0 && magnitude[i] > thr) {
        c[n].x = x; c[n].y = r; c[n].label = get_label(quantized[i]); c[n].score = magnitude[i]; c[n].order = n;
        ++n;
      }
    }
  int ok = n >= num_features;
  if (ok) {
    qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);
    select_scattered(c, n, num_features, (float)(n / num_features + 1), out);
  }
  free(c);
  free(local);
  return ok;
}

/* DepthNormalPyramid::extractTemplate (:747-825) */
int orc_extract_template_depth(const uint8_t *normal, const uint8_t *mask, int w, int h, int extract_threshold,
                               int num_features, orc_feature *out)
{
  const size_t npx = (size_t)w * h;
  uint8_t *local = NULL;
  if (mask) { local = (uint8_t *)malloc(npx); orc_erode_rect(mask, w, h, 2, local); }
  float *dist = (float *)malloc(sizeof(float) * npx * 8);
  uint8_t *temp = (uint8_t *)malloc(npx);
  for (int k = 0; k < 8; ++k) {
    for (size_t i = 0; i < npx; ++i) temp[i] = (uint8_t)((!local || local[i]) ? (normal[i] & (1 << k)) : 0);
    orc_distance_transform_c3(temp, w, h, dist + npx * k);
  }
  int label_counts[8] = {0};
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * npx);
  int n = 0;
  for (int r = 0; r < h; ++r)
    for (int x = 0; x < w; ++x) {
      const size_t i = (size_t)r * w + x;
      const int q = normal[i];
      if ((!local || local[i]) && q != 0 && q != 255) {
        const int label = get_label(q);
        if (label < 0) continue;                             /* cannot happen: normal is one-hot */
        const float score = dist[npx * label + i];
        if (score >= (float)extract_threshold) {
          c[n].x = x; c[n].y = r; c[n].label = label; c[n].score = score; c[n].order = n;
          ++n;
          ++label_counts[label];
        }
      }
    }
  int ok = n >= num_features;
  if (ok) {
    for (int i = 0; i < n; ++i) c[i].score /= (float)label_counts[c[i].label];
    qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);
    float area = (float)npx;
    if (local) { size_t nz = 0; for (size_t i = 0; i < npx; ++i) nz += local[i] != 0; area = (float)nz; }
    select_scattered(c, n, num_features, sqrtf(area) / sqrtf((float)num_features) + 1.5f, out);
  }
  free(c);
  free(temp);
  free(dist);
  free(local);
  return ok;
}

/* cropTemplates (:52-96) on `n` templates sharing one flat feature array; returns the bounding box {x, y, w, h} */
void orc_crop_templates(orc_template *t, int n, orc_feature *feats, int bb[4])
{
  int min_x = INT_MAX, min_y = INT_MAX, max_x = INT_MIN, max_y = INT_MIN;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < t[i].feat_count; ++j) {
      const int x = feats[t[i].feat_begin + j].x << t[i].pyramid_level, y = feats[t[i].feat_begin + j].y << t[i].pyramid_level;
      if (x < min_x) min_x = x;
      if (y < min_y) min_y = y;
      if (x > max_x) max_x = x;
      if (y > max_y) max_y = y;
    }
  if (min_x % 2 == 1) --min_x;
  if (min_y % 2 == 1) --min_y;
  for (int i = 0; i < n; ++i) {
    t[i].width = (max_x - min_x) >> t[i].pyramid_level;
    t[i].height = (max_y - min_y) >> t[i].pyramid_level;
    t[i].offset_x = min_x >> t[i].pyramid_level;
    t[i].offset_y = min_y >> t[i].pyramid_level;
    for (int j = 0; j < t[i].feat_count; ++j) {
      feats[t[i].feat_begin + j].x -= t[i].offset_x;
      feats[t[i].feat_begin + j].y -= t[i].offset_y;
    }
  }
  bb[0] = min_x; bb[1] = min_y; bb[2] = max_x - min_x; bb[3] = max_y - min_y;
}

/* Detector::addTemplate (:1579-1615) with the two default modalities (ColorGradient(10, 63, 55), DepthNormal(2000, 50,
 * 63, 2); :515-519, :827-832).  mask may be NULL (= empty object_mask).  templates: levels * 2 entries ordered
 * [l * 2 + m]; feats must hold levels * 2 * 63 entries.  Returns 0, or -1 when a level has too few candidates. */
int orc_add_template(const uint8_t *bgr, const uint16_t *depth, const uint8_t *mask, int w0, int h0, int levels,
                     orc_template *templates, orc_feature *feats, int bb[4])
{
  const int M = 2;
  int nfeat = 0, rc = 0;
  for (int l = 0; l < levels * M; ++l) { templates[l].feat_begin = 0; templates[l].feat_count = 0; }
  /* modality 0: ColorGradientPyramid (pyrDown :434-453) */
  {
    int w = w0, h = h0, nf = 63;
    uint8_t *src = (uint8_t *)malloc((size_t)w * h * 3), *mk = NULL;
    memcpy(src, bgr, (size_t)w * h * 3);
    if (mask) { mk = (uint8_t *)malloc((size_t)w * h); memcpy(mk, mask, (size_t)w * h); }
    for (int l = 0; l < levels && rc == 0; ++l) {
      if (l > 0) {
        nf /= 2;
        uint8_t *ns = (uint8_t *)malloc((size_t)(w / 2) * (h / 2) * 3);
        orc_pyrdown_bgr(src, w, h, ns);
        free(src);
        src = ns;
        if (mk) { uint8_t *nm = (uint8_t *)malloc((size_t)(w / 2) * (h / 2)); orc_resize_nn_half(mk, w, h, nm); free(mk); mk = nm; }
        w /= 2;
        h /= 2;
      }
      uint8_t *q = (uint8_t *)malloc((size_t)w * h);
      float *mag = (float *)malloc(sizeof(float) * (size_t)w * h);
      orc_quantized_orientations(src, w, h, 10.0f, q, mag);
      orc_template *t = &templates[l * M + 0];
      t->pyramid_level = l; t->width = t->height = -1; t->offset_x = t->offset_y = 0;
      t->feat_begin = (l * M + 0) * 63;
      if (orc_extract_template_color(q, mag, mk, w, h, 55.0f, nf, feats + t->feat_begin)) t->feat_count = nf; else rc = -1;
      free(q);
      free(mag);
    }
    free(src);
    free(mk);
  }
  /* modality 1: DepthNormalPyramid (pyrDown :721-739) */
  if (rc == 0) {
    int w = w0, h = h0, nf = 63, ext = 2;
    uint8_t *normal = (uint8_t *)malloc((size_t)w * h), *mk = NULL;
    orc_quantized_normals(depth, w, h, 2000, 50, normal);
    if (mask) { mk = (uint8_t *)malloc((size_t)w * h); memcpy(mk, mask, (size_t)w * h); }
    for (int l = 0; l < levels && rc == 0; ++l) {
      if (l > 0) {
        nf /= 2;
        ext /= 2;
        uint8_t *nn = (uint8_t *)malloc((size_t)(w / 2) * (h / 2));
        orc_resize_nn_half(normal, w, h, nn);
        free(normal);
        normal = nn;
        if (mk) { uint8_t *nm = (uint8_t *)malloc((size_t)(w / 2) * (h / 2)); orc_resize_nn_half(mk, w, h, nm); free(mk); mk = nm; }
        w /= 2;
        h /= 2;
      }
      orc_template *t = &templates[l * M + 1];
      t->pyramid_level = l; t->width = t->height = -1; t->offset_x = t->offset_y = 0;
      t->feat_begin = (l * M + 1) * 63;
      if (orc_extract_template_depth(normal, mk, w, h, ext, nf, feats + t->feat_begin)) t->feat_count = nf; else rc = -1;
    }
    free(normal);
    free(mk);
  }
  (void)nfeat;
  if (rc == 0) orc_crop_templates(templates, levels * M, feats, bb);
  return rc;
}
