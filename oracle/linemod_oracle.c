/*
 * linemod_oracle.c -- ORACLE (test infrastructure, not product): CPU restatement of the online
 * matching half of cup_linemod::Detector (linemod/linemod.cpp:882-1577 of /root/reference).
 * PARITY UNPINNED (see fealess_oracle.h).  Plain C99, scalar fallbacks of the reference's
 * SSE loops (the #else branches are the semantic definition).
 */
#include "fealess_oracle.h"
#include <stdlib.h>
#include <string.h>

/* SIMILARITY_LUT (linemod.cpp:970).  This fork's own table: response of orientation `ori` to a
 * set of spread orientations = max over set bits i of g(circular distance(i, ori)) with
 * g(0)=4, g(1)=2, g(2)=1, else 0; entry [32*ori + n] covers the low nibble n (bits 0-3), entry
 * [32*ori + 16 + n] the high nibble (bits 4-7).  Regenerated here from that rule; the unit test
 * tests/test_oracle_cpu.py::test_tables_equal_reference_text checks it byte for byte against the reference text when present. */
void orc_similarity_lut(uint8_t lut[256])
{
  static const uint8_t g[8] = {4, 2, 1, 0, 0, 0, 1, 2};
  for (int ori = 0; ori < 8; ++ori)
    for (int half = 0; half < 2; ++half)
      for (int n = 0; n < 16; ++n) {
        int best = 0;
        for (int b = 0; b < 4; ++b)
          if (n & (1 << b)) {
            int bit = b + 4 * half;
            int v = g[(bit - ori) & 7];
            if (v > best) best = v;
          }
        lut[32 * ori + 16 * half + n] = (uint8_t)best;
      }
}

/* spread (linemod.cpp:950-965): dst(y,x) = OR_{r<T,c<T} src(y+r, x+c), clipped bottom/right. */
void orc_spread(const uint8_t *src, int w, int h, int T, uint8_t *dst)
{
  memset(dst, 0, (size_t)w * h);
  for (int r = 0; r < T; ++r)
    for (int c = 0; c < T; ++c)
      /* orUnaligned8u(&src(r,c), ..., dst, ..., w - c, h - r)  (linemod.cpp:961) */
      for (int y = 0; y < h - r; ++y) {
        const uint8_t *s = src + (size_t)(y + r) * w + c;
        uint8_t *d = dst + (size_t)y * w;
        for (int x = 0; x < w - c; ++x) d[x] |= s[x];
      }
}

/* computeResponseMaps (linemod.cpp:979-1048, scalar branch :1034-1046). maps8 = 8 * w*h. */
void orc_response_maps(const uint8_t *spread, int w, int h, uint8_t *maps8)
{
  uint8_t lut[256];
  orc_similarity_lut(lut);
  size_t n = (size_t)w * h;
  for (int ori = 0; ori < 8; ++ori) {
    const uint8_t *lo = lut + 32 * ori, *hi = lo + 16;
    uint8_t *m = maps8 + ori * n;
    for (size_t i = 0; i < n; ++i) {
      uint8_t a = lo[spread[i] & 15], b = hi[(spread[i] & 240) >> 4];
      m[i] = a > b ? a : b;
    }
  }
}

/* linearize (linemod.cpp:1060-1088) */
int orc_linearize(const uint8_t *map, int w, int h, int T, uint8_t *out)
{
  if (h % T != 0 || w % T != 0) return -1;           /* CV_Assert :1062-1063 */
  uint8_t *memory = out;
  for (int r_start = 0; r_start < T; ++r_start)
    for (int c_start = 0; c_start < T; ++c_start)
      for (int r = r_start; r < h; r += T)
        for (int c = c_start; c < w; c += T) *memory++ = map[(size_t)r * w + c];
  return 0;
}

/* Layout of the linear memories of one (level, modality) in the oracle and in the HIP path:
 *   [label 0..7][grid 0..T*T-1][ (w/T)*(h/T) bytes ] followed, per label, by a zero pad.
 * The reference keeps one continuous T*T x (W*H) Mat per label (linemod.cpp:1068), so an
 * over-read past a grid row (quirk Q2 of SURVEY.md section 8: feature y == template height with
 * height % T == 0; or a 16x16 patch leaving the map) lands in the next grid row -- reproduced
 * here -- and past the last grid row it is undefined behaviour in the reference; here it reads
 * the zero pad (documented deviation: defined instead of UB). */
size_t orc_lm_label_stride(int w, int h, int T)
{
  size_t W = (size_t)(w / T), H = (size_t)(h / T);
  size_t pad = W * H + 16 * W + 64;
  size_t s = (size_t)T * T * W * H + pad;
  return (s + 63) & ~(size_t)63;
}

int orc_build_linear_memories(const uint8_t *quantized, int w, int h, int T, uint8_t *out)
{
  if ((w * h) % 16 != 0) return -1;                  /* CV_Assert linemod.cpp:981 */
  if (h % T != 0 || w % T != 0) return -1;
  size_t n = (size_t)w * h, stride = orc_lm_label_stride(w, h, T);
  uint8_t *spread = (uint8_t *)malloc(n), *maps = (uint8_t *)malloc(8 * n);
  orc_spread(quantized, w, h, T, spread);
  orc_response_maps(spread, w, h, maps);
  memset(out, 0, 8 * stride);
  for (int l = 0; l < 8; ++l) orc_linearize(maps + l * n, w, h, T, out + l * stride);
  free(spread);
  free(maps);
  return 0;
}

/* accessLinearMemory (linemod.cpp:1094-1117): offset of feature f inside its label's block */
static size_t lm_offset(const orc_feature *f, int T, int W, size_t WH)
{
  int grid_index = (f->y % T) * T + (f->x % T);
  int lm_index = (f->y / T) * W + (f->x / T);
  return (size_t)grid_index * WH + (size_t)lm_index;
}

/* similarity (linemod.cpp:1130-1214) */
int orc_similarity(const uint8_t *lm8, const orc_template *t, const orc_feature *feats,
                   int w, int h, int T, uint8_t *dst)
{
  if (t->feat_count > 63) return -1;                 /* CV_Assert :1137 */
  int W = w / T, H = h / T;
  size_t WH = (size_t)W * H, stride = orc_lm_label_stride(w, h, T);
  int wf = (t->width - 1) / T + 1;
  int hf = (t->height - 1) / T + 1;
  int span_x = W - wf, span_y = H - hf;
  int template_positions = span_y * W + span_x + 1;  /* :1155 */
  memset(dst, 0, WH);
  for (int i = 0; i < t->feat_count; ++i) {
    const orc_feature *f = feats + t->feat_begin + i;
    if (f->x < 0 || f->x >= w || f->y < 0 || f->y >= h) continue;   /* :1179 */
    const uint8_t *lm_ptr = lm8 + (size_t)f->label * stride + lm_offset(f, T, W, WH);
    for (int j = 0; j < template_positions; ++j) dst[j] = (uint8_t)(dst[j] + lm_ptr[j]);
  }
  return 0;
}

/* similarityLocal (linemod.cpp:1226-1300) */
int orc_similarity_local(const uint8_t *lm8, const orc_template *t, const orc_feature *feats,
                         int w, int h, int T, int cx, int cy, uint8_t *dst)
{
  if (t->feat_count > 63) return -1;                 /* CV_Assert :1231 */
  int W = w / T, H = h / T;
  size_t WH = (size_t)W * H, stride = orc_lm_label_stride(w, h, T);
  memset(dst, 0, 256);
  int offset_x = (cx / T - 8) * T;                   /* C division truncates toward zero, :1240 */
  int offset_y = (cy / T - 8) * T;
  for (int i = 0; i < t->feat_count; ++i) {
    orc_feature f = feats[t->feat_begin + i];
    f.x += offset_x;
    f.y += offset_y;
    if (f.x < 0 || f.y < 0 || f.x >= w || f.y >= h) continue;       /* :1257 */
    const uint8_t *lm_ptr = lm8 + (size_t)f.label * stride + lm_offset(&f, T, W, WH);
    uint8_t *d = dst;
    for (int row = 0; row < 16; ++row) {             /* scalar branch :1290-1297 */
      for (int col = 0; col < 16; ++col) d[col] = (uint8_t)(d[col] + lm_ptr[col]);
      d += 16;
      lm_ptr += W;
    }
  }
  return 0;
}

/* addSimilarities (linemod.cpp:1322-1338) for any number of modalities */
static void add_similarities(const uint8_t *const *sims, int M, size_t n, uint16_t *dst)
{
  for (size_t i = 0; i < n; ++i) {
    unsigned s = 0;
    for (int m = 0; m < M; ++m) s += sims[m][i];
    dst[i] = (uint16_t)(s > 65535u ? 65535u : s);    /* cv::add saturates; never reached (<= 252*M) */
  }
}

int orc_total_similarity(const uint8_t *const *lm_level, const orc_bank *bank, int pyramid,
                         int w, int h, int T, uint16_t *dst)
{
  int M = bank->modalities, L = bank->levels;
  size_t WH = (size_t)(w / T) * (h / T);
  uint8_t *buf = (uint8_t *)malloc(WH * M);
  const uint8_t *sims[16];
  const orc_template *tp = bank->templates + (size_t)pyramid * L * M;
  int lowest_start = L * M - M;
  for (int m = 0; m < M; ++m) {
    sims[m] = buf + m * WH;
    if (orc_similarity(lm_level[m], tp + lowest_start + m, bank->features, w, h, T, buf + m * WH)) {
      free(buf);
      return -1;
    }
  }
  add_similarities(sims, M, WH, dst);
  free(buf);
  return 0;
}

/* ---- growable match vector ---- */
typedef struct { orc_match *v; int n, cap; } mvec;
static void mv_push(mvec *a, orc_match m)
{
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 256;
    a->v = (orc_match *)realloc(a->v, sizeof(orc_match) * a->cap);
  }
  a->v[a->n++] = m;
}

/* Detector::matchClass (linemod.cpp:1451-1577).
 * lm[l*M+m] = padded linear memories of level l, modality m; sizes w_l x h_l. */
static int match_class(const uint8_t *const *lm, const int *ws, const int *hs,
                       int levels, int M, const int *T_at_level, float threshold,
                       const orc_bank *bank, int class_idx, mvec *matches)
{
  int wl = ws[levels - 1], hl = hs[levels - 1], lowest_T = T_at_level[levels - 1];
  int Wl = wl / lowest_T, Hl = hl / lowest_T;
  uint16_t *total = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)Wl * Hl);
  uint8_t loc[16][256];
  int rc = 0;
  for (int template_id = 0; template_id < bank->n_pyramids && !rc; ++template_id) {
    const orc_template *tp = bank->templates + (size_t)template_id * levels * M;
    int lowest_start = levels * M - M;
    int num_features = 0;
    for (int i = 0; i < M; ++i) num_features += tp[lowest_start + i].feat_count;   /* :1474 */
    if (orc_total_similarity(lm + (size_t)(levels - 1) * M, bank, template_id, wl, hl, lowest_T, total)) {
      rc = -1;
      break;
    }
    /* :1487 -- evaluated in float exactly as written */
    int raw_threshold = (int)(2 * num_features + (threshold / 100.f) * (2 * num_features) + 0.5f);

    mvec cand = {0, 0, 0};
    for (int r = 0; r < Hl; ++r)
      for (int c = 0; c < Wl; ++c) {
        int raw_score = total[(size_t)r * Wl + c];
        if (raw_score > raw_threshold) {
          int offset = lowest_T / 2 + (lowest_T % 2 - 1);
          orc_match m;
          m.x = c * lowest_T + offset;
          m.y = r * lowest_T + offset;
          m.similarity = (raw_score * 100.f) / (4 * num_features) + 0.5f;        /* :1502 */
          m.class_idx = class_idx;
          m.template_id = template_id;
          mv_push(&cand, m);
        }
      }

    for (int l = levels - 2; l >= 0 && !rc; --l) {                                /* :1509 */
      int T = T_at_level[l];
      int start = l * M;
      int w = ws[l], h = hs[l];
      int border = 8 * T;
      int offset = T / 2 + (T % 2 - 1);
      int max_x = w - tp[start].width - border;
      int max_y = h - tp[start].height - border;
      for (int mi = 0; mi < cand.n; ++mi) {
        orc_match *m2 = &cand.v[mi];
        int x = m2->x * 2 + 1, y = m2->y * 2 + 1;
        x = x > border ? x : border;
        y = y > border ? y : border;
        x = x < max_x ? x : max_x;
        y = y < max_y ? y : max_y;
        int numFeatures = 0;
        for (int i = 0; i < M; ++i) {
          const orc_template *templ = tp + start + i;
          numFeatures += templ->feat_count;
          if (orc_similarity_local(lm[(size_t)l * M + i], templ, bank->features, w, h, T, x, y, loc[i])) {
            rc = -1;
            break;
          }
        }
        if (rc) break;
        int best_score = 0, best_r = -1, best_c = -1;
        for (int r = 0; r < 16; ++r)
          for (int c = 0; c < 16; ++c) {
            int score = 0;
            for (int i = 0; i < M; ++i) score += loc[i][r * 16 + c];
            if (score > best_score) { best_score = score; best_r = r; best_c = c; }
          }
        m2->x = (x / T - 8 + best_c) * T + offset;                                /* :1564 */
        m2->y = (y / T - 8 + best_r) * T + offset;
        m2->similarity = (best_score * 100.f) / (4 * numFeatures);                /* :1566 */
      }
      int k = 0;                                                                  /* remove_if :1570 */
      for (int mi = 0; mi < cand.n; ++mi)
        if (!(cand.v[mi].similarity < threshold)) cand.v[k++] = cand.v[mi];
      cand.n = k;
    }
    for (int mi = 0; mi < cand.n; ++mi) mv_push(matches, cand.v[mi]);
    free(cand.v);
  }
  free(total);
  return rc;
}

/* Match::operator< (linemod.hpp:262-269) extended to a total order.  std::sort is unstable and the
 * comparator only looks at (similarity, template_id), so the reference's order among equal
 * keys is unspecified; (class, y, x) ascending is the canonical choice of this repo (Q5). */
static int match_cmp(const void *pa, const void *pb)
{
  const orc_match *a = (const orc_match *)pa, *b = (const orc_match *)pb;
  if (a->similarity != b->similarity) return a->similarity > b->similarity ? -1 : 1;
  if (a->template_id != b->template_id) return a->template_id < b->template_id ? -1 : 1;
  if (a->class_idx != b->class_idx) return a->class_idx < b->class_idx ? -1 : 1;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  if (a->x != b->x) return a->x < b->x ? -1 : 1;
  return 0;
}

int orc_sort_unique(orc_match *m, int n)
{
  if (n <= 0) return 0;
  qsort(m, (size_t)n, sizeof(orc_match), match_cmp);
  int k = 0;                                         /* std::unique with Match::operator== */
  for (int i = 1; i < n; ++i) {
    const orc_match *p = &m[k], *q = &m[i];
    int eq = p->x == q->x && p->y == q->y && p->similarity == q->similarity &&
             p->class_idx == q->class_idx;           /* ignores template_id, linemod.hpp:271-274 */
    if (!eq) m[++k] = m[i];
  }
  return k + 1;
}

/* Detector::match (linemod.cpp:1356-1441) downstream of the quantizers */
int orc_match_quantized(const uint8_t *const *quantized, int w0, int h0,
                        int levels, int modalities, const int *T_at_level,
                        const orc_bank *banks, int n_classes, float threshold,
                        orc_match *out, int cap, int *n_total)
{
  int M = modalities, rc = 0;
  int ws[16], hs[16];
  uint8_t **lm = (uint8_t **)calloc((size_t)levels * M, sizeof(uint8_t *));
  for (int l = 0; l < levels && !rc; ++l) {
    ws[l] = w0 >> l;
    hs[l] = h0 >> l;
    for (int m = 0; m < M && !rc; ++m) {
      lm[l * M + m] = (uint8_t *)malloc(8 * orc_lm_label_stride(ws[l], hs[l], T_at_level[l]));
      rc = orc_build_linear_memories(quantized[l * M + m], ws[l], hs[l], T_at_level[l], lm[l * M + m]);
    }
  }
  mvec matches = {0, 0, 0};
  for (int c = 0; c < n_classes && !rc; ++c)
    rc = match_class((const uint8_t *const *)lm, ws, hs, levels, M, T_at_level, threshold,
                     &banks[c], c, &matches);
  int n = 0;
  if (!rc) {
    n = orc_sort_unique(matches.v, matches.n);
    if (n_total) *n_total = n;
    if (n > cap) n = cap;
    if (n > 0) memcpy(out, matches.v, sizeof(orc_match) * (size_t)n);
  }
  for (int i = 0; i < levels * M; ++i) free(lm[i]);
  free(lm);
  free(matches.v);
  return rc ? -1 : n;
}
