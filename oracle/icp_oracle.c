/*
 * icp_oracle.c -- ORACLE (test infrastructure, not product): CPU restatement of the ICP half of
 * the rlvc/FEALESS hot path: cup_d2pc::depthTo3d (ICP/depth_to_3d.cpp:99-137,190-269),
 * matToVec / is_vec3f_valid / scale_mat_vec3f (ICP/common.cpp:261-266,382-425), detection()
 * (ICP/detection.cpp:11-254, live lines), icpCloudToCloud_Ex and helpers (ICP/ICP.cpp:8-111,
 * 193-279,617-809) and CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:86-204).
 *
 * PARITY UNPINNED.  cv::SVD (OpenCV JacobiSVDImpl_<float>) and the FLANN KDTreeSingleIndex
 * exact 1-NN are un-vendored third-party code, restated from their published algorithms.
 * Build with -ffp-contract=off: each float expression is one IEEE binary32 operation per
 * operator, in the order written (the reference is built for baseline x86-64: no FMA).
 */
#define _POSIX_C_SOURCE 199309L   /* clock_gettime under -std=c99 */
#include "fealess_oracle.h"
#include <math.h>
#include <time.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* cup_d2pc::depthTo3d, CV_16UC1 input, float K (depth_to_3d.cpp:99-137, 190-221, 244-269)      */
void orc_depth_to_3d(const uint16_t *depth, int w, int h, double fx, double fy, double cx,
                     double cy, float *out)
{
  /* K.convertTo(K_new, CV_32F) (:202); setCamIntrinsic builds K from doubles (common.cpp:374-379) */
  const float fxf = (float)fx, fyf = (float)fy, ox = (float)cx, oy = (float)cy;
  const float inv_fx = 1.0f / fxf, inv_fy = 1.0f / fyf;          /* :103-104 */
  const float scale = (float)(1 / 1000.0);                        /* convertTo(.., 1/1000.0) :257 */
  for (int y = 0; y < h; ++y) {
    float yc = ((float)y - oy) * inv_fy;                          /* :121 */
    for (int x = 0; x < w; ++x) {
      float xc = ((float)x - ox) * inv_fx;                        /* :119 */
      uint16_t d = depth[(size_t)y * w + x];
      float z = d == 0 ? NAN : (float)d * scale;                  /* :257-259 */
      float *p = out + ((size_t)y * w + x) * 3;
      p[0] = xc * z;                                              /* :132-134 */
      p[1] = yc * z;
      p[2] = z;
    }
  }
}

static inline int vec_valid(const float *v) { return v[2] <= 900.0f; }   /* common.cpp:261-266 */

/* ------------------------------------------------------------------------------------------ */
/* cv::SVD::compute on a 3x3 CV_32F matrix: OpenCV's JacobiSVDImpl_<float> (one-sided Jacobi on
 * the rows of At = A^T, double accumulators, eps = 2*FLT_EPSILON, max(m,30) sweeps), restated.
 * hypot(p, beta) is written sqrt(p*p + beta*beta) so that the HIP path can reproduce it bit for
 * bit (documented deviation; both are within 1 ulp of each other in double). */
void orc_svd3(const float A[9], float Wout[3], float U[9], float Vt[9])
{
  const int m = 3, n = 3;
  float At[9];
  double W[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) At[i * 3 + j] = A[j * 3 + i];
  const float eps = FLT_EPSILON * 2;
  const double minval = FLT_MIN;
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) { float t = At[i * 3 + k]; sd += (double)t * t; }
    W[i] = sd;
    for (int k = 0; k < n; ++k) Vt[i * 3 + k] = 0;
    Vt[i * 3 + i] = 1;
  }
  for (int iter = 0; iter < 30; ++iter) {
    int changed = 0;
    for (int i = 0; i < n - 1; ++i)
      for (int j = i + 1; j < n; ++j) {
        float *Ai = At + i * 3, *Aj = At + j * 3;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; ++k) p += (double)Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt((double)a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = sqrt(p * p + beta * beta);
        float c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = (float)sqrt(delta / gamma);
          c = (float)(p / (gamma * s * 2));
        } else {
          c = (float)sqrt((gamma + beta) / (gamma * 2));
          s = (float)(p / (gamma * c * 2));
        }
        a = b = 0;
        for (int k = 0; k < m; ++k) {
          float t0 = c * Ai[k] + s * Aj[k];
          float t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += (double)t0 * t0;
          b += (double)t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = 1;
        float *Vi = Vt + i * 3, *Vj = Vt + j * 3;
        for (int k = 0; k < n; ++k) {
          float t0 = c * Vi[k] + s * Vj[k];
          float t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) { float t = At[i * 3 + k]; sd += (double)t * t; }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < n - 1; ++i) {
    int j = i;
    for (int k = i + 1; k < n; ++k)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      for (int k = 0; k < m; ++k) { float f = At[i * 3 + k]; At[i * 3 + k] = At[j * 3 + k]; At[j * 3 + k] = f; }
      for (int k = 0; k < n; ++k) { float f = Vt[i * 3 + k]; Vt[i * 3 + k] = Vt[j * 3 + k]; Vt[j * 3 + k] = f; }
    }
  }
  for (int i = 0; i < n; ++i) Wout[i] = (float)W[i];
  for (int i = 0; i < n; ++i) {
    double sd = W[i];
    /* OpenCV regenerates a random orthogonal vector when sd <= FLT_MIN (rank-deficient input);
     * not reproduced: the row is zeroed instead (a degenerate cloud; R is then not a rotation in
     * the reference either). */
    float s = (float)(sd > minval ? 1 / sd : 0.);
    for (int k = 0; k < m; ++k) At[i * 3 + k] *= s;
  }
  /* rows of At are the left singular vectors: U = At^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) U[i * 3 + j] = At[j * 3 + i];
}

/* ------------------------------------------------------------------------------------------ */
/* exact 1-NN, squared L2 as cvflann::L2_Simple<float>: ((dx*dx + dy*dy) + dz*dz) in float.     */
static inline float d2f(const float *a, const float *b)
{
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  float r = dx * dx;
  r += dy * dy;
  r += dz * dz;
  return r;
}

static void nn_brute(const float *ref, int n_ref, const float *q, int *idx, float *dist)
{
  int bi = -1;
  float bd = INFINITY;
  for (int j = 0; j < n_ref; ++j) {
    float d = d2f(q, ref + 3 * j);
    if (d < bd) { bd = d; bi = j; }          /* ties keep the lowest index */
  }
  *idx = bi;
  *dist = bi >= 0 ? bd : NAN;
}

/* kd-tree with leaf size 15 (KDTreeSingleIndexParams(15), ICP.cpp:658), exact search */
typedef struct { int left, right, begin, end; float lo[3], hi[3]; } kd_node;
typedef struct { kd_node *nodes; int n_nodes, cap; int *perm; const float *pts; } kd_tree;
static __thread const float *g_sort_pts;   /* thread-local: bench.py times the oracle on several host threads */
static __thread int g_sort_dim;
static int cmp_dim(const void *a, const void *b)
{
  float x = g_sort_pts[3 * *(const int *)a + g_sort_dim], y = g_sort_pts[3 * *(const int *)b + g_sort_dim];
  if (x < y) return -1;
  if (x > y) return 1;
  return *(const int *)a - *(const int *)b;
}
static int kd_build(kd_tree *t, int begin, int end)
{
  if (t->n_nodes == t->cap) { t->cap *= 2; t->nodes = (kd_node *)realloc(t->nodes, sizeof(kd_node) * t->cap); }
  int id = t->n_nodes++;
  kd_node nd;
  nd.left = nd.right = -1;
  nd.begin = begin;
  nd.end = end;
  for (int d = 0; d < 3; ++d) { nd.lo[d] = INFINITY; nd.hi[d] = -INFINITY; }
  for (int i = begin; i < end; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = t->pts[3 * t->perm[i] + d];
      if (v < nd.lo[d]) nd.lo[d] = v;
      if (v > nd.hi[d]) nd.hi[d] = v;
    }
  if (end - begin > 15) {
    int dim = 0;
    float ext = nd.hi[0] - nd.lo[0];
    for (int d = 1; d < 3; ++d)
      if (nd.hi[d] - nd.lo[d] > ext) { ext = nd.hi[d] - nd.lo[d]; dim = d; }
    g_sort_pts = t->pts;
    g_sort_dim = dim;
    qsort(t->perm + begin, (size_t)(end - begin), sizeof(int), cmp_dim);
    int mid = (begin + end) / 2;
    t->nodes[id] = nd;
    int l = kd_build(t, begin, mid);
    int r = kd_build(t, mid, end);
    t->nodes[id].left = l;
    t->nodes[id].right = r;
  } else {
    t->nodes[id] = nd;
  }
  return id;
}
static double box_lb(const kd_node *nd, const float *q)
{
  double s = 0;
  for (int d = 0; d < 3; ++d) {
    double v = q[d] < nd->lo[d] ? (double)nd->lo[d] - q[d] : (q[d] > nd->hi[d] ? (double)q[d] - nd->hi[d] : 0.0);
    s += v * v;
  }
  return s;
}
static void kd_search(const kd_tree *t, int id, const float *q, int *bi, float *bd)
{
  const kd_node *nd = &t->nodes[id];
  /* conservative prune: the float d2 may round below the real value by a few ulp */
  if (*bi >= 0 && box_lb(nd, q) > (double)*bd * (1.0 + 1e-5) + 1e-30) return;
  if (nd->left < 0) {
    for (int i = nd->begin; i < nd->end; ++i) {
      int j = t->perm[i];
      float d = d2f(q, t->pts + 3 * j);
      if (d < *bd || (d == *bd && j < *bi)) { *bd = d; *bi = j; }
    }
    return;
  }
  double ll = box_lb(&t->nodes[nd->left], q), lr = box_lb(&t->nodes[nd->right], q);
  if (ll <= lr) { kd_search(t, nd->left, q, bi, bd); kd_search(t, nd->right, q, bi, bd); }
  else { kd_search(t, nd->right, q, bi, bd); kd_search(t, nd->left, q, bi, bd); }
}

/* ------------------------------------------------------------------------------------------ */
/* Matx33f * Vec3f as cv::Matx: s = 0; s += a(i,k)*b(k) (float), k = 0..2 */
static inline void mat_vec(const float *R, const float *v, float *o)
{
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    s += R[i * 3 + 0] * v[0];
    s += R[i * 3 + 1] * v[1];
    s += R[i * 3 + 2] * v[2];
    o[i] = s;
  }
}
static inline void mat_mat(const float *A, const float *B, float *O)
{
  float t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      t[i * 3 + j] = s;
    }
  memcpy(O, t, sizeof(t));
}

/* getL2distClouds (ICP.cpp:68-111) */
static float l2dist_clouds(const float *model, const float *ref, int n, float *dist_mean,
                           float dist_thr, int accum64)
{
  int nbr_inliers = 0, counter = 0;
  float ratio = 0.0f, dm = 0.0f;
  double dm64 = 0.0;
  for (int i = 0; i < n; ++i) {
    const float *a = model + 3 * i, *b = ref + 3 * i;
    if (!vec_valid(b)) continue;
    if (!vec_valid(a)) continue;
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    /* cv::norm(Vec3f): double accumulation of squares, sqrt in double (:88) */
    float dist = (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
    if (dist <= dist_thr) {
      dm += dist;
      dm64 += dist;
      ++nbr_inliers;
    }
    ++counter;
  }
  if (counter > 0) {
    if (accum64) dm = (float)(dm64 / (double)nbr_inliers);
    else dm /= (float)nbr_inliers;                       /* 0/0 -> NaN ends the loop (Q9) */
    ratio = (float)nbr_inliers / (float)counter;
  } else
    dm = FLT_MAX;
  *dist_mean = dm;
  return ratio;
}

static int all_finite(const float *v, int n)
{
  for (int i = 0; i < n; ++i)
    if (!isfinite(v[i])) return 0;
  return 1;
}

/* icpCloudToCloud_Ex (ICP.cpp:617-809) */
int orc_icp(const float *ref, int n_ref, const float *model, int n_model,
            int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
            int accum64, int use_kdtree, orc_icp_result *res, float *trace, int trace_cap)
{
  memset(res, 0, sizeof(*res));            /* cv::Matx33f R; cv::Vec3f T; are zero-initialised */
  if (n_model < 3 || n_ref < 3) { res->dist_mean = -1.0f; return -1; }          /* :633-638 */
  if (n_ref < n_model) { res->dist_mean = -1.0f; return -2; }  /* reference would read out of bounds */
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0};

  kd_tree tree;
  memset(&tree, 0, sizeof(tree));
  if (use_kdtree) {
    tree.cap = 2 * (n_ref / 8 + 2);
    tree.nodes = (kd_node *)malloc(sizeof(kd_node) * tree.cap);
    tree.perm = (int *)malloc(sizeof(int) * n_ref);
    tree.pts = ref;
    for (int i = 0; i < n_ref; ++i) tree.perm[i] = i;
    kd_build(&tree, 0, n_ref);
  }

  /* copyPoints(pts_model, pts_model_tmp) (:666-667): invalid points stay Vec3f() = 0 */
  float *mt = (float *)calloc((size_t)n_model * 3, sizeof(float));
  for (int i = 0; i < n_model; ++i)
    if (vec_valid(model + 3 * i)) memcpy(mt + 3 * i, model + 3 * i, 12);
  float *cm = (float *)malloc(sizeof(float) * 3 * (size_t)n_model);
  float *cr = (float *)malloc(sizeof(float) * 3 * (size_t)n_ref);

  float dist_mean = 0.0f;
  float px = l2dist_clouds(mt, ref, n_model, &dist_mean, FLT_MAX, accum64);      /* :670 */
  float dist_diff = FLT_MAX;
  int iter = 0, n_corr_last = 0;
  while ((dist_mean > dist_mean_thr) && (dist_diff > dist_diff_thr) && (iter < icp_it_thr)) {
    ++iter;
    int ncm = 0, ncr = 0;
    if (iter == 1) {                                     /* :700-704 */
      memset(cm, 0, sizeof(float) * 3 * (size_t)n_model);
      memset(cr, 0, sizeof(float) * 3 * (size_t)n_ref);
      for (int i = 0; i < n_model; ++i)
        if (vec_valid(mt + 3 * i)) memcpy(cm + 3 * i, mt + 3 * i, 12);
      for (int i = 0; i < n_ref; ++i)
        if (vec_valid(ref + 3 * i)) memcpy(cr + 3 * i, ref + 3 * i, 12);
      ncm = n_model;
      ncr = n_ref;
    } else {                                             /* PointsCorresponding :193-279 */
      float thr = 3 * dist_mean;                         /* squared distance vs 3*mean (Q9) */
      for (int i = 0; i < n_model; ++i) {
        int idx = -1;
        float d = INFINITY;
        if (use_kdtree) { kd_search(&tree, 0, mt + 3 * i, &idx, &d); if (idx < 0) d = NAN; }
        else nn_brute(ref, n_ref, mt + 3 * i, &idx, &d);
        if (d <= thr) {
          memcpy(cm + 3 * ncm, mt + 3 * i, 12);
          memcpy(cr + 3 * ncr, ref + 3 * idx, 12);
          ++ncm;
          ++ncr;
        }
      }
    }
    n_corr_last = ncm;
    if (ncr < 3 || ncm < 3) { iter = icp_it_thr; continue; }                     /* :711-715 */

    float mc[3] = {0, 0, 0}, rc[3] = {0, 0, 0}, C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!accum64) {
      for (int i = 0; i < ncm; ++i) { mc[0] += cm[3 * i]; mc[1] += cm[3 * i + 1]; mc[2] += cm[3 * i + 2]; }   /* getMean :8-25 */
      for (int i = 0; i < ncr; ++i) { rc[0] += cr[3 * i]; rc[1] += cr[3 * i + 1]; rc[2] += cr[3 * i + 2]; }
      for (int k = 0; k < 3; ++k) { mc[k] /= (float)ncm; rc[k] /= (float)ncr; }
      for (int i = 0; i < ncm; ++i)                                                 /* :731-735 */
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) C[a * 3 + b] += cm[3 * i + a] * cr[3 * i + b];
    } else {
      double m64[3] = {0, 0, 0}, r64[3] = {0, 0, 0}, C64[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < ncm; ++i) for (int k = 0; k < 3; ++k) m64[k] += cm[3 * i + k];
      for (int i = 0; i < ncr; ++i) for (int k = 0; k < 3; ++k) r64[k] += cr[3 * i + k];
      for (int i = 0; i < ncm; ++i)
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) C64[a * 3 + b] += (double)cm[3 * i + a] * (double)cr[3 * i + b];
      for (int k = 0; k < 3; ++k) { mc[k] = (float)(m64[k] / ncm); rc[k] = (float)(r64[k] / ncr); }
      for (int k = 0; k < 9; ++k) C[k] = (float)C64[k];
    }

    float w[3], u[9], vt[9], Ropt[9], Topt[3];
    orc_svd3(C, w, u, vt);                                                          /* :742 */
    /* R_optimal = Mat(vt.t() * u.t()) (:744): cv::gemm on CV_32F accumulates in double */
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += (double)vt[k * 3 + i] * (double)u[j * 3 + k];
        Ropt[i * 3 + j] = (float)s;
      }
    float Rm[3];
    mat_vec(Ropt, mc, Rm);
    for (int k = 0; k < 3; ++k) Topt[k] = rc[k] - Rm[k];                            /* :747 */
    if (trace && iter <= trace_cap) {
      float *tr = trace + 23 * (iter - 1);
      tr[0] = (float)ncm;
      memcpy(tr + 2, C, 36);
      memcpy(tr + 11, Ropt, 36);
      memcpy(tr + 20, Topt, 12);
      tr[1] = NAN;
    }
    if (!all_finite(Ropt, 9) || !all_finite(Topt, 3)) continue;                     /* :748-749 */

    for (int i = 0; i < n_model; ++i) {                  /* transformPoints in place :28-45, :756 */
      float *p = mt + 3 * i;
      if (!vec_valid(p)) continue;
      float o[3];
      mat_vec(Ropt, p, o);
      p[0] = o[0] + Topt[0];
      p[1] = o[1] + Topt[1];
      p[2] = o[2] + Topt[2];
    }
    dist_diff = dist_mean;                                                          /* :778-780 */
    float thr = 3 * dist_mean;
    px = l2dist_clouds(mt, ref, n_model, &dist_mean, thr, accum64);
    dist_diff -= dist_mean;
    if (trace && iter <= trace_cap) trace[23 * (iter - 1) + 1] = dist_mean;

    float RT[3];                                                                    /* :793-797 */
    mat_vec(Ropt, T, RT);
    for (int k = 0; k < 3; ++k) T[k] = RT[k] + Topt[k];
    mat_mat(Ropt, R, R);
  }
  memcpy(res->R, R, sizeof(R));
  memcpy(res->T, T, sizeof(T));
  res->dist_mean = dist_mean;
  res->px_ratio = px;
  res->iters = iter;
  res->n_corr_last = n_corr_last;
  free(mt);
  free(cm);
  free(cr);
  free(tree.nodes);
  free(tree.perm);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* detection() (ICP/detection.cpp:28-44, 112-114, 162-206, 225-234) */
int orc_detection(const uint16_t *model_depth, const uint16_t *scene_depth, int w, int h,
                  double fx, double fy, double cx, double cy,
                  const int rect_model[4], const int rect_ref[4],
                  int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
                  const float r_match[9], const float t_match[3],
                  int accum64, int use_kdtree, orc_detection_result *res)
{
  memset(res, 0, sizeof(*res));
  /* Q10: a rect leaving the image throws in the reference (cv::Mat ROI assert, :43-44) */
  const int *rm = rect_model, *rr = rect_ref;
  if (rm[0] < 0 || rm[1] < 0 || rm[2] < 0 || rm[3] < 0 || rm[0] + rm[2] > w || rm[1] + rm[3] > h) return -3;
  if (rr[0] < 0 || rr[1] < 0 || rr[2] < 0 || rr[3] < 0 || rr[0] + rr[2] > w || rr[1] + rr[3] > h) return -3;
  if (rm[2] != rr[2] || rm[3] != rr[3]) return -3;
  size_t n = (size_t)w * h;
  float *p_ref = (float *)malloc(sizeof(float) * 3 * n), *p_mod = (float *)malloc(sizeof(float) * 3 * n);
  orc_depth_to_3d(scene_depth, w, h, fx, fy, cx, cy, p_ref);                       /* :31-32 */
  orc_depth_to_3d(model_depth, w, h, 608, 608, 320, 240, p_mod);                   /* :35-36, common.cpp:358 */
  for (size_t i = 0; i < 3 * n; ++i) { p_ref[i] = p_ref[i] * 1000; p_mod[i] = p_mod[i] * 1000; }  /* :39-40 */

  int cw = rm[2], ch = rm[3];
  float *pts_ref = (float *)malloc(sizeof(float) * 3 * (size_t)(cw * ch + 1));
  float *pts_mod = (float *)malloc(sizeof(float) * 3 * (size_t)(cw * ch + 1));
  int np = 0;
  for (int y = 0; y < ch; ++y)                                                     /* matToVec common.cpp:382-405 */
    for (int x = 0; x < cw; ++x) {
      const float *a = p_ref + 3 * ((size_t)(rr[1] + y) * w + rr[0] + x);
      const float *b = p_mod + 3 * ((size_t)(rm[1] + y) * w + rm[0] + x);
      if (!vec_valid(a)) continue;
      if (!vec_valid(b)) continue;
      memcpy(pts_ref + 3 * np, a, 12);
      memcpy(pts_mod + 3 * np, b, 12);
      ++np;
    }
  res->n_points = np;
  float mc[3] = {0, 0, 0}, rc[3] = {0, 0, 0};                                      /* getMean x2 :165-166 */
  if (!accum64) {
    for (int i = 0; i < np; ++i) for (int k = 0; k < 3; ++k) { mc[k] += pts_mod[3 * i + k]; }
    for (int i = 0; i < np; ++i) for (int k = 0; k < 3; ++k) { rc[k] += pts_ref[3 * i + k]; }
    if (np > 0) for (int k = 0; k < 3; ++k) { mc[k] /= (float)np; rc[k] /= (float)np; }
  } else {
    double m64[3] = {0, 0, 0}, r64[3] = {0, 0, 0};
    for (int i = 0; i < np; ++i) for (int k = 0; k < 3; ++k) { m64[k] += pts_mod[3 * i + k]; r64[k] += pts_ref[3 * i + k]; }
    if (np > 0) for (int k = 0; k < 3; ++k) { mc[k] = (float)(m64[k] / np); rc[k] = (float)(r64[k] / np); }
  }
  float t_tmp[3], t_init[3];
  for (int k = 0; k < 3; ++k) { t_tmp[k] = rc[k] - mc[k]; t_init[k] = t_tmp[k] + t_match[k]; }   /* :177,:199 */
  for (int i = 0; i < np; ++i) {                                                   /* transformPoints(I, t_tmp) :206 */
    float *p = pts_mod + 3 * i;
    if (!vec_valid(p)) continue;
    static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float o[3];
    mat_vec(I, p, o);
    p[0] = o[0] + t_tmp[0];
    p[1] = o[1] + t_tmp[1];
    p[2] = o[2] + t_tmp[2];
  }
  orc_icp(pts_ref, np, pts_mod, np, icp_it_thr, dist_mean_thr, dist_diff_thr, accum64, use_kdtree,
          &res->icp, NULL, 0);                                                     /* :228 */
  float Rt[3];
  mat_vec(res->icp.R, t_init, Rt);                                                 /* :232-234 */
  for (int k = 0; k < 3; ++k) res->T_final[k] = Rt[k] + res->icp.T[k];
  mat_mat(res->icp.R, r_match, res->R_final);
  free(p_ref);
  free(p_mod);
  free(pts_ref);
  free(pts_mod);
  return 0;
}

/* CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:86-204), input already 640 wide */
/* the part of Recognition() after the match has been chosen (obj_reco_lmicp.cpp:111-197) */
/* wall clock of the stage timers (cv::getTickCount in the reference) */
static __thread double orc_stage_ms[2];
static double stage_clock_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static int recognition_refine(const orc_match best, const uint16_t *depth, int w, int h, double fx, double fy, double cx, double cy,
                              const orc_bank *bank, const float *poses13, const uint16_t *const *model_depths_01mm,
                              int icp_it_thr, float dist_mean_thr, float dist_diff_thr, int accum64, int use_kdtree,
                              orc_recognition_result *res)
{
  res->found = 1;
  res->best = best;
  const orc_template *t0 = bank->templates + (size_t)best.template_id * bank->levels * bank->modalities;
  int rect_model[4] = {t0->offset_x, t0->offset_y, t0->width, t0->height};          /* :127-132 */
  int rect_ref[4] = {best.x, best.y, t0->width, t0->height};
  const float *p = poses13 + 13 * (size_t)best.template_id;                        /* :141-152 */
  float r_match[9], t_match[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) r_match[i * 3 + j] = p[i * 4 + j];
    t_match[i] = p[i * 4 + 3];
  }
  /* depImg_model_raw.convertTo(.., CV_16UC1, 0.1) (:188): float multiply, round half even, saturate */
  size_t npx = (size_t)w * h;
  uint16_t *md = (uint16_t *)malloc(sizeof(uint16_t) * npx);
  const uint16_t *src = model_depths_01mm[best.template_id];
  for (size_t i = 0; i < npx; ++i) {
    long v = lrintf((float)src[i] * 0.1f);
    md[i] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
  }
  int rc = orc_detection(md, depth, w, h, fx, fy, cx, cy, rect_model, rect_ref, icp_it_thr,
                         dist_mean_thr, dist_diff_thr, r_match, t_match, accum64, use_kdtree, &res->det);
  free(md);
  if (rc) return rc;
  for (int i = 0; i < 3; ++i) {                                                    /* Convert :20-30 */
    for (int j = 0; j < 3; ++j) res->pose[i * 4 + j] = res->det.R_final[i * 3 + j];
    res->pose[i * 4 + 3] = res->det.T_final[i];
  }
  res->pose[12] = res->pose[13] = res->pose[14] = 0;
  res->pose[15] = 1;
  return 0;
}

int orc_recognition(const uint8_t *bgr, const uint16_t *depth, int w, int h,
                    double fx, double fy, double cx, double cy,
                    int levels, const int *T_at_level, const orc_bank *bank,
                    const float *poses13, const uint16_t *const *model_depths_01mm,
                    float threshold, int icp_it_thr, float dist_mean_thr, float dist_diff_thr,
                    int accum64, int use_kdtree, orc_recognition_result *res)
{
  memset(res, 0, sizeof(*res));
  orc_match best;
  int n_total = 0;
  /* the reference's own timer points: "Time of linemod" = Detector::match (obj_reco_lmicp.cpp:88,124-125),
   * "Time of ICP" = everything after it up to the pose (:126,201-202) */
  const double t0 = stage_clock_ms();
  orc_stage_ms[0] = orc_stage_ms[1] = 0.0;
  int n = orc_match_images(bgr, depth, w, h, levels, T_at_level, bank, 1, threshold, &best, 1, &n_total, NULL);
  const double t1 = stage_clock_ms();
  orc_stage_ms[0] = t1 - t0;
  if (n < 0) return -1;                                  /* ERROR_INVALID_PARAM :102-105 */
  res->n_matches = n_total;
  if (n == 0) return 0;                                  /* :106-109 */
  const int rc = recognition_refine(best, depth, w, h, fx, fy, cx, cy, bank, poses13, model_depths_01mm, icp_it_thr, dist_mean_thr,
                                    dist_diff_thr, accum64, use_kdtree, res);
  orc_stage_ms[1] = stage_clock_ms() - t1;
  return rc;
}

/* the two stage times (ms) of the calling thread's last orc_recognition: [0] "Time of linemod", [1] "Time of ICP" */
void orc_last_stage_ms(double out[2])
{
  out[0] = orc_stage_ms[0];
  out[1] = orc_stage_ms[1];
}

/* SURVEY 8f rank 3 -- the multi-hypothesis pipeline the reference's dead code sketches: the same refinement for the
 * first k matches instead of matches[0] only (results[r] for r < min(k, n_matches); returns that count or -1) ... */
int orc_recognition_topk(const uint8_t *bgr, const uint16_t *depth, int w, int h, double fx, double fy, double cx, double cy,
                         int levels, const int *T_at_level, const orc_bank *bank, const float *poses13,
                         const uint16_t *const *model_depths_01mm, float threshold, int icp_it_thr, float dist_mean_thr,
                         float dist_diff_thr, int accum64, int use_kdtree, int k, orc_recognition_result *results)
{
  orc_match *m = (orc_match *)malloc(sizeof(orc_match) * (size_t)(k > 0 ? k : 1));
  int n_total = 0;
  int n = orc_match_images(bgr, depth, w, h, levels, T_at_level, bank, 1, threshold, m, k, &n_total, NULL);
  if (n < 0) { free(m); return -1; }
  for (int r = 0; r < n; ++r) {
    memset(&results[r], 0, sizeof(results[r]));
    results[r].n_matches = n_total;
    int rc = recognition_refine(m[r], depth, w, h, fx, fy, cx, cy, bank, poses13, model_depths_01mm, icp_it_thr, dist_mean_thr,
                                dist_diff_thr, accum64, use_kdtree, &results[r]);
    if (rc) results[r].found = 0;                        /* e.g. Q10: the crop leaves the image -- this hypothesis is skipped */
  }
  free(m);
  return n;
}

/* ... and nonMaximumSuppression (ICP/NMS.cpp:6-40) over the refined hypotheses, in list order: a hypothesis not yet
 * absorbed opens a group; every later one within th_obj_dist of the group's CURRENT best (cv::norm of the t
 * difference, double accumulate) is absorbed, and replaces the best if it has more than 0.85 x the opener's points
 * and a smaller ICP distance.  Writes the index of each group's winner; returns the number of groups. */
int orc_nms(const orc_recognition_result *objs, int n, float th_obj_dist, int *winners)
{
  char *done = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
  int n_out = 0;
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    int o = i;
    const int size_th = (int)((float)objs[i].det.n_points * 0.85);
    for (int j = i + 1; j < n; ++j) {
      if (done[j]) continue;
      double s = 0;
      for (int c = 0; c < 3; ++c) { double d = (double)objs[o].det.T_final[c] - (double)objs[j].det.T_final[c]; s += d * d; }
      if (sqrt(s) < th_obj_dist) {
        done[j] = 1;
        if (objs[j].det.n_points > size_th && objs[j].det.icp.dist_mean < objs[o].det.icp.dist_mean) o = j;
      }
    }
    winners[n_out++] = o;
  }
  free(done);
  return n_out;
}
