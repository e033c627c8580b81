// fl_icp.hip -- gfx950 kernels for the ICP half of the FEALESS hot path.
//
// Replaces (reference paths): cup_d2pc::depthTo3d (ICP/depth_to_3d.cpp:99-137,190-269),
// scale_mat_vec3f / matToVec / is_vec3f_valid (ICP/common.cpp:261-266,382-425), detection()
// (ICP/detection.cpp:11-254), icpCloudToCloud_Ex + getMean/transformPoints/getL2distClouds/
// PointsCorresponding (ICP/ICP.cpp:8-111,193-279,617-809) and the tail of
// CObjRecoLmICP::Recognition (CadReco/obj_reco_lmicp.cpp:111-199).
//
// Execution model: ONE workgroup owns one frame's whole refinement -- crop back-projection,
// paired-valid compaction, centroid pre-alignment, every ICP iteration and the final pose -- in
// a single launch; a batch is a grid of such workgroups.  No host round trip and no inter-
// workgroup synchronisation exists anywhere in the path: phases inside a frame are separated by
// workgroup barriers only.  Parallelism comes from frames (hundreds to thousands resident in
// 288 GB HBM), which is what lets the reference's strictly sequential float32 sums be kept:
//   * FL_ICP_PARITY: each of the 15 centroid/covariance scalars (and the distance sum) is a
//     float32 chain accumulated in the reference's order by its own lane; dropped pairs add an
//     exact +0.0f so the chain is branch-free.  Bit-identical to the reference's arithmetic.
//   * FL_ICP_FAST: per-thread partials + fixed-shape fp64 tree, rounded once to float32.
//   * FL_ICP_POINT_TO_PLANE (no reference counterpart): NN pairs from the first
//     iteration on, 27 fp64 sums of the linearised point-to-plane system, 6x6 Cholesky + Rodrigues.
// Nearest neighbours: the reference's FLANN kd-tree (exact 1-NN, eps 0) is replaced by a uniform
// x/y cell grid over the static reference cloud built once per frame; a query only visits the
// cells within sqrt(3*dist_mean) because farther neighbours are discarded anyway
// (PointsCorresponding keeps d^2 <= 3*dist_mean, ICP.cpp:268,708).  Distances use L2_Simple's
// float expression ((dx*dx + dy*dy) + dz*dz); ties go to the lowest index.
// Built with -ffp-contract=off: one IEEE binary32/64 operation per operator.
#include "fl_internal.h"
#include <float.h>
#include <math.h>
#include <string.h>
#include <type_traits>

#ifndef FL_ICP_BS
#define FL_ICP_BS 256
#endif
#define ICP_BS FL_ICP_BS         // threads per frame workgroup (64 chain lanes + ICP_BS - 64 producers in parity mode)
#ifdef FL_ICP_PHASES
#define TSTAMP(k) do { if (threadIdx.x == 0) { long long now_ = clock64(); S.tacc[k] += now_ - S.tlast; S.tlast = now_; } } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif
#define ICP_MAX_THREADS ICP_BS
#ifndef FL_ICP_WPE
#define FL_ICP_WPE 5               // waves per SIMD the recognition kernel is compiled for (4 -> 128 VGPRs, 5 -> 96):
                                  // measured +4 % frames/s at 5 workgroups per CU (LDS: 5 x 27 KB)
#endif
#ifndef FL_ICP_FAST_F32
#define FL_ICP_FAST_F32 1          // FL_ICP_FAST keeps its per-thread partial sums (~60 terms) in float32, like the point-to-plane mode;
                                  // the cross-thread tree is fp64.  0: fp64 partials (161 VGPRs, 3 workgroups per CU: 18.6 ms per 1280 frames)
#endif
#ifndef FL_ICP_FAST_WPE
#define FL_ICP_FAST_WPE 5          // waves per SIMD the FL_ICP_FAST kernel is compiled for (96 VGPRs with float32 partials: 15.8 ms per 1280 frames)
#endif
#ifndef FL_ICP_PLANE_WPE
#define FL_ICP_PLANE_WPE 5         // waves per SIMD the point-to-plane kernel is compiled for (96 VGPRs, 5 workgroups per CU; ICP ms per 1280 frames: 3: 41.0, 4: 37.3, 5: 36.3)
#endif
// waves per SIMD kernel k_icp_pipeline<MODE> is compiled for (minimum; the maximum is 4 unless more is asked for)
#define ICP_MODE_WPE(MODE) ((MODE) == FL_ICP_PARITY ? FL_ICP_WPE : ((MODE) == FL_ICP_POINT_TO_PLANE ? FL_ICP_PLANE_WPE : FL_ICP_FAST_WPE))
#ifndef FL_ICP_NB
#define FL_ICP_NB 10              // candidates fetched per round trip of the NN search (measured: 8..20)
#endif
#define ICP_TQ (ICP_BS - 64)        // rows per LDS tile: virtual wave 0 chains, the other three produce one row per thread
#define ICP_TS (ICP_TQ + 4)         // tile column stride (floats): 16-byte aligned columns for ds_read_b128; the +4 keeps
                                  // the 16 chain lanes of a b128 read on distinct bank groups

// HBM layout of one frame's ICP workspace (n = capacity in points):
//   ref   n x 3 f32   reference cloud, index order (pairing + iteration 1)
//   mod   n x 3 f32   model cloud, transformed in place every iteration
//   sref  n x float4  reference cloud sorted by grid cell, w = original index (bit pattern)
//   nn     n x i32    nearest reference index j of model point i (kept pair: j, dropped: -1)
//   bnd    n x f32    upper bound on the distance from model point i to its nearest reference point
//   cell_start / cell_cur   CSR offsets of the x/y cell grid
//   nrm   n x 3 f32   unit normals of the reference cloud, index order (FL_ICP_POINT_TO_PLANE only; 0 = unknown)
struct IcpWsLayout {
  size_t ref, mod, sref, nn, bnd, cell_start, cell_cur, nrm, total;
  int ncell_max;
};
static __host__ __device__ inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static __host__ __device__ inline IcpWsLayout icp_layout(int n)
{
  IcpWsLayout L;
  size_t o = 0, nn = (size_t)(n > 0 ? n : 1);
  L.ncell_max = n + 4096;
  L.ref = o; o = al256(o + 12 * nn);
  L.mod = o; o = al256(o + 12 * nn);
  L.sref = o; o = al256(o + 16 * nn);
  L.nn = o; o = al256(o + 4 * nn);
  L.bnd = o; o = al256(o + 4 * nn);
  L.cell_start = o; o = al256(o + 4 * ((size_t)L.ncell_max + 1));
  L.cell_cur = o; o = al256(o + 4 * (size_t)L.ncell_max);
  L.nrm = o; o = al256(o + 12 * nn);
  L.total = o;
  return L;
}
size_t fl_icp_ws_bytes(int n_pts_max) { return icp_layout(n_pts_max).total; }

struct IcpJob {          // what one workgroup needs besides its workspace
  int kind;              // 0: recognition (read the frame's best match), 1: detection, 2: clouds given
  int n_ref, n_model;    // kind 2
  int rect_model[4], rect_ref[4];      // kind 1
  float r_match[9], t_match[3];        // kind 1
  const uint16_t *model_depth;         // kind 1 (mm)
  const uint16_t *scene_depth;         // kind 1 (mm)
};

struct IcpArgs {
  uint8_t *ws;           // frame 0's ICP workspace
  size_t ws_stride;
  int n_max;
  // scene
  int w, h;
  float fx, fy, cx, cy;  // (float)dFx ... as Mat_<float> K would hold them (common.cpp:374-379)
  int it_thr;
  float dmt, ddt;
  int mode;
  IcpJob job;            // kinds 1, 2
  // kind 0: recognition batch
  const uint8_t *frame_ws;   // detector frame workspaces
  size_t frame_stride;       // their stride (= ws_stride in the per-frame pipeline)
  int ranks;                 // hypotheses per frame: workgroup b refines match (b % ranks) of frame (b / ranks)
  size_t off_count, off_match;
  const uint16_t *scene_base;   // frame i's depth = scene_base + i*scene_stride (bytes)
  size_t scene_stride;
  const FlPyrInfo *pyr;
  const int *class_first;
  const float *poses;
  const uint16_t *const *depth_ptrs;
  fl_recognition_result *results;
};

struct IcpShared {
  float R[9], T[3], Ropt[9], Topt[3];
  float dist_mean, dist_diff, px, thr;
  int iter, n_corr, go, ok;
  float xmin, ymin, inv_c;
  int GX, GY, nsorted;
  float sums[16];
  double dsum[ICP_BS / 64][32];   // [wave][scalar]
  int iscan[ICP_BS / 64 + 1];
  int iscan2[2][ICP_BS / 64];          // crop_clouds: per-wave kept counts, double-buffered
  float fred[4][ICP_BS / 64];
  int n, rect_m[4], rect_r[4], status, g;
  // double-buffered LDS tiles feeding the sequential float32 chains (FL_ICP_PARITY):
  // prod[b][k][r] = scalar k (9 products, 3 model coords, 3 reference coords, pad) of row r of tile b;
  // +1 column of padding puts the 16 chain lanes on 16 different banks
  alignas(16) float prod[2][15][ICP_TS];
  alignas(16) float dtile[2][ICP_TQ];
#ifdef FL_ICP_PHASES
  long long tacc[16], tlast, tkernel;   // tkernel: clock at kernel entry (k_icp_pipeline)
#endif
};

__device__ __forceinline__ bool vvalid(float z) { return z <= 900.0f; }      // common.cpp:261-266

// ---- block-level helpers (every thread of the workgroup must call) ---------------------------
__device__ __forceinline__ int block_sum_int(IcpShared &S, int v)
{
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) S.iscan[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += S.iscan[i];
  return t;
}

__device__ __forceinline__ double shfl_xor_d(double v, int s)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, s, 64);
  hi = __shfl_xor(hi, s, 64);
  return __hiloint2double(hi, lo);
}

// fixed-shape fp64 reduction of NS scalars per thread -> S.sums[k] as float is NOT done here: the
// caller rounds.  Result (double) valid in thread 0..NS-1's return slot via S.dsum[0][k].
template <int NS>
__device__ __forceinline__ void block_sum_double(IcpShared &S, double *v)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double x = v[k];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x += shfl_xor_d(x, s);
    v[k] = x;
  }
  __syncthreads();
  if (lane == 0)
    for (int k = 0; k < NS; ++k) S.dsum[wave][k] = v[k];
  __syncthreads();
  if (threadIdx.x < NS) {
    double t = 0;
    for (int i = 0; i < nw; ++i) t += S.dsum[i][threadIdx.x];
    S.dsum[0][threadIdx.x] = t;
  }
  __syncthreads();
}

// ---- 3x3 helpers with cv::Matx float semantics (s = 0; s += a*b ...) --------------------------
__device__ __forceinline__ void mat_vec(const float *R, const float *v, float *o)
{
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    s += R[i * 3 + 0] * v[0];
    s += R[i * 3 + 1] * v[1];
    s += R[i * 3 + 2] * v[2];
    o[i] = s;
  }
}
__device__ __forceinline__ void mat_mat(const float *A, const float *B, float *O)
{
  float t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      t[i * 3 + j] = s;
    }
  for (int i = 0; i < 9; ++i) O[i] = t[i];
}

// cv::SVD::compute on 3x3 CV_32F: OpenCV JacobiSVDImpl_<float> restated (same text as
// oracle/icp_oracle.c orc_svd3, which documents the one deviation: hypot -> sqrt(p*p+b*b)).
__device__ __noinline__ void svd3(const float *A, float *U, float *Vt)
{
  float At[9];
  double W[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) At[i * 3 + j] = A[j * 3 + i];
  const float eps = FLT_EPSILON * 2;
  const double minval = FLT_MIN;
  for (int i = 0; i < 3; ++i) {
    double sd = 0;
    for (int k = 0; k < 3; ++k) { float t = At[i * 3 + k]; sd += (double)t * t; }
    W[i] = sd;
    for (int k = 0; k < 3; ++k) Vt[i * 3 + k] = 0;
    Vt[i * 3 + i] = 1;
  }
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
    for (int i = 0; i < 2; ++i)
      for (int j = i + 1; j < 3; ++j) {
        float *Ai = At + i * 3, *Aj = At + j * 3;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < 3; ++k) p += (double)Ai[k] * Aj[k];
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
        for (int k = 0; k < 3; ++k) {
          float t0 = c * Ai[k] + s * Aj[k];
          float t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += (double)t0 * t0;
          b += (double)t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        float *Vi = Vt + i * 3, *Vj = Vt + j * 3;
        for (int k = 0; k < 3; ++k) {
          float t0 = c * Vi[k] + s * Vj[k];
          float t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < 3; ++i) {
    double sd = 0;
    for (int k = 0; k < 3; ++k) { float t = At[i * 3 + k]; sd += (double)t * t; }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < 2; ++i) {
    int j = i;
    for (int k = i + 1; k < 3; ++k)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      for (int k = 0; k < 3; ++k) { float f = At[i * 3 + k]; At[i * 3 + k] = At[j * 3 + k]; At[j * 3 + k] = f; }
      for (int k = 0; k < 3; ++k) { float f = Vt[i * 3 + k]; Vt[i * 3 + k] = Vt[j * 3 + k]; Vt[j * 3 + k] = f; }
    }
  }
  for (int i = 0; i < 3; ++i) {
    double sd = W[i];
    float s = (float)(sd > minval ? 1 / sd : 0.);
    for (int k = 0; k < 3; ++k) At[i * 3 + k] *= s;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) U[i * 3 + j] = At[j * 3 + i];
}

__device__ __forceinline__ bool finite_all(const float *v, int n)
{
  for (int i = 0; i < n; ++i)
    if (!isfinite(v[i])) return false;
  return true;
}

// ---- sequential float32 chains (FL_ICP_PARITY) ------------------------------------------------
// one chain step over an LDS tile: acc += col[0], col[1], ... col[rows-1], strictly in order.
// The adds are one dependent chain; what made a tile slow was the LDS read latency in front of every 16 of
// them.  The column (16-byte aligned) is read 16 rows at a time with four ds_read_b128, two batches in flight:
// batch B is issued before batch A is added and vice versa, so the reads overlap the chain.
__device__ __forceinline__ void chain_load16(const float *col, float4 (&v)[4])
{
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = *(const float4 *)(col + 4 * u);
}
__device__ __forceinline__ float chain_add16(const float4 (&v)[4], float acc)
{
#pragma unroll
  for (int u = 0; u < 4; ++u) { acc += v[u].x; acc += v[u].y; acc += v[u].z; acc += v[u].w; }
  return acc;
}
__device__ __forceinline__ float chain_tile(const float *col, int rows, float acc)
{
  int r = 0;
  if (rows >= 16) {
    float4 a[4], b[4];
    chain_load16(col, a);
    for (; r + 48 <= rows; r += 32) {
      chain_load16(col + r + 16, b);
      acc = chain_add16(a, acc);
      chain_load16(col + r + 32, a);
      acc = chain_add16(b, acc);
    }
    // here: batch at r is loaded in a; 16 <= rows - r < 48
    if (r + 32 <= rows) {
      chain_load16(col + r + 16, b);
      acc = chain_add16(a, acc);
      acc = chain_add16(b, acc);
      r += 32;
    } else {
      acc = chain_add16(a, acc);
      r += 16;
    }
  }
  for (; r < rows; ++r) acc += col[r];
  return acc;
}

// uniform base + 32-bit unsigned byte offset: one VGPR per address (global_load ... v_off, s[base]) instead of a
// sign-extended 64-bit pointer pair -- the search keeps 24 addresses in flight
template <typename T>
__device__ __forceinline__ T ld_u32(const T *__restrict__ base, int idx)
{
  return *(const T *)((const char *)base + (size_t)((unsigned)idx * (unsigned)sizeof(T)));
}

// one 12-byte load (global_load_dwordx3) for a point instead of three dword loads: the phases are bound by the
// number of vector-memory instructions as much as by anything else
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3_u32(const float *__restrict__ base, int i)
{
  F3 v;
  __builtin_memcpy(&v, (const char *)base + (size_t)((unsigned)i * 12u), 12);
  return v;
}

// An UPPER bound of sqrt(x) for the search-radius bookkeeping (bnd[]): the hardware's 1-ulp v_sqrt_f32 inflated past
// its error (and past a flushed denormal) instead of the ~15-instruction correctly rounded sqrtf.  Any over-estimate
// only widens the visited area; the nearest neighbour found is the same.
__device__ __forceinline__ float sqrt_upper(float x) { return __builtin_amdgcn_sqrtf(x) * 1.000001f + 1.1e-19f; }

// ---- uniform x/y grid over the reference cloud --------------------------------------------------
__device__ __forceinline__ int cell_of(float v, float vmin, float inv_c, int G)
{
  float t = floorf((v - vmin) * inv_c);
  int c = t < 0.f ? 0 : (t > (float)(G - 1) ? G - 1 : (int)t);
  return c;
}

__device__ __forceinline__ void build_grid(IcpShared &S, const float *ref, int n_ref, float4 *sref, int *cell_start, int *cell_cur,
                           int ncell_max)
{
  // bounding box of the finite points
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (int i = threadIdx.x; i < n_ref; i += blockDim.x) {
    const F3 p3 = ld3_u32(ref, i);
    const float x = p3.x, y = p3.y, z = p3.z;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      xmin = fminf(xmin, x);
      xmax = fmaxf(xmax, x);
      ymin = fminf(ymin, y);
      ymax = fmaxf(ymax, y);
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    xmin = fminf(xmin, __shfl_xor(xmin, s, 64));
    xmax = fmaxf(xmax, __shfl_xor(xmax, s, 64));
    ymin = fminf(ymin, __shfl_xor(ymin, s, 64));
    ymax = fmaxf(ymax, __shfl_xor(ymax, s, 64));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const int wv = threadIdx.x >> 6;
    S.fred[0][wv] = xmin;
    S.fred[1][wv] = xmax;
    S.fred[2][wv] = ymin;
    S.fred[3][wv] = ymax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      xmin = fminf(xmin, S.fred[0][i]);
      xmax = fmaxf(xmax, S.fred[1][i]);
      ymin = fminf(ymin, S.fred[2][i]);
      ymax = fmaxf(ymax, S.fred[3][i]);
    }
    if (!(xmax >= xmin)) { xmin = xmax = 0.f; ymin = ymax = 0.f; }
    const float dx = xmax - xmin, dy = ymax - ymin;
    float c = sqrtf((dx * dy) / (float)(n_ref > 0 ? n_ref : 1));   // about one point per cell on a dense surface
    // scale-free guards: a (nearly) collinear cloud gets cells of extent / sqrt(n); coincident points one cell
    const float ext = fmaxf(dx, dy);
    if (!(c > ext * 1e-4f)) c = ext / sqrtf((float)(n_ref > 0 ? n_ref : 1));
    if (!(c > 0.f) || !isfinite(c)) c = 1.0f;
    int GX, GY;
    for (;;) {
      GX = (int)(dx / c) + 1;
      GY = (int)(dy / c) + 1;
      if ((long long)GX * GY <= ncell_max) break;
      c *= 1.5f;
    }
    S.xmin = xmin;
    S.ymin = ymin;
    S.inv_c = 1.0f / c;
    S.GX = GX;
    S.GY = GY;
  }
  __syncthreads();
  const int ncell = S.GX * S.GY;
  for (int i = threadIdx.x; i < ncell; i += blockDim.x) cell_cur[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_ref; i += blockDim.x) {
    const F3 p3 = ld3_u32(ref, i);
    const float x = p3.x, y = p3.y, z = p3.z;
    if (isfinite(x) && isfinite(y) && isfinite(z))
      atomicAdd(&cell_cur[cell_of(y, S.ymin, S.inv_c, S.GY) * S.GX + cell_of(x, S.xmin, S.inv_c, S.GX)], 1);
  }
  __syncthreads();
  // exclusive scan of the counts -> cell_start (and cell_cur, the scatter cursors): four consecutive cells per thread,
  // a shuffle scan inside the wave, the wave totals through a double-buffered LDS slot -- one barrier per 1024 cells
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    int run = 0, step = 0;
    for (int base = 0; base < ncell; base += 4 * blockDim.x, ++step) {
      const int i0 = base + 4 * threadIdx.x;
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = i0 + u < ncell ? cell_cur[i0 + u] : 0;
      const int mine = (c[0] + c[1]) + (c[2] + c[3]);
      int inc = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
      }
      int *slot = S.iscan2[step & 1];
      if (lane == 63) slot[wv] = inc;
      __syncthreads();
      int before = 0, total = 0;
      for (int w = 0; w < nwv; ++w) { const int t = slot[w]; before += w < wv ? t : 0; total += t; }
      int ex = run + before + inc - mine;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + u < ncell) { cell_start[i0 + u] = ex; cell_cur[i0 + u] = ex; }
        ex += c[u];
      }
      run += total;
    }
    if (threadIdx.x == 0) { cell_start[ncell] = run; S.nsorted = run; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_ref; i += blockDim.x) {
    const F3 p3 = ld3_u32(ref, i);
    const float x = p3.x, y = p3.y, z = p3.z;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      const int slot = atomicAdd(&cell_cur[cell_of(y, S.ymin, S.inv_c, S.GY) * S.GX + cell_of(x, S.xmin, S.inv_c, S.GX)], 1);
      sref[slot] = make_float4(x, y, z, __int_as_float(i));
    }
  }
  __syncthreads();
}

// ---- exact 1-NN within squared distance thr (float compare as the reference's `dists <= dist_thr`)
// cell ranges a query has to visit; false if the query cannot have a neighbour at all.
// `bnd` is an upper bound on the distance to SOME reference point (last iteration's partner plus how far the
// query moved since, see l2dist_phase): it only shrinks the visited area -- every point within that distance
// is still seen, so the result is the exact nearest neighbour.  r_thr = sqrtf(thr).
// the grid parameters as wave-uniform scalars (SGPRs): read from LDS they would each cost a VGPR in the search loop
struct NnGrid {
  float xmin, ymin, inv_c;
  int GX, GY, nsorted;
};
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ NnGrid nn_grid(const IcpShared &S)
{
  NnGrid g;
  g.xmin = uniform_f(S.xmin);
  g.ymin = uniform_f(S.ymin);
  g.inv_c = uniform_f(S.inv_c);
  g.GX = __builtin_amdgcn_readfirstlane(S.GX);
  g.GY = __builtin_amdgcn_readfirstlane(S.GY);
  g.nsorted = __builtin_amdgcn_readfirstlane(S.nsorted);
  return g;
}

__device__ __forceinline__ bool nn_ranges(const NnGrid &S, float qx, float qy, float qz, float thr, float r_thr, float bnd,
                                          int *cx0, int *cx1, int *cy0, int *cy1)
{
  if (!(thr >= 0.f && isfinite(qx) && isfinite(qy) && isfinite(qz))) return false;
  const float lim = fminf(bnd, r_thr);                   // NaN bnd -> r_thr
  // conservative radius: float rounding of d2, of the coordinate differences and of the bound's own
  // arithmetic is orders of magnitude below the relative margins
  const float r = lim * 1.0001f + 2e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz)) + 1e-30f;
  *cx0 = 0; *cx1 = S.GX - 1; *cy0 = 0; *cy1 = S.GY - 1;
  if (isfinite(r)) {
    *cx0 = cell_of(qx - r, S.xmin, S.inv_c, S.GX);
    *cx1 = cell_of(qx + r, S.xmin, S.inv_c, S.GX);
    *cy0 = cell_of(qy - r, S.ymin, S.inv_c, S.GY);
    *cy1 = cell_of(qy + r, S.ymin, S.inv_c, S.GY);
  }
  return true;
}

// Branch-free running minimum: (d2, index) packed as d2's bit pattern (non-negative floats order
// like unsigned integers) in the high word and the reference index in the low word, so one
// 64-bit unsigned min implements "smaller distance, ties to the lower index" exactly.
#define NN_CONSIDER(P)                                                                            \
  {                                                                                               \
    const float dx = qx - (P).x, dy = qy - (P).y, dz = qz - (P).z;                                \
    float d = dx * dx; /* cvflann::L2_Simple<float> */                                            \
    d += dy * dy;                                                                                 \
    d += dz * dz;                                                                                 \
    const unsigned long long key_ =                                                               \
        ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int((P).w);         \
    best = key_ < best ? key_ : best;                                                             \
  }
#define NN_CONSIDER_IF(P, PRED)                                                                   \
  {                                                                                               \
    const float dx = qx - (P).x, dy = qy - (P).y, dz = qz - (P).z;                                \
    float d = dx * dx;                                                                            \
    d += dy * dy;                                                                                 \
    d += dz * dz;                                                                                 \
    const unsigned long long key_ =                                                               \
        ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int((P).w);         \
    best = ((PRED) && key_ < best) ? key_ : best;                                                 \
  }
#define NN_KEY_NONE 0xFFFFFFFFFFFFFFFFull
#define NN_UNPACK(best, bi, bd)                                            \
  {                                                                        \
    const bool found_ = (best) != NN_KEY_NONE;                             \
    *(bi) = found_ ? (int)((best) & 0xFFFFFFFFull) : -1;                   \
    *(bd) = found_ ? __uint_as_float((unsigned)((best) >> 32)) : NAN;      \
  }

// search of the cell rows [cy0, cy1] x [cx0, cx1]; the grid and the sorted cloud are L2-resident
__device__ __forceinline__ void nn_search_global(const NnGrid &S, const float4 *__restrict__ sref,
                                                 const int *__restrict__ cell_start, float qx, float qy, float qz, int cx0,
                                                 int cx1, int cy0, int cy1, int *bi, float *bd)
{
  unsigned long long best = NN_KEY_NONE;
  const int last = S.nsorted - 1;
  if (last < 0) { NN_UNPACK(best, bi, bd) return; }
  // The search is latency-bound and a wave pays for its slowest lane, so round trips are what counts:
  // the headers of 4 grid rows (8 loads) are fetched together, then the candidates of all 4 row segments
  // are enumerated as ONE flat list, FL_ICP_NB per round trip -- a lane needs ceil(total / NB) rounds however the
  // candidates are spread over the rows.  Slots past the end of the list are NOT masked: they read points that
  // follow the last row segment (clamped to the cloud), and looking at extra reference points never changes the
  // answer -- the minimum over a superset that still contains every point within the search radius is the same
  // nearest neighbour, ties to the lowest index included.  (Chunked per-row reads with no index mapping at all
  // were measured too: fewer instructions per candidate, but a wave then pays for its longest ROW, not its
  // longest list: 20.3 ms vs 17.6 ms.)
  for (int cy = cy0; cy <= cy1; cy += 4) {
    int rb[4], re[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cyu = min(cy + u, cy1);
      rb[u] = ld_u32(cell_start, cyu * S.GX + cx0);      // cells of a row are contiguous
      re[u] = ld_u32(cell_start, cyu * S.GX + cx1 + 1);
    }
#pragma unroll
    for (int u = 1; u < 4; ++u)
      if (cy + u > cy1) re[u] = rb[u];                  // predicated: keeps rb/re in registers
    // flat index k -> slot k + adj[u] for pre[u] <= k < pre[u + 1]
    const int pre1 = re[0] - rb[0], pre2 = pre1 + (re[1] - rb[1]), pre3 = pre2 + (re[2] - rb[2]);
    const int tot = pre3 + (re[3] - rb[3]);
    const int adj0 = rb[0], adj1 = rb[1] - pre1, adj2 = rb[2] - pre2, adj3 = rb[3] - pre3;
    for (int base = 0; base < tot; base += FL_ICP_NB) {
      float4 p[FL_ICP_NB];
#pragma unroll
      for (int v = 0; v < FL_ICP_NB; ++v) {
        const int k = base + v;
        int adj = k >= pre1 ? adj1 : adj0;
        adj = k >= pre2 ? adj2 : adj;
        adj = k >= pre3 ? adj3 : adj;
        p[v] = ld_u32(sref, min(k + adj, last));
      }
#pragma unroll
      for (int v = 0; v < FL_ICP_NB; ++v) NN_CONSIDER(p[v])
    }
  }
  NN_UNPACK(best, bi, bd)
}

// ---- getL2distClouds (ICP.cpp:68-111) over the index-paired clouds, optionally fused with the
// in-place transformPoints that precedes it (ICP.cpp:28-45, 756).  FL_ICP_PARITY: waves 1-3 write
// the per-point terms into double-buffered LDS tiles while lane 0 of wave 0 adds the previous tile
// in index order (the reference's `dist_mean += dist` chain); one barrier per tile.
template <int MODE>
__device__ __forceinline__ void l2dist_phase(IcpShared &S, float *mod, const float *ref, float *bnd, int n, float thr,
                                             const float *Ropt, const float *Topt)
{
  constexpr bool parity = MODE == FL_ICP_PARITY;
  const int TQ = parity ? ICP_BS - 64 : ICP_BS;          // rows per tile: wave 0 only chains in parity mode
  const int slot = parity ? (int)threadIdx.x - 64 : (int)threadIdx.x;
  int counter = 0, inl = 0;
  double dsum[1] = {0.0};
  float acc = 0.0f;
  const int ntiles = (n + TQ - 1) / TQ;
  // The phase is one memory round trip + one barrier per tile, so the next tile's (coalesced) loads are
  // issued before this tile is processed: the round trip overlaps the chain of the previous tile.
  float pa[3] = {0.f, 0.f, 0.f}, pb[3] = {0.f, 0.f, 0.f}, pbnd = 0.f;
  if (slot >= 0 && slot < n) {
    const int i = slot;
    { const F3 v3_ = ld3_u32(mod, i); pa[0] = v3_.x; pa[1] = v3_.y; pa[2] = v3_.z; }
    { const F3 v3_ = ld3_u32(ref, i); pb[0] = v3_.x; pb[1] = v3_.y; pb[2] = v3_.z; }
    if (Ropt) pbnd = ld_u32(bnd, i);
  }
  for (int t = 0; t < ntiles; ++t) {
    if (slot >= 0) {
      const int i = t * TQ + slot;
      float term = 0.0f;
      float a[3] = {pa[0], pa[1], pa[2]};
      const float b0 = pb[0], b1 = pb[1], b2 = pb[2];
      const float bprev = pbnd;
      {
        const int in = min(i + TQ, n - 1);                // clamped: unused past the end (this thread owns row i + TQ)
        { const F3 v3_ = ld3_u32(mod, in); pa[0] = v3_.x; pa[1] = v3_.y; pa[2] = v3_.z; }
        { const F3 v3_ = ld3_u32(ref, in); pb[0] = v3_.x; pb[1] = v3_.y; pb[2] = v3_.z; }
        if (Ropt) pbnd = ld_u32(bnd, in);
      }
      if (i < n) {
        if (Ropt) {
          float move = 0.0f;
          if (vvalid(a[2])) {                             // transformPoints in place (:28-45, :756)
            float o[3];
            mat_vec(Ropt, a, o);
            o[0] += Topt[0];
            o[1] += Topt[1];
            o[2] += Topt[2];
            const float mx = o[0] - a[0], my = o[1] - a[1], mz = o[2] - a[2];
            move = sqrt_upper(mx * mx + my * my + mz * mz);
            a[0] = o[0];
            a[1] = o[1];
            a[2] = o[2];
            mod[3 * i] = a[0];
            mod[3 * i + 1] = a[1];
            mod[3 * i + 2] = a[2];
          }
          bnd[i] = bprev + move;                          // triangle inequality: still reaches the old partner
        } else {
          // first bound: the index pair (n_ref >= n_model); NaN/inf simply disable the bound
          const float ex = a[0] - b0, ey = a[1] - b1, ez = a[2] - b2;
          bnd[i] = sqrt_upper(ex * ex + ey * ey + ez * ez);
        }
        if (vvalid(b2) && vvalid(a[2])) {
          const float dx = a[0] - b0, dy = a[1] - b1, dz = a[2] - b2;
          // cv::norm(Vec3f): squares accumulated in double, sqrt in double, stored to float (:88)
          const float dist = (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
          if (dist <= thr) { term = dist; ++inl; dsum[0] += (double)dist; }
          ++counter;
        }
      }
      if (parity) S.dtile[t & 1][slot] = term;            // non-inliers add an exact +0.0f
    } else if (t > 0 && threadIdx.x == 0) {
      acc = chain_tile(S.dtile[(t - 1) & 1], min(TQ, n - (t - 1) * TQ), acc);
    }
    if (parity) __syncthreads();

  }
  if (parity && ntiles > 0 && threadIdx.x == 0)
    acc = chain_tile(S.dtile[(ntiles - 1) & 1], min(TQ, n - (ntiles - 1) * TQ), acc);
  counter = block_sum_int(S, counter);
  inl = block_sum_int(S, inl);
  float dm;
  if (parity) {
    if (threadIdx.x == 0) S.sums[0] = acc;
    __syncthreads();
    dm = S.sums[0];
    if (counter > 0) dm /= (float)inl;                   // 0/0 -> NaN ends the loop (Q9)
  } else {
    block_sum_double<1>(S, dsum);
    dm = counter > 0 ? (float)(S.dsum[0][0] / (double)inl) : 0.f;
  }
  if (threadIdx.x == 0) {
    if (counter > 0) {
      S.dist_mean = dm;
      S.px = (float)inl / (float)counter;
    } else {
      S.dist_mean = FLT_MAX;
      S.px = 0.0f;
    }
  }
  __syncthreads();
}

// ---- icpCloudToCloud_Ex (ICP.cpp:617-809) --------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void icp_run(IcpShared &S, uint8_t *wsb, const IcpWsLayout &L, int n_ref, int n_model,
                        int it_thr, float dmt, float ddt, fl_icp_result *res)
{
  float *ref = (float *)(wsb + L.ref), *mod = (float *)(wsb + L.mod);
  float4 *sref = (float4 *)(wsb + L.sref);
  int *nn = (int *)(wsb + L.nn);
  float *bnd = (float *)(wsb + L.bnd);
  const float *nrm = (const float *)(wsb + L.nrm);
  int *cell_start = (int *)(wsb + L.cell_start), *cell_cur = (int *)(wsb + L.cell_cur);
  constexpr bool plane = MODE == FL_ICP_POINT_TO_PLANE;
  constexpr int NSUM = plane ? 27 : 15;

  if (threadIdx.x == 0) {
    for (int i = 0; i < 9; ++i) S.R[i] = 0.f;           // cv::Matx33f R; cv::Vec3f T; zero-initialised
    for (int i = 0; i < 3; ++i) S.T[i] = 0.f;
    S.iter = 0;
    S.n_corr = 0;
    S.px = 0.f;
#ifdef FL_ICP_PHASES
    for (int i = 0; i < 16; ++i) S.tacc[i] = 0;
    S.tlast = clock64();
    S.tacc[6] = S.tlast - S.tkernel;
#endif
  }
  __syncthreads();
  if (n_model < 3 || n_ref < 3 || n_ref < n_model) {    // :633-638 (n_ref < n_model: reference reads OOB)
    if (threadIdx.x == 0) {
      for (int i = 0; i < 9; ++i) res->R[i] = 0.f;
      for (int i = 0; i < 3; ++i) res->T[i] = 0.f;
      res->dist_mean = -1.0f;
      res->px_ratio = 0.f;
      res->iters = 0;
      res->n_corr_last = 0;
    }
    __syncthreads();
    return;
  }
  build_grid(S, ref, n_ref, sref, cell_start, cell_cur, L.ncell_max);
  TSTAMP(0);
  // copyPoints(pts_model, pts_model_tmp) (:666-667): invalid points become Vec3f() = 0
  for (int i = threadIdx.x; i < n_model; i += blockDim.x)
    if (!vvalid(mod[3 * i + 2])) { mod[3 * i] = 0.f; mod[3 * i + 1] = 0.f; mod[3 * i + 2] = 0.f; }
  if (threadIdx.x == 0) {
    S.R[0] = S.R[4] = S.R[8] = 1.f;                      // R = eye, T = 0 (:644-645)
    S.dist_diff = FLT_MAX;
  }
  __syncthreads();
  l2dist_phase<MODE>(S, mod, ref, bnd, n_model, FLT_MAX, nullptr, nullptr);             // :670

  for (;;) {
    if (threadIdx.x == 0) S.go = (S.dist_mean > dmt) && (S.dist_diff > ddt) && (S.iter < it_thr);   // :684
    __syncthreads();
    if (!S.go) break;
    // point-to-plane gates pairs at distance 3*dist_mean; the reference compares the SQUARED distance
    // with 3*dist_mean (Q9), which parity/fast keep
    if (threadIdx.x == 0) { ++S.iter; S.thr = plane ? (3 * S.dist_mean) * (3 * S.dist_mean) : 3 * S.dist_mean; }
    __syncthreads();
    const int iter = S.iter;
    const float thr = S.thr;
    const int rows = iter == 1 ? n_ref : n_model;
    constexpr bool parity = MODE == FL_ICP_PARITY;
    constexpr int mode = MODE;
    int kept = 0;
    const bool index_pairs = iter == 1 && !plane;       // :700-704
    // fast: fp64 per-thread partials.  point-to-plane: the thread's ~60 terms are summed in float32 (27 registers
    // instead of 54 -- what lets the kernel run 5 workgroups per CU like the parity one) and only the cross-thread tree
    // runs in fp64; the 6x6 system is re-linearised every iteration, so a 1e-6 relative error in a sum is immaterial.
    typename std::conditional<plane || FL_ICP_FAST_F32 != 0, float, double>::type ds[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) ds[k] = 0.0;
    if (!index_pairs) {
      // Phase A1 -- PointsCorresponding (:193-279): all four waves search.  bnd[i] (distance to the partner found
      // last time plus the motion since; initially the index pair) bounds the search radius, so a converging
      // cloud visits only a handful of candidates per point; the result is still the exact 1-NN.  Every load
      // ahead of the search is coalesced and the next query's are issued before this query's search.
      const float r_thr = uniform_f(sqrtf(thr));
      const NnGrid G = nn_grid(S);
      int i = threadIdx.x;
      float qx = 0.f, qy = 0.f, qz = 0.f, qb = 0.f;
      if (i < n_model) { const F3 q3 = ld3_u32(mod, i); qx = q3.x; qy = q3.y; qz = q3.z; qb = ld_u32(bnd, i); }
      for (; i < n_model; i += ICP_BS) {
        const int in = min(i + ICP_BS, n_model - 1);       // clamped: unused past the end
        const F3 nq3 = ld3_u32(mod, in);
        const float nqx = nq3.x, nqy = nq3.y, nqz = nq3.z, nqb = ld_u32(bnd, in);
        int cx0, cx1, cy0, cy1, j = -1;
        float d = NAN;
        if (nn_ranges(G, qx, qy, qz, thr, r_thr, qb, &cx0, &cx1, &cy0, &cy1))
          nn_search_global(G, sref, cell_start, qx, qy, qz, cx0, cx1, cy0, cy1, &j, &d);
        const bool keep = d <= thr;                       // dists[i][0] <= dist_thr (:268)
        if (keep) ++kept;
        nn[i] = keep ? j : -1;
        if (j >= 0) bnd[i] = sqrt_upper(d);               // else: the old partner is still within qb
        if (plane && keep) {
          // linearised point-to-plane: residual e = n.(m - r), Jacobian row J = [m x n, n] wrt (omega, t);
          // ds = upper triangle of sum J J^T (21) followed by sum J e (6)
          const F3 rv = ld3_u32(ref, j), nv = ld3_u32(nrm, j);
          const float n0 = nv.x, n1 = nv.y, n2 = nv.z, m0 = qx, m1 = qy, m2 = qz;
          const float J[6] = {m1 * n2 - m2 * n1, m2 * n0 - m0 * n2, m0 * n1 - m1 * n0, n0, n1, n2};
          const float e = n0 * (m0 - rv.x) + n1 * (m1 - rv.y) + n2 * (m2 - rv.z);
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) ds[q++] += J[a] * J[b];
#pragma unroll
          for (int a = 0; a < 6; ++a) ds[21 + a] += J[a] * e;
        }
        if (!parity && !plane && keep) {
          const float m[3] = {qx, qy, qz}, r[3] = {ref[3 * j], ref[3 * j + 1], ref[3 * j + 2]};
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) ds[a * 3 + b] += (double)m[a] * (double)r[b];
#pragma unroll
          for (int q = 0; q < 3; ++q) { ds[9 + q] += (double)m[q]; ds[12 + q] += (double)r[q]; }
        }
        qx = nqx; qy = nqy; qz = nqz; qb = nqb;
      }
      __syncthreads();                                   // nn[] complete
      TSTAMP(2);
    }
    // Phase A2 (iteration 1: the only phase): rows in index order.  Parity mode: waves 1-3 write
    // the 15 scalars of each row into double-buffered LDS tiles, wave 0 adds the previous tile
    // in row order -- getMean (:8-25) and the covariance loop (:731-735) as 15 float32 chains.
    const int TQ = parity ? ICP_BS - 64 : ICP_BS;
    const int slot = parity ? (int)threadIdx.x - 64 : (int)threadIdx.x;
    const int ntiles = (parity || index_pairs) ? (rows + TQ - 1) / TQ : 0;
    float acc = 0.0f;                                    // chain accumulator of lane k < 15 of wave 0
    if (parity && iter > 1) {
      // Two-deep register pipeline of the producers: a tile costs nn/mod loads (coalesced) and then the
      // dependent gather ref[j]; tile t + 2's loads and tile t + 1's gather are in flight while tile t is
      // written, so both round trips overlap the chains instead of adding up per tile.
      int j1 = -1, j2 = -1;
      float m1[3] = {0.f, 0.f, 0.f}, r1[3] = {0.f, 0.f, 0.f}, m2[3] = {0.f, 0.f, 0.f};
      if (slot >= 0) {
        const int i0 = slot, i1 = TQ + slot;
        if (i0 < rows) { j1 = ld_u32(nn, i0); { const F3 v3_ = ld3_u32(mod, i0); m1[0] = v3_.x; m1[1] = v3_.y; m1[2] = v3_.z; } }
        if (i1 < rows) { j2 = ld_u32(nn, i1); { const F3 v3_ = ld3_u32(mod, i1); m2[0] = v3_.x; m2[1] = v3_.y; m2[2] = v3_.z; } }
        const int g = max(j1, 0);
        { const F3 v3_ = ld3_u32(ref, g); r1[0] = v3_.x; r1[1] = v3_.y; r1[2] = v3_.z; }
      }
      for (int t = 0; t < ntiles; ++t) {
        if (slot >= 0) {
          const int i3 = min((t + 2) * TQ + slot, rows - 1);     // clamped: unused past the end
          const bool in3 = (t + 2) * TQ + slot < rows;
          int j3 = ld_u32(nn, i3);
          const F3 m3v = ld3_u32(mod, i3);
          const float m3[3] = {m3v.x, m3v.y, m3v.z};
          j3 = in3 ? j3 : -1;
          const int g = max(j2, 0);
          const F3 r2v = ld3_u32(ref, g);
          const float r2[3] = {r2v.x, r2v.y, r2v.z};
          const bool have = j1 >= 0;                     // dropped pairs contribute an exact +0.0f: (+0) * (+0)
          float (*tile)[ICP_TS] = S.prod[t & 1];
          float mm[3], rr[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) { mm[q] = have ? m1[q] : 0.0f; rr[q] = have ? r1[q] : 0.0f; }
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) tile[a * 3 + b][slot] = mm[a] * rr[b];   // (*it_s) * (*it_ref).t()
#pragma unroll
          for (int q = 0; q < 3; ++q) { tile[9 + q][slot] = mm[q]; tile[12 + q][slot] = rr[q]; }
          j1 = j2;
#pragma unroll
          for (int q = 0; q < 3; ++q) { m1[q] = m2[q]; r1[q] = r2[q]; m2[q] = m3[q]; }
          j2 = j3;
        } else if (t > 0 && threadIdx.x < 15) {
          acc = chain_tile(S.prod[(t - 1) & 1][threadIdx.x], min(TQ, rows - (t - 1) * TQ), acc);
        }
        __syncthreads();
      }
    } else
    for (int t = 0; t < ntiles; ++t) {
      if (slot >= 0) {
        const int i = t * TQ + slot;
        float m[3] = {0.f, 0.f, 0.f}, r[3] = {0.f, 0.f, 0.f};
        bool have_m = false, have_pair = false;
        if (i < rows) {
          if (iter == 1) {                               // :700-704: index pairs, invalid -> 0
            if (i < n_model) {
              have_m = true;
              have_pair = true;
              if (vvalid(mod[3 * i + 2])) { m[0] = mod[3 * i]; m[1] = mod[3 * i + 1]; m[2] = mod[3 * i + 2]; }
            }
            if (vvalid(ref[3 * i + 2])) { r[0] = ref[3 * i]; r[1] = ref[3 * i + 1]; r[2] = ref[3 * i + 2]; }
          } else {
            const int j = nn[i];
            if (j >= 0) {
              have_m = have_pair = true;
              m[0] = mod[3 * i]; m[1] = mod[3 * i + 1]; m[2] = mod[3 * i + 2];
              r[0] = ref[3 * j]; r[1] = ref[3 * j + 1]; r[2] = ref[3 * j + 2];
            }
          }
        }
        if (parity) {
          // dropped pairs contribute an exact +0.0f, so the chains are branch-free
          float (*tile)[ICP_TS] = S.prod[t & 1];
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) tile[a * 3 + b][slot] = have_pair ? m[a] * r[b] : 0.0f;   // (*it_s) * (*it_ref).t()
#pragma unroll
          for (int q = 0; q < 3; ++q) { tile[9 + q][slot] = have_m ? m[q] : 0.0f; tile[12 + q][slot] = r[q]; }
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) ds[a * 3 + b] += (double)m[a] * (double)r[b] * (have_pair ? 1.0 : 0.0);
#pragma unroll
          for (int q = 0; q < 3; ++q) { ds[9 + q] += have_m ? (double)m[q] : 0.0; ds[12 + q] += (double)r[q]; }
        }
      } else if (t > 0 && threadIdx.x < 15) {
        acc = chain_tile(S.prod[(t - 1) & 1][threadIdx.x], min(TQ, rows - (t - 1) * TQ), acc);
      }
      if (parity) __syncthreads();
    }
    if (parity && ntiles > 0 && threadIdx.x < 15)
      acc = chain_tile(S.prod[(ntiles - 1) & 1][threadIdx.x], min(TQ, rows - (ntiles - 1) * TQ), acc);
    kept = block_sum_int(S, kept);
    TSTAMP(3);
    const int ncm = index_pairs ? n_model : kept, ncr = index_pairs ? n_ref : kept;
    if (threadIdx.x == 0) S.n_corr = ncm;
    if (ncr < 3 || ncm < 3) {                            // :711-715
      __syncthreads();
      if (threadIdx.x == 0) S.iter = it_thr;
      __syncthreads();
      continue;
    }
    if (mode == FL_ICP_PARITY) {
      if (threadIdx.x < 15) S.sums[threadIdx.x] = acc;
      __syncthreads();
    } else {
      double dd[NSUM];
#pragma unroll
      for (int k = 0; k < NSUM; ++k) dd[k] = (double)ds[k];
      block_sum_double<NSUM>(S, dd);
    }
    if (plane) {
      if (threadIdx.x == 0) {
        double A[6][6], x[6], tr = 0.0;
        int q = 0;
        for (int a = 0; a < 6; ++a)
          for (int b = a; b < 6; ++b) A[a][b] = A[b][a] = S.dsum[0][q++];
        for (int a = 0; a < 6; ++a) { x[a] = -S.dsum[0][21 + a]; tr += A[a][a]; }
        bool ok = tr > 0.0 && tr < 1.0e300;
        for (int a = 0; a < 6; ++a) A[a][a] += 1.0e-12 * tr;          // keeps a barely constrained direction finite
        for (int c = 0; c < 6 && ok; ++c) {                            // Cholesky A = L L^T (lower triangle in place)
          double d = A[c][c];
          for (int k = 0; k < c; ++k) d -= A[c][k] * A[c][k];
          if (!(d > 1.0e-13 * tr)) { ok = false; break; }              // a direction the pairs do not constrain
          d = sqrt(d);
          A[c][c] = d;
          for (int r = c + 1; r < 6; ++r) {
            double v = A[r][c];
            for (int k = 0; k < c; ++k) v -= A[r][k] * A[c][k];
            A[r][c] = v / d;
          }
        }
        if (ok) {
          for (int r = 0; r < 6; ++r) {                                // L y = b
            double v = x[r];
            for (int k = 0; k < r; ++k) v -= A[r][k] * x[k];
            x[r] = v / A[r][r];
          }
          for (int r = 5; r >= 0; --r) {                               // L^T x = y
            double v = x[r];
            for (int k = r + 1; k < 6; ++k) v -= A[k][r] * x[k];
            x[r] = v / A[r][r];
          }
          // Rodrigues: R = I + (sin t / t) K + ((1 - cos t) / t^2) K^2, K = [omega]x
          const double wx = x[0], wy = x[1], wz = x[2], t2 = wx * wx + wy * wy + wz * wz, t = sqrt(t2);
          const double sa = t > 1.0e-9 ? sin(t) / t : 1.0 - t2 / 6.0, sb = t > 1.0e-9 ? (1.0 - cos(t)) / t2 : 0.5 - t2 / 24.0;
          const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
              double k2 = 0;
              for (int k = 0; k < 3; ++k) k2 += K[i * 3 + k] * K[k * 3 + j];
              S.Ropt[i * 3 + j] = (float)((i == j ? 1.0 : 0.0) + sa * K[i * 3 + j] + sb * k2);
            }
          for (int k = 0; k < 3; ++k) S.Topt[k] = (float)x[3 + k];
        }
        S.ok = ok && finite_all(S.Ropt, 9) && finite_all(S.Topt, 3);
      }
    } else
    if (threadIdx.x == 0) {
      float C[9], mc[3], rc[3];
      if (mode == FL_ICP_PARITY) {
        for (int k = 0; k < 9; ++k) C[k] = S.sums[k];
        for (int k = 0; k < 3; ++k) { mc[k] = S.sums[9 + k] / (float)ncm; rc[k] = S.sums[12 + k] / (float)ncr; }   // getMean :19-24
      } else {
        for (int k = 0; k < 9; ++k) C[k] = (float)S.dsum[0][k];
        for (int k = 0; k < 3; ++k) { mc[k] = (float)(S.dsum[0][9 + k] / ncm); rc[k] = (float)(S.dsum[0][12 + k] / ncr); }
      }
      float u[9], vt[9];
      svd3(C, u, vt);                                    // :742
      for (int i = 0; i < 3; ++i)                        // R_optimal = Mat(vt.t() * u.t()) :744 (gemm: double acc)
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += (double)vt[k * 3 + i] * (double)u[j * 3 + k];
          S.Ropt[i * 3 + j] = (float)s;
        }
      float Rm[3];
      mat_vec(S.Ropt, mc, Rm);
      for (int k = 0; k < 3; ++k) S.Topt[k] = rc[k] - Rm[k];                      // :747
      S.ok = finite_all(S.Ropt, 9) && finite_all(S.Topt, 3);                      // checkRange :748
    }
    __syncthreads();
    TSTAMP(4);
    if (!S.ok) continue;                                 // :749
    float Ro[9], To[3];
    for (int k = 0; k < 9; ++k) Ro[k] = S.Ropt[k];
    for (int k = 0; k < 3; ++k) To[k] = S.Topt[k];
    const float old_mean = S.dist_mean;
    __syncthreads();
    l2dist_phase<MODE>(S, mod, ref, bnd, n_model, 3 * old_mean, Ro, To);           // :756, :778-780
    TSTAMP(5);
    if (threadIdx.x == 0) {
      S.dist_diff = old_mean - S.dist_mean;
      float RT[3];                                       // :793-797
      mat_vec(S.Ropt, S.T, RT);
      for (int k = 0; k < 3; ++k) S.T[k] = RT[k] + S.Topt[k];
      mat_mat(S.Ropt, S.R, S.R);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int i = 0; i < 9; ++i) res->R[i] = S.R[i];
    for (int i = 0; i < 3; ++i) res->T[i] = S.T[i];
    res->dist_mean = S.dist_mean;
    res->px_ratio = S.px;
    res->iters = S.iter;
    res->n_corr_last = S.n_corr;
#ifdef FL_ICP_PHASES
    // dev build only: phase cycles (grid, -, A1, A2, svd, B, before icp_run, whole kernel) of this workgroup instead of R
    for (int i = 0; i < 6; ++i) res->R[i] = (float)S.tacc[i];
    res->R[6] = (float)S.tacc[6];
    res->R[7] = (float)(clock64() - S.tkernel);
#endif
  }
  __syncthreads();
}

// ---- detection() front half: crop back-projection + paired-valid compaction ----------------------
// Unit surface normal at scene pixel (x, y) for FL_ICP_POINT_TO_PLANE (no reference counterpart): least-squares
// depth gradient (zu, zv) over the (2r+1)^2 window -- on a complete window the fit decouples into
// zu = sum du*z / sum du^2 -- then n = Pu x Pv of the back-projected surface P(u,v) = ((u-cx)/fx*z, (v-cy)/fy*z, z).
// A window that leaves the image, holds a missing return or crosses a depth step (> 2 % + 2 mm of the centre
// depth) gives n = 0: the point still takes part in the NN search but not in the 6x6 system.
#define ICP_NRM_R 3
__device__ __forceinline__ void scene_normal(const uint16_t *__restrict__ scene, int w, int h, int x, int y, float fx, float fy,
                                             float cx, float cy, float *n)
{
  n[0] = n[1] = n[2] = 0.f;
  if (x < ICP_NRM_R || y < ICP_NRM_R || x + ICP_NRM_R >= w || y + ICP_NRM_R >= h) return;
  const float zc = (float)scene[(size_t)y * w + x], gate = 0.02f * zc + 2.0f;
  float su = 0.f, sv = 0.f;
  bool ok = zc > 0.f;
  for (int dv = -ICP_NRM_R; dv <= ICP_NRM_R; ++dv)
    for (int du = -ICP_NRM_R; du <= ICP_NRM_R; ++du) {
      const float z = (float)scene[(size_t)(y + dv) * w + x + du];
      ok = ok && z > 0.f && fabsf(z - zc) <= gate;
      su += (float)du * z;
      sv += (float)dv * z;
    }
  if (!ok) return;
  const float s2 = (float)((2 * ICP_NRM_R + 1) * ICP_NRM_R * (ICP_NRM_R + 1) * (2 * ICP_NRM_R + 1) / 3);   // sum du^2 over the window
  const float zu = su / s2, zv = sv / s2, X = (x - cx) / fx, Y = (y - cy) / fy;
  const float pu[3] = {zc / fx + X * zu, Y * zu, zu}, pv[3] = {X * zv, zc / fy + Y * zv, zv};
  const float c[3] = {pu[1] * pv[2] - pu[2] * pv[1], pu[2] * pv[0] - pu[0] * pv[2], pu[0] * pv[1] - pu[1] * pv[0]};
  const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  if (!(len > 0.f)) return;
  n[0] = c[0] / len; n[1] = c[1] / len; n[2] = c[2] / len;
}

__device__ __forceinline__ int crop_clouds(IcpShared &S, const IcpArgs &a, const uint16_t *scene, const uint16_t *model, bool model_01mm,
                           const int *rm, const int *rr, float *ref, float *mod, float *nrm)
{
  const int cw = rm[2], ch = rm[3], np = cw * ch;
  const float inv_fx = 1.0f / a.fx, inv_fy = 1.0f / a.fy;                        // depth_to_3d.cpp:103-104
  const float minv_fx = 1.0f / 608.f, minv_fy = 1.0f / 608.f;                   // initInternalMat common.cpp:358
  const float zs = (float)(1 / 1000.0);
  // Row-major compaction of the paired-valid pixels.  Per 256 pixels: rank inside the wave by ballot + mbcnt, the four
  // wave counts through a double-buffered LDS slot -- ONE barrier per step (the block-wide scan it replaces took
  // five, and with 5 workgroups per CU the barriers were the cost); every thread keeps the running total itself.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const float inv_cw = 1.0f / (float)cw;                   // p / cw below: exact for p < 2^20, cw <= 2^10 (see div_small)
  int kept_before = 0, step = 0;
  __syncthreads();
  for (int base = 0; base < np; base += blockDim.x, ++step) {
    const int p = base + threadIdx.x;
    float A[3] = {0, 0, 0}, B[3] = {0, 0, 0};
    int keep = 0;
    if (p < np) {
      const int y = np < (1 << 20) && cw <= 1024 ? (int)(((float)p + 0.5f) * inv_cw) : p / cw, x = p - y * cw;
      const int sx = rr[0] + x, sy = rr[1] + y, mx = rm[0] + x, my = rm[1] + y;
      const unsigned ds = scene[(size_t)sy * a.w + sx];
      unsigned dm = model[(size_t)my * a.w + mx];
      if (model_01mm) {                                  // convertTo(CV_16UC1, 0.1) obj_reco_lmicp.cpp:188
        int v = __float2int_rn((float)dm * 0.1f);
        dm = (unsigned)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
      }
      const float zsf = ds == 0 ? NAN : (float)ds * zs;                           // rescaleDepth :257-259
      const float zmf = dm == 0 ? NAN : (float)dm * zs;
      A[0] = ((((float)sx - a.cx) * inv_fx) * zsf) * 1000;                        // :119,:132; scale_mat_vec3f
      A[1] = ((((float)sy - a.cy) * inv_fy) * zsf) * 1000;
      A[2] = zsf * 1000;
      B[0] = ((((float)mx - 320.f) * minv_fx) * zmf) * 1000;
      B[1] = ((((float)my - 240.f) * minv_fy) * zmf) * 1000;
      B[2] = zmf * 1000;
      keep = vvalid(A[2]) && vvalid(B[2]);                                        // matToVec common.cpp:382-405
    }
    const unsigned long long bal = __ballot(keep);
    const int in_wave = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    int *slot = S.iscan2[step & 1];
    if (lane == 0) slot[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < nwv; ++i) { const int c = slot[i]; before += i < wv ? c : 0; total += c; }
    if (keep) {
      const int k = kept_before + before + in_wave;
      ref[3 * k] = A[0]; ref[3 * k + 1] = A[1]; ref[3 * k + 2] = A[2];
      mod[3 * k] = B[0]; mod[3 * k + 1] = B[1]; mod[3 * k + 2] = B[2];
      if (nrm) {
        float nv[3];
        scene_normal(scene, a.w, a.h, rr[0] + p % cw, rr[1] + p / cw, a.fx, a.fy, a.cx, a.cy, nv);
        nrm[3 * k] = nv[0]; nrm[3 * k + 1] = nv[1]; nrm[3 * k + 2] = nv[2];
      }
    }
    kept_before += total;
  }
  __syncthreads();                                         // the clouds are complete for every thread
  return kept_before;
}

// icpCloudToCloud_Ex on clouds the host staged in the workspace (fl_icp)
template <int MODE>
__global__ __launch_bounds__(ICP_MAX_THREADS) void k_icp_clouds(IcpArgs a)
{
  extern __shared__ __align__(16) uint8_t icp_smem[];
  IcpShared &S = *(IcpShared *)icp_smem;
  const IcpWsLayout L = icp_layout(a.n_max);
  uint8_t *wsb = a.ws + (size_t)blockIdx.x * a.ws_stride;
  icp_run<MODE>(S, wsb, L, a.job.n_ref, a.job.n_model, a.it_thr, a.dmt, a.ddt, &a.results[blockIdx.x].det.icp);
}

template <int MODE>
__global__ __launch_bounds__(ICP_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(ICP_MODE_WPE(MODE), ICP_MODE_WPE(MODE) > 4 ? ICP_MODE_WPE(MODE) : (MODE == FL_ICP_PARITY ? FL_ICP_WPE : 4)))) void k_icp_pipeline(IcpArgs a)
{
  extern __shared__ __align__(16) uint8_t icp_smem[];
  IcpShared &S = *(IcpShared *)icp_smem;
#ifdef FL_ICP_PHASES
  if (threadIdx.x == 0) S.tkernel = clock64();
#endif
  const int job = blockIdx.x, frame = a.job.kind == 0 ? job / a.ranks : job, rank = a.job.kind == 0 ? job % a.ranks : 0;
  const IcpWsLayout L = icp_layout(a.n_max);
  uint8_t *wsb = a.ws + (size_t)job * a.ws_stride;
  float *ref = (float *)(wsb + L.ref), *mod = (float *)(wsb + L.mod);


  fl_recognition_result *res = &a.results[job];
  const uint16_t *scene, *model;
  float r_match[9], t_match[3];
  bool model_01mm;
  if (a.job.kind == 1) {
    if (threadIdx.x == 0) {
      for (int k = 0; k < 4; ++k) { S.rect_m[k] = a.job.rect_model[k]; S.rect_r[k] = a.job.rect_ref[k]; }
      S.status = FL_OK;
      res->status = FL_OK;
      res->found = 1;
    }
    scene = a.job.scene_depth;
    model = a.job.model_depth;
    for (int k = 0; k < 9; ++k) r_match[k] = a.job.r_match[k];
    for (int k = 0; k < 3; ++k) t_match[k] = a.job.t_match[k];
    model_01mm = false;
    __syncthreads();
  } else {
    // CObjRecoLmICP::Recognition after Detector::match (obj_reco_lmicp.cpp:106-152)
    const uint8_t *fws = a.frame_ws + (size_t)frame * a.frame_stride;
    const int *counters = (const int *)(fws + a.off_count);
    const fl_match *matches = (const fl_match *)(fws + a.off_match);
    if (threadIdx.x == 0) {
      S.status = FL_OK;
      S.g = -1;
      res->n_matches = counters[1];
      res->found = 0;
      res->status = FL_OK;
      if (counters[2]) S.status = res->status = FL_ERR_OVERFLOW;
      else if (counters[1] > rank) {
        const fl_match best = matches[rank];             // matches[0] :111 (rank > 0: multi-hypothesis extension)
        res->best = best;
        const int g = a.class_first[best.class_idx] + best.template_id;
        const FlPyrInfo pi = a.pyr[g];
        S.g = g;
        S.rect_m[0] = pi.off_x0; S.rect_m[1] = pi.off_y0; S.rect_m[2] = pi.width0; S.rect_m[3] = pi.height0;   // :129
        S.rect_r[0] = best.x; S.rect_r[1] = best.y; S.rect_r[2] = pi.width0; S.rect_r[3] = pi.height0;         // :130-132
        if (!a.depth_ptrs[g]) S.status = res->status = FL_ERR_STATE;      // no depth/<id>.png uploaded
      }
    }
    __syncthreads();
    if (S.g < 0 || S.status != FL_OK) return;            // vtResult stays empty (:106-109)
    const int g = S.g;
    scene = (const uint16_t *)((const uint8_t *)a.scene_base + (size_t)frame * a.scene_stride);
    model = a.depth_ptrs[g];
    const float *p = a.poses + 13 * (size_t)g;           // :141-152
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) r_match[i * 3 + j] = p[i * 4 + j];
      t_match[i] = p[i * 4 + 3];
    }
    model_01mm = true;
  }
  // Q10: a rect leaving the image is a cv::Mat ROI assertion in the reference (detection.cpp:43-44)
  {
    const int *rm = S.rect_m, *rr = S.rect_r;
    const bool bad = rm[0] < 0 || rm[1] < 0 || rm[2] < 0 || rm[3] < 0 || rm[0] + rm[2] > a.w || rm[1] + rm[3] > a.h ||
                     rr[0] < 0 || rr[1] < 0 || rr[2] < 0 || rr[3] < 0 || rr[0] + rr[2] > a.w || rr[1] + rr[3] > a.h ||
                     rm[2] != rr[2] || rm[3] != rr[3] || (long long)rm[2] * rm[3] > a.n_max;
    if (bad) {
      if (threadIdx.x == 0) { res->status = FL_ERR_ASSERT; res->det.status = FL_ERR_ASSERT; res->found = 0; }
      return;
    }
  }
  const int np = crop_clouds(S, a, scene, model, model_01mm, S.rect_m, S.rect_r, ref, mod,
                             MODE == FL_ICP_POINT_TO_PLANE ? (float *)(wsb + L.nrm) : nullptr);
  // getMean x2 (detection.cpp:165-166), t_match_tmp = r - m (:177), t_init (:199)
  float mc[3] = {0, 0, 0}, rc[3] = {0, 0, 0};
  if (MODE == FL_ICP_PARITY) {
    // getMean x2 as six float32 chains in index order, fed like phase A2: waves 1..3 load 192 rows per tile
    // (coalesced 12-byte loads) into the LDS tiles, lanes 0..5 of wave 0 add the previous tile
    {
      const int slot = (int)threadIdx.x - 64, ntiles = (np + ICP_TQ - 1) / ICP_TQ;
      float acc = 0.0f;
      for (int t = 0; t < ntiles; ++t) {
        if (slot >= 0) {
          const int i = t * ICP_TQ + slot;
          F3 m3 = {0.f, 0.f, 0.f}, r3 = {0.f, 0.f, 0.f};
          if (i < np) { m3 = ld3_u32(mod, i); r3 = ld3_u32(ref, i); }
          float (*tile)[ICP_TS] = S.prod[t & 1];
          tile[0][slot] = m3.x; tile[1][slot] = m3.y; tile[2][slot] = m3.z;
          tile[3][slot] = r3.x; tile[4][slot] = r3.y; tile[5][slot] = r3.z;
        } else if (t > 0 && threadIdx.x < 6) {
          acc = chain_tile(S.prod[(t - 1) & 1][threadIdx.x], min(ICP_TQ, np - (t - 1) * ICP_TQ), acc);
        }
        __syncthreads();
      }
      if (ntiles > 0 && threadIdx.x < 6)
        acc = chain_tile(S.prod[(ntiles - 1) & 1][threadIdx.x], min(ICP_TQ, np - (ntiles - 1) * ICP_TQ), acc);
      if (threadIdx.x < 6) S.sums[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) { mc[k] = S.sums[k]; rc[k] = S.sums[3 + k]; }
    if (np > 0)
      for (int k = 0; k < 3; ++k) { mc[k] /= (float)np; rc[k] /= (float)np; }
  } else {
    double ds[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < np; i += blockDim.x)
      for (int k = 0; k < 3; ++k) { ds[k] += mod[3 * i + k]; ds[3 + k] += ref[3 * i + k]; }
    block_sum_double<6>(S, ds);
    if (np > 0)
      for (int k = 0; k < 3; ++k) { mc[k] = (float)(S.dsum[0][k] / np); rc[k] = (float)(S.dsum[0][3 + k] / np); }
  }
  float t_tmp[3], t_init[3];
  for (int k = 0; k < 3; ++k) { t_tmp[k] = rc[k] - mc[k]; t_init[k] = t_tmp[k] + t_match[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < np; i += blockDim.x) {   // transformPoints(pts_mod, I, t_match_tmp) :206
    if (!vvalid(mod[3 * i + 2])) continue;
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float v[3] = {mod[3 * i], mod[3 * i + 1], mod[3 * i + 2]}, o[3];
    mat_vec(I, v, o);
    mod[3 * i] = o[0] + t_tmp[0];
    mod[3 * i + 1] = o[1] + t_tmp[1];
    mod[3 * i + 2] = o[2] + t_tmp[2];
  }
  __syncthreads();
  icp_run<MODE>(S, wsb, L, np, np, a.it_thr, a.dmt, a.ddt, &res->det.icp);      // :228
  if (threadIdx.x == 0) {
    const fl_icp_result &ic = res->det.icp;
    float Rt[3];
    mat_vec(ic.R, t_init, Rt);                           // T_final = R*t_init + T, R_final = R*r_match :232-234
    for (int k = 0; k < 3; ++k) res->det.T_final[k] = Rt[k] + ic.T[k];
    mat_mat(ic.R, r_match, res->det.R_final);
    res->det.n_points = np;
    res->det.status = FL_OK;
    for (int i = 0; i < 3; ++i) {                        // Convert() obj_reco_lmicp.cpp:20-30
      for (int j = 0; j < 3; ++j) res->pose[i * 4 + j] = res->det.R_final[i * 3 + j];
      res->pose[i * 4 + 3] = res->det.T_final[i];
    }
    res->pose[12] = res->pose[13] = res->pose[14] = 0.f;
    res->pose[15] = 1.f;
    res->found = 1;
  }
}

// ------------------------------------------------------------------------------------------
// cup_d2pc::depthTo3d full frame (the stage entry point; the pipeline only back-projects crops)
__global__ __launch_bounds__(256) void k_depth_to_3d(const uint16_t *__restrict__ depth, int w, int h, float fx, float fy,
                                                     float cx, float cy, float *__restrict__ out)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
  const unsigned d = depth[(size_t)y * w + x];
  const float z = d == 0 ? NAN : (float)d * (float)(1 / 1000.0);
  float *p = out + ((size_t)y * w + x) * 3;
  p[0] = (((float)x - cx) * inv_fx) * z;
  p[1] = (((float)y - cy) * inv_fy) * z;
  p[2] = z;
}

extern "C" int fl_depth_to_3d(fl_context *ctx, const uint16_t *depth, int w, int h, double fx, double fy, double cx,
                              double cy, float *out, int mem)
{
  if (!ctx || !depth || !out || w <= 0 || h <= 0) return FL_ERR_INVALID;
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nin = (size_t)w * h * 2, nout = (size_t)w * h * 12;
  const uint16_t *din = depth;
  float *dout = out;
  if (mem == FL_MEM_HOST) {
    void *s = nullptr;
    int rc = fl_scratch(ctx, fl_align(nin, 256) + nout, &s);
    if (rc) return rc;
    FL_HIP(ctx, hipMemcpyAsync(s, depth, nin, hipMemcpyHostToDevice, ctx->stream));
    din = (const uint16_t *)s;
    dout = (float *)((uint8_t *)s + fl_align(nin, 256));
  }
  dim3 grid((w + 63) / 64, (h + 3) / 4);
  hipLaunchKernelGGL(k_depth_to_3d, grid, dim3(256), 0, ctx->stream, din, w, h, (float)fx, (float)fy, (float)cx, (float)cy,
                     dout);
  FL_HIP(ctx, hipGetLastError());
  if (mem == FL_MEM_HOST) {
    FL_HIP(ctx, hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, ctx->stream));
    FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return FL_OK;
}

static int icp_threads(int) { return ICP_BS; }
static size_t icp_lds_bytes() { return (sizeof(IcpShared) + 15) & ~(size_t)15; }
static_assert(sizeof(IcpShared) + 16 <= 160 * 1024 / FL_ICP_WPE, "IcpShared must leave room for FL_ICP_WPE workgroups per CU");
template <typename K>
static int icp_launch_one(fl_context *ctx, K kern, int n_frames, const IcpArgs &a)
{
  const size_t lds = icp_lds_bytes();
  FL_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(n_frames), dim3(icp_threads(n_frames)), lds, ctx->stream, a);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}
static int icp_launch(fl_context *ctx, int n_frames, const IcpArgs &a)
{
  if (a.mode != FL_ICP_PARITY && a.mode != FL_ICP_FAST && a.mode != FL_ICP_POINT_TO_PLANE)
    return fl_set_error(ctx, FL_ERR_INVALID, "unknown fl_icp_mode");
  if (a.job.kind == 2) {
    if (a.mode == FL_ICP_POINT_TO_PLANE) return icp_launch_one(ctx, k_icp_clouds<FL_ICP_POINT_TO_PLANE>, n_frames, a);
    return a.mode == FL_ICP_FAST ? icp_launch_one(ctx, k_icp_clouds<FL_ICP_FAST>, n_frames, a)
                                 : icp_launch_one(ctx, k_icp_clouds<FL_ICP_PARITY>, n_frames, a);
  }
  if (a.mode == FL_ICP_POINT_TO_PLANE) return icp_launch_one(ctx, k_icp_pipeline<FL_ICP_POINT_TO_PLANE>, n_frames, a);
  return a.mode == FL_ICP_FAST ? icp_launch_one(ctx, k_icp_pipeline<FL_ICP_FAST>, n_frames, a)
                               : icp_launch_one(ctx, k_icp_pipeline<FL_ICP_PARITY>, n_frames, a);
}

static int icp_clouds(fl_context *ctx, const float *ref, const float *ref_normals, int n_ref, const float *model, int n_model,
                      int icp_it_thr, float dist_mean_thr, float dist_diff_thr, int icp_mode, int mem, fl_icp_result *res)
{
  if (!ctx || !res || n_ref < 0 || n_model < 0 || (n_ref && !ref) || (n_model && !model)) return FL_ERR_INVALID;
  if (icp_mode == FL_ICP_POINT_TO_PLANE && n_ref && !ref_normals)
    return fl_set_error(ctx, FL_ERR_INVALID, "FL_ICP_POINT_TO_PLANE needs reference normals: use fl_icp_point_to_plane or fl_detection");
  FL_HIP(ctx, hipSetDevice(ctx->device));
  const int n_max = n_ref > n_model ? n_ref : n_model;
  const IcpWsLayout L = icp_layout(n_max);
  void *s = nullptr;
  int rc = fl_scratch(ctx, L.total + 4096, &s);
  if (rc) return rc;
  uint8_t *wsb = (uint8_t *)s;
  fl_recognition_result *dres = (fl_recognition_result *)(wsb + L.total);
  const hipMemcpyKind kind = mem == FL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (n_ref) FL_HIP(ctx, hipMemcpyAsync(wsb + L.ref, ref, 12 * (size_t)n_ref, kind, ctx->stream));
  if (n_model) FL_HIP(ctx, hipMemcpyAsync(wsb + L.mod, model, 12 * (size_t)n_model, kind, ctx->stream));
  if (n_ref && ref_normals) FL_HIP(ctx, hipMemcpyAsync(wsb + L.nrm, ref_normals, 12 * (size_t)n_ref, kind, ctx->stream));
  IcpArgs a;
  memset(&a, 0, sizeof(a));
  a.ws = wsb;
  a.ws_stride = 0;
  a.n_max = n_max;
  a.it_thr = icp_it_thr;
  a.dmt = dist_mean_thr;
  a.ddt = dist_diff_thr;
  a.mode = icp_mode;
  a.job.kind = 2;
  a.job.n_ref = n_ref;
  a.job.n_model = n_model;
  a.results = dres;
  rc = icp_launch(ctx, 1, a);
  if (rc) return rc;
  fl_recognition_result *h = nullptr;
  rc = fl_pinned(ctx, sizeof(*h), (void **)&h);
  if (rc) return rc;
  FL_HIP(ctx, hipMemcpyAsync(h, dres, sizeof(*h), hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *res = h->det.icp;
  return FL_OK;
}

extern "C" int fl_icp(fl_context *ctx, const float *ref, int n_ref, const float *model, int n_model, int icp_it_thr,
                      float dist_mean_thr, float dist_diff_thr, int icp_mode, int mem, fl_icp_result *res)
{
  return icp_clouds(ctx, ref, nullptr, n_ref, model, n_model, icp_it_thr, dist_mean_thr, dist_diff_thr, icp_mode, mem, res);
}

extern "C" int fl_icp_point_to_plane(fl_context *ctx, const float *ref, const float *ref_normals, int n_ref, const float *model,
                                     int n_model, int icp_it_thr, float dist_mean_thr, float dist_diff_thr, int mem,
                                     fl_icp_result *res)
{
  return icp_clouds(ctx, ref, ref_normals, n_ref, model, n_model, icp_it_thr, dist_mean_thr, dist_diff_thr,
                    FL_ICP_POINT_TO_PLANE, mem, res);
}

extern "C" int fl_detection(fl_context *ctx, const uint16_t *model_depth, const uint16_t *scene_depth, int w, int h,
                            const fl_intrinsics *K, const int rect_model[4], const int rect_ref[4], int icp_it_thr,
                            float dist_mean_thr, float dist_diff_thr, const float r_match[9], const float t_match[3],
                            int icp_mode, int mem, fl_detection_result *res)
{
  if (!ctx || !model_depth || !scene_depth || !K || !rect_model || !rect_ref || !r_match || !t_match || !res || w <= 0 ||
      h <= 0)
    return FL_ERR_INVALID;
  FL_HIP(ctx, hipSetDevice(ctx->device));
  long long area = (long long)rect_model[2] * rect_model[3];
  if (area < 0 || area > (long long)w * h) area = 0;     // the kernel reports FL_ERR_ASSERT for bad rects
  const int n_max = (int)(area > 0 ? area : 1);
  const IcpWsLayout L = icp_layout(n_max);
  const size_t img = fl_align((size_t)w * h * 2, 256);
  void *s = nullptr;
  int rc = fl_scratch(ctx, L.total + 4096 + 2 * img, &s);
  if (rc) return rc;
  uint8_t *wsb = (uint8_t *)s;
  fl_recognition_result *dres = (fl_recognition_result *)(wsb + L.total);
  const uint16_t *dm = model_depth, *dsn = scene_depth;
  if (mem == FL_MEM_HOST) {
    uint8_t *b = wsb + L.total + 4096;
    FL_HIP(ctx, hipMemcpyAsync(b, model_depth, (size_t)w * h * 2, hipMemcpyHostToDevice, ctx->stream));
    FL_HIP(ctx, hipMemcpyAsync(b + img, scene_depth, (size_t)w * h * 2, hipMemcpyHostToDevice, ctx->stream));
    dm = (const uint16_t *)b;
    dsn = (const uint16_t *)(b + img);
  }
  FL_HIP(ctx, hipMemsetAsync(dres, 0, sizeof(*dres), ctx->stream));
  IcpArgs a;
  memset(&a, 0, sizeof(a));
  a.ws = wsb;
  a.n_max = n_max;
  a.w = w;
  a.h = h;
  a.fx = (float)K->fx;
  a.fy = (float)K->fy;
  a.cx = (float)K->cx;
  a.cy = (float)K->cy;
  a.it_thr = icp_it_thr;
  a.dmt = dist_mean_thr;
  a.ddt = dist_diff_thr;
  a.mode = icp_mode;
  a.job.kind = 1;
  for (int k = 0; k < 4; ++k) { a.job.rect_model[k] = rect_model[k]; a.job.rect_ref[k] = rect_ref[k]; }
  for (int k = 0; k < 9; ++k) a.job.r_match[k] = r_match[k];
  for (int k = 0; k < 3; ++k) a.job.t_match[k] = t_match[k];
  a.job.model_depth = dm;
  a.job.scene_depth = dsn;
  a.results = dres;
  rc = icp_launch(ctx, 1, a);
  if (rc) return rc;
  fl_recognition_result *hres = nullptr;
  rc = fl_pinned(ctx, sizeof(*hres), (void **)&hres);
  if (rc) return rc;
  FL_HIP(ctx, hipMemcpyAsync(hres, dres, sizeof(*hres), hipMemcpyDeviceToHost, ctx->stream));
  FL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *res = hres->det;
  if (hres->status == FL_ERR_ASSERT) {
    res->status = FL_ERR_ASSERT;
    return fl_set_error(ctx, FL_ERR_ASSERT, "crop rectangle leaves the image (cv::Mat ROI assert, detection.cpp:43-44)");
  }
  return FL_OK;
}

// batch: one workgroup per frame of the detector workspace
// Multi-hypothesis refinement: the first `k` matches of each of n_frames frames are refined by
// n_frames * k workgroups of the same kernel, each with an ICP workspace of its own in `ws`.
int fl_launch_detection_topk(fl_detector *det, int n_frames, int k, const fl_intrinsics *K, const fl_recognition_params *p,
                             const uint16_t *depth, size_t depth_stride, uint8_t *ws, fl_recognition_result *d_results)
{
  fl_context *ctx = det->ctx;
  IcpArgs a;
  memset(&a, 0, sizeof(a));
  a.ws = ws;
  a.ws_stride = fl_icp_ws_bytes(det->n_pts_max);
  a.n_max = det->n_pts_max;
  a.w = det->w0;
  a.h = det->h0;
  a.fx = (float)K->fx;
  a.fy = (float)K->fy;
  a.cx = (float)K->cx;
  a.cy = (float)K->cy;
  a.it_thr = p->icp_it_thr;
  a.dmt = p->dist_mean_thr;
  a.ddt = p->dist_diff_thr;
  a.mode = p->icp_mode;
  a.job.kind = 0;
  a.frame_ws = det->d_ws;
  a.frame_stride = det->ws_stride;
  a.ranks = k;
  a.scene_base = depth;
  a.scene_stride = depth_stride;
  a.off_count = det->off_count;
  a.off_match = det->off_match;
  a.pyr = det->d_pyr;
  a.class_first = det->d_class_first;
  a.poses = det->d_poses;
  a.depth_ptrs = det->d_depth_ptrs;
  a.results = d_results;
  return icp_launch(ctx, n_frames * k, a);
}

int fl_launch_detection_batch(fl_detector *det, int n_frames, const fl_intrinsics *K, const fl_recognition_params *p,
                              const uint16_t *depth, size_t depth_stride)
{
  fl_context *ctx = det->ctx;
  IcpArgs a;
  memset(&a, 0, sizeof(a));
  a.ws = det->d_ws + det->off_icp;
  a.ws_stride = det->ws_stride;
  a.n_max = det->n_pts_max;
  a.w = det->w0;
  a.h = det->h0;
  a.fx = (float)K->fx;
  a.fy = (float)K->fy;
  a.cx = (float)K->cx;
  a.cy = (float)K->cy;
  a.it_thr = p->icp_it_thr;
  a.dmt = p->dist_mean_thr;
  a.ddt = p->dist_diff_thr;
  a.mode = p->icp_mode;
  a.job.kind = 0;
  a.frame_ws = det->d_ws;
  a.frame_stride = det->ws_stride;
  a.ranks = 1;
  a.scene_base = depth;
  a.scene_stride = depth_stride;
  a.off_count = det->off_count;
  a.off_match = det->off_match;
  a.pyr = det->d_pyr;
  a.class_first = det->d_class_first;
  a.poses = det->d_poses;
  a.depth_ptrs = det->d_depth_ptrs;
  a.results = det->d_results;
  return icp_launch(ctx, n_frames, a);
}
