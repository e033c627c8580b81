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
// The workgroup is 256 threads at full batches (built for 4 and for 5 workgroups per CU: icp_small_wpe picks) and 1024 threads
// when the batch would leave CUs idle anyway (up to two frames per CU; a camera-rate caller: 1-8 frames): the search and the
// chain producers get 4x the lanes, the next iteration's search runs underneath the dist_mean chain ("Search ahead of the
// distance chain" in icp_run), and a frame's latency drops to about what its two sequential chains cost.
//
// Nearest neighbours: the reference's FLANN kd-tree (exact 1-NN, eps 0) is replaced by a uniform
// x/y cell grid over the static reference cloud built once per frame; a query only visits the
// cells within sqrt(3*dist_mean) because farther neighbours are discarded anyway
// (PointsCorresponding keeps d^2 <= 3*dist_mean, ICP.cpp:268,708).  Distances use L2_Simple's
// float expression ((dx*dx + dy*dy) + dz*dz); ties go to the lowest index.
//
// Organised search (the recognition / detection() pipeline, where both clouds are back-projected crops).  The grid, its
// CSR headers and their gathers are not needed there: the reference cloud is kept as an IMAGE (crop pixel -> 12-byte point,
// a point at infinity where the pixel was dropped; a point is named by its pixel), and the reference points within distance r of a query
// q can only come from the pixels its ball projects to -- u in [fx (qx -+ r) / (qz +- r)], likewise v -- a window of a
// few pixels.  Queries are taken in 16x4-pixel tile order (a permutation built once per frame), so the 64 queries of a
// wave share a compact union window; the wave stages that window into its share of the (idle) chain tiles in LDS with a
// handful of coalesced loads and every lane enumerates its own window from LDS.  When the union does not fit (the first
// iterations, where sqrt(3 dist_mean) is several pixels) the same enumeration reads the image from L2 instead.
// fl_icp() (caller-supplied clouds, no image structure) keeps the grid search.
// Built with -ffp-contract=off: one IEEE binary32/64 operation per operator.
#include "fl_internal.h"
#include <float.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include <mutex>
#include <set>
#include <utility>

#ifndef FL_ICP_BS_SMALL
#define FL_ICP_BS_SMALL 256
#endif
#define ICP_BS_SMALL FL_ICP_BS_SMALL
#define ICP_BS_WIDE 1024
#define ICP_DT 512                 // terms per block of the deferred dist_mean chain (two float4 per lane of the chain wave)
#ifdef FL_ICP_PHASES
#define TSTAMP(k) do { if (threadIdx.x == 0) { long long now_ = clock64(); S.tacc[k] += now_ - S.tlast; S.tlast = now_; } } while (0)
// chain phases: what the chain wave spends adding and at the tile barrier (S.hist[k], [k + 1]), and what the first producer
// wave spends at the tile barriers (S.hist[k]; one barrier in `every` is timed); units of 16 cycles
#define CH_STAMP_BEGIN long long c_add = 0, c_bar = 0, c_last = clock64()
#define CH_STAMP(v) { const long long now_ = clock64(); v += now_ - c_last; c_last = now_; }
#define CH_STAMP_END(k) { if (threadIdx.x == 0) { S.hist[k] += (unsigned)(c_add >> 4); S.hist[(k) + 1] += (unsigned)(c_bar >> 4); } }
#define PR_STAMP_BARRIER(k, every) { const long long b0_ = clock64(); tile_barrier(); if (threadIdx.x == 64) S.hist[k] += (unsigned)(((clock64() - b0_) * (every)) >> 4); }
#else
#define TSTAMP(k) do { } while (0)
#define CH_STAMP_BEGIN
#define CH_STAMP(v) { }
#define CH_STAMP_END(k) { }
#define PR_STAMP_BARRIER(k, every) tile_barrier()
#endif
#ifndef FL_ICP_WPE
#define FL_ICP_WPE 4               // waves per SIMD the default 256-thread recognition kernel is compiled for (4 -> 128 VGPRs, 5 -> 96; a
                                  // second instance for 5 is always built, see k_icp_pipeline).  ICP us per frame, one box: mid-round
                                  // 5 @ 1280 frames 11.65, 4 @ 2048 10.4 (the 96-VGPR build spilled in the search loop); after the scan
                                  // batches lost 10 instructions and a few registers: 5 @ 2560 9.76, 4 @ 2048 10.15
#endif
#ifndef FL_ICP_FAST_F32
#define FL_ICP_FAST_F32 1          // FL_ICP_FAST keeps its per-thread partial sums (~60 terms) in float32, like the point-to-plane mode;
                                  // the cross-thread tree is fp64.  0: fp64 partials
#endif
#ifndef FL_ICP_FAST_WPE
#define FL_ICP_FAST_WPE 4
#endif
#ifndef FL_ICP_PLANE_WPE
#define FL_ICP_PLANE_WPE 4
#endif
// waves per SIMD kernel k_icp_pipeline<MODE, 256> is compiled for
#define ICP_MODE_WPE(MODE) ((MODE) == FL_ICP_PARITY ? FL_ICP_WPE : ((MODE) == FL_ICP_POINT_TO_PLANE ? FL_ICP_PLANE_WPE : FL_ICP_FAST_WPE))
#ifndef FL_ICP_NBQ
#define FL_ICP_NBQ 4              // organised search: candidate positions fetched per batch (4 VGPRs each)
#endif
#ifndef FL_ICP_NBUF_SMALL
#define FL_ICP_NBUF_SMALL 4       // 16-row register batches of a chain in the 256-thread kernel.  Round 3: 2 / 3 / 4 measured the same; since
                                  // the dist_mean chain runs over blocks of four tiles (FL_ICP_BMACRO) 2 / 3 / 4 / 6 give 29.5 / 29.2 / 29.1 /
                                  // 38.1 ms per 4096 frames (6: the batches spill)
#endif
#ifndef FL_ICP_BSUM
#define FL_ICP_BSUM 1             // parity mode: the dist_mean chain (non-negative terms) adds whole blocks exactly (chain_block_nonneg) and falls
                                  // back to the term-by-term chain only where a block holds a rounding tie or crosses a binade.  1: in the
                                  // 1024-thread kernel (a frame alone on its CU waits for that chain: ICP 3.07 -> 2.71 ms per 8 frames);
                                  // 2: in the 256-thread kernel too (measured at 4096 frames: 34.1 against 33.5 ms -- with four workgroups
                                  // per CU nobody waits for the chain, and a phase B that runs as fast as memory lets it only takes the
                                  // memory side from the co-resident workgroups' phases); 0: always term by term
#endif
#ifndef FL_ICP_ZIMG
#define FL_ICP_ZIMG 1             // organised search, parity mode: the staged rectangles and the partner gather read a 4-byte image (the
                                  // pixel's depth factor, NaN where the pixel was dropped) and rebuild the 12-byte point by crop_clouds'
                                  // own expression -- bit for bit the point the 12-byte image holds -- instead of reading it
#endif
#ifndef FL_ICP_TILE_W
#define FL_ICP_TILE_W 16          // organised search: the queries of a step come from FL_ICP_TILE_W x (64 / FL_ICP_TILE_W)-pixel tiles.  Wider tiles =
                                  // longer contiguous runs in the tile-ordered gathers / scatters (mod, bnd, nn: 192 instead of 96 bytes per row piece)
                                  // against a larger staged rectangle; measured at 4096 frames, ICP ms per launch: 4 x 16 35.4, 8 x 8 33.9 / 33.8,
                                  // 16 x 4 33.6 / 33.6, 32 x 2 35.8
#endif
#ifndef FL_ICP_BMACRO
#define FL_ICP_BMACRO 4           // dist_mean phase, 256-thread parity kernel: tiles per barrier (l2dist_phase)
#endif
#ifndef FL_ICP_TILE_COLMAJOR
#define FL_ICP_TILE_COLMAJOR 1    // the order of the pixels inside a search tile (build_tile_order)
#endif
#ifndef FL_ICP_BPD
#define FL_ICP_BPD 1              // dist_mean phase: tiles of (mod, ref, bnd) loads a producer thread keeps in flight (1 or 2)
#endif
#ifndef FL_ICP_ALLPROD
#define FL_ICP_ALLPROD 1          // with the block sums in the 256-thread kernel: the chain wave produces rows too (tiles of 256 rows)
#endif
#ifndef FL_ICP_CHAIN_SIMD
#define FL_ICP_CHAIN_SIMD 0       // 256-thread kernel: elect the chain wave so that the chain waves of a CU's workgroups sit on different SIMDs
                                  // (chain_elect).  Measured at 4096 frames, one box: 33.49 against 33.50 ms per launch -- SIMD issue
                                  // slots are not what the launch waits for (profiles/README.md, round 4) -- so wave 0 chains; 1 builds it in
#endif
#ifndef FL_ICP_SPEC
#define FL_ICP_SPEC 1             // parity mode, organised search: the next iteration's search runs while the chain wave adds dist_mean
                                  // (1: in the 1024-thread kernel, 2: in both, 0: off)
#endif
#define ICP_STAGE_CAP 384         // points a wave stages per search step: six passes of 64
#ifndef FL_ICP_NB
#define FL_ICP_NB 10              // candidates fetched per round trip of the NN search (measured: 8..20)
#endif

// HBM layout of one frame's ICP workspace (n = capacity in points):
//   ref   n x 3 f32   reference cloud, index order (pairing + iteration 1)
//   mod   n x 3 f32   model cloud, transformed in place every iteration
//   sref  (n+4) x 16 bytes.  grid search: reference cloud sorted by grid cell as float4 (index bits, X, Y, Z);
//                     organised search: the reference IMAGE, crop pixel p -> X, Y, Z (3 x f32; +inf where the pixel was
//                     dropped by the paired compaction), then the image of the points' indices (i32, NN_IDX_NONE where dropped)
//   zimg   (n+4) x f32  organised search: crop pixel p -> rescaleDepth's z factor of the scene pixel (depth * (1 / 1000.0)), NaN where
//                     the paired compaction dropped the pixel: the point is (((sx - cx) / fx... crop_clouds' expression) of it
//   nn     n x i32    nearest reference point j of model point i (kept pair: j, dropped: -1): its index (grid search) or its
//                     crop pixel (organised search)
//   bnd    n x 16 bit upper bound on the distance from model point i to its nearest reference point: the top half of the
//                     float32 pattern, rounded up (bnd_ld / bnd_st)
//   nd     n x f32    organised search, parity mode: squared distance to nn[i] (the search runs ahead of the threshold it is
//                     compared with, see "Search ahead of the distance chain")
//   dterm  n x f32    parity mode: the terms of getL2distClouds' dist_mean chain, index order (0 for a dropped pair)
//   perm   n x i32    organised search: model indices in 16x4-pixel tile order, column by column inside a tile (build_tile_order)
//   cell_start / cell_cur   CSR offsets of the x/y cell grid (grid search)
//   nrm   n x 3 f32   unit normals of the reference cloud, index order (FL_ICP_POINT_TO_PLANE only; 0 = unknown)
struct IcpWsLayout {
  size_t ref, mod, sref, zimg, nn, bnd, nd, dterm, perm, cell_start, cell_cur, nrm, total;
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
  L.sref = o; o = al256(o + 16 * (nn + 4));          // + NN_OVERRUN points at infinity behind the reference image
  L.zimg = o; o = al256(o + 4 * (nn + 4));           // organised search: crop pixel -> depth factor (float), + the overrun guard
  L.nn = o; o = al256(o + 4 * nn);
  L.bnd = o; o = al256(o + 4 * nn);
  L.nd = o; o = al256(o + 4 * nn);
  L.dterm = o; o = al256(o + 4 * nn + 4 * ICP_DT);     // read in whole blocks of ICP_DT terms
  L.perm = o; o = al256(o + 4 * nn);
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
  const FlRefineJob *jobs;     // kind 0 with caller-chosen matches (fl_refine_matches): job b refines jobs[b].match on frame jobs[b].frame
  const int *order;            // kind 0 batches: workgroup b runs job order[b] (longest first, see k_icp_order); null: job b
  unsigned *cu_chain;          // per-CU bookings of the chain waves' SIMDs (chain_elect); null: wave 0 chains
};

// LDS state of one frame workgroup of BS_ threads.  Parity mode: virtual wave 0 chains, the other BS/64 - 1 waves
// produce one tile row per thread.
template <int BS_>
struct IcpSharedT {
  static constexpr int BS = BS_;
  static constexpr int NW = BS_ / 64;
  // Parity mode: wave 0 runs the chains, producer waves write one tile row per thread.  A 16-wave workgroup keeps the
  // waves that share wave 0's SIMD (waves 4, 8, 12: the SPI deals a workgroup's waves round-robin over the 4 SIMDs) out
  // of the producer role, so the chain wave has its SIMD's issue slots to itself.
  static constexpr int NPROD = NW >= 8 ? NW - NW / 4 : NW - 1;
  static constexpr int CHAIN_NBUF = BS_ >= 1024 ? 4 : FL_ICP_NBUF_SMALL;
  static constexpr int TQ = NPROD * 64;   // rows per LDS tile
  // Which wave chains.  The 1024-thread workgroup: wave 0.  The 256-thread one: wave `cw`, elected per workgroup so that the
  // chain waves of the workgroups that share a CU sit on DIFFERENT SIMDs (chain_elect): a chain wave issues one dependent
  // add per 8 cycles -- half of its SIMD's issue slots for one to fifteen useful lanes -- and two of them on one SIMD leave
  // the other two waves of that SIMD the scraps while the neighbouring SIMDs idle.
  int cw;
  int wsimd[NW];                         // chain_elect: the SIMD each wave runs on
  int cu_slot, cu_simd;                  // chain_elect: what this workgroup booked in the per-CU table (cu_slot < 0: nothing)
  // tile row of this thread, or -1 (chain wave / idle wave)
  __device__ __forceinline__ int producer_slot() const
  {
    const int t = (int)threadIdx.x, w = t >> 6;
    if (NW >= 8) return (w & 3) == 0 ? -1 : t - 64 * (1 + (w >> 2));
    const int v = (w - __builtin_amdgcn_readfirstlane(cw) - 1) & (NW - 1);     // producers are virtual waves 0 .. NW - 2
    return v == NW - 1 ? -1 : v * 64 + (t & 63);
  }
  // lane of this thread in the chain wave, or -1
  __device__ __forceinline__ int chain_lane() const
  {
    const int t = (int)threadIdx.x;
    if (NW >= 8) return t < 64 ? t : -1;
    return (t >> 6) == __builtin_amdgcn_readfirstlane(cw) ? (t & 63) : -1;
  }
  static constexpr int TS = TQ + 4;       // tile column stride (floats): 16-byte aligned columns for ds_read_b128; the +4
                                          // keeps the 16 chain lanes of a b128 read on distinct bank groups
  float R[9], T[3], Ropt[9], Topt[3];
  float dist_mean, dist_diff, px, thr;
  int iter, n_corr, go, ok;
  int cnt, inl;                          // deferred dist_mean chain: valid pairs / inliers of the terms in dterm[]
  int stop;                              // the chain wave found the loop over: the search running ahead is abandoned
  int a1_next;                           // organised search: next unclaimed query (64 per step)
  float xmin, ymin, inv_c;
  int GX, GY, nsorted;
  float sums[16];
  double dsum[NW][32];                  // [wave][scalar]
  int iscan[NW + 1];
  int iscan2[2][NW];                    // crop_clouds / CSR scan: per-wave counts, double-buffered
  float fred[4][NW];
  int n, rect_m[4], rect_r[4], status, g;
  // double-buffered LDS tiles feeding the sequential float32 chains (FL_ICP_PARITY):
  // prod[b][k][r] = scalar k (9 products, 3 model coords, 3 reference coords) of row r of tile b
  alignas(16) float prod[2][15][TS];
  // rows per tile of the dist_mean phase: with the exact block sums the chain wave of a 4-wave workgroup has time to produce too
  static constexpr bool BSUM = FL_ICP_BSUM == 2 || (FL_ICP_BSUM == 1 && NW >= 8);   // exact block sums of the dist_mean chain in this kernel
  static constexpr bool ALLPROD = BSUM && FL_ICP_ALLPROD && NW < 8;
  static constexpr int DTQ = ALLPROD ? BS_ : TQ;
  alignas(16) float dtile[2][DTQ];
  // the organised search stages ICP_STAGE_CAP points per wave in the tile region above: a workgroup with few waves has small
  // tiles (TQ rows), so the region is padded up to what its waves stage
  static constexpr int STAGE_FLOATS = NW * 384 * 4, TILE_FLOATS = 2 * 15 * TS + 2 * DTQ;
  alignas(16) float stage_pad[STAGE_FLOATS > TILE_FLOATS ? STAGE_FLOATS - TILE_FLOATS : 4];
  alignas(16) float dchain[2][ICP_DT];   // the deferred dist_mean chain's staging (chain wave only)
#ifdef FL_ICP_PHASES
  long long tacc[16], tlast, tkernel;   // tkernel: clock at kernel entry (k_icp_pipeline)
  long long wall0;                      // wall_clock64() (constant 100 MHz) at kernel entry
  unsigned long long stime[8];         // FL_ICP_PHASES: cycles of the search step's segments, summed over the workgroup's waves
  unsigned hist[40];                    // organised search: [0,16) union (W class x H class), [16,26) largest lane window height, [26,30) width class
#endif
};

__device__ __forceinline__ bool vvalid(float z) { return z <= 900.0f; }      // common.cpp:261-266
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// float -> int as the hardware converts: saturating, NaN -> 0 (a C cast of an out-of-range value is undefined)
__device__ __forceinline__ int cvt_i32_sat(float f)
{
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}

// ---- block-level helpers (every thread of the workgroup must call) ---------------------------
template <class SH>
__device__ __forceinline__ int block_sum_int(SH &S, int v)
{
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) S.iscan[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int i = 0; i < SH::NW; ++i) t += S.iscan[i];
  return t;
}

__device__ __forceinline__ double shfl_xor_d(double v, int s)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, s, 64);
  hi = __shfl_xor(hi, s, 64);
  return __hiloint2double(hi, lo);
}

// fixed-shape fp64 reduction of NS scalars per thread; the result (double) is left in S.dsum[0][k], the caller rounds
template <int NS, class SH>
__device__ __forceinline__ void block_sum_double(SH &S, double *v)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    for (int i = 0; i < SH::NW; ++i) t += S.dsum[i][threadIdx.x];
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
// NBUF 16-row batches of the chain's column are kept in registers: NBUF - 1 of them are in flight while one is added (an
// LDS read takes longer than the 16 dependent adds of a batch).  Measured: 4 in the 1024-thread kernel (128 VGPRs:
// 9.5 -> 8.7 M cycles per frame at batch 1), 2 in the 256-thread one (96 VGPRs: 4 spills, 18 -> 21 ms per 1280 frames).
template <int NBUF>
__device__ __forceinline__ float chain_tile(const float *col, int rows, float acc)
{
  const int nb = rows >> 4;                              // full batches
  int b = 0;
  if (nb >= NBUF) {
    float4 buf[NBUF][4];
#pragma unroll
    for (int u = 0; u < NBUF; ++u) chain_load16(col + 16 * u, buf[u]);
    for (; b + 2 * NBUF <= nb; b += NBUF) {              // steady state: every batch added is replaced by the one NBUF ahead
#pragma unroll
      for (int u = 0; u < NBUF; ++u) {
        acc = chain_add16(buf[u], acc);
        chain_load16(col + 16 * (b + NBUF + u), buf[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < NBUF; ++u) acc = chain_add16(buf[u], acc);
    b += NBUF;
  }
  for (; b < nb; ++b) {                                  // fewer than NBUF batches are left (last tile only)
    float4 one[4];
    chain_load16(col + 16 * b, one);
    acc = chain_add16(one, acc);
  }
  for (int r = nb << 4; r < rows; ++r) acc += col[r];
  return acc;
}

// ---- exact block sums of a float32 chain with non-negative terms (getL2distClouds' dist_mean, ICP.cpp:68-111) ----------
// acc <- fl(fl(fl(acc + x0) + x1) + ...) is a chain of dependent adds (8.25 cycles each, one useful lane).  But while the
// running sum stays inside ONE binade [2^e, 2^(e+1)), every float it can take is a multiple of u = 2^(e-23), and
// fl(s + x) = s + rne(x / u) u for a multiple s of u -- unless x / u lies exactly half way between two integers (then the
// parity of s decides).  So for a block of terms x_k >= 0 (the sum only grows): if no x_k / u is a tie and
// acc / u + sum_k rne(x_k / u) < 2^24, the chain's result after the block is exactly that integer times u, whatever the
// order -- a wave-wide integer reduction instead of 64 x N dependent adds.  Ties (about 1 term in 2^(e - e_x)) and binade
// crossings (log2 of the sum over a whole cloud) make the block fall back to the term-by-term chain.  Bit-identical by
// construction; x * 2^(23-e) is exact (a power of two; an underflow rounds to a value below 1/2, which still rounds to 0).
struct BlkScale {
  float scale, ulp;      // 2^(23-e), 2^(e-23)
  int ai;                // acc / ulp, in [2^23, 2^24)
  bool ok;               // acc is a positive normal float whose scale and ulp are normal too
};
__device__ __forceinline__ BlkScale blk_scale(float acc)
{
  BlkScale b;
  const unsigned eb = __float_as_uint(acc) >> 23;          // sign + exponent: a negative, zero, denormal, inf or NaN acc fails the test
  b.ok = eb >= 24u && eb <= 254u;
  const unsigned e2 = b.ok ? eb : 127u;
  b.scale = __uint_as_float((277u - e2) << 23);
  b.ulp = __uint_as_float((e2 - 23u) << 23);
  b.ai = cvt_i32_sat(acc * b.scale);
  return b;
}
#define BLK_LANE_MAX 2097152.0f                             // 2^21: 768 terms below it sum to less than 2^31
__device__ __forceinline__ void blk_term(const BlkScale &b, float x, int &isum, bool &bad)
{
  const float y = x * b.scale, q = __builtin_rintf(y), f = y - q;
  bad = bad || !(x >= 0.0f) || !(y < BLK_LANE_MAX) || __builtin_fabsf(f) == 0.5f;
  isum += cvt_i32_sat(q);
}
// wave-wide: true and the new sum in acc if the block could be added exactly
__device__ __forceinline__ bool blk_finish(const BlkScale &b, int isum, bool bad, float &acc)
{
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) isum += __shfl_xor(isum, s, 64);
  const int ni = b.ai + isum;
  if (!b.ok || __ballot(bad) != 0ull || ni >= (1 << 24) || isum < 0) return false;
  acc = (float)ni * b.ulp;
  return true;
}
// all 64 lanes of the chain wave: acc + col[0 .. rows), as the term-by-term chain would leave it; every lane returns the sum
template <int NBUF>
__device__ __forceinline__ float chain_block_nonneg(const float *col, int rows, float acc)
{
  const int lane = threadIdx.x & 63;
  const BlkScale b = blk_scale(acc);
  int isum = 0;
  bool bad = false;
  for (int r = lane; r < rows; r += 64) blk_term(b, col[r], isum, bad);
  if (blk_finish(b, isum, bad, acc)) return acc;
  float r = acc;
  if (lane == 0) r = chain_tile<NBUF>(col, rows, acc);
  return uniform_f(r);
}

// The deferred dist_mean chain (executed by ONE wave, all 64 lanes enter): dterm[0 .. n) added strictly in order by lane 0.
// The wave streams the terms from HBM ICP_DT at a time (two coalesced float4 per lane, the next block in flight while
// this one is added) through its own LDS buffer `buf` (2 x ICP_DT floats); the array is padded to whole blocks.
template <int NBUF>
__device__ __forceinline__ float chain_deferred(const float *__restrict__ dterm, float (*buf)[ICP_DT], int n)
{
  const int lane = threadIdx.x & 63;
  const int nblk = (n + ICP_DT - 1) / ICP_DT;
  float acc = 0.0f;
  float4 r0, r1;
  if (nblk > 0) {
    r0 = *(const float4 *)(dterm + 4 * lane);
    r1 = *(const float4 *)(dterm + 256 + 4 * lane);
  }
  for (int b = 0; b < nblk; ++b) {
    const int cnt = min(ICP_DT, n - b * ICP_DT);
    const float4 c0 = r0, c1 = r1;
    if (b + 1 < nblk) {
      r0 = *(const float4 *)(dterm + (size_t)(b + 1) * ICP_DT + 4 * lane);
      r1 = *(const float4 *)(dterm + (size_t)(b + 1) * ICP_DT + 256 + 4 * lane);
    }
#if FL_ICP_BSUM
    {
      // the block straight from the registers it arrived in, as one exact integer sum (chain_block_nonneg); the words behind
      // term n - 1 of the last block are padding
      const BlkScale bs = blk_scale(acc);
      int isum = 0;
      bool bad = false;
      const float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) blk_term(bs, (k < 4 ? 4 * lane + k : 256 + 4 * lane + k - 4) < cnt ? v[k] : 0.0f, isum, bad);
      if (blk_finish(bs, isum, bad, acc)) continue;
    }
#endif
    float *dst = buf[b & 1];
    *(float4 *)(dst + 4 * lane) = c0;
    *(float4 *)(dst + 256 + 4 * lane) = c1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) acc = chain_tile<NBUF>(dst, cnt, acc);
    acc = uniform_f(acc);
  }
  return acc;                                            // every lane holds the sum
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
  unsigned i3;                                           // 12 i as (2 i + i) << 2: the compiler folds the C form back into a quarter-rate v_mul_lo_u32
  asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(i3) : "v"(i));
  __builtin_memcpy(&v, (const char *)base + (size_t)(i3 << 2), 12);
  return v;
}

// bnd[] is only ever an UPPER bound (a wider search radius visits more pixels, the neighbour found is the same), so it is kept
// as the top 16 bits of its float32 pattern, rounded UP: 8 of the ~155 bytes the kernel moves per point and iteration.
// Saturating: a finite value rounds up to at most +inf (0x7F80), and ANY NaN payload is stored as the canonical 0x7FC0 -- "no
// bound" -- instead of carrying into the exponent or the sign (0x7FFFxxxx + 0xFFFF would wrap to 0x8000 = a bound of -0).
// bnd values are non-negative by construction (distances and sums of distances).
typedef uint16_t bnd_t;
// (three instructions: every pattern at or above the canonical NaN -- the negative ones included -- is clamped to it first;
// the NaNs below it round up to a NaN no larger than it; FLT_MAX rounds up to +inf)
__host__ __device__ __forceinline__ uint16_t bnd_pack(unsigned bits)
{
  return (uint16_t)(((bits < 0x7FC00000u ? bits : 0x7FC00000u) + 0xFFFFu) >> 16);
}
extern "C" unsigned fl_dev_bnd_pack(unsigned bits) { return bnd_pack(bits); }      // for tests/test_abi_cpu.py (host arithmetic, no GPU)
__device__ __forceinline__ float bnd_ld(const bnd_t *__restrict__ b, int i) { return __uint_as_float((unsigned)ld_u32(b, i) << 16); }
__device__ __forceinline__ void bnd_st(bnd_t *b, int i, float v) { b[i] = bnd_pack(__float_as_uint(v)); }

// An UPPER bound of sqrt(x) for the search-radius bookkeeping (bnd[]): the hardware's 1-ulp v_sqrt_f32 inflated past
// its error (and past a flushed denormal) instead of the ~15-instruction correctly rounded sqrtf.  Any over-estimate
// only widens the visited area; the nearest neighbour found is the same.
__device__ __forceinline__ float sqrt_upper(float x) { return __builtin_amdgcn_sqrtf(x) * 1.000001f + 1.1e-19f; }

// A searchable reference point is a float4 (index bits, X, Y, Z): the index comes FIRST so that the 64-bit key
// (index low, d2 high) can be formed in the two registers the load put the index and X into -- X is dead once dx is
// computed -- without a register move per candidate.
__device__ __forceinline__ float4 nn_point(float x, float y, float z, int index) { return make_float4(__int_as_float(index), x, y, z); }
__device__ __forceinline__ int nn_point_index(const float4 &p) { return __float_as_int(p.x); }
#ifndef FL_ICP_BATCH_CLAMP
#define FL_ICP_BATCH_CLAMP 1
#endif
#define NN_OVERRUN 3                // readable points behind the last position of a staged window / of the reference image
#define NN_IDX_NONE 0x7fffffff      // index stored with a dropped pixel of the reference image (real indices are below it)

// ---- uniform x/y grid over the reference cloud --------------------------------------------------
__device__ __forceinline__ int cell_of(float v, float vmin, float inv_c, int G)
{
  float t = floorf((v - vmin) * inv_c);
  int c = t < 0.f ? 0 : (t > (float)(G - 1) ? G - 1 : (int)t);
  return c;
}

template <class SH>
__device__ __forceinline__ void build_grid(SH &S, const float *ref, int n_ref, float4 *sref, int *cell_start, int *cell_cur,
                           int ncell_max)
{
  constexpr int BS = SH::BS, NW = SH::NW;
  // bounding box of the finite points
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (int i = threadIdx.x; i < n_ref; i += BS) {
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
    for (int i = 1; i < NW; ++i) {
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
  for (int i = threadIdx.x; i < ncell; i += BS) cell_cur[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_ref; i += BS) {
    const F3 p3 = ld3_u32(ref, i);
    const float x = p3.x, y = p3.y, z = p3.z;
    if (isfinite(x) && isfinite(y) && isfinite(z))
      atomicAdd(&cell_cur[cell_of(y, S.ymin, S.inv_c, S.GY) * S.GX + cell_of(x, S.xmin, S.inv_c, S.GX)], 1);
  }
  __syncthreads();
  // exclusive scan of the counts -> cell_start (and cell_cur, the scatter cursors): four consecutive cells per thread,
  // a shuffle scan inside the wave, the wave totals through a double-buffered LDS slot -- one barrier per 4*BS cells
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int run = 0, step = 0;
    for (int base = 0; base < ncell; base += 4 * BS, ++step) {
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
#pragma unroll
      for (int w = 0; w < NW; ++w) { const int t = slot[w]; before += w < wv ? t : 0; total += t; }
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
  for (int i = threadIdx.x; i < n_ref; i += BS) {
    const F3 p3 = ld3_u32(ref, i);
    const float x = p3.x, y = p3.y, z = p3.z;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      const int slot = atomicAdd(&cell_cur[cell_of(y, S.ymin, S.inv_c, S.GY) * S.GX + cell_of(x, S.xmin, S.inv_c, S.GX)], 1);
      sref[slot] = nn_point(x, y, z, i);
    }
  }
  __syncthreads();
}

// ---- exact nearest neighbours within a window -------------------------------------------------------
// the grid parameters as wave-uniform scalars (SGPRs): read from LDS they would each cost a VGPR in the search loop
struct NnGrid {
  float xmin, ymin, inv_c;
  int GX, GY, nsorted;
};
template <class SH>
__device__ __forceinline__ NnGrid nn_grid(const SH &S)
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

// the radius every reference point within distance `lim` of the query lies within, inflated past the float rounding of
// d2, of the coordinate differences and of the bound's own arithmetic (orders of magnitude below the relative margins)
__device__ __forceinline__ float nn_radius(float qx, float qy, float qz, float lim)
{
  return lim * 1.0001f + 2e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz)) + 1e-30f;
}

// (d2, index) packed as d2's bit pattern (non-negative floats order like unsigned integers) in the high word and the
// reference index in the low word: one 64-bit unsigned minimum implements "smaller distance, ties to the lower
// index" exactly.
#define NN_KEY_NONE 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ unsigned long long nn_key(float qx, float qy, float qz, const float4 &p)
{
  const float dx = qx - p.y, dy = qy - p.z, dz = qz - p.w;
  float d = dx * dx;                                     // cvflann::L2_Simple<float>
  d += dy * dy;
  d += dz * dz;
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.x);
}
#define NN_CONSIDER(P) { const unsigned long long key_ = nn_key(qx, qy, qz, (P)); best = key_ < best ? key_ : best; }
// a key whose index is NN_IDX_NONE (a dropped pixel: d2 = inf) or NN_KEY_NONE itself means "nothing found"
#define NN_UNPACK(best, bi, bd)                                            \
  {                                                                        \
    const unsigned lo_ = (unsigned)((best) & 0xFFFFFFFFull);               \
    const bool found_ = lo_ < (unsigned)NN_IDX_NONE;                       \
    *(bi) = found_ ? (int)lo_ : -1;                                        \
    *(bd) = found_ ? __uint_as_float((unsigned)((best) >> 32)) : NAN;      \
  }

// ---- grid search (fl_icp: caller-supplied clouds) -------------------------------------------------------
// exact 1-NN among the points of the cell rows [cy0, cy1] x [cx0, cx1]; the grid and the sorted cloud are L2-resident
__device__ __forceinline__ void nn_search_grid(const NnGrid &S, const float4 *__restrict__ sref,
                                               const int *__restrict__ cell_start, float qx, float qy, float qz, float r,
                                               int *bi, float *bd)
{
  unsigned long long best = NN_KEY_NONE;
  const int last = S.nsorted - 1;
  if (last < 0) { NN_UNPACK(best, bi, bd) return; }
  int cx0 = 0, cx1 = S.GX - 1, cy0 = 0, cy1 = S.GY - 1;
  if (isfinite(r)) {
    cx0 = cell_of(qx - r, S.xmin, S.inv_c, S.GX);
    cx1 = cell_of(qx + r, S.xmin, S.inv_c, S.GX);
    cy0 = cell_of(qy - r, S.ymin, S.inv_c, S.GY);
    cy1 = cell_of(qy + r, S.ymin, S.inv_c, S.GY);
  }
  // The search is latency-bound and a wave pays for its slowest lane, so round trips are what counts:
  // the headers of 4 grid rows (8 loads) are fetched together, then the candidates of all 4 row segments
  // are enumerated as ONE flat list, FL_ICP_NB per round trip -- a lane needs ceil(total / NB) rounds however the
  // candidates are spread over the rows.  Slots past the end of the list are NOT masked: they read points that
  // follow the last row segment (clamped to the cloud), and looking at extra reference points never changes the
  // answer -- the minimum over a superset that still contains every point within the search radius is the same
  // nearest neighbour, ties to the lowest index included.
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

// ---- organised search (recognition / detection: the reference cloud is a back-projected crop) ------------
// The reference cloud of the organised search is an IMAGE of 12-byte points (crop pixel p -> X, Y, Z; +inf where the paired
// compaction dropped the pixel) followed by an image of their indices.  The search identifies a reference point by its PIXEL
// (nn[] holds pixel positions there; ties between equal distances go to the lower pixel, which is the lower index: the
// compaction is row-major), so the staged records need no index from memory -- 12 instead of 16 bytes per staged point, of a
// kernel that is bound by the bytes it moves -- and phase A2 gathers the partner from the image instead of from ref[].
// idximg is read once per frame (tile order) and by the point-to-plane mode (normals are stored by index).
__host__ __device__ __forceinline__ size_t org_idximg_offset(int pixels) { return (size_t)12 * (size_t)(pixels + 4); }
struct OrgGeom {
  int cw, ch;            // crop size: rimg[v * cw + u] holds the point of crop pixel (u, v)
  float offu, offv;      // scene pixel of crop pixel (0, 0) minus the principal point
  float fx, fy;
  float cwm, chm;        // (float)(cw - 1), (float)(ch - 1)
  int sx0, sy0;          // scene pixel of crop pixel (0, 0)
  float cx, cy, inv_fx, inv_fy;   // as crop_clouds uses them: (float)K.cx, 1.0f / (float)K.fx ...
};
// The reference point of crop pixel (u, v) from its depth factor zsf, by crop_clouds' own expression (depth_to_3d.cpp:119,132 +
// scale_mat_vec3f): bit for bit what the 12-byte image holds.  A dropped pixel (zsf = NaN) gives a NaN point, whose distance
// is NaN: its key orders behind every real one.
// element of the depth-factor image: the factor itself (float, NaN = dropped) or (FL_ICP_ZIMG == 2) the scene's 16-bit depth
// (0 = dropped, which rescaleDepth turns into NaN anyway: depth_to_3d.cpp:257-259)
#if FL_ICP_ZIMG == 2
typedef uint16_t zimg_t;
__device__ __forceinline__ float zimg_ld(const zimg_t *__restrict__ z, int i)
{
  const unsigned d = ld_u32(z, i);
  return d == 0 ? NAN : (float)d * (float)(1 / 1000.0);
}
__device__ __forceinline__ zimg_t zimg_pack(bool keep, unsigned ds, float) { return (zimg_t)(keep ? ds : 0u); }
#else
typedef float zimg_t;
__device__ __forceinline__ float zimg_ld(const zimg_t *__restrict__ z, int i) { return ld_u32(z, i); }
__device__ __forceinline__ zimg_t zimg_pack(bool keep, unsigned, float zsf) { return keep ? zsf : NAN; }
#endif
// (suf, svf): the SCENE pixel as floats, (float)(g.sx0 + u) and (float)(g.sy0 + v)
__device__ __forceinline__ F3 org_point_f(const OrgGeom &g, float suf, float svf, float zsf)
{
  F3 p;
  p.x = (((suf - g.cx) * g.inv_fx) * zsf) * 1000;
  p.y = (((svf - g.cy) * g.inv_fy) * zsf) * 1000;
  p.z = zsf * 1000;
  return p;
}
__device__ __forceinline__ F3 org_point(const OrgGeom &g, int u, int v, float zsf)
{
  return org_point_f(g, (float)(g.sx0 + u), (float)(g.sy0 + v), zsf);
}
// whole-wave minimum / maximum of an int by DPP (row_shr 1, 2, 4, 8, row_bcast 15 / 31), returned as a wave-uniform value
#define FL_DPP_RED(OP, IDENT)                                                                                 \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(IDENT), v, 0x111, 0xF, 0xF, false));                            \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(IDENT), v, 0x112, 0xF, 0xF, false));                            \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(IDENT), v, 0x114, 0xF, 0xF, false));                            \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(IDENT), v, 0x118, 0xF, 0xF, false));                            \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(IDENT), v, 0x142, 0xA, 0xF, false));                            \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(IDENT), v, 0x143, 0xC, 0xF, false));                            \
  return __builtin_amdgcn_readlane(v, 63);
__device__ __forceinline__ int wave_min_i(int v) { FL_DPP_RED(min, 0x7fffffff) }
__device__ __forceinline__ int wave_max_i(int v) { FL_DPP_RED(max, (int)0x80000000) }

// N whole-wave maxima at once (a minimum is the maximum of the negated values): the DPP steps of the N reductions are
// interleaved, so the wait states a DPP read needs after the VALU write of its source are filled by the other reductions'
// steps instead of s_nop -- six serial reductions cost 6 x (6 + 6 nops + 2) issue slots, five interleaved ones 5 x 7
template <int N>
__device__ __forceinline__ void wave_max_multi(int (&v)[N])
{
#define FL_DPP_STEP(CTRL, RMASK)                                                                                       \
  _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_)                                                                     \
    v[k_] = max(v[k_], __builtin_amdgcn_update_dpp((int)0x80000000, v[k_], CTRL, RMASK, 0xF, false));
  FL_DPP_STEP(0x111, 0xF)
  FL_DPP_STEP(0x112, 0xF)
  FL_DPP_STEP(0x114, 0xF)
  FL_DPP_STEP(0x118, 0xF)
  FL_DPP_STEP(0x142, 0xA)
  FL_DPP_STEP(0x143, 0xC)
#undef FL_DPP_STEP
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = __builtin_amdgcn_readlane(v[k], 63);
}

// Six whole-wave maxima with gfx950's lane-swap instructions: v_permlane32_swap exchanges the upper half of one register with
// the lower half of another, so ONE maximum of the swapped pair folds the two halves of BOTH values (value A's partials now live
// in lanes 0-31, value B's in 32-63); v_permlane16_swap does the same for 16-lane rows.  Four values folded into the four rows
// of one register and two into the halves of another need 4 + 5 DPP steps in all where six separate reductions need 36:
// 19 vector instructions instead of 36 (+ their wait states) per search step.
__device__ __forceinline__ void wave_max6(int (&v)[6])
{
  auto fold32 = [](int a, int b) {                         // lanes 0-31: max over a's halves, lanes 32-63: over b's
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    return max((int)r[0], (int)r[1]);
  };
  const int m01 = fold32(v[0], v[1]), m23 = fold32(v[2], v[3]);
  int m45 = fold32(v[4], v[5]);
  const auto q = __builtin_amdgcn_permlane16_swap((unsigned)m01, (unsigned)m23, false, false);
  int n = max((int)q[0], (int)q[1]);                        // rows 0..3: v[0], v[2], v[1], v[3] (16 partials each)
#define FL_DPP_STEP2(CTRL)                                                                                             \
  n = max(n, __builtin_amdgcn_update_dpp((int)0x80000000, n, CTRL, 0xF, 0xF, false));                                  \
  m45 = max(m45, __builtin_amdgcn_update_dpp((int)0x80000000, m45, CTRL, 0xF, 0xF, false));
  FL_DPP_STEP2(0x111)
  FL_DPP_STEP2(0x112)
  FL_DPP_STEP2(0x114)
  FL_DPP_STEP2(0x118)
#undef FL_DPP_STEP2
  m45 = max(m45, __builtin_amdgcn_update_dpp((int)0x80000000, m45, 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1, 3
  v[0] = __builtin_amdgcn_readlane(n, 15);
  v[2] = __builtin_amdgcn_readlane(n, 31);
  v[1] = __builtin_amdgcn_readlane(n, 47);
  v[3] = __builtin_amdgcn_readlane(n, 63);
  v[4] = __builtin_amdgcn_readlane(m45, 31);
  v[5] = __builtin_amdgcn_readlane(m45, 63);
}
// dev / tests: both reductions on one wavefront of inputs (tests/test_gpu_icp.py checks them against each other and numpy)
__global__ void k_dev_wave_max6(const int *in, int *out)
{
  int a[6], b[6];
  for (int k = 0; k < 6; ++k) a[k] = b[k] = in[k * 64 + threadIdx.x];
  wave_max6(a);
  wave_max_multi(b);
  if (threadIdx.x == 0)
    for (int k = 0; k < 6; ++k) { out[k] = a[k]; out[6 + k] = b[k]; }
}

// host entry for the test: 6 x 64 ints in, 6 (wave_max6) + 6 (wave_max_multi) out; plain HIP calls on the null stream
extern "C" int fl_dev_wave_max6(const int *in_host, int *out_host)
{
  int *d_in = nullptr, *d_out = nullptr;
  if (hipMalloc(&d_in, 6 * 64 * sizeof(int)) != hipSuccess || hipMalloc(&d_out, 12 * sizeof(int)) != hipSuccess) { (void)hipFree(d_in); return FL_ERR_HIP; }
  bool ok = hipMemcpy(d_in, in_host, 6 * 64 * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_dev_wave_max6, dim3(1), dim3(64), 0, 0, d_in, d_out);
    ok = hipGetLastError() == hipSuccess && hipMemcpy(out_host, d_out, 12 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return ok ? FL_OK : FL_ERR_HIP;
}

// The crop pixels whose points can lie within distance r of q: a point (X, Y, Z) of pixel (su, sv) satisfies
// su - cx = X fx / Z up to float rounding (it was generated as X = ((su - cx) / fx) Z), and |X - qx|, |Z - qz| <= r.
// The 0.01-pixel slop is an order of magnitude above that rounding (5e-7 relative on |su - cx| <= 2000 pixels).  An empty window has u_lo > u_hi.
// (the constants folded -- cul / cuh = offu +- slop, cvl / cvh = offv +- slop, wave-uniform -- and the clamping left to the
// saturating float -> int conversion; returns whether the window holds a pixel)
// Branch-free form of org_window2 for the pipelined step: every lane computes the projection (a lane whose radius is not finite
// or reaches Z <= 1 computes garbage and selects the whole crop, a lane that is not queryable selects nothing), so that a step
// has no exec-mask regions in front of its reductions.  Same windows.
__device__ __forceinline__ bool org_window2_flat(const OrgGeom &g, float cul, float cuh, float cvl, float cvh, float qx, float qy, float qz, float r,
                                                 bool queryable, int &u_lo, int &u_hi, int &v_lo, int &v_hi)
{
  const float zlo = qz - r, zhi = qz + r;
  const bool narrow = isfinite(r) && zlo > 1.0f;         // otherwise the whole crop (valid points have 0 < Z <= 900)
  const float ilo = __builtin_amdgcn_rcpf(zlo), ihi = __builtin_amdgcn_rcpf(zhi);
  const float xlo = qx - r, xhi = qx + r, ylo = qy - r, yhi = qy + r;
  const int iul = cvt_i32_sat(ceilf((xlo * (xlo < 0.f ? ilo : ihi)) * g.fx - cul));
  const int iuh = cvt_i32_sat(floorf((xhi * (xhi > 0.f ? ilo : ihi)) * g.fx - cuh));
  const int ivl = cvt_i32_sat(ceilf((ylo * (ylo < 0.f ? ilo : ihi)) * g.fy - cvl));
  const int ivh = cvt_i32_sat(floorf((yhi * (yhi > 0.f ? ilo : ihi)) * g.fy - cvh));
  u_lo = narrow ? max(iul, 0) : 0;
  u_hi = narrow ? min(iuh, g.cw - 1) : g.cw - 1;
  v_lo = narrow ? max(ivl, 0) : 0;
  v_hi = narrow ? min(ivh, g.ch - 1) : g.ch - 1;
  return queryable & (u_lo <= u_hi) & (v_lo <= v_hi);
}
__device__ __forceinline__ bool org_window2(const OrgGeom &g, float cul, float cuh, float cvl, float cvh, float qx, float qy, float qz, float r,
                                            int &u_lo, int &u_hi, int &v_lo, int &v_hi)
{
  u_lo = 0; u_hi = g.cw - 1; v_lo = 0; v_hi = g.ch - 1;
  const float zlo = qz - r, zhi = qz + r;
  if (isfinite(r) && zlo > 1.0f) {                       // otherwise the whole crop (valid points have 0 < Z <= 900)
    const float ilo = __builtin_amdgcn_rcpf(zlo), ihi = __builtin_amdgcn_rcpf(zhi);
    const float xlo = qx - r, xhi = qx + r, ylo = qy - r, yhi = qy + r;
    const int iul = cvt_i32_sat(ceilf((xlo * (xlo < 0.f ? ilo : ihi)) * g.fx - cul));
    const int iuh = cvt_i32_sat(floorf((xhi * (xhi > 0.f ? ilo : ihi)) * g.fx - cuh));
    const int ivl = cvt_i32_sat(ceilf((ylo * (ylo < 0.f ? ilo : ihi)) * g.fy - cvl));
    const int ivh = cvt_i32_sat(floorf((yhi * (yhi > 0.f ? ilo : ihi)) * g.fy - cvh));
    u_lo = max(iul, 0);
    u_hi = min(iuh, g.cw - 1);
    v_lo = max(ivl, 0);
    v_hi = min(ivh, g.ch - 1);
  }
  return u_lo <= u_hi && v_lo <= v_hi;
}

// every lane's window enumerated in lockstep, maxh rows of maxw positions, FL_ICP_NBQ positions per batch (lanes with a
// smaller window re-read their own last column / row: duplicates do not change a minimum);
// fetch((v - ov) * RS + (u - ou)) returns the point of crop pixel (u, v)
template <typename F>
__device__ __forceinline__ unsigned long long org_scan(F fetch, int RS, int ou, int ov, float qx, float qy, float qz, int u_lo, int u_hi,
                                                       int v_lo, int v_hi, int maxw, int maxh)
{
  constexpr int NBQ = FL_ICP_NBQ;
  unsigned long long best = NN_KEY_NONE;
  const int wl = u_hi - u_lo, hl = v_hi - v_lo;
  const int b0 = (v_lo - ov) * RS + (u_lo - ou);
#if FL_ICP_BATCH_CLAMP
  // A batch is NBQ CONSECUTIVE positions from a clamped start (one address and immediate offsets instead of a clamp and
  // an address per position).  A window narrower than NBQ reads up to NBQ - 1 = NN_OVERRUN positions past its right edge:
  // the next pixels of the row, the start of the next row, or -- behind the last row -- the points the caller keeps there
  // (copies of the last staged point / points at infinity behind the image).  All of them are reference points of this
  // frame or points at infinity, and looking at more reference points never changes the nearest one.
  static_assert(NBQ - 1 <= NN_OVERRUN, "the overrun guard behind the staged window / the image is NN_OVERRUN points");
  const int wlc = max(wl - (NBQ - 1), 0);
#endif
  for (int dv = 0; dv < maxh; ++dv) {
    const int rb = b0 + min(dv, hl) * RS;
    for (int du = 0; du < maxw; du += NBQ) {               // one batch at a time: the other waves of the SIMD cover the LDS latency
      float4 cur[NBQ];
#if FL_ICP_BATCH_CLAMP
      const int bb = rb + min(du, wlc);
#pragma unroll
      for (int e = 0; e < NBQ; ++e) cur[e] = fetch(bb + e);
#else
#pragma unroll
      for (int e = 0; e < NBQ; ++e) cur[e] = fetch(rb + min(du + e, wl));
#endif
#pragma unroll
      for (int e = 0; e < NBQ; ++e) NN_CONSIDER(cur[e])
    }
  }
  return best;
}

// ---- getL2distClouds (ICP.cpp:68-111) over the index-paired clouds, optionally fused with the
// in-place transformPoints that precedes it (ICP.cpp:28-45, 756).  FL_ICP_PARITY: the producer waves write
// the per-point terms into double-buffered LDS tiles while lane 0 of wave 0 adds the previous tile
// in index order (the reference's `dist_mean += dist` chain); one barrier per tile.
// Also maintains bnd[i], an upper bound on the distance from model point i to SOME reference point: the index pair
// before the first iteration (Ropt == nullptr), afterwards the bound grows by how far point i moved (the search
// replaces it by the distance it found).
// DEFER (parity mode, organised search): the terms go to dterm[] in HBM instead of LDS tiles and are NOT added here;
// every thread produces, the valid / inlier counts are left in S.cnt / S.inl, and the chain wave adds the terms later
// (chain_deferred) while the other waves already search for the next iteration.
template <int MODE, bool DEFER, class SH>
__device__ __forceinline__ float l2dist_phase(SH &S, float *mod, const float *ref, bnd_t *bnd, float *dterm, int n, float thr,
                                             const float *Ropt, const float *Topt)
{
  constexpr bool parity = MODE == FL_ICP_PARITY && !DEFER;   // chains in this phase
  // With the exact block sums (FL_ICP_BSUM) the chain costs its wave a few dozen instructions per tile, so in the 4-wave
  // workgroup the chain wave produces as well: tiles of BS rows, four producer waves instead of three.
  constexpr bool allprod = parity && SH::ALLPROD;
  constexpr int TQ = parity && !allprod ? SH::TQ : SH::BS;   // rows per tile
  const int slot = parity && !allprod ? S.producer_slot() : (int)threadIdx.x;
  const int clane = parity ? S.chain_lane() : -1;        // lane of this thread in the chain wave, or -1
  int counter = 0, inl = 0;
  double dsum[1] = {0.0};
  float acc = 0.0f;
  const int ntiles = (n + TQ - 1) / TQ;
  // 256-thread parity kernel: the chain wave and the producers meet at a barrier every KB tiles, not every tile.  The chain is
  // one lane adding one float per row; at a barrier per 192 rows it spent 2 - 3 of its ~11 cycles per row there and in the LDS
  // round trip that opens each tile (phase stamps: 11.3 -> 9.5 cycles per row inside blocks of 4, 8.6 inside blocks of 8 -- but
  // the chain starts a block after the producers and ends a block after them, which is why 4 beats 8 and 15; blocks graded
  // 1, 1, 2, 2, 4, 4, 8 ... from both ends cost more in bookkeeping than they saved).  The terms of a block go to one of two
  // buffers of KB tiles in the (idle) tiles of phase A2.
  constexpr int KB = parity && !allprod && !SH::BSUM && SH::NW < 8 ? FL_ICP_BMACRO : 1;
  static_assert(KB == 1 || sizeof(S.prod) >= (size_t)2 * KB * TQ * sizeof(float), "two blocks of KB tiles of terms fit the A2 tiles");
  // Block 0 is tile 0 alone (the chain starts after one tile, not after KB), block m >= 1 the tiles (m - 1) KB + 1 .. m KB.
  auto dbuf = [&](int t) -> float * {                    // where the terms of tile t go
    if (KB == 1) return S.dtile[t & 1];
    const int m = (t + KB - 1) / KB, first = m == 0 ? 0 : (m - 1) * KB + 1;
    return (float *)&S.prod[0][0][0] + ((m & 1) * KB + (t - first)) * TQ;
  };
  // The phase is one memory round trip + one barrier per tile, so the next tile's (coalesced) loads are issued before this
  // tile is processed.  Two register sets trade roles by unrolling the tile loop twice -- never by moving registers, and
  // nothing computes with a loaded value in the iteration that issued its load: either makes the compiler drain the whole
  // memory queue (s_waitcnt vmcnt(0)) once per tile (see the A2 phase in icp_run).  With chains in this phase (parity) the
  // tile barrier is a raw s_barrier behind an lgkmcnt wait: the LDS tile must be complete, the global stores of this
  // phase (mod, bnd) need not be before the phase ends.
  struct Row { F3 a, b; float bnd; };
  auto row_load = [&](Row &w, int t) {
    const int i = min(t * TQ + slot, n - 1);              // clamped: unused past the end
    w.a = ld3_u32(mod, i);
    w.b = ld3_u32(ref, i);
    w.bnd = Ropt ? bnd_ld(bnd, i) : 0.0f;
  };
  auto row_process = [&](const Row &w, int t) {
    const int i = t * TQ + slot;
    float term = 0.0f;
    float a[3] = {w.a.x, w.a.y, w.a.z};
    const float b0 = w.b.x, b1 = w.b.y, b2 = w.b.z;
    if (i < n) {
      if (Ropt) {
        float move = 0.0f;
        if (vvalid(a[2])) {                               // transformPoints in place (:28-45, :756)
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
        bnd_st(bnd, i, w.bnd + move);                     // triangle inequality: still reaches the old partner
      } else {
        // first bound: the index pair (n_ref >= n_model); NaN/inf simply disable the bound
        const float ex = a[0] - b0, ey = a[1] - b1, ez = a[2] - b2;
        bnd_st(bnd, i, sqrt_upper(ex * ex + ey * ey + ez * ez));
      }
      if (vvalid(b2) && vvalid(a[2])) {
        const float dx = a[0] - b0, dy = a[1] - b1, dz = a[2] - b2;
        // cv::norm(Vec3f): squares accumulated in double, sqrt in double, stored to float (:88)
        const float dist = (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
        if (dist <= thr) { term = dist; ++inl; dsum[0] += (double)dist; }
        ++counter;
      }
    }
    if (parity) dbuf(t)[slot] = term;                     // non-inliers add an exact +0.0f
    if (DEFER && i < n) dterm[i] = term;
  };
  auto tile_barrier = [&]() {
    if (parity) {
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // s_waitcnt lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
    }
  };
  // tile t (block t of KB tiles) is complete in LDS: the chain wave adds it (every lane of the wave enters)
  auto chain_step = [&](int t) {
    const int first = KB == 1 || t == 0 ? t : (t - 1) * KB + 1, last = KB == 1 || t == 0 ? t : t * KB;   // the block's tiles
    const float *col = dbuf(first);
    const int rows = min((last + 1) * TQ, n) - first * TQ;
    if (SH::BSUM) acc = chain_block_nonneg<SH::CHAIN_NBUF>(col, rows, acc);
    else if (clane == 0) acc = chain_tile<SH::CHAIN_NBUF>(col, rows, acc);
  };
  // blocks of tiles; barriers inside the phase: behind every COMPLETE block (tile 0, tile KB, tile 2 KB, ...)
  const int nblocks = KB == 1 ? ntiles : (ntiles > 0 ? (ntiles - 1 + KB - 1) / KB + 1 : 0);
  const int nbar = KB == 1 ? ntiles : (ntiles > 0 ? (ntiles - 1) / KB + 1 : 0);
  auto block_barrier = [&](int t) {                       // behind tile t: the barrier that closes a block
    if (KB == 1 || t % KB == 0) tile_barrier();
  };
  const bool chain_wave = __builtin_amdgcn_readfirstlane(clane) >= 0;
  // producer loop: `before_barrier(t)` runs behind the rows of tile t (the chain wave's block sum of tile t - 1 when every wave
  // produces).  FL_ICP_BPD tiles of loads in flight per thread: register sets that trade roles by unrolling (see above).
  auto produce = [&](auto &&before_barrier) {
#if FL_ICP_BPD >= 2
    Row X, Y, Z;
    if (ntiles > 0) { row_load(X, 0); row_load(Y, 1); }
    for (int t = 0; t < ntiles; t += 3) {
      row_load(Z, t + 2);
      row_process(X, t);
      before_barrier(t);
      block_barrier(t);
      if (t + 1 < ntiles) {
        row_load(X, t + 3);
        row_process(Y, t + 1);
        before_barrier(t + 1);
        block_barrier(t + 1);
      }
      if (t + 2 < ntiles) {
        row_load(Y, t + 4);
        row_process(Z, t + 2);
        before_barrier(t + 2);
        block_barrier(t + 2);
      }
    }
#else
    Row A, B;
    if (ntiles > 0) row_load(A, 0);
    for (int t = 0; t < ntiles; t += 2) {
      row_load(B, t + 1);
      row_process(A, t);
      before_barrier(t);
      block_barrier(t);
      if (t + 1 < ntiles) {
        row_load(A, t + 2);
        row_process(B, t + 1);
        before_barrier(t + 1);
        block_barrier(t + 1);
      }
    }
#endif
  };
  if (allprod) {
    // every wave produces; the chain wave adds tile t - 1 (complete since the last barrier, the other buffer) behind its own rows of tile t
    produce([&](int t) { if (chain_wave && t > 0) chain_step(t - 1); });
  } else if (chain_wave) {
    // the chain wave's own loop (see the A2 phase): one barrier per tile like the producers' below
    CH_STAMP_BEGIN;
    for (int t = 0; t < nbar; ++t) {
      if (t > 0) chain_step(t - 1);
      CH_STAMP(c_add);
      tile_barrier();
      CH_STAMP(c_bar);
    }
    CH_STAMP_END(26);
  } else if (slot >= 0) {
    produce([&](int) {});
  } else {
    for (int t = 0; t < nbar; ++t) tile_barrier();         // a wave that neither chains nor produces (1024-thread workgroup)
  }
  __syncthreads();                                         // the phase's stores (mod, bnd, dterm) are visible to the workgroup
  if (parity && chain_wave)                                // the blocks the loop above has not added: the last one, or the last two
    for (int t = max((allprod ? nblocks : nbar) - 1, 0); t < nblocks; ++t) chain_step(t);
  counter = block_sum_int(S, counter);
  inl = block_sum_int(S, inl);
  if (DEFER) {
    // An upper bound of the dist_mean the chain will produce, from a parallel fp64 sum P of the same terms: the float32
    // chain c of n non-negative terms satisfies c <= (1 + n u) E, u = 2^-24, E the exact sum (and P = E to 1e-12), and
    // the division adds one rounding.
    block_sum_double<1>(S, dsum);
    if (threadIdx.x == 0) { S.cnt = counter; S.inl = inl; }
    const double P = S.dsum[0][0];
    __syncthreads();
    if (counter <= 0) return INFINITY;                   // dist_mean = FLT_MAX
    return (float)(P / (double)inl * (1.0 + 1.3e-7 * (double)n + 1.0e-6)) * 1.000001f;      // 0 inliers: NaN, as dist_mean will be
  }
  float dm;
  if (parity) {
    if (clane == 0) S.sums[0] = acc;
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
  return 0.0f;
}

// ---- icpCloudToCloud_Ex (ICP.cpp:617-809) --------------------------------------------------------
// ORG: the reference cloud is also available as the image `og` describes (sref = image of points + image of indices, perm = tile order), see
// "Organised search" at the top; otherwise the grid is built here and searched.
template <int MODE, bool ORG, class SH>
__device__ __forceinline__ void icp_run(SH &S, uint8_t *wsb, const IcpWsLayout &L, int n_ref, int n_model,
                        int it_thr, float dmt, float ddt, fl_icp_result *res, const OrgGeom &og)
{
  constexpr int BS = SH::BS, NW = SH::NW;
  float *ref = (float *)(wsb + L.ref), *mod = (float *)(wsb + L.mod);
  float4 *sref = (float4 *)(wsb + L.sref);
  // where the partner of nn[i] = j is read from: the organised search names it by its crop pixel (the image of points),
  // the grid search by its index (ref[]); idximg maps a pixel to its index (point-to-plane: normals are stored by index)
  const float *jsrc = ORG ? (const float *)sref : ref;
  const int *idximg = ORG ? (const int *)((const uint8_t *)sref + org_idximg_offset(og.cw * og.ch)) : nullptr;
  int *nn = (int *)(wsb + L.nn);
  bnd_t *bnd = (bnd_t *)(wsb + L.bnd);
  float *nd = (float *)(wsb + L.nd), *dterm = (float *)(wsb + L.dterm);
  const int *perm = (const int *)(wsb + L.perm);
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
    for (int i = 0; i < 40; ++i) S.hist[i] = 0;
    for (int i = 0; i < 8; ++i) S.stime[i] = 0;
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
  if (!ORG) build_grid(S, ref, n_ref, sref, cell_start, cell_cur, L.ncell_max);
  TSTAMP(0);
  // copyPoints(pts_model, pts_model_tmp) (:666-667): invalid points become Vec3f() = 0
  for (int i = threadIdx.x; i < n_model; i += BS)
    if (!vvalid(mod[3 * i + 2])) { mod[3 * i] = 0.f; mod[3 * i + 1] = 0.f; mod[3 * i + 2] = 0.f; }
  if (threadIdx.x == 0) {
    S.R[0] = S.R[4] = S.R[8] = 1.f;                      // R = eye, T = 0 (:644-645)
    S.dist_diff = FLT_MAX;
  }
  __syncthreads();
  // Search ahead of the distance chain (parity mode, organised search).  getL2distClouds' dist_mean is one float32 chain over
  // all points: alone it costs a quarter of an iteration during which every wave but one waits.  So the terms are written
  // to dterm[] (l2dist_phase<DEFER>), and while the chain wave adds them the other waves already run PointsCorresponding for
  // the NEXT iteration -- against a threshold that is not known yet.  It needs none: the search radius of a point is
  // min(bnd[i], r_lim) with r_lim >= sqrt(3 dist_mean) whatever the chain will find (l2dist_phase returns an upper bound
  // of its result, a fraction of a percent above a parallel sum of the same terms), the exact neighbour and its distance
  // are stored (nn[], nd[]), and `d <= 3 dist_mean` (:268) is applied where the pairs are consumed.  A wider radius only
  // visits more pixels; the neighbour found is the same.  If the chain ends the loop, the search is abandoned (S.stop).
  // Only the 1024-thread workgroup (batches that leave CUs idle) does this: with four 256-thread workgroups per CU the
  // other workgroups already fill the chain's shadow, and the extra traffic (dterm, nd) costs 7 % there (profiles/README.md).
  constexpr bool SPEC = (FL_ICP_SPEC == 2 || (FL_ICP_SPEC == 1 && BS >= ICP_BS_WIDE)) && MODE == FL_ICP_PARITY && ORG;
  float mean_ub = l2dist_phase<MODE, SPEC>(S, mod, ref, bnd, dterm, n_model, FLT_MAX, nullptr, nullptr);             // :670
  int pending = SPEC ? 1 : 0;                            // 1: the initial distances await their chain, 2: an iteration's
  bool have_nn = false;                                  // nn[] / nd[] hold the neighbours of the current model cloud
  float old_mean = 0.0f;                                 // dist_mean before the pending distances (pending == 2)

  // ---- organised search: PointsCorresponding (:193-279) for every model point, exact 1-NN within min(bnd[i], r_lim) ----
  // 64 queries of one (or two adjacent) 16x4-pixel tiles per wave and step: window, staging, scan.  Steps are claimed from a
  // workgroup counter (S.a1_next, reset by the caller) where a wave joins late (SPEC), four steps ahead of the one being
  // scanned, so that the loads of the next steps (perm -> mod, bnd) are in flight; otherwise wave w takes steps w, w + NW, ...
  // found(active, i, qx, qy, qz, j, d): j = -1, d = NaN when no reference point lies within the radius.
  // Around the distance scan of a step (which is what the step is for: 4 rows x 46 instructions per batch of four
  // positions) round 2 spent as many instructions again; what is left of that:
  //  * the union rectangle is staged in whole passes of 64 points, as many as it needs (a compile-time count per case), the
  //    point of a slot by a reciprocal multiply: no division, no per-pass predicate, no separate guard points (measured:
  //    rows padded to 16 / 32 / 64 points need no address arithmetic at all, but nine steps in ten are 14 - 29 pixels wide
  //    and would stage twice the points: 26.8 against 25.4 ms per 2560 frames, profiles/README.md);
  //  * the five wave reductions (union rectangle, tallest lane window) are interleaved (wave_max_multi), the batches per
  //    row come from two ballots;
  //  * the window arithmetic has its constants folded and leaves the clamping to the saturating float -> int conversion.
  auto org_search = [&](const float r_lim, const bool poll_stop, auto &&found) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    static_assert(sizeof(S.prod) + sizeof(S.dtile) + sizeof(S.stage_pad) >= (size_t)NW * ICP_STAGE_CAP * 16, "the chain tiles (idle during the search) hold every wave's staged rows");
    static_assert(offsetof(SH, stage_pad) == offsetof(SH, dtile) + sizeof(S.dtile), "prod, dtile and stage_pad are one contiguous region");
    static_assert(offsetof(SH, dtile) == offsetof(SH, prod) + sizeof(S.prod), "prod and dtile are one contiguous region");
    float4 *stage = (float4 *)&S.prod[0][0][0] + wv * ICP_STAGE_CAP;
    const float *rimg = (const float *)sref;
    // (the 1024-thread kernel keeps the 12-byte image: a frame alone on its CU waits for the rebuilt points -- 2.92 against 2.67 ms
    // per 8 frames -- where four workgroups per CU gain from the bytes: 32.9 against 34.2 ms per 4096)
    constexpr bool ZIMG = FL_ICP_ZIMG && MODE == FL_ICP_PARITY && NW < 8;
    const zimg_t *zimg = (const zimg_t *)(wsb + L.zimg);
    const int last_s = n_model - 1;
    const float cul = uniform_f(og.offu + 0.01f), cuh = uniform_f(og.offu - 0.01f), cvl = uniform_f(og.offv + 0.01f), cvh = uniform_f(og.offv - 0.01f);
    constexpr int stride = NW * 64;
    int static_next = wv * 64 + 4 * stride;
    auto claim = [&](int count) {                          // `count` queries off the workgroup's list (wave-uniform result)
      int v = 0;
      if (lane == 0) v = atomicAdd(&S.a1_next, count);
      return __builtin_amdgcn_readfirstlane(v);
    };
    auto next_step = [&]() {
      if (SPEC) return claim(64);
      const int v = static_next;
      static_next += stride;
      return v;
    };
    const int sdist = SPEC ? 64 : stride;
    int sb0 = SPEC ? claim(256) : wv * 64, sb1 = sb0 + sdist, sb2 = sb0 + 2 * sdist, sb3 = sb0 + 3 * sdist;
    int i_c = ld_u32(perm, min(sb0 + lane, last_s));
    int i_n = ld_u32(perm, min(sb1 + lane, last_s));
    int i_nn = ld_u32(perm, min(sb2 + lane, last_s));
    F3 q_c = ld3_u32(mod, i_c), q_n = ld3_u32(mod, i_n);
    float b_c = bnd_ld(bnd, i_c), b_n = bnd_ld(bnd, i_n);
    // A step's results are handed to found() (which stores them) at the head of the NEXT step, in front of that step's
    // staging loads: vmcnt counts loads and stores in one in-order queue, so stores issued at the end of a step were what the
    // wait at the head of the next one waited for (20 % of the phase); issued here, the wait that follows them is the one
    // for the staging loads, which covers them for free.
    bool pend = false, p_active = false;
    int p_i = 0, p_j = -1;
    float p_qx = 0.f, p_qy = 0.f, p_qz = 0.f, p_d = NAN;
#ifdef FL_ICP_PHASES
    long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = clock64();
#define ST_STAMP(k) { const long long now_ = clock64(); st_acc[k] += now_ - st_last; st_last = now_; }
#else
#define ST_STAMP(k) { }
#endif
    while (sb0 < n_model) {
      if (poll_stop && *(volatile int *)&S.stop) break;
      const int sb4 = next_step();
      const int i = i_c;
      const float qx = q_c.x, qy = q_c.y, qz = q_c.z;
#ifdef FL_ICP_PHASES
      asm volatile("" :: "v"(qx), "v"(b_c));             // the query and its bound have arrived
#endif
      ST_STAMP(0)
      if (pend) { found(p_active, p_i, p_qx, p_qy, p_qz, p_j, p_d); pend = false; }
      const bool active = sb0 + lane < n_model;
      const bool queryable = active && r_lim >= 0.f && isfinite(qx) && isfinite(qy) && isfinite(qz);
      // ---- this lane's window, the union rectangle, the tallest window ----
      int u_lo = 1, u_hi = 0, v_lo = 1, v_hi = 0;
      bool some = false;
      if (queryable) some = org_window2(og, cul, cuh, cvl, cvh, qx, qy, qz, nn_radius(qx, qy, qz, fminf(b_c, r_lim)), u_lo, u_hi, v_lo, v_hi);   // NaN bnd -> r_lim
      const int big = 0x3fffffff;
      int red[5] = {some ? -u_lo : -big, some ? u_hi : -1, some ? -v_lo : -big, some ? v_hi : -1, some ? v_hi - v_lo + 1 : 0};
      wave_max_multi(red);
      const int U0 = -red[0], U1 = red[1], V0 = -red[2], V1 = red[3], maxh = red[4];
      const bool any = U1 >= U0;                            // wave-uniform: some lane has a window
      ST_STAMP(1)
      // the loads of the step after next (the next one's are in flight)
      const F3 q_nn = ld3_u32(mod, i_nn);
      const float b_nn = bnd_ld(bnd, i_nn);
      const int i_nnn = ld_u32(perm, min(sb3 + lane, last_s));
      int j = -1;
      float d = NAN;
      if (any) {
        if (!some) { u_lo = u_hi = U0; v_lo = v_hi = V0; }     // lanes without a window look at one point of the union: a real
                                                             // reference point beyond their radius, which the gate drops
        const int wl = u_hi - u_lo, hl = v_hi - v_lo;
        // 4-wide batches per row of the widest lane window: 1 or 2 by ballot, beyond that by a reduction (first iterations)
        int nbw = 1;
        if (__ballot(wl > 3) != 0ull) nbw = __ballot(wl > 7) == 0ull ? 2 : (wave_max_i(wl) >> 2) + 1;
        const int W = U1 - U0 + 1, H = V1 - V0 + 1, area = W * H;
        // The union rectangle is staged row by row at its own width: LDS slot k = lane + 64 p holds its point (k / W, k % W).
        // Whole passes of 64 slots are staged, enough for the rectangle and the 3 slots a 4-wide batch may run past its last
        // row; slots past the rectangle hold further points of the image (or the points at infinity behind it), which is
        // all a scan may ever look at: a real reference point of this frame or infinity never changes the nearest one.
        const int npneed = (area + 3 + 63) >> 6, npass = npneed <= 2 ? 2 : (npneed <= 4 ? npneed : 6);
        const bool staged = npneed <= 6;
#ifdef FL_ICP_PHASES
        if (lane == 0) {
          atomicAdd((unsigned long long *)&S.tacc[8], 1ull);
          atomicAdd((unsigned long long *)&S.tacc[9], (unsigned long long)(4 * nbw * maxh));
          atomicAdd((unsigned long long *)&S.tacc[10], staged ? 0ull : 1ull);
          atomicAdd((unsigned long long *)&S.tacc[11], (unsigned long long)(staged ? npass * 64 : 0));
          if (S.iter <= 3) atomicAdd((unsigned long long *)&S.tacc[12], (unsigned long long)(4 * nbw * maxh));
          const int wcl = W <= 13 ? 0 : (W <= 29 ? 1 : (W <= 61 ? 2 : 3)), hcl = H <= 5 ? 0 : (H <= 10 ? 1 : (H <= 20 ? 2 : 3));
          atomicAdd(&S.hist[wcl * 4 + hcl], 1u);
          atomicAdd(&S.hist[16 + min(maxh, 6) - 1], 1u);   // (bins 22..24 and 26..28 carry the chain phases' stamps)
        }
#endif
        unsigned long long best = NN_KEY_NONE;
        if (staged) {
          // k / W by reciprocal: (k + 0.5) / W stays 0.5 / W away from the integers, three orders of magnitude more than the
          // error of v_rcp_f32 and the product (k < 448)
          const float invW = uniform_f(__builtin_amdgcn_rcpf((float)W));
          const int base = (int)__umul24((unsigned)V0, (unsigned)og.cw) + U0, last_pt = og.cw * og.ch + NN_OVERRUN - 1;
          auto stage_passes = [&](auto np_) {
            constexpr int NP = decltype(np_)::value;
            float4 R[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const int row = (int)(((float)(lane + 64 * p) + 0.5f) * invW), col = lane + 64 * p - row * W;
              const int pos = min((int)__umul24((unsigned)row, (unsigned)og.cw) + col + base, last_pt);
              if (ZIMG) {
                // 4 bytes per staged point instead of 12: the pixel's depth factor; its point rebuilt by crop_clouds' expression
                // (pixel (U0 + col, V0 + row); a slot clamped to the guard behind the image reads NaN whatever its pixel)
                const float zf = zimg_ld(zimg, pos);
                const F3 pt = org_point(og, U0 + col, V0 + row, zf);
                R[p] = nn_point(pt.x, pt.y, pt.z, zf != zf ? NN_IDX_NONE : pos);
              } else {
                const F3 pt = ld3_u32(rimg, pos);
                R[p] = nn_point(pt.x, pt.y, pt.z, pt.x == INFINITY ? NN_IDX_NONE : pos);
              }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) stage[lane + 64 * p] = R[p];
          };
          if (npass == 3) stage_passes(std::integral_constant<int, 3>());
          else if (npass == 4) stage_passes(std::integral_constant<int, 4>());
          else if (npass == 2) stage_passes(std::integral_constant<int, 2>());
          else stage_passes(std::integral_constant<int, 6>());
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          ST_STAMP(2)
          // scan: maxh rows (a lane with fewer re-reads its last one) of nbw batches of 4 consecutive points (a window narrower
          // than 4 reads on into the next points of its row, of the next row, or of the slots behind the rectangle)
          static_assert(FL_ICP_NBQ == 4 && NN_OVERRUN == 3, "the staged scan is written for 4-wide batches (cur[4], wl - 3, area + 3): "
                                                            "build org_scan's FL_ICP_NBQ variants only with this path rewritten to match");
          const float4 *row0 = stage + (v_lo - V0) * W + (u_lo - U0);
          if (nbw == 1) {                                      // every lane's window is at most 4 wide: one batch per row
            for (int dv = 0; dv < maxh; ++dv) {
              const float4 *bp = row0 + min(dv, hl) * W;
              float4 cur[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) cur[e] = bp[e];
#pragma unroll
              for (int e = 0; e < 4; ++e) NN_CONSIDER(cur[e])
            }
          } else {
            const int wlc = max(wl - 3, 0);
            for (int dv = 0; dv < maxh; ++dv) {
              const float4 *rowp = row0 + min(dv, hl) * W;
              for (int du = 0; du < 4 * nbw; du += 4) {
                const float4 *bp = rowp + min(du, wlc);
                float4 cur[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[e] = bp[e];
#pragma unroll
                for (int e = 0; e < 4; ++e) NN_CONSIDER(cur[e])
              }
            }
          }
        } else {
          best = org_scan([&](int idx) { const F3 pt = ld3_u32(rimg, idx); return nn_point(pt.x, pt.y, pt.z, pt.x == INFINITY ? NN_IDX_NONE : idx); },
                          og.cw, 0, 0, qx, qy, qz, u_lo, u_hi, v_lo, v_hi, 4 * nbw, maxh);
        }
        if (queryable) NN_UNPACK(best, &j, &d)
#ifdef FL_ICP_PHASES
        asm volatile("" :: "v"(j), "v"(d));
#endif
        ST_STAMP(3)
      }
      pend = true; p_active = active; p_i = i; p_qx = qx; p_qy = qy; p_qz = qz; p_j = j; p_d = d;
      i_c = i_n; q_c = q_n; b_c = b_n;
      i_n = i_nn; q_n = q_nn; b_n = b_nn;
      i_nn = i_nnn;
      sb0 = sb1; sb1 = sb2; sb2 = sb3; sb3 = sb4;
      ST_STAMP(4)
    }
    if (pend) found(p_active, p_i, p_qx, p_qy, p_qz, p_j, p_d);
#ifdef FL_ICP_PHASES
    if (lane == 0)
      for (int k = 0; k < 5; ++k) atomicAdd(&S.stime[k], (unsigned long long)st_acc[k]);
#endif
#undef ST_STAMP
  };
  // (Measured in round 3 and not kept, profiles/README.md: the same search software-pipelined over its steps -- the next
  // step's rectangle fetched by LDS-DMA while this one is scanned -- shortens the search phase by a fifth and lengthens the
  // chain phases by as much; the rectangle staged as 2-byte depths with the points rebuilt in registers, an eighth of the
  // staged bytes, is slower still: the conversion sits on the step's critical path.)
  // ---- the same search with the staging one step ahead (256-thread parity kernel) -------------------------------------------
  // With the 4-byte image a step's staged data is a handful of dwords per lane, so the NEXT step can be prepared -- its
  // windows, its union rectangle, its staging loads issued -- before this step is scanned: the loads' round trip (a third of a
  // step) runs underneath the scan instead of in front of it.  Two prepared-step states trade roles by unrolling the loop
  // twice (never by moves: see the chain phases).  A state issues its loads in wave-uniform pairs of passes (FL_ICP_PIPE_COND; the
  // first version always issued ICP_PIPE_NP, slots past the rectangle reading the guard behind the image, for fear that a branch
  // between issue and use would make the compiler drain the queue: it does not -- the wait in front of the first use is
  // counted for the shortest path -- and with the column-major tiles seven steps in ten need only two passes).  Same windows,
  // same candidates, same keys: bit-identical to org_search.
#ifndef FL_ICP_PIPE
#define FL_ICP_PIPE 1
#endif
#define ICP_PIPE_NP 6
#ifndef FL_ICP_PIPE_COND
#define FL_ICP_PIPE_COND 1
#endif

  // Slot s = lane + 64 p of a staged W-wide rectangle -> its row and column and the crop position base + row * cw + col, in
  // float32: every value is an integer below 2^24 (small_crop), so each product, fma and sum is exact, at full rate, where the
  // integer forms cost two quarter-rate multiplies per slot (v_mul_lo_u32, v_mad_u64_u32).
  struct SlotPos { float row, col, pos; };
  const float cwf = (float)og.cw;
  const bool small_crop = (long long)og.cw * og.ch + NN_OVERRUN < (1 << 24);
  auto slot_pos = [&](float slotf, float Wf, float invW, float basef) {
    SlotPos r;
    r.row = __builtin_truncf((slotf + 0.5f) * invW);
    r.col = __builtin_fmaf(-r.row, Wf, slotf);
    r.pos = __builtin_fmaf(r.row, cwf, r.col + basef);
    return r;
  };
  struct Prep {
    int i;                                   // the lane's query (model index) and point
    float qx, qy, qz;
    bool active, queryable;
    int u_lo, u_hi, v_lo, v_hi;              // its window (one pixel of the union if it has none)
    int U0, V0, W, maxh, nbw, npass;         // wave-uniform: union rectangle, tallest window, batches per row, passes
    bool w3;                                 // no lane window is wider than 3 pixels: batches of 3 positions
#ifdef FL_ICP_PHASES
    int wlmax;
#endif
    bool any, staged;
    float z[ICP_PIPE_NP];                    // depth factors of the staged slots lane + 64 p (in flight until finish)
  };
  auto org_search_pipe = [&](const float r_lim, auto &&found) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    float4 *stage = (float4 *)&S.prod[0][0][0] + wv * ICP_STAGE_CAP;
    const float *rimg = (const float *)sref;
    const zimg_t *zimg = (const zimg_t *)(wsb + L.zimg);
    const int last_s = n_model - 1, last_pt = og.cw * og.ch + NN_OVERRUN - 1;
    const float lanef = (float)lane;
    const float cul = uniform_f(og.offu + 0.01f), cuh = uniform_f(og.offu - 0.01f), cvl = uniform_f(og.offv + 0.01f), cvh = uniform_f(og.offv - 0.01f);
    constexpr int stride = NW * 64;
    // windows, union and staging loads of the step whose queries start at sb
    auto prepare = [&](Prep &P, int sb, int i, const F3 &q, float b) {
      P.i = i; P.qx = q.x; P.qy = q.y; P.qz = q.z;
      P.active = sb + lane < n_model;
      P.queryable = P.active && r_lim >= 0.f && isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
      int u_lo, u_hi, v_lo, v_hi;
      const bool some = org_window2_flat(og, cul, cuh, cvl, cvh, q.x, q.y, q.z, nn_radius(q.x, q.y, q.z, fminf(b, r_lim)), P.queryable,
                                         u_lo, u_hi, v_lo, v_hi);
      // union rectangle, tallest and widest lane window (a lane without a window: one pixel of the union, width 0).  A lane
      // without a window contributes the crop's far corners -- neutral among real windows, and a step in which NO lane has one
      // still gets a pixel of the crop as its "union" without a single wave-uniform select (each costs four scalar instructions)
      int red[6] = {some ? -u_lo : 1 - og.cw, some ? u_hi : 0, some ? -v_lo : 1 - og.ch, some ? v_hi : 0, some ? v_hi - v_lo + 1 : 0,
                    some ? u_hi - u_lo : 0};
      wave_max6(red);
      const int U0 = -red[0], U1 = red[1], V0 = -red[2], V1 = red[3];
      P.maxh = red[4];
      P.any = red[4] > 0;                                        // some lane has a window
      if (!some) { u_lo = u_hi = U0; v_lo = v_hi = V0; }
      P.u_lo = u_lo; P.u_hi = u_hi; P.v_lo = v_lo; P.v_hi = v_hi;
      P.nbw = (red[5] >> 2) + 1;                                 // batches of four positions per row: 1 up to width 3, 2 up to 7, ...
      P.w3 = red[5] <= 2;
#ifdef FL_ICP_PHASES
      P.wlmax = red[5];
#endif
      const int W = max(U1 - U0 + 1, 1), H = max(V1 - V0 + 1, 1), area = W * H;
      const int npneed = (area + 3 + 63) >> 6;
      P.npass = npneed <= 2 ? 2 : npneed;
      P.staged = P.any && npneed <= ICP_PIPE_NP && small_crop;
      P.U0 = U0; P.V0 = V0; P.W = W;
      const float Wf = (float)W, invW = uniform_f(__builtin_amdgcn_rcpf(Wf));
      const float basef = (float)((int)__umul24((unsigned)P.V0, (unsigned)og.cw) + P.U0);
#if FL_ICP_PIPE_COND
      // two passes always, the others in pairs where the rectangle needs them (wave-uniform: with the column-major tiles seven
      // steps in ten need two)
      auto ldz = [&](int p) {
        const SlotPos sp = slot_pos(lanef + (float)(64 * p), Wf, invW, basef);
        P.z[p] = zimg_ld(zimg, P.staged ? min((int)sp.pos, last_pt) : last_pt);
      };
      ldz(0);
      ldz(1);
      if (P.staged && P.npass > 2) { ldz(2); ldz(3); }
      if (P.staged && P.npass > 4) { ldz(4); ldz(5); }
#else
      const int np_eff = P.staged ? P.npass : 0;               // passes that hold the rectangle; the others read the guard
#pragma unroll
      for (int p = 0; p < ICP_PIPE_NP; ++p) {
        const SlotPos sp = slot_pos(lanef + (float)(64 * p), Wf, invW, basef);
        P.z[p] = zimg_ld(zimg, p < np_eff ? min((int)sp.pos, last_pt) : last_pt);
      }
#endif
    };
    // the step itself: rebuild and stage its rectangle, scan, unpack
    auto finish = [&](const Prep &P, int &j, float &d) {
      j = -1;
      d = NAN;
      if (!P.any) return;
      const float qx = P.qx, qy = P.qy, qz = P.qz;
      const int u_lo = P.u_lo, u_hi = P.u_hi, v_lo = P.v_lo, v_hi = P.v_hi, W = P.W, U0 = P.U0, V0 = P.V0, maxh = P.maxh, nbw = P.nbw;
      const int wl = u_hi - u_lo, hl = v_hi - v_lo;
#ifdef FL_ICP_PHASES
      {                                                    // dev: steps, scanned / staged positions, what the lanes' own windows hold
        int own = P.queryable ? (wl + 1) * (hl + 1) : 0;
        for (int sft = 32; sft >= 1; sft >>= 1) own += __shfl_xor(own, sft, 64);
        if (lane == 0) {
          atomicAdd((unsigned long long *)&S.tacc[8], 1ull);
          atomicAdd((unsigned long long *)&S.tacc[9], (unsigned long long)((P.w3 ? 3 : 4 * nbw) * maxh));
          atomicAdd((unsigned long long *)&S.tacc[10], P.staged ? 0ull : 1ull);
          atomicAdd((unsigned long long *)&S.tacc[11], (unsigned long long)(P.staged ? P.npass * 64 : 0));
          atomicAdd((unsigned long long *)&S.tacc[12], (unsigned long long)own);     // summed over the 64 lanes
          atomicAdd(&S.hist[16 + min(maxh, 6) - 1], 1u);   // (bins 22..24 and 26..28 carry the chain phases' stamps)
          atomicAdd(&S.hist[5 + min(nbw, 3) - 1], 1u);     // bins 5..7: 1, 2, 3+ batches per row
          atomicAdd(&S.hist[8 + min(P.npass, 4) - 2], 1u); // bins 8..10: 2, 3, 4+ staged passes
          atomicAdd(&S.stime[min(P.wlmax, 4)], 1ull);      // widest lane window of the step: 1, 2, 3, 4, 5+ pixels
        }
      }
#endif
      unsigned long long best = NN_KEY_NONE;
      if (P.staged) {
        const float Wf = (float)W, invW = uniform_f(__builtin_amdgcn_rcpf(Wf));
        const float basef = (float)((int)__umul24((unsigned)V0, (unsigned)og.cw) + U0);
        const float suf = (float)(og.sx0 + U0), svf = (float)(og.sy0 + V0);   // scene pixel of the rectangle's corner
        auto put = [&](int p) {
          const SlotPos sp = slot_pos(lanef + (float)(64 * p), Wf, invW, basef);
          const float zf = P.z[p];
          const F3 pt = org_point_f(og, suf + sp.col, svf + sp.row, zf);
          stage[lane + 64 * p] = nn_point(pt.x, pt.y, pt.z, zf != zf ? NN_IDX_NONE : min((int)sp.pos, last_pt));
        };
        put(0);
        put(1);
        if (P.npass > 2) put(2);
        if (P.npass > 3) put(3);
        if (P.npass > 4) { put(4); put(5); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 *row0 = stage + __mul24(v_lo - V0, W) + (u_lo - U0);
        if (P.w3) {                                        // (nearly half of the steps: a quarter of their scan saved)
          for (int dv = 0; dv < maxh; ++dv) {
            const float4 *bp = row0 + __mul24(min(dv, hl), W);
            float4 cur[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) cur[e] = bp[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) NN_CONSIDER(cur[e])
          }
        } else if (nbw == 1) {
          for (int dv = 0; dv < maxh; ++dv) {
            const float4 *bp = row0 + __mul24(min(dv, hl), W);
            float4 cur[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[e] = bp[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) NN_CONSIDER(cur[e])
          }
        } else {
          const int wlc = max(wl - 3, 0);
          for (int dv = 0; dv < maxh; ++dv) {
            const float4 *rowp = row0 + __mul24(min(dv, hl), W);
            for (int du = 0; du < 4 * nbw; du += 4) {
              const float4 *bp = rowp + min(du, wlc);
              float4 cur[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) cur[e] = bp[e];
#pragma unroll
              for (int e = 0; e < 4; ++e) NN_CONSIDER(cur[e])
            }
          }
        }
      } else {
        best = org_scan([&](int idx) { const F3 pt = ld3_u32(rimg, idx); return nn_point(pt.x, pt.y, pt.z, pt.x == INFINITY ? NN_IDX_NONE : idx); },
                        og.cw, 0, 0, qx, qy, qz, u_lo, u_hi, v_lo, v_hi, 4 * nbw, maxh);
      }
      if (P.queryable) NN_UNPACK(best, &j, &d)
    };
    int sb0 = wv * 64, sb1 = sb0 + stride, sb2 = sb0 + 2 * stride, sb3 = sb0 + 3 * stride;
    if (!(sb0 < n_model)) return;
    int i_c = ld_u32(perm, min(sb0 + lane, last_s));
    int i_n = ld_u32(perm, min(sb1 + lane, last_s));
    int i_nn = ld_u32(perm, min(sb2 + lane, last_s));
    F3 q_c = ld3_u32(mod, i_c), q_n = ld3_u32(mod, i_n);
    float b_c = bnd_ld(bnd, i_c), b_n = bnd_ld(bnd, i_n);
    Prep A, B;
    prepare(A, sb0, i_c, q_c, b_c);
    bool pend = false, p_active = false;
    int p_i = 0, p_j = -1;
    float p_qx = 0.f, p_qy = 0.f, p_qz = 0.f, p_d = NAN;
    // one step: the results of the step before are stored, the queries of the step after next requested, the NEXT step prepared
    // (its loads issued), then THIS step finished
#define ICP_PIPE_STEP(CUR, NXT)                                                                                        \
    {                                                                                                                  \
      if (pend) { found(p_active, p_i, p_qx, p_qy, p_qz, p_j, p_d); pend = false; }                                    \
      const F3 q_nn = ld3_u32(mod, i_nn);                                                                              \
      const float b_nn = bnd_ld(bnd, i_nn);                                                                            \
      const int i_nnn = ld_u32(perm, min(sb3 + lane, last_s));                                                         \
      prepare(NXT, sb1, i_n, q_n, b_n);                                                                                \
      int j_; float d_;                                                                                                \
      finish(CUR, j_, d_);                                                                                             \
      pend = true; p_active = CUR.active; p_i = CUR.i; p_qx = CUR.qx; p_qy = CUR.qy; p_qz = CUR.qz; p_j = j_; p_d = d_; \
      i_n = i_nn; q_n = q_nn; b_n = b_nn; i_nn = i_nnn;                                                                \
      sb0 = sb1; sb1 = sb2; sb2 = sb3; sb3 += stride;                                                                  \
    }
    for (;;) {
      ICP_PIPE_STEP(A, B)
      if (!(sb0 < n_model)) break;
      ICP_PIPE_STEP(B, A)
      if (!(sb0 < n_model)) break;
    }
#undef ICP_PIPE_STEP
    if (pend) found(p_active, p_i, p_qx, p_qy, p_qz, p_j, p_d);
  };
#define ORG_SEARCH org_search
  // the deferred dist_mean chain of the pending distances (chain wave), then -- every wave -- the search for the next iteration
  auto chain_and_search = [&](const int pend, const bool want, const float r_lim, const float old_mean_) {
    if (pend && threadIdx.x < 64) {
      const float acc = chain_deferred<SH::CHAIN_NBUF>(dterm, S.dchain, n_model);
      if (threadIdx.x == 0) {
        const int counter = S.cnt, inl = S.inl;
        float dm = acc;
        if (counter > 0) dm /= (float)inl;               // 0/0 -> NaN ends the loop (Q9)
        const float new_mean = counter > 0 ? dm : FLT_MAX;
        S.px = counter > 0 ? (float)inl / (float)counter : 0.0f;
        if (pend == 2) {
          S.dist_diff = old_mean_ - new_mean;
          float RT[3];                                   // :793-797
          mat_vec(S.Ropt, S.T, RT);
          for (int k = 0; k < 3; ++k) S.T[k] = RT[k] + S.Topt[k];
          mat_mat(S.Ropt, S.R, S.R);
        }
        S.dist_mean = new_mean;
        const bool go = (new_mean > dmt) && (S.dist_diff > ddt) && (S.iter < it_thr);       // :684, as the loop head will find it
        *(volatile int *)&S.stop = go ? 0 : 1;
      }
    }
    if (want)
      ORG_SEARCH(r_lim, true, [&](bool active, int i, float, float, float, int j, float d) {
        if (active) {
          nn[i] = j;
          nd[i] = d;
          if (j >= 0) bnd_st(bnd, i, sqrt_upper(d));             // else: the old partner is still within the old bound
        }
      });
  };


  for (;;) {
    if (SPEC) {
      const int it_done = __builtin_amdgcn_readfirstlane(S.iter);
      const bool want = !have_nn && it_done >= 1 && it_done < it_thr;   // the next iteration, if there is one, searches
      if (pending || want) {
        const float r_lim = pending ? sqrtf(3.0f * mean_ub) * 1.00001f : sqrtf(3 * S.dist_mean);
        if (threadIdx.x == 0) { S.a1_next = 0; S.stop = 0; }
        __syncthreads();
        chain_and_search(pending, want, uniform_f(r_lim), old_mean);
        __syncthreads();                                 // dist_mean, nn[], nd[], bnd[] complete
        TSTAMP(2);
        if (want) have_nn = true;
        pending = 0;
      }
    }
    if (threadIdx.x == 0) S.go = (S.dist_mean > dmt) && (S.dist_diff > ddt) && (S.iter < it_thr);   // :684
    __syncthreads();
    if (!S.go) break;
    // point-to-plane gates pairs at distance 3*dist_mean; the reference compares the SQUARED distance
    // with 3*dist_mean (Q9), which parity/fast keep
    if (threadIdx.x == 0) { ++S.iter; S.thr = plane ? (3 * S.dist_mean) * (3 * S.dist_mean) : 3 * S.dist_mean; S.a1_next = 0; }
    __syncthreads();
    const int iter = __builtin_amdgcn_readfirstlane(S.iter);
    const float thr = uniform_f(S.thr);
    const int rows = iter == 1 ? n_ref : n_model;
    constexpr bool parity = MODE == FL_ICP_PARITY;
    constexpr int mode = MODE;
    int kept = 0;
    const bool index_pairs = iter == 1 && !plane;       // :700-704
    // fast: per-thread partials.  point-to-plane: the thread's ~60 terms are summed in float32 (27 registers
    // instead of 54) and only the cross-thread tree runs in fp64; the 6x6 system is re-linearised every
    // iteration, so a 1e-6 relative error in a sum is immaterial.
    typename std::conditional<plane || FL_ICP_FAST_F32 != 0, float, double>::type ds[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) ds[k] = 0.0;
    // a kept pair (model point m, reference point r with index j) enters the sums of the modes that do not chain
    auto pair_sums = [&](float m0, float m1, float m2, float r0, float r1, float r2, int j) {
      if (plane) {
        // linearised point-to-plane: residual e = n.(m - r), Jacobian row J = [m x n, n] wrt (omega, t);
        // ds = upper triangle of sum J J^T (21) followed by sum J e (6)
        const F3 nv = ld3_u32(nrm, ORG ? ld_u32(idximg, j) : j);
        const float n0 = nv.x, n1 = nv.y, n2 = nv.z;
        const float J[6] = {m1 * n2 - m2 * n1, m2 * n0 - m0 * n2, m0 * n1 - m1 * n0, n0, n1, n2};
        const float e = n0 * (m0 - r0) + n1 * (m1 - r1) + n2 * (m2 - r2);
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) ds[q++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) ds[21 + a] += J[a] * e;
      } else if (!parity) {
        const float m[3] = {m0, m1, m2}, r[3] = {r0, r1, r2};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) ds[a * 3 + b] += (double)m[a] * (double)r[b];
#pragma unroll
        for (int q = 0; q < 3; ++q) { ds[9 + q] += (double)m[q]; ds[12 + q] += (double)r[q]; }
      }
    };
    if (!index_pairs && !SPEC) {
      // Phase A1 -- PointsCorresponding (:193-279): all waves search.  bnd[i] (distance to the partner found
      // last time plus the motion since; initially the index pair) bounds the search radius, so a converging
      // cloud visits only a handful of candidates per point; the result is still the exact 1-NN.
      const float r_thr = uniform_f(sqrtf(thr));
      if (ORG) {
        constexpr bool PIPE = FL_ICP_PIPE && FL_ICP_ZIMG && MODE == FL_ICP_PARITY && NW < 8;
        auto on_found = [&](bool active, int i, float qx, float qy, float qz, int j, float d) {
            const bool keep = d <= thr;                       // dists[i][0] <= dist_thr (:268)
            if (active) {
              nn[i] = keep ? j : -1;
              if (j >= 0) bnd_st(bnd, i, sqrt_upper(d));             // else: the old partner is still within the old bound
            }
            if (keep) {
              ++kept;
              if (!parity) { const F3 rv = ld3_u32(jsrc, j); pair_sums(qx, qy, qz, rv.x, rv.y, rv.z, j); }
            }
          };
        if (!SPEC) {
          if (PIPE) org_search_pipe(r_thr, on_found);
          else ORG_SEARCH(r_thr, false, on_found);
        }
      } else {
        const NnGrid G = nn_grid(S);
        int i = threadIdx.x;
        float qx = 0.f, qy = 0.f, qz = 0.f, qb = 0.f;
        if (i < n_model) { const F3 q3 = ld3_u32(mod, i); qx = q3.x; qy = q3.y; qz = q3.z; qb = bnd_ld(bnd, i); }
        for (; i < n_model; i += BS) {
          const int in = min(i + BS, n_model - 1);           // clamped: unused past the end
          const F3 nq3 = ld3_u32(mod, in);
          const float nqx = nq3.x, nqy = nq3.y, nqz = nq3.z, nqb = bnd_ld(bnd, in);
          int j = -1;
          float d = NAN;
          if (thr >= 0.f && isfinite(qx) && isfinite(qy) && isfinite(qz))
            nn_search_grid(G, sref, cell_start, qx, qy, qz, nn_radius(qx, qy, qz, fminf(qb, r_thr)), &j, &d);
          const bool keep = d <= thr;                       // dists[i][0] <= dist_thr (:268)
          nn[i] = keep ? j : -1;
          if (j >= 0) bnd_st(bnd, i, sqrt_upper(d));               // else: the old partner is still within qb
          if (keep) {
            ++kept;
            if (!parity) { const F3 rv = ld3_u32(ref, j); pair_sums(qx, qy, qz, rv.x, rv.y, rv.z, j); }
          }
          qx = nqx; qy = nqy; qz = nqz; qb = nqb;
        }
      }
      __syncthreads();                                   // nn[] complete
      TSTAMP(2);
    }
    // Phase A2 (iteration 1: the only phase): rows in index order.  Parity mode: the producer waves write
    // the 15 scalars of each row into double-buffered LDS tiles, wave 0 adds the previous tile
    // in row order -- getMean (:8-25) and the covariance loop (:731-735) as 15 float32 chains.
    constexpr int TQ = parity ? SH::TQ : BS;
    const int slot = parity ? S.producer_slot() : (int)threadIdx.x;
    const int clane = parity ? S.chain_lane() : -1;      // lane of this thread in the chain wave, or -1
    const int ntiles = (parity || index_pairs) ? (rows + TQ - 1) / TQ : 0;
    float acc = 0.0f;                                    // chain accumulator of lane k < 15 of wave 0
    if (parity && iter > 1) {
      // Register pipeline of the producers: a tile row costs its nn / mod loads (coalesced) and then the dependent gather
      // ref[j]; with thousands of frames in flight both come from HBM (thousands of cycles under load) while the chain
      // needs a tile every TQ * 8 cycles.  Three tiles are in flight per thread -- loaded (t + 2), gathered (t + 1), written
      // (t) -- in three register sets that trade roles by UNROLLING the tile loop three times, never by moving registers:
      // a rotation by moves reads the registers of loads that were issued in the same iteration, and the compiler answers it
      // with s_waitcnt vmcnt(0) -- every tile then waits for a full memory round trip and the prefetch is void (that was
      // the state of round 2: "1 / 2 / 3 / 4 tiles ahead measure the same").  For the same reason nothing computes with a
      // loaded value in the iteration that issued its load (the kept-pair test is applied where the row is gathered), and
      // the tile barrier is a raw s_barrier behind an lgkmcnt wait: the LDS tile must be complete, the loads in flight and
      // this phase's own global stores need not be (__syncthreads would drain them: it is a fence).
      struct Row {                                         // one tile row of this thread on its way to LDS
        int j;                                             // nearest reference index (valid from the gather on)
        float d;                                           // SPEC: its squared distance
        bool in;                                           // the row exists (below `rows`)
        F3 m, r;                                           // model point, reference point (FL_ICP_ZIMG: r.x = its depth factor until row_write)
      };
      constexpr bool ZIMG = FL_ICP_ZIMG && ORG && NW < 8;
      const zimg_t *zimg = (const zimg_t *)(wsb + L.zimg);
      const float inv_cw = uniform_f(1.0f / (float)max(og.cw, 1));
      auto row_load = [&](Row &w, int t) {                 // issue only: nothing here reads what it loads
        const int i = t * TQ + slot;
        w.in = i < rows;
        const int ic = min(i, rows - 1);                   // clamped: unused past the end
        w.j = ld_u32(nn, ic);
        w.d = SPEC ? ld_u32(nd, ic) : 0.0f;
        w.m = ld3_u32(mod, ic);
      };
      auto row_gather = [&](Row &w) {                      // SPEC: nn[] holds the neighbour whatever its distance; the pair is kept
        if (!(w.in && (!SPEC || w.d <= thr))) w.j = -1;    // if d <= dist_thr (:268; NaN = none found)
        if (ZIMG) w.r.x = zimg_ld(zimg, max(w.j, 0));      // 4 (2) bytes: the partner pixel's depth factor (rebuilt in row_write)
        else w.r = ld3_u32(jsrc, max(w.j, 0));
      };
      auto row_write = [&](const Row &w, int t) {
        const bool have = w.j >= 0;                        // dropped pairs contribute an exact +0.0f: (+0) * (+0)
        if (SPEC && have) ++kept;
        float (*tile)[SH::TS] = S.prod[t & 1];
        F3 rp = w.r;
        if (ZIMG) {                                        // the partner's point from its pixel and depth factor (org_point)
          // its crop row by a float reciprocal and one correction step instead of an integer division (three quarter-rate
          // multiplies): exact for every pixel index (the estimate is off by at most one below 2^24 pixels; beyond: divide)
          const int jj = max(w.j, 0);
          int vv, uu;
          if (small_crop) {
            vv = (int)(((float)jj + 0.5f) * inv_cw);
            uu = jj - __mul24(vv, og.cw);
            if (uu < 0) { --vv; uu += og.cw; } else if (uu >= og.cw) { ++vv; uu -= og.cw; }
          } else {
            vv = jj / og.cw;
            uu = jj - vv * og.cw;
          }
          rp = org_point(og, uu, vv, w.r.x);
        }
        const float mm[3] = {have ? w.m.x : 0.0f, have ? w.m.y : 0.0f, have ? w.m.z : 0.0f};
        const float rr[3] = {have ? rp.x : 0.0f, have ? rp.y : 0.0f, have ? rp.z : 0.0f};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) tile[a * 3 + b][slot] = mm[a] * rr[b];   // (*it_s) * (*it_ref).t()
#pragma unroll
        for (int q = 0; q < 3; ++q) { tile[9 + q][slot] = mm[q]; tile[12 + q][slot] = rr[q]; }
      };
      auto tile_barrier = [&]() {                          // LDS writes of this wave complete, then the workgroup barrier
        __builtin_amdgcn_s_waitcnt(0xC07F);                // s_waitcnt lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
      };
      const bool chain_wave = __builtin_amdgcn_readfirstlane(clane) >= 0;
      if (chain_wave) {
        // the chain wave's own loop: one barrier per tile like the producers' below
        CH_STAMP_BEGIN;
        for (int t = 0; t < ntiles; ++t) {
          if (t > 0 && clane < 15)
            acc = chain_tile<SH::CHAIN_NBUF>(S.prod[(t - 1) & 1][clane], min(TQ, rows - (t - 1) * TQ), acc);
          CH_STAMP(c_add);
          tile_barrier();
          CH_STAMP(c_bar);
        }
        CH_STAMP_END(22);
      } else if (slot >= 0) {
        Row X, Y, Z;
        row_load(X, 0);
        row_load(Y, 1);
        row_gather(X);
        for (int t = 0; t < ntiles; t += 3) {
          row_load(Z, t + 2); row_gather(Y); row_write(X, t);
          PR_STAMP_BARRIER(24, 3);
          if (t + 1 < ntiles) {
            row_load(X, t + 3); row_gather(Z); row_write(Y, t + 1);
            tile_barrier();
          }
          if (t + 2 < ntiles) {
            row_load(Y, t + 4); row_gather(X); row_write(Z, t + 2);
            tile_barrier();
          }
        }
      } else {
        for (int t = 0; t < ntiles; ++t) tile_barrier();   // a wave that neither chains nor produces (1024-thread workgroup)
      }
      __syncthreads();                                     // (the loads still in flight belong to rows past the end)
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
              r[0] = jsrc[3 * j]; r[1] = jsrc[3 * j + 1]; r[2] = jsrc[3 * j + 2];
            }
          }
        }
        if (parity) {
          // dropped pairs contribute an exact +0.0f, so the chains are branch-free
          float (*tile)[SH::TS] = S.prod[t & 1];
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
      } else if (t > 0 && clane >= 0 && clane < 15) {
        acc = chain_tile<SH::CHAIN_NBUF>(S.prod[(t - 1) & 1][clane], min(TQ, rows - (t - 1) * TQ), acc);
      }
      if (parity) __syncthreads();
    }
    if (parity && ntiles > 0 && clane >= 0 && clane < 15)
      acc = chain_tile<SH::CHAIN_NBUF>(S.prod[(ntiles - 1) & 1][clane], min(TQ, rows - (ntiles - 1) * TQ), acc);
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
      if (clane >= 0 && clane < 15) S.sums[clane] = acc;
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
          const double Kx[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
              double k2 = 0;
              for (int k = 0; k < 3; ++k) k2 += Kx[i * 3 + k] * Kx[k * 3 + j];
              S.Ropt[i * 3 + j] = (float)((i == j ? 1.0 : 0.0) + sa * Kx[i * 3 + j] + sb * k2);
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
    old_mean = S.dist_mean;
    __syncthreads();
    mean_ub = l2dist_phase<MODE, SPEC>(S, mod, ref, bnd, dterm, n_model, 3 * old_mean, Ro, To);           // :756, :778-780
    TSTAMP(5);
    if (SPEC) {                                          // the chain, dist_diff and the pose update follow at the loop head
      pending = 2;
      have_nn = false;
      continue;
    }
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
    res->R[8] = (float)S.tacc[8];                       // organised search: steps, candidates, fallbacks
    res->T[0] = (float)S.tacc[9];
    res->T[1] = (float)S.tacc[10];
    res->T[2] = (float)S.tacc[11];
    res->dist_mean = (float)S.tacc[12];
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

template <class SH>
__device__ __forceinline__ int crop_clouds(SH &S, const IcpArgs &a, const uint16_t *scene, const uint16_t *model, bool model_01mm,
                           const int *rm, const int *rr, float *ref, float *mod, float *nrm, float *rimg, int *idximg, zimg_t *zimg)
{
  constexpr int BS = SH::BS, NW = SH::NW;
  const int cw = rm[2], ch = rm[3], np = cw * ch;
  const float inv_fx = 1.0f / a.fx, inv_fy = 1.0f / a.fy;                        // depth_to_3d.cpp:103-104
  const float minv_fx = 1.0f / 608.f, minv_fy = 1.0f / 608.f;                   // initInternalMat common.cpp:358
  const float zs = (float)(1 / 1000.0);
  // Row-major compaction of the paired-valid pixels.  Per BS pixels: rank inside the wave by ballot + mbcnt, the
  // wave counts through a double-buffered LDS slot -- ONE barrier per step; every thread keeps the running total itself.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float inv_cw = 1.0f / (float)cw;                   // p / cw below: exact for p < 2^20, cw <= 2^10 (see div_small)
  int kept_before = 0, step = 0;
  __syncthreads();
  for (int base = 0; base < np; base += BS, ++step) {
    const int p = base + threadIdx.x;
    float A[3] = {0, 0, 0}, B[3] = {0, 0, 0};
    float zsf_keep = NAN;                                   // the scene pixel's depth factor (what org_point rebuilds A from)
    unsigned ds_keep = 0;
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
      zsf_keep = zsf;
      ds_keep = ds;
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
#pragma unroll
    for (int i = 0; i < NW; ++i) { const int c = slot[i]; before += i < wv ? c : 0; total += c; }
    // the reference cloud as an image (organised search): the point or a point at infinity, and its index
    if (p < np) {
      const F3 pt = {keep ? A[0] : INFINITY, keep ? A[1] : INFINITY, keep ? A[2] : INFINITY};
      __builtin_memcpy(rimg + 3 * p, &pt, 12);
      idximg[p] = keep ? kept_before + before + in_wave : NN_IDX_NONE;
      if (zimg) zimg[p] = zimg_pack(keep, ds_keep, zsf_keep);
    }
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
  if (threadIdx.x < 3 * NN_OVERRUN) rimg[3 * np + threadIdx.x] = INFINITY;     // overrun guard: points at infinity behind the image (org_scan)
  if (zimg && threadIdx.x < NN_OVERRUN + 1) zimg[np + threadIdx.x] = zimg_pack(false, 0u, NAN);
  __syncthreads();                                         // the clouds are complete for every thread
  return kept_before;
}

// The model indices in tile order (FL_ICP_TILE_W x 64 / FL_ICP_TILE_W pixels: 16 x 4) (tile rows alternately left-to-right and right-to-left, so consecutive tiles
// are neighbours; inside a tile column by column: FL_ICP_TILE_COLMAJOR): the 64 queries a wave takes per step then project into a
// compact window of the reference image.
// Index k of crop pixel p is idximg[p] (the paired compaction keeps the same pixels of both clouds).
template <class SH>
__device__ __forceinline__ void build_tile_order(SH &S, const int *idximg, int cw, int ch, int n, int *perm)
{
  constexpr int BS = SH::BS, NW = SH::NW;
  int *cnt = (int *)&S.prod[0][0][0];
  constexpr int CAP = (int)(sizeof(S.prod) / 4);
  constexpr int TW = FL_ICP_TILE_W, TH = 64 / FL_ICP_TILE_W;
  static_assert(TW * TH == 64 && (TW & (TW - 1)) == 0, "a tile is one wavefront of pixels");
  const int ntx = (cw + TW - 1) / TW, nty = (ch + TH - 1) / TH, ntile = ntx * nty;
  if (ntile > CAP) {                                       // more tiles than the scratch holds (crops beyond 1.5 Mpixel): index order
    for (int i = threadIdx.x; i < n; i += BS) perm[i] = i;
    __syncthreads();
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float inv_ntx = 1.0f / (float)ntx;
  auto tile_index = [&](int t, bool &has) {
    const int ty = (int)(((float)t + 0.5f) * inv_ntx);
    int tx = t - ty * ntx;
    if (ty & 1) tx = ntx - 1 - tx;
#if FL_ICP_TILE_COLMAJOR
    // column by column inside a tile, in the direction the tile row is walked: 64 consecutive entries of perm[] -- a search
    // step -- then always cover ONE contiguous run of columns of the tile row (row by row, a step that starts in the middle of
    // a tile takes its lower rows, the next tile and the upper rows of the one after: up to three tiles wide)
    const int c = lane / TH, x = tx * TW + ((ty & 1) ? TW - 1 - c : c), y = ty * TH + (lane & (TH - 1));
#else
    const int x = tx * TW + (lane & (TW - 1)), y = ty * TH + lane / TW;
#endif
    int k = NN_IDX_NONE;
    if (x < cw && y < ch) k = idximg[y * cw + x];
    has = (unsigned)k < (unsigned)NN_IDX_NONE;
    return k;
  };
  for (int t = wv; t < ntile; t += NW) {
    bool has;
    tile_index(t, has);
    const unsigned long long bal = __ballot(has);
    if (lane == 0) cnt[t] = __popcll(bal);
  }
  __syncthreads();
  // exclusive scan of cnt[0 .. ntile) in place: a contiguous chunk per thread, the thread totals scanned by shuffles and
  // one LDS round
  {
    const int per = (ntile + BS - 1) / BS, lo = min((int)threadIdx.x * per, ntile), hi = min(lo + per, ntile);
    int mine = 0;
    for (int t = lo; t < hi; ++t) mine += cnt[t];
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(inc, d, 64);
      if (lane >= d) inc += v;
    }
    if (lane == 63) S.iscan[wv] = inc;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) before += w < wv ? S.iscan[w] : 0;
    int ex = before + inc - mine;
    for (int t = lo; t < hi; ++t) { const int c = cnt[t]; cnt[t] = ex; ex += c; }
  }
  __syncthreads();
  for (int t = wv; t < ntile; t += NW) {
    bool has;
    const int k = tile_index(t, has);
    const unsigned long long bal = __ballot(has);
    if (has) perm[cnt[t] + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = k;
  }
  __syncthreads();
}

// ---- which wave chains (256-thread workgroups) ---------------------------------------------------------------------------
// Four (or five) of these workgroups share a CU, one wave of each on every SIMD, and each has ONE wave that spends the two
// chain phases issuing a dependent add every 8 cycles: half a SIMD's issue slots.  Were it always wave 0, the SIMD a CU's
// chain waves land on would be whatever the dispatcher's rotation made it -- two or three on one SIMD as often as not, with
// that SIMD saturated and its neighbours idle.  So a workgroup books its chain SIMD in a per-CU table (4 x 8-bit counts, keyed
// by XCC_ID and HW_ID's SE / SH / CU fields): the least booked SIMD among those its waves run on, released at the end.
// Purely a scheduling choice: which wave adds does not change what is added, or in which order.
template <class SH>
__device__ __forceinline__ void chain_elect(SH &S, unsigned *cu_chain)
{
  if (threadIdx.x == 0) { S.cw = 0; S.cu_slot = -1; S.cu_simd = 0; }
  if (SH::NW >= 8 || !FL_ICP_CHAIN_SIMD || !cu_chain) { __syncthreads(); return; }
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: SIMD_ID [5:4], CU_ID [11:8], SH_ID [12], SE_ID [15:13]
  if ((threadIdx.x & 63) == 0) S.wsimd[threadIdx.x >> 6] = (int)((hw >> 4) & 3u);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID
    const int slot = (int)(((xcc << 8) | ((hw >> 8) & 0xffu)) & (FL_CU_TABLE - 1));
    unsigned have = 0;
    for (int w = 0; w < SH::NW; ++w) have |= 1u << S.wsimd[w];
    unsigned old = atomicAdd(&cu_chain[slot], 0u), best_s = 0;
    for (;;) {
      unsigned best_c = 256;
      for (unsigned sd = 0; sd < 4; ++sd) {
        const unsigned c = (old >> (8 * sd)) & 0xffu;
        if (((have >> sd) & 1u) && c < best_c) { best_c = c; best_s = sd; }
      }
      if (best_c >= 255u) { best_s = 4; break; }                            // a stale table: do not book
      const unsigned seen = atomicCAS(&cu_chain[slot], old, old + (1u << (8 * best_s)));
      if (seen == old) break;
      old = seen;
    }
    if (best_s < 4) {
      S.cu_slot = slot;
      S.cu_simd = (int)best_s;
      for (int w = SH::NW - 1; w >= 0; --w)
        if (S.wsimd[w] == (int)best_s) S.cw = w;
    }
  }
  __syncthreads();
}
template <class SH>
__device__ __forceinline__ void chain_release(SH &S, unsigned *cu_chain)
{
  if (threadIdx.x == 0 && cu_chain && S.cu_slot >= 0) { atomicSub(&cu_chain[S.cu_slot], 1u << (8 * S.cu_simd)); S.cu_slot = -1; }
}

// waves per SIMD a kernel instance is compiled for: the 256-thread one shares a CU with up to ICP_MODE_WPE - 1 others,
// a 1024-thread workgroup is 4 waves per SIMD by itself
#define ICP_WPE(MODE, BS) ((BS) == ICP_BS_SMALL ? ICP_MODE_WPE(MODE) : (BS) / 256)

// icpCloudToCloud_Ex on clouds the host staged in the workspace (fl_icp)
template <int MODE, int BS>
__global__ __launch_bounds__(BS) void k_icp_clouds(IcpArgs a)
{
  extern __shared__ __align__(16) uint8_t icp_smem[];
  using SH = IcpSharedT<BS>;
  SH &S = *(SH *)icp_smem;
  const IcpWsLayout L = icp_layout(a.n_max);
  uint8_t *wsb = a.ws + (size_t)blockIdx.x * a.ws_stride;
  const OrgGeom none = {0, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0, 0, 0.f, 0.f, 0.f, 0.f};
  chain_elect(S, nullptr);
  icp_run<MODE, false>(S, wsb, L, a.job.n_ref, a.job.n_model, a.it_thr, a.dmt, a.ddt, &a.results[blockIdx.x].det.icp, none);
}

// WPE: waves per SIMD this instance is compiled for.  The 256-thread parity kernel exists twice, for 4 (128 VGPRs) and for 5
// (96 VGPRs, a few more spills) workgroups per CU: 5 per CU finish a full round of 5 x #CUs frames 4 % faster per frame,
// 4 per CU win when the batch is not a multiple of that (icp_small_wpe).
template <int MODE, int BS, int WPE = ICP_WPE(MODE, BS)>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(WPE, WPE > 4 ? WPE : 4)))
void k_icp_pipeline(IcpArgs a)
{
  extern __shared__ __align__(16) uint8_t icp_smem[];
  using SH = IcpSharedT<BS>;
  SH &S = *(SH *)icp_smem;
#ifdef FL_ICP_PHASES
  if (threadIdx.x == 0) { S.tkernel = clock64(); S.wall0 = wall_clock64(); }
#endif
  const int job = a.order ? a.order[blockIdx.x] : (int)blockIdx.x, rank = a.job.kind == 0 && !a.jobs ? job % a.ranks : 0;
  const int frame = a.job.kind == 0 ? (a.jobs ? a.jobs[job].frame : job / a.ranks) : job;
  const IcpWsLayout L = icp_layout(a.n_max);
  uint8_t *wsb = a.ws + (size_t)job * a.ws_stride;
  float *ref = (float *)(wsb + L.ref), *mod = (float *)(wsb + L.mod);

  fl_recognition_result *res = &a.results[job];
  if (a.job.kind == 0 && a.jobs && frame < 0) return;    // fl_refine_selected: this frame's winner belongs to another rank (uniform per workgroup)
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
      res->n_matches = a.jobs ? 1 : counters[1];
      res->found = 0;
      res->status = FL_OK;
      if (!a.jobs && counters[2]) S.status = res->status = FL_ERR_OVERFLOW;
      else if (a.jobs || counters[1] > rank) {
        const fl_match best = a.jobs ? a.jobs[job].match : matches[rank];   // matches[0] :111 (rank > 0: multi-hypothesis extension)
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
  chain_elect(S, a.cu_chain);                            // (behind the last early return: what is booked here is released below)
  float *rimg = (float *)(wsb + L.sref);                  // image of 12-byte points, then the image of their indices (16 bytes per pixel in all)
  const int np = crop_clouds(S, a, scene, model, model_01mm, S.rect_m, S.rect_r, ref, mod,
                             MODE == FL_ICP_POINT_TO_PLANE ? (float *)(wsb + L.nrm) : nullptr, rimg,
                             (int *)((uint8_t *)rimg + org_idximg_offset(S.rect_m[2] * S.rect_m[3])),
                             FL_ICP_ZIMG && MODE == FL_ICP_PARITY && BS < ICP_BS_WIDE ? (zimg_t *)(wsb + L.zimg) : nullptr);
  // wave-uniform values read from LDS are VGPRs unless told otherwise: the search loop keeps them in SGPRs
  const OrgGeom og = {__builtin_amdgcn_readfirstlane(S.rect_r[2]), __builtin_amdgcn_readfirstlane(S.rect_r[3]),
                      uniform_f((float)S.rect_r[0] - a.cx), uniform_f((float)S.rect_r[1] - a.cy), a.fx, a.fy,
                      uniform_f((float)(S.rect_r[2] - 1)), uniform_f((float)(S.rect_r[3] - 1)),
                      __builtin_amdgcn_readfirstlane(S.rect_r[0]), __builtin_amdgcn_readfirstlane(S.rect_r[1]), a.cx, a.cy,
                      uniform_f(1.0f / a.fx), uniform_f(1.0f / a.fy)};
  build_tile_order(S, (const int *)((const uint8_t *)rimg + org_idximg_offset(og.cw * og.ch)), og.cw, og.ch, np, (int *)(wsb + L.perm));
  // getMean x2 (detection.cpp:165-166), t_match_tmp = r - m (:177), t_init (:199)
  float mc[3] = {0, 0, 0}, rc[3] = {0, 0, 0};
  if (MODE == FL_ICP_PARITY) {
    // getMean x2 as six float32 chains in index order, fed like phase A2: the producer waves load TQ rows per tile
    // (coalesced 12-byte loads) into the LDS tiles, lanes 0..5 of wave 0 add the previous tile
    {
      constexpr int TQ = SH::TQ;
      const int slot = S.producer_slot(), clane = S.chain_lane(), ntiles = (np + TQ - 1) / TQ;
      float acc = 0.0f;
      for (int t = 0; t < ntiles; ++t) {
        if (slot >= 0) {
          const int i = t * TQ + slot;
          F3 m3 = {0.f, 0.f, 0.f}, r3 = {0.f, 0.f, 0.f};
          if (i < np) { m3 = ld3_u32(mod, i); r3 = ld3_u32(ref, i); }
          float (*tile)[SH::TS] = S.prod[t & 1];
          tile[0][slot] = m3.x; tile[1][slot] = m3.y; tile[2][slot] = m3.z;
          tile[3][slot] = r3.x; tile[4][slot] = r3.y; tile[5][slot] = r3.z;
        } else if (t > 0 && clane >= 0 && clane < 6) {
          acc = chain_tile<SH::CHAIN_NBUF>(S.prod[(t - 1) & 1][clane], min(TQ, np - (t - 1) * TQ), acc);
        }
        __syncthreads();
      }
      if (ntiles > 0 && clane >= 0 && clane < 6)
        acc = chain_tile<SH::CHAIN_NBUF>(S.prod[(ntiles - 1) & 1][clane], min(TQ, np - (ntiles - 1) * TQ), acc);
      if (clane >= 0 && clane < 6) S.sums[clane] = acc;
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) { mc[k] = S.sums[k]; rc[k] = S.sums[3 + k]; }
    if (np > 0)
      for (int k = 0; k < 3; ++k) { mc[k] /= (float)np; rc[k] /= (float)np; }
  } else {
    double ds[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < np; i += BS)
      for (int k = 0; k < 3; ++k) { ds[k] += mod[3 * i + k]; ds[3 + k] += ref[3 * i + k]; }
    block_sum_double<6>(S, ds);
    if (np > 0)
      for (int k = 0; k < 3; ++k) { mc[k] = (float)(S.dsum[0][k] / np); rc[k] = (float)(S.dsum[0][3 + k] / np); }
  }
  float t_tmp[3], t_init[3];
  for (int k = 0; k < 3; ++k) { t_tmp[k] = rc[k] - mc[k]; t_init[k] = t_tmp[k] + t_match[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < np; i += BS) {           // transformPoints(pts_mod, I, t_match_tmp) :206
    if (!vvalid(mod[3 * i + 2])) continue;
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float v[3] = {mod[3 * i], mod[3 * i + 1], mod[3 * i + 2]}, o[3];
    mat_vec(I, v, o);
    mod[3 * i] = o[0] + t_tmp[0];
    mod[3 * i + 1] = o[1] + t_tmp[1];
    mod[3 * i + 2] = o[2] + t_tmp[2];
  }
  __syncthreads();
  icp_run<MODE, true>(S, wsb, L, np, np, a.it_thr, a.dmt, a.ddt, &res->det.icp, og);      // :228
  chain_release(S, a.cu_chain);
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
#ifdef FL_ICP_PHASES
    // dev build only: histograms of the organised search instead of the pose
    for (int i = 0; i < 16; ++i) res->pose[i] = (float)S.hist[i];
    for (int i = 0; i < 9; ++i) res->det.R_final[i] = (float)S.hist[16 + i];
    for (int i = 0; i < 3; ++i) res->det.T_final[i] = (float)S.hist[26 + i];
    for (int i = 0; i < 5; ++i) res->pose[11 + i] = (float)S.stime[i];   // the last five union classes (wider than 29 pixels) make room
    // start / end of this workgroup on the 100 MHz wall clock (two 24-bit halves each: a float holds 24 bits), and its CU
    {
      const long long w1 = wall_clock64();
      res->pose[0] = (float)((S.wall0 >> 24) & 0xffffff); res->pose[1] = (float)(S.wall0 & 0xffffff);
      res->pose[2] = (float)((w1 >> 24) & 0xffffff); res->pose[3] = (float)(w1 & 0xffffff);
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      res->pose[4] = (float)(hwid & 0xffffff);
    }
#endif
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

// ---- longest job first ------------------------------------------------------------------------------------------------
// A batch of a few thousand frames is two or three rounds of workgroups on the chip's slots, and a frame's refinement takes
// time in proportion to its cloud (13 k - 17 k points on the bench scenes: lifetimes of 6.4 - 10.9 ms in one launch).  The
// hardware deals workgroups to slots as they free up, in blockIdx order, so with jobs in frame order the last ones to start
// are average jobs behind the slowest first-round ones, and a launch of 2048 frames took 20.0 ms for 2 x 8.5 ms of mean
// work per slot (85 % of the slots' time used; profiles/README.md).  Dealt longest first, the long jobs run in the first
// round and the short ones fill in behind them.  k_icp_count estimates a job's size as the number of pixels of its crop
// that are valid in both depth images (what crop_clouds will keep); k_icp_order sorts the jobs by it, descending.
__global__ __launch_bounds__(256) void k_icp_count(IcpArgs a, int *__restrict__ size)
{
  __shared__ int part[4];
  const int job = blockIdx.x;
  const uint8_t *fws = a.frame_ws + (size_t)job * a.frame_stride;
  const int *counters = (const int *)(fws + a.off_count);
  const fl_match *matches = (const fl_match *)(fws + a.off_match);
  int n = 0;
  if (!counters[2] && counters[1] > 0) {
    const fl_match best = matches[0];
    const int g = a.class_first[best.class_idx] + best.template_id;
    const FlPyrInfo pi = a.pyr[g];
    const uint16_t *model = a.depth_ptrs[g];
    const uint16_t *scene = (const uint16_t *)((const uint8_t *)a.scene_base + (size_t)job * a.scene_stride);
    const int cw = pi.width0, ch = pi.height0;
    const bool ok = model && pi.off_x0 >= 0 && pi.off_y0 >= 0 && best.x >= 0 && best.y >= 0 && cw > 0 && ch > 0 && pi.off_x0 + cw <= a.w &&
                    pi.off_y0 + ch <= a.h && best.x + cw <= a.w && best.y + ch <= a.h;
    if (ok)
      for (int y = threadIdx.x >> 6; y < ch; y += 4)
        for (int x = threadIdx.x & 63; x < cw; x += 64) {
          const unsigned ds = scene[(size_t)(best.y + y) * a.w + best.x + x], dm = model[(size_t)(pi.off_y0 + y) * a.w + pi.off_x0 + x];
          n += (ds != 0 && ds <= 900 && dm != 0 && dm <= 9004) ? 1 : 0;      // z <= 900 mm in both (the render is in 0.1 mm): an estimate
        }
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) n += __shfl_xor(n, sft, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) size[job] = part[0] + part[1] + part[2] + part[3];
}

// order[0 .. n) = the jobs by size, descending (ties by job index): one workgroup, bitonic sort of 64-bit keys in LDS
#define ICP_ORDER_MAX 8192
__global__ __launch_bounds__(1024) void k_icp_order(const int *__restrict__ size, int n, int *__restrict__ order)
{
  extern __shared__ unsigned long long okeys[];
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = threadIdx.x; i < m; i += 1024)
    okeys[i] = i < n ? ((unsigned long long)(unsigned)(0x7fffffff - size[i]) << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = okeys[i], y = okeys[l];
          if (((i & k) == 0) == (x > y)) { okeys[i] = y; okeys[l] = x; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < n; i += 1024) order[i] = (int)(unsigned)(okeys[i] & 0xffffffffull);
}

#define ICP_WPE_MAX 5
static_assert(sizeof(IcpSharedT<ICP_BS_SMALL>) + 16 <= 160 * 1024 / ICP_WPE_MAX, "IcpSharedT<256> must leave room for 5 workgroups per CU");
static_assert(sizeof(IcpSharedT<ICP_BS_WIDE>) + 16 <= 160 * 1024, "IcpSharedT<1024> must fit the CU's LDS");
template <int BS, typename K>
static int icp_launch_one(fl_context *ctx, K kern, int n_jobs, const IcpArgs &a)
{
#ifdef FL_ICP_LDS_PAD                                      // dev builds: extra dynamic LDS per workgroup = fewer workgroups per CU (occupancy experiment)
  size_t lds = ((sizeof(IcpSharedT<BS>) + 15) & ~(size_t)15) + (BS == ICP_BS_SMALL ? (size_t)(FL_ICP_LDS_PAD) : 0);
#else
  size_t lds = (sizeof(IcpSharedT<BS>) + 15) & ~(size_t)15;
#endif
  // option icp_wg_per_cu = 1 .. 3: no more than that many 256-thread workgroups per CU (the launch asks for so much LDS that
  // one more does not fit) -- leaves registers and LDS to the kernels of ANOTHER stream (a second pipeline's LINEMOD stages)
  const long cap = ctx->opt.icp_wg_per_cu;
  if (BS == ICP_BS_SMALL && cap >= 1 && cap <= 3) {
    const size_t want = ((size_t)(160 * 1024) / (size_t)cap - 256) & ~(size_t)15;
    if (want > lds) lds = want;
  }
  const size_t attr = lds > (size_t)(82 * 1024) ? lds : (size_t)(82 * 1024);
  {                                                        // the attribute is set once per kernel and device, not per launch
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (!done.count({(const void *)kern, ctx->device})) {
      FL_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(attr > (size_t)(160 * 1024) ? (size_t)(160 * 1024) : attr)));
      done.insert({(const void *)kern, ctx->device});
    }
  }
  hipLaunchKernelGGL(kern, dim3(n_jobs), dim3(BS), lds, ctx->stream, a);
  FL_HIP(ctx, hipGetLastError());
  return FL_OK;
}
// Workgroup width by batch size: with no more jobs than CUs every job runs alone on its CU whatever its width, so it
// gets the 1024-thread kernel; two rounds of it still beat two 256-thread workgroups per CU (measured, ICP ms per launch,
// 1024- vs 256-thread: 256 jobs 3.6 / 6.8, 384: 6.4 / 7.2, 512: 6.8 / 7.7, 768: 10.0 / 8.9, 1024: 13.2 / 11.1).
// option icp_wide = 0 / 1 forces one or the other (fl_context_set_option)
static bool icp_wide(fl_context *ctx, int n_jobs)
{
  if (ctx->opt.icp_wide == 0) return false;
  if (ctx->opt.icp_wide == 1) return true;
  return n_jobs <= 2 * ctx->cus;
}
// Workgroups per CU of the 256-thread parity kernel: whichever of 4 and 5 needs fewer rounds by the measured cost of a full
// round (5 per CU: about 1.25 x the time of 4 per CU for 1.25 x the frames -- the 96-VGPR build spills more; round 3, jobs
// dealt longest first, ICP ms per launch at 4 / 5 per CU: 2560 frames 24.9 / 24.0, 3840: 33.4 / 33.2, 4096: 35.0 / 36.5,
// 5120: 46.7 / 47.7, 6144: 56.0 / 58.5).  Option icp_occ = 4 / 5 forces one (fl_context_set_option).
static int icp_small_wpe(fl_context *ctx, int n_jobs)
{
  if (ctx->opt.icp_occ == 4 || ctx->opt.icp_occ == 5) return (int)ctx->opt.icp_occ;
  const int cus = ctx->cus;
  const int r4 = (n_jobs + 4 * cus - 1) / (4 * cus), r5 = (n_jobs + 5 * cus - 1) / (5 * cus);
  return 1.25 * r5 < 1.0 * r4 ? 5 : 4;
}
template <int MODE>
static int icp_launch_mode(fl_context *ctx, int n_jobs, const IcpArgs &a)
{
  const bool wide = icp_wide(ctx, n_jobs);
  if (MODE == FL_ICP_PARITY && a.job.kind != 2 && !wide && icp_small_wpe(ctx, n_jobs) == 5)
    return icp_launch_one<ICP_BS_SMALL>(ctx, k_icp_pipeline<MODE, ICP_BS_SMALL, 5>, n_jobs, a);
  if (a.job.kind == 2)
    return wide ? icp_launch_one<ICP_BS_WIDE>(ctx, k_icp_clouds<MODE, ICP_BS_WIDE>, n_jobs, a)
                : icp_launch_one<ICP_BS_SMALL>(ctx, k_icp_clouds<MODE, ICP_BS_SMALL>, n_jobs, a);
  return wide ? icp_launch_one<ICP_BS_WIDE>(ctx, k_icp_pipeline<MODE, ICP_BS_WIDE>, n_jobs, a)
              : icp_launch_one<ICP_BS_SMALL>(ctx, k_icp_pipeline<MODE, ICP_BS_SMALL>, n_jobs, a);
}
static int icp_launch(fl_context *ctx, int n_jobs, const IcpArgs &a)
{
  switch (a.mode) {
    case FL_ICP_PARITY: return icp_launch_mode<FL_ICP_PARITY>(ctx, n_jobs, a);
    case FL_ICP_FAST: return icp_launch_mode<FL_ICP_FAST>(ctx, n_jobs, a);
    case FL_ICP_POINT_TO_PLANE: return icp_launch_mode<FL_ICP_POINT_TO_PLANE>(ctx, n_jobs, a);
    default: return fl_set_error(ctx, FL_ERR_INVALID, "unknown fl_icp_mode");
  }
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
  a.cu_chain = ctx->d_cu_chain;
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
  a.cu_chain = ctx->d_cu_chain;
  return icp_launch(ctx, n_frames * k, a);
}

// refinement of caller-chosen matches (template-sharded recognition: the rank that owns the winning template refines it):
// job b = (frame, match) read from the device array `jobs`, ICP workspace and result slot b
int fl_launch_detection_jobs(fl_detector *det, int n_jobs, const FlRefineJob *d_jobs, const fl_intrinsics *K, const fl_recognition_params *p,
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
  a.jobs = d_jobs;
  a.cu_chain = ctx->d_cu_chain;
  return icp_launch(ctx, n_jobs, a);
}

// once per detector (fl_detector_finalize): the job-order buffer and k_icp_order's LDS size, so that no launch path allocates
// (an implicit device synchronisation) or sets function attributes
int fl_icp_prepare(fl_detector *det)
{
  fl_context *ctx = det->ctx;
  if (det->max_batch > 4 * ctx->cus && !det->d_icp_order)      // (used for batches of up to ICP_ORDER_MAX frames)
    FL_HIP(ctx, hipMalloc((void **)&det->d_icp_order, sizeof(int) * 2 * (size_t)det->max_batch));
  FL_HIP(ctx, hipFuncSetAttribute((const void *)k_icp_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned long long) * ICP_ORDER_MAX)));
  return FL_OK;
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
  a.cu_chain = ctx->d_cu_chain;
  // more jobs than the chip has slots: deal them longest first (see k_icp_order); option icp_order = 0 keeps the frame order
  if (n_frames > 4 * ctx->cus && n_frames <= ICP_ORDER_MAX && det->d_icp_order && ctx->opt.icp_order != 0) {
    int *d_size = det->d_icp_order + det->max_batch;
    hipLaunchKernelGGL(k_icp_count, dim3(n_frames), dim3(256), 0, ctx->stream, a, d_size);
    int m = 1;
    while (m < n_frames) m <<= 1;
    hipLaunchKernelGGL(k_icp_order, dim3(1), dim3(1024), sizeof(unsigned long long) * (size_t)m, ctx->stream, (const int *)d_size, n_frames,
                       det->d_icp_order);
    FL_HIP(ctx, hipGetLastError());
    a.order = det->d_icp_order;
  }
  return icp_launch(ctx, n_frames, a);
}

