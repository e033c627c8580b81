// dev probe: what does ONE dependent float32 add of a sequential chain cost on gfx950 -- alone on its SIMD, next to 1..3
// other waves of the same SIMD that issue independent VALU work, and with s_setprio raised on the chain wave?
// (the ICP kernel's chain phases run 12 - 13 cycles per add with 4 waves per SIMD; this says how much of that is the chain's
// own latency and how much is the issue slots the neighbours take)
// hipcc --offload-arch=gfx950 -O2 tools/probes/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// one 1024-thread workgroup = 16 waves, 4 per SIMD.  Wave 0 chains; of the other waves on ITS SIMD the first `hogs` spin on
// independent (hog_kind 0) or LDS-reading (1) work until the chain is done; every other wave waits at the end.
__global__ __launch_bounds__(1024) void k_chain(unsigned long long *out, int n16, int hogs, int hog_kind, int prio, int with_lds)
{
  __shared__ float s_x[1024];
  __shared__ volatile int s_done;
  __shared__ int s_simd0, s_rank[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned simd = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3;     // HW_REG_HW_ID.SIMD_ID
  s_x[threadIdx.x] = 1.0f + threadIdx.x * 1e-7f;
  if (threadIdx.x == 0) { s_done = 0; s_simd0 = (int)simd; }
  __syncthreads();
  if (lane == 0) s_rank[wave] = (int)simd == s_simd0 ? 1 : 0;
  __syncthreads();
  int rank = 0;                                          // how many waves before this one share wave 0's SIMD
  for (int w = 1; w < wave; ++w) rank += s_rank[w];
  const bool on0 = wave > 0 && (int)simd == s_simd0;
  if (wave == 0) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    float acc = 0.0f;
    const float x = s_x[lane];
    const long long t0 = clock64();
    for (int i = 0; i < n16; ++i) {
      if (with_lds) {
        // the ICP chain's shape: four b128 LDS reads, sixteen dependent adds
        const float4 a = *(const float4 *)&s_x[(i * 16) & 1008], b = *(const float4 *)&s_x[((i * 16) & 1008) + 4],
                     c = *(const float4 *)&s_x[((i * 16) & 1008) + 8], d = *(const float4 *)&s_x[((i * 16) & 1008) + 12];
        acc += a.x; acc += a.y; acc += a.z; acc += a.w; acc += b.x; acc += b.y; acc += b.z; acc += b.w;
        acc += c.x; acc += c.y; acc += c.z; acc += c.w; acc += d.x; acc += d.y; acc += d.z; acc += d.w;
      } else {
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     : "+v"(acc) : "v"(x));
      }
    }
    const long long t1 = clock64();
    if (prio) __builtin_amdgcn_s_setprio(0);
    if (lane == 0) { out[0] = (unsigned long long)(t1 - t0); out[1] = (unsigned long long)__float_as_uint(acc); s_done = 1; }
  } else if (on0 && rank < hogs) {
    float a0 = lane, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    unsigned long long iters = 0;
    while (!s_done) {
      for (int k = 0; k < 32; ++k) {
        if (hog_kind == 0) {
          a0 = a0 * 1.0001f + 1.0f; a1 = a1 * 1.0001f + 1.0f; a2 = a2 * 1.0001f + 1.0f; a3 = a3 * 1.0001f + 1.0f;
          a4 = a4 * 1.0001f + 1.0f; a5 = a5 * 1.0001f + 1.0f; a6 = a6 * 1.0001f + 1.0f; a7 = a7 * 1.0001f + 1.0f;
        } else {
          a0 += s_x[(lane * 17 + k * 64 + (int)a1) & 1023]; a1 += 1.0f;
        }
      }
      ++iters;
    }
    if (lane == 0) out[2 + rank] = iters * 32 * (hog_kind == 0 ? 8 : 2);
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[15] = 1;
  }
}

int main()
{
  unsigned long long *d, h[16];
  hipMalloc(&d, sizeof h);
  const int n16 = 1 << 14;
  printf("%-8s %-5s %-9s %-5s %-5s %12s %14s\n", "chain", "hogs", "hog kind", "prio", "", "cycles/add", "hog VALU/add");
  for (int with_lds = 0; with_lds < 2; ++with_lds)
    for (int hog_kind = 0; hog_kind < 2; ++hog_kind)
      for (int hogs = 0; hogs <= 3; ++hogs)
        for (int prio = 0; prio < 2; ++prio) {
          if (hogs == 0 && (hog_kind || prio)) continue;
          hipMemset(d, 0, sizeof h);
          for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_chain, dim3(1), dim3(1024), 0, 0, d, n16, hogs, hog_kind, prio, with_lds);
          hipDeviceSynchronize();
          hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
          const double adds = 16.0 * n16;
          printf("%-8s %-5d %-9s %-5d %-5s %12.2f %14.2f\n", with_lds ? "lds+add" : "add", hogs, hog_kind ? "lds" : "valu", prio, "",
                 h[0] / adds, (double)(h[2] + h[3] + h[4]) / adds);
        }
  return 0;
}
