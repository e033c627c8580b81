// dev probe: which SIMD does wave w of a 256-thread workgroup land on, and how do 4 co-resident workgroups of a CU line up?
// hipcc --offload-arch=gfx950 -O2 tools/probes/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__global__ __launch_bounds__(256) void k_probe(unsigned *out, int spin)
{
  extern __shared__ char lds[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID[3:0]
  long long t0 = clock64();
  float a = threadIdx.x;
  while (clock64() - t0 < spin) a = a * 1.0001f + 1.0f;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc | (a == 123.0f ? 1u << 31 : 0);
  }
  if (a == 77.0f) lds[threadIdx.x] = 1;
}
int main()
{
  const int nb = 2048;
  unsigned *d, *h = (unsigned *)malloc(nb * 8 * 4);
  hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 38 * 1024);
  hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 38 * 1024, 0, d, 2000000);
  hipDeviceSynchronize();
  hipMemcpy(h, d, nb * 8 * 4, hipMemcpyDeviceToHost);
  int hist[4][4];
  memset(hist, 0, sizeof hist);
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 4; ++w) hist[w][(h[(b * 4 + w) * 2] >> 4) & 3]++;
  for (int w = 0; w < 4; ++w) printf("wave %d on simd 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 12; ++b) {
    const unsigned hw = h[b * 8], x = h[b * 8 + 1] & 15;
    printf("wg %d: xcc %u se %u sh %u cu %u | simd of waves:", b, x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
    for (int w = 0; w < 4; ++w) printf(" %u(slot %u)", (h[(b * 4 + w) * 2] >> 4) & 3, h[(b * 4 + w) * 2] & 15);
    printf("\n");
  }
  // co-resident workgroups: same (xcc, se, sh, cu) among the first 1024
  int same = 0;
  for (int b = 0; b < 1024; ++b) {
    const unsigned key = (h[b * 8] & 0xff00) | ((h[b * 8 + 1] & 15) << 16);
    int n = 0;
    for (int c = 0; c < 1024; ++c) n += ((h[c * 8] & 0xff00) | ((h[c * 8 + 1] & 15) << 16)) == key;
    same += n;
    if (b < 4) { printf("wg %d shares its CU with:", b); for (int c = 0; c < 1024; ++c) if (((h[c * 8] & 0xff00) | ((h[c * 8 + 1] & 15) << 16)) == key) printf(" %d", c); printf("\n"); }
  }
  printf("mean workgroups per CU key among the first 1024: %.2f\n", same / 1024.0);
  return 0;
}
