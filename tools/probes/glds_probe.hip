// dev only: where does a sub-dword LDS-DMA (global_load_lds_ushort) put each lane's element, and does its M0 base reach
// beyond 64 KB of the 160 KB LDS?  (gfx950)  Answer (MI355X): lane k's element lands zero-extended in dword k of the
// destination (base + 4 k); see the printed second block for the high base.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint16_t *src, uint32_t *out, unsigned word0)
{
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 128; i += 64) lds[word0 + i] = 0xdeadbeefu;
  __syncthreads();
  const uint16_t *p = src + 3 * threadIdx.x;            // lane k fetches element 3k (value 3k + 1000)
  const unsigned dst = (unsigned)(uintptr_t)&lds[word0 + 16];
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 128; i += 64) out[i] = lds[word0 + i];
  if (threadIdx.x == 0) out[128] = dst;
}
int main()
{
  uint16_t h[256]; for (int i = 0; i < 256; ++i) h[i] = (uint16_t)(i + 1000);
  uint16_t *d; uint32_t *o; uint32_t ho[129];
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(ho));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (unsigned word0 : {0u, 30000u}) {                  // byte offsets 0 and 120000
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 150 * 1024, 0, d, o, word0);
    (void)hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    printf("word0 %u, dst byte address 0x%x\n", word0, ho[128]);
    for (int i = 8; i < 40; ++i) printf("%08x%s", ho[i], (i % 8) == 7 ? "\n" : " ");
  }
  return 0;
}
