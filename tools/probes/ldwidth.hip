// dev microbenchmark: cost of wave-wide global loads by width, per-lane stride and alignment (L2-resident buffer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
template <int BYTES>
__global__ __launch_bounds__(256) void k_ld(const unsigned char *p, int lane_stride, int offset, int iters, size_t span, unsigned *sink)
{
  const size_t wave = (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6));
  size_t base = (wave * 4096) % span + (size_t)(threadIdx.x & 63) * lane_stride + offset;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    if (BYTES == 16) { uint4 v; __builtin_memcpy(&v, p + base, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    else if (BYTES == 12) { unsigned v[3]; __builtin_memcpy(v, p + base, 12); acc += v[0] ^ v[1] ^ v[2]; }
    else if (BYTES == 8) { uint2 v; __builtin_memcpy(&v, p + base, 8); acc += v.x ^ v.y; }
    else if (BYTES == 4) { unsigned v; __builtin_memcpy(&v, p + base, 4); acc += v; }
    else if (BYTES == 2) { unsigned short v; __builtin_memcpy(&v, p + base, 2); acc += v; }
    else { acc += p[base]; }
    base += 8192;
    if (base + 4096 >= span) base -= span - 8192;
  }
  if (acc == 0x12345) sink[0] = acc;
}
template <int BYTES>
static void run(const unsigned char *p, int stride, int off, size_t span, unsigned *sink)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 8, iters = 400;
  hipLaunchKernelGGL(k_ld<BYTES>, dim3(blocks), dim3(256), 0, 0, p, stride, off, 10, span, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_ld<BYTES>, dim3(blocks), dim3(256), 0, 0, p, stride, off, iters, span, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instrs = (double)blocks * 4 * iters;     // wave-instructions
  printf("width %2d B  lane stride %3d B  offset %d : %.2f ns per wave-instruction per CU  (%.0f GB/s requested)\n", BYTES, stride, off,
         ms * 1e6 / (instrs / 256), instrs * 64 * BYTES / (ms * 1e-3) / 1e9);
}
int main()
{
  const size_t span = 24u << 20;                          // 24 MB: mostly L2/MALL resident
  unsigned char *p; unsigned *sink;
  hipMalloc(&p, span + (1 << 20)); hipMemset(p, 1, span + (1 << 20)); hipMalloc(&sink, 4);
  run<16>(p, 16, 0, span, sink); run<16>(p, 16, 4, span, sink); run<16>(p, 16, 1, span, sink);
  run<16>(p, 6, 0, span, sink); run<16>(p, 24, 0, span, sink); run<16>(p, 24, 8, span, sink); run<16>(p, 5, 0, span, sink);
  run<12>(p, 12, 0, span, sink);
  run<8>(p, 8, 0, span, sink); run<8>(p, 6, 0, span, sink);
  run<4>(p, 4, 0, span, sink); run<4>(p, 6, 0, span, sink); run<4>(p, 12, 0, span, sink); run<4>(p, 64, 0, span, sink);
  run<2>(p, 2, 0, span, sink); run<2>(p, 6, 0, span, sink);
  run<1>(p, 1, 0, span, sink); run<1>(p, 3, 0, span, sink); run<1>(p, 5, 0, span, sink); run<1>(p, 64, 0, span, sink);
  return 0;
}
