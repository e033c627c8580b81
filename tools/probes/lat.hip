// dev microbenchmark: latency of one coalesced dword load per wave, fresh lines at a given stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_lat(const float *p, size_t stride_f, int iters, size_t span_f, long long *out, float *sink)
{
  size_t off = ((size_t)blockIdx.x * 1315423911u) % span_f;
  off = (off / 64) * 64 + threadIdx.x;
  long long tot = 0;
  float acc = 0;
  for (int i = 0; i < iters; ++i) {
    long long t0 = clock64();
    float v = p[off];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += v;
    long long t1 = clock64();
    tot += t1 - t0;
    off += stride_f;
    if (off >= span_f) off -= span_f;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = tot / iters;
  if (acc == 12345.f) sink[0] = acc;
}
int main(int argc, char **argv)
{
  size_t bytes = (size_t)1200 << 20;
  float *p; long long *out; float *sink;
  hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
  hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 4);
  long long h[4096];
  size_t strides[] = {256, 4096, 65536, 1 << 20, (2 << 20) + 256, (size_t)(16 << 20) + 256};
  int blocks[] = {1, 256, 1024, 4096};
  for (int b : blocks)
    for (size_t st : strides) {
      hipLaunchKernelGGL(k_lat, dim3(b), dim3(64), 0, 0, p, st / 4, 200, bytes / 4, out, sink);
      hipDeviceSynchronize();
      hipMemcpy(h, out, 8 * b, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < b; ++i) s += h[i];
      printf("blocks %4d stride %9zu B : avg latency %.0f cycles\n", b, st, s / b);
    }
  return 0;
}
