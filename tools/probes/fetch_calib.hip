// dev probe: how does rocprofv3's FETCH_SIZE count the access widths the ICP kernel uses?  Each kernel reads a buffer far
// larger than the Infinity Cache exactly once; compare the counter (KiB) with the bytes printed here.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fetch_calib -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct F3 { float x, y, z; };

__global__ __launch_bounds__(256) void k_read_b32(const float *p, size_t n, float *out)
{
  float acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_b96(const float *p, size_t n3, float *out)
{
  float acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n3; i += (size_t)gridDim.x * 256) {
    F3 v;
    __builtin_memcpy(&v, p + 3 * i, 12);
    acc += v.x + v.y + v.z;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_b128(const f32x4 *p, size_t n4, float *out)
{
  float acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
// the search's staging shape: a wave fetches 4 rows of 16 consecutive float4 of a 160-wide image (256 B runs, 2560 B apart)
__global__ __launch_bounds__(256) void k_read_rows16(const f32x4 *p, size_t n4, float *out)
{
  float acc = 0;
  const int lane = threadIdx.x & 63;
  const size_t waves = (size_t)gridDim.x * 4, wave = blockIdx.x * 4ull + (threadIdx.x >> 6);
  // wave w, step s reads the 16-wide column block (s * waves + w): rows r..r+3 of a 160-wide image tile
  const size_t blocks = n4 / 25600 * 400;                 // whole 160 x 160 images only (a partial last image would run past the buffer)
  for (size_t b = wave; b < blocks; b += waves) {
    const size_t img = b / (10 * 40), rem = b % (10 * 40), cb = rem % 10, rb = rem / 10;     // 160 x 160 image = 10 x 40 blocks of 16 x 4
    const size_t idx = img * 25600 + (rb * 4 + (lane >> 4)) * 160 + cb * 16 + (lane & 15);
    const f32x4 v = p[idx];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
// phase A2's gather: 12-byte points at an index that wanders around i (neighbouring lanes mostly neighbouring points)
__global__ __launch_bounds__(256) void k_gather_b96(const float *p, size_t n3, float *out)
{
  float acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n3; i += (size_t)gridDim.x * 256) {
    const unsigned h = (unsigned)i * 2654435761u;
    size_t j = i + (h >> 29) * 160 + ((h >> 26) & 7);      // up to 7 rows of 160 and 7 points away
    if (j >= n3) j = i;
    F3 v;
    __builtin_memcpy(&v, p + 3 * j, 12);
    acc += v.x + v.y + v.z;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write_b96(float *p, size_t n3)
{
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n3; i += (size_t)gridDim.x * 256) {
    const F3 v = {1.f, 2.f, 3.f};
    __builtin_memcpy(p + 3 * i, &v, 12);
  }
}

int main()
{
  const size_t bytes = 6ull << 30;                          // 6 GiB, divisible by 12 and 16
  float *buf, *out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 64);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  const int grid = 256 * 16;
  hipLaunchKernelGGL(k_read_b32, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, out);
  hipLaunchKernelGGL(k_read_b96, dim3(grid), dim3(256), 0, 0, buf, bytes / 12, out);
  hipLaunchKernelGGL(k_read_b128, dim3(grid), dim3(256), 0, 0, (const f32x4 *)buf, bytes / 16, out);
  hipLaunchKernelGGL(k_read_rows16, dim3(grid), dim3(256), 0, 0, (const f32x4 *)buf, bytes / 16, out);
  hipLaunchKernelGGL(k_gather_b96, dim3(grid), dim3(256), 0, 0, buf, bytes / 12, out);
  hipLaunchKernelGGL(k_write_b96, dim3(grid), dim3(256), 0, 0, buf, bytes / 12);
  hipDeviceSynchronize();
  printf("every kernel touches %.0f KiB once (k_gather_b96: the same count of 12-byte reads at wandering addresses)\n", bytes / 1024.0);
  return 0;
}
