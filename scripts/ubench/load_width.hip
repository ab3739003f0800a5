// L2 -> CU load rate by access width: every thread sums NLOAD values read lane-contiguously from a small (L2-resident) or a
// large (HBM) buffer, as 8-byte or 16-byte loads per lane. Question behind it (profiles/r03_q_two_level.txt): is the inverse
// row kernel's ~12 TB/s of eight-byte L2 loads the vector memory path's limit, and would 16-byte loads double it?
//   hipcc --offload-arch=gfx950 -O3 -o load_width load_width.hip
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename V, int NLOAD>
__global__ __launch_bounds__(256) void stream_kernel(const V *__restrict__ buf, size_t words, float *out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  float acc = 0.0f;
#pragma unroll 16
  for (int k = 0; k < NLOAD; ++k) {
    const V v = buf[i % words];
    acc += v.x + v.y;
    if constexpr (sizeof(V) == 16) {
      acc += v.z + v.w;
    }
    i += stride;
  }
  if (acc == 12345.678f) {
    out[0] = acc;
  }
}

template <typename V, int NLOAD>
double run(const void *buf, size_t bytes, float *out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const size_t words = bytes / sizeof(V);
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL((stream_kernel<V, NLOAD>), dim3(grid), dim3(256), 0, 0, static_cast<const V *>(buf), words, out);
  }
  hipEventRecord(a, 0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL((stream_kernel<V, NLOAD>), dim3(grid), dim3(256), 0, 0, static_cast<const V *>(buf), words, out);
  }
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double total = static_cast<double>(grid) * 256 * NLOAD * sizeof(V) * reps;
  return total / (ms * 1e-3) / 1e12;
}

int main() {
  void *small, *large;
  float *out;
  const size_t smallBytes = 2u << 20, largeBytes = 1u << 30;
  hipMalloc(&small, smallBytes);
  hipMalloc(&large, largeBytes);
  hipMalloc(&out, 4);
  hipMemset(small, 0, smallBytes);
  hipMemset(large, 0, largeBytes);
  const int grid = 256 * 16;  // 16 workgroups of 256 threads per CU queued
  std::printf("TB/s       8-byte loads   16-byte loads   (x loads per thread)\n");
  std::printf("L2-resident (2 MB)   %6.2f x80     %6.2f x40\n", run<float2, 80>(small, smallBytes, out, grid),
              run<float4, 40>(small, smallBytes, out, grid));
  std::printf("L2-resident (2 MB)   %6.2f x160    %6.2f x80\n", run<float2, 160>(small, smallBytes, out, grid),
              run<float4, 80>(small, smallBytes, out, grid));
  std::printf("HBM (1 GB)           %6.2f x80     %6.2f x40\n", run<float2, 80>(large, largeBytes, out, grid),
              run<float4, 40>(large, largeBytes, out, grid));
  return 0;
}
