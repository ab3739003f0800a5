// What does timing a kernel cost on the stream, and do events attached to the dispatch itself (hipExtLaunchKernelGGL)
// read the same duration as a hipEventRecord pair around it?
//   build: hipcc --offload-arch=gfx950 -O2 -o ext_launch_events ext_launch_events.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("%s -> %s\n", #x, hipGetErrorString(e_));                    \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned *sink) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned acc = 0;
  while (__builtin_readcyclecounter() - t0 < ticks) {
    acc += 1;
  }
  if (acc == 0xffffffffu) {
    *sink = acc;
  }
}
__global__ void tiny_kernel(unsigned *sink) {
  if (threadIdx.x == 1023) {
    *sink = 1;
  }
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned *sink;
  CK(hipMalloc(&sink, 4));
  const int n = 400;
  std::vector<hipEvent_t> a(n), b(n);
  for (int i = 0; i < n; ++i) {
    CK(hipEventCreate(&a[i]));
    CK(hipEventCreate(&b[i]));
  }
  const unsigned long long ticks = 250000;  // ~ the headline kernel
  for (int mode = 0; mode < 4; ++mode) {
    // 0 plain; 1 hipEventRecord pair; 2 events on the dispatch; 3 plain again
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipStreamSynchronize(st));
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; ++i) {
        if (mode == 1) {
          CK(hipEventRecord(a[i], st));
        }
        if (mode == 2) {
          hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, st, a[i], b[i], 0, ticks, sink);
        } else {
          hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, st, ticks, sink);
        }
        if (mode == 1) {
          CK(hipEventRecord(b[i], st));
        }
        hipLaunchKernelGGL(tiny_kernel, dim3(8), dim3(256), 0, st, sink);  // the history carry's place
      }
      CK(hipStreamSynchronize(st));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
      best = us < best ? us : best;
    }
    double sum = 0, mn = 1e9, mx = 0;
    if (mode == 1 || mode == 2) {
      for (int i = 0; i < n; ++i) {
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a[i], b[i]));
        sum += ms;
        mn = ms < mn ? ms : mn;
        mx = ms > mx ? ms : mx;
      }
    }
    std::printf("mode %d  wall per step %8.2f us   event-measured kernel avg %8.2f min %8.2f max %8.2f us\n", mode, best,
                sum / n * 1e3, mn * 1e3, mx * 1e3);
  }
  return 0;
}
