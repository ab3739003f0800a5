// LDS data-path rates on gfx950 by access width: ds_read/ds_write b32, b64, b128 with every lane on its own consecutive
// address (conflict free), 1 / 2 / 4 waves per SIMD. Prints s_memtime ticks per wave-instruction per CU and bytes per tick
// (calibration: a v_add_f32 wave-instruction at 4 waves per SIMD costs 1.72 ticks, profiles/r02_a_*).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_rates.hip -o gpurun_out/lds_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int BYTES, int WRITE>
__global__ void lds_kernel(float *out, unsigned long long *cycles, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < blockDim.x * 8 * (BYTES / 4); i += blockDim.x) {
    lds[i] = static_cast<float>(i);
  }
  __syncthreads();
  // 8 slots of blockDim.x * BYTES bytes; lane address = slot * blockDim.x * BYTES + tid * BYTES
  const unsigned base = static_cast<unsigned>(tid) * BYTES;
  const unsigned step = blockDim.x * BYTES;
  f4v acc = {static_cast<float>(tid), 1.0f, 2.0f, 3.0f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned a = base + step * (j & 7);
      if constexpr (WRITE) {
        if constexpr (BYTES == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(acc.x) : "memory");
        if constexpr (BYTES == 8) {
          const f2v v = {acc.x, acc.y};
          asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
        }
        if constexpr (BYTES == 16) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(acc) : "memory");
      } else {
        if constexpr (BYTES == 4) {
          float v;
          asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory");
          asm volatile("" ::"v"(v));
        }
        if constexpr (BYTES == 8) {
          f2v v;
          asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
          asm volatile("" ::"v"(v));
        }
        if constexpr (BYTES == 16) {
          f4v v;
          asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
          asm volatile("" ::"v"(v));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = acc.x + lds[tid];
  if ((tid & 63) == 0) {
    cycles[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
  }
}

template <typename K>
double run(K kernel, int threads, size_t lds, int iters) {
  const int grid = 256;
  float *out;
  unsigned long long *cyc;
  hipMalloc(&out, sizeof(float) * grid * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * grid * (threads / 64));
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, out, cyc, iters);
  }
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * (threads / 64));
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += static_cast<double>(v);
  hipFree(out);
  hipFree(cyc);
  const double per_wave = sum / h.size();  // ticks one wave spent in the loop = ticks the CU spent on all waves' accesses
  return per_wave / (static_cast<double>(iters) * 16 * (threads / 64));  // per wave-instruction, CU level
}

template <int BYTES>
void report(int threads) {
  const size_t lds = static_cast<size_t>(threads) * 8 * BYTES;
  const double r = run(lds_kernel<BYTES, 0>, threads, lds, 500), w = run(lds_kernel<BYTES, 1>, threads, lds, 500);
  std::printf("   b%-3d  read %6.2f ticks/wave-instr = %6.1f B/tick    write %6.2f ticks/wave-instr = %6.1f B/tick\n", BYTES * 8,
              r, 64.0 * BYTES / r, w, 64.0 * BYTES / w);
}

int main() {
  std::printf("LDS rates per CU (s_memtime ticks; 1 workgroup per CU, 256 CUs; 1 v_add_f32 wave-instr @4 waves/SIMD = 1.72)\n");
  for (int threads : {256, 512, 1024}) {
    std::printf("-- %d threads per workgroup = %d wave(s) per SIMD\n", threads, threads / 256);
    report<4>(threads);
    report<8>(threads);
    report<16>(threads);
  }
  return 0;
}
