// Micro-benchmark (measurement tool, not product): issue rate of scalar vs packed fp32 VALU
// instructions on gfx950 at 1, 2 and 4 waves per SIMD, to decide whether hand-placed
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 butterflies can beat v_add_f32 / v_fma_f32 ones
// in the FFT passes (VERDICT r01 "next round" 2.iii). Also ds_read_b64 / ds_write_b64 rates.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o scripts/ubench/valu_rates
//   ./valu_rates        -> one line per (instruction, waves/SIMD): cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum Op { ADD, FMA, MUL, PKADD, PKFMA, PKMUL, MIX_ADD_PKADD };

template <int OP>
__global__ void valu_kernel(float *out, unsigned long long *cycles, int iters) {
  float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.0f, a5 = 5.0f, a6 = 6.0f, a7 = 7.0f;
  float b0 = 0.5f, b1 = 1.5f, b2 = 2.5f, b3 = 3.5f, b4 = 4.5f, b5 = 5.5f, b6 = 6.5f, b7 = 7.5f;
  const float c = 1.0000001f, d = 1e-9f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
  const f2 pc = {c, c}, pd = {d, d};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == ADD) {
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(d));)
    } else if constexpr (OP == FMA) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(c), "v"(d));)
    } else if constexpr (OP == MUL) {
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(c));)
    } else if constexpr (OP == PKADD) {
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                        : "v"(pd));)
    } else if constexpr (OP == PKFMA) {
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                        : "v"(pc), "v"(pd));)
    } else if constexpr (OP == PKMUL) {
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                        : "v"(pc));)
    } else {
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_pk_add_f32 %4, %4, %9\n v_add_f32 %1, %1, %8\n v_pk_add_f32 %5, %5, %9\n"
                        "v_add_f32 %2, %2, %8\n v_pk_add_f32 %6, %6, %9\n v_add_f32 %3, %3, %8\n v_pk_add_f32 %7, %7, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                        : "v"(d), "v"(pd));)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
            p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  }
}

// LDS: every lane reads / writes 8 bytes at consecutive addresses (conflict free), 64 per iteration
template <int WRITE>
__global__ void lds_kernel(float *out, unsigned long long *cycles, int iters) {
  extern __shared__ float2 lds[];
  const int tid = threadIdx.x;
  lds[tid] = make_float2(tid, 1.0f);
  __syncthreads();
  float2 acc = make_float2(0.0f, 0.0f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int idx = (tid + j * 64) & (blockDim.x * 8 - 1);
      if constexpr (WRITE) {
        lds[idx] = make_float2(acc.x + j, acc.y);
      } else {
        const float2 v = lds[idx];
        acc.x += v.x;
        acc.y += v.y;
      }
    }
    if constexpr (WRITE) {
      acc.x += 1.0f;
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = acc.x + acc.y + lds[tid].x;
  if ((tid & 63) == 0) {
    cycles[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
  }
}

template <typename K>
double run(K kernel, int threads, size_t lds, int iters, int per_iter) {
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
  const double per_wave = sum / h.size();                       // cycles one wave spent in the loop
  const double waves_per_simd = threads / 64 / 4.0;
  // cycles of SIMD time per wave-instruction = elapsed / (instructions issued on that SIMD)
  return per_wave / (static_cast<double>(iters) * per_iter * (waves_per_simd < 1 ? 1 : waves_per_simd));
}

int main() {
  const int iters = 2000;
  const char *names[] = {"v_add_f32", "v_fma_f32", "v_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                         "v_add_f32+v_pk_add_f32 alternating"};
  std::printf("cycles of SIMD time per wave-instruction (s_memtime ticks; 1 workgroup per CU, 256 CUs)\n");
  for (int threads : {256, 512, 1024}) {
    std::printf("-- %d threads per workgroup = %d wave(s) per SIMD\n", threads, threads / 256);
    double r[7];
    r[0] = run(valu_kernel<ADD>, threads, 0, iters, 64);
    r[1] = run(valu_kernel<FMA>, threads, 0, iters, 64);
    r[2] = run(valu_kernel<MUL>, threads, 0, iters, 64);
    r[3] = run(valu_kernel<PKADD>, threads, 0, iters, 64);
    r[4] = run(valu_kernel<PKFMA>, threads, 0, iters, 64);
    r[5] = run(valu_kernel<PKMUL>, threads, 0, iters, 64);
    r[6] = run(valu_kernel<MIX_ADD_PKADD>, threads, 0, iters, 64);
    for (int i = 0; i < 7; ++i) {
      std::printf("   %-38s %6.2f\n", names[i], r[i]);
    }
    const size_t lds = static_cast<size_t>(threads) * 8 * 8;
    std::printf("   %-38s %6.2f  (per CU: x/4 per 512 B)\n", "ds_read_b64", run(lds_kernel<0>, threads, lds, 200, 64));
    std::printf("   %-38s %6.2f\n", "ds_write_b64", run(lds_kernel<1>, threads, lds, 200, 64));
  }
  return 0;
}
