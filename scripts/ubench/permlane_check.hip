// Check (on the GPU) that two v_permlane32_swap_b32 exchange a register pair with lane ^ 32, against __shfl_xor.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float2* a, float2* o, float2* ref) {
  int i = threadIdx.x;
  float2 v = a[i];
  float x = v.x * 1.0f, y = v.y * 1.0f;
  // after the pair: y = partner's x, x = partner's y
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\ts_nop 1" : "+v"(x), "+v"(y));
  o[i] = make_float2(y + 0.0f, x + 0.0f);
  ref[i] = make_float2(__shfl_xor(v.x, 32), __shfl_xor(v.y, 32));
}
int main() {
  float2 h[128], r[128], e[128];
  for (int i = 0; i < 128; ++i) h[i] = make_float2(i, 1000 + i);
  float2 *a, *o, *f;
  hipMalloc(&a, sizeof(h)); hipMalloc(&o, sizeof(h)); hipMalloc(&f, sizeof(h));
  hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, a, o, f);
  hipMemcpy(r, o, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(e, f, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) {
    if (r[i].x != e[i].x || r[i].y != e[i].y) {
      if (bad < 6) std::printf("lane %d: got (%g, %g) want (%g, %g)\n", i, r[i].x, r[i].y, e[i].x, e[i].y);
      ++bad;
    }
  }
  std::printf(bad ? "MISMATCH %d\n" : "OK\n", bad);
  return bad != 0;
}
