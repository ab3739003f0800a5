// Check (on the GPU) what cooperative frames relies on for SPEED (never for correctness): that hardware register XCC_ID
// (s_getreg_b32 hwreg 20, bits 3:0) names the XCD a workgroup runs on, and that workgroups are dealt round-robin over
// the XCDs by their linear id (id % 8). Also: an agent-scope atomic counter incremented by every workgroup adds up
// (atomics of ONE XCD on one line are coherent; of several XCDs on one line they may not be -- printed, not required).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *xcc, unsigned *count) {
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;
    __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  const int n = 4096;
  unsigned *d, *c, h[n], hc = 0;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&c, 4);
  hipMemset(c, 0, 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d, c);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost);
  int hist[16] = {0}, match = 0;
  for (int i = 0; i < n; ++i) {
    hist[h[i] & 15]++;
    match += (h[i] == static_cast<unsigned>(i % 8));
  }
  std::printf("XCC_ID histogram over %d workgroups:", n);
  for (int i = 0; i < 16; ++i) std::printf(" %d", hist[i]);
  std::printf("\nworkgroups whose XCC_ID == id %% 8: %d of %d\n", match, n);
  std::printf("agent-scope atomic counter across XCDs: %u of %d\n", hc, n);
  return 0;
}
