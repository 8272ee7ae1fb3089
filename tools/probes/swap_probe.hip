// swap_probe.hip — what v_permlane16_swap_b32 does to a wave64 on gfx950: a = lane, b = 100 + lane before; both printed after
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/swap_probe.hip -o tools/probes/swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc((void**)&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("first  (was lane):      "); for (int i = 0; i < 64; ++i) printf("%u ", h[i]); printf("\n");
  printf("second (was 100+lane):  "); for (int i = 0; i < 64; ++i) printf("%u ", h[64 + i]); printf("\n");
  return 0;
}
