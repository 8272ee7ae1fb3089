#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void mf(float* out) {
  const int l = threadIdx.x;
  bf8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * (l >> 5) + j;
    a[j] = (__bf16)(float)((l & 31) + 1) * (k == 3 || k == 12 ? 1.f : 0.f) ;     // A[m][k] = (m+1) at k in {3,12}
    b[j] = (__bf16)(float)(k == 3 ? 1.f : (k == 12 ? 64.f : 0.f)) * (float)1.f;  // B[k][n]: 1 at k=3, 64 at k=12 ... times 1
    if (k == 3 || k == 12) b[j] = (__bf16)((k == 3 ? 1.f : 64.f));
  }
  // make B depend on n too: scale by (n odd ? 2 : 1)
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)((float)b[j] * ((l & 1) ? 2.f : 1.f));
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
__global__ void swp(int* out) {
  const int l = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(l, 1000 + l, false, false);
  out[l * 2] = r[0]; out[l * 2 + 1] = r[1];
}
int main() {
  { float* d; hipMalloc(&d, 64 * 16 * 4); float h[1024];
    hipLaunchKernelGGL(mf, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int n = l & 31, m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const float want = (m + 1) * 65.f * ((n & 1) ? 2.f : 1.f);
      if (h[l * 16 + r] != want) { if (bad < 8) printf("mfma32 mismatch l=%d r=%d got %g want %g\n", l, r, h[l * 16 + r], want); ++bad; }
    }
    printf("mfma 32x32x16 layout check: %d mismatches\n", bad);
    int* di; hipMalloc(&di, 512); int hi[128];
    hipLaunchKernelGGL(swp, dim3(1), dim3(64), 0, 0, di); hipMemcpy(hi, di, 512, hipMemcpyDeviceToHost);
    printf("permlane32_swap(l, 1000+l): lane0 -> (%d,%d) lane5 -> (%d,%d) lane32 -> (%d,%d) lane37 -> (%d,%d)\n", hi[0], hi[1], hi[10], hi[11], hi[64], hi[65], hi[74], hi[75]);
  }
  int h[64]; short o[256];
  int* d; short* dout;
  hipMalloc(&d, 256); hipMalloc(&dout, 512);
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15;
      if (variant == 0) h[l] = g * 64 + i * 4;                       // contiguous 128 B per group
      if (variant == 1) h[l] = g * 1024 + (i >> 2) * 128 + (i & 3) * 4;   // 4 rows stride 128 elems, 16 cols
      if (variant == 2) h[l] = g * 16 + (i >> 2) * 256 + (i & 3) * 4;    // groups adjacent in columns
    }
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) { printf("L%02d a=%4d:", l, h[l]); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); printf(l % 4 == 3 ? "\n" : "   "); }
  }
  return 0;
}
