// mx_probe.hip — what v_mfma_scale_f32_32x32x64_f8f6f4 does with its operands on gfx950 (one wave, crafted inputs):
//   exp 0  all ones, unit scales                        -> 64 everywhere if fp8 decode / accumulation are as assumed
//   exp 1  A half-0 lanes = 1, half-1 lanes = 0         -> 32 everywhere if a lane's 32 bytes pair with the same lane-half of B
//   exp 2  exp 1 + scale of A lane (i0 = 5, h = 0) x 2  -> row 5 = 64 if a lane's scale covers exactly its own 32 bytes
//   exp 3  exp 0 + A lane (5, 1) scale x 2              -> row 5 = 96
//   exp 4  A[i][slot s] = (s == s0), B[j][slot s] = s+1 -> D = s0 + 1: slot s of A pairs with slot s of B (within a half)
//   exp 5  scale word bytes (127, 128, 129, 130), op_sel 0..3 -> which byte each op_sel form picks
//   exp 6  same for the B-side scale
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mx_probe.hip -o tools/probes/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int OPA, int OPB>
__global__ void k(const v8i* a, const v8i* b, const unsigned* sa, const unsigned* sb, float* d) {
  const int l = threadIdx.x;
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, OPA, (int)sa[l], OPB, (int)sb[l]);
  // C/D: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5); operands swapped in our kernels is irrelevant here: row index = A row
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// the asm form the GEMM kernel uses
template <int HI>
__global__ void kasm(const v8i* a, const v8i* b, const unsigned* sa, const unsigned* sb, float* d) {
  const int l = threadIdx.x;
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  v8i av = a[l], bv = b[l];
  unsigned s0 = sa[l], s1 = sb[l];
  if (HI == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(av), "v"(bv), "v"(s0), "v"(s1));
  else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(av), "v"(bv), "v"(s0), "v"(s1));
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

static unsigned char A[64][32], B[64][32];
static unsigned SA[64], SB[64];
static float D[32 * 32];
static void *da, *db, *dsa, *dsb, *dd;
const unsigned char ONE = 0x38;      // e4m3 1.0
static unsigned char f8(int v) {     // small integers 0..8 in e4m3
  const unsigned char t[] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return t[v];
}
template <typename F> static void run(F launch) {
  hipMemcpy(da, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(db, B, sizeof(B), hipMemcpyHostToDevice);
  hipMemcpy(dsa, SA, sizeof(SA), hipMemcpyHostToDevice); hipMemcpy(dsb, SB, sizeof(SB), hipMemcpyHostToDevice);
  launch();
  hipDeviceSynchronize();
  hipMemcpy(D, dd, sizeof(D), hipMemcpyDeviceToHost);
}
static void show(const char* name) {
  printf("%s: D[0][0..3] = %g %g %g %g | D[5][0..3] = %g %g %g %g | D[5][31] = %g D[31][5] = %g\n", name, D[0], D[1], D[2], D[3], D[5 * 32], D[5 * 32 + 1], D[5 * 32 + 2],
         D[5 * 32 + 3], D[5 * 32 + 31], D[31 * 32 + 5]);
}
#define GO(OPA, OPB) [&] { hipLaunchKernelGGL((k<OPA, OPB>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const unsigned*)dsa, (const unsigned*)dsb, (float*)dd); }
int main() {
  hipMalloc(&da, sizeof(A)); hipMalloc(&db, sizeof(B)); hipMalloc(&dsa, sizeof(SA)); hipMalloc(&dsb, sizeof(SB)); hipMalloc(&dd, sizeof(D));
  auto ones = [&] { memset(A, ONE, sizeof(A)); memset(B, ONE, sizeof(B)); for (int l = 0; l < 64; ++l) SA[l] = SB[l] = 127u * 0x01010101u; };
  ones(); run(GO(0, 0)); show("exp0 all ones (want 64)");
  ones(); for (int l = 32; l < 64; ++l) memset(A[l], 0, 32); run(GO(0, 0)); show("exp1 A half-1 zero (want 32)");
  SA[5] = 128u * 0x01010101u; run(GO(0, 0)); show("exp2 exp1 + scale(A lane 5) x2 (want row 5 = 64)");
  ones(); SA[32 + 5] = 128u * 0x01010101u; run(GO(0, 0)); show("exp3 ones + scale(A lane 37) x2 (want row 5 = 96)");
  for (int s0 = 0; s0 < 32; s0 += 9) {
    ones();
    for (int l = 0; l < 64; ++l) for (int s = 0; s < 32; ++s) { A[l][s] = (l < 32 && s == s0) ? ONE : 0; B[l][s] = f8(s % 8 + 1); }
    run(GO(0, 0));
    char nm[96]; snprintf(nm, sizeof nm, "exp4 A one-hot slot %d in half 0, B slot s = s%%8+1 (want %d)", s0, s0 % 8 + 1); show(nm);
  }
  { ones(); for (int l = 0; l < 64; ++l) for (int s = 0; s < 32; ++s) { A[l][s] = (l >= 32 && s == 3) ? ONE : 0; B[l][s] = (l >= 32) ? f8(s % 8 + 1) : f8(8); }
    run(GO(0, 0)); show("exp4b A one-hot slot 3 in half 1, B half 1 slot s = s%8+1, half 0 = 8 (want 4)"); }
  ones(); for (int l = 0; l < 64; ++l) SA[l] = 127u | (128u << 8) | (129u << 16) | (130u << 24);
  run(GO(0, 0)); show("exp5 A scale bytes 127..130, op_sel 0 (want 64)");
  run(GO(1, 0)); show("exp5 op_sel 1 (want 128)");
  run(GO(2, 0)); show("exp5 op_sel 2 (want 256)");
  run(GO(3, 0)); show("exp5 op_sel 3 (want 512)");
  ones(); for (int l = 0; l < 64; ++l) SB[l] = 127u | (128u << 8) | (129u << 16) | (130u << 24);
  run(GO(0, 2)); show("exp6 B scale bytes, op_sel_b 2 (want 256)");
  ones(); for (int l = 0; l < 64; ++l) SA[l] = SB[l] = 127u | (128u << 8) | (129u << 16) | (130u << 24);
  run([&] { hipLaunchKernelGGL((kasm<0>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const unsigned*)dsa, (const unsigned*)dsb, (float*)dd); }); show("asm op_sel_hi:[0,0,0] (want 64)");
  run([&] { hipLaunchKernelGGL((kasm<1>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const unsigned*)dsa, (const unsigned*)dsb, (float*)dd); }); show("asm op_sel_hi:[1,1,0] (want 64 * 4 * 4 = 1024)");
  // scale in a NON-uniform pattern per lane, shifted words as the kernel does: lane l word = (127 + (l & 1)) in byte 0
  ones(); for (int l = 0; l < 64; ++l) SA[l] = (127u + (unsigned)(l & 1)) | 0x7f7f7f00u; run(GO(0, 0));
  printf("exp7 A scale alternates by lane: D[0][0] = %g (want 64) D[1][0] = %g (want 128) D[2][0] = %g D[3][0] = %g\n", D[0], D[32], D[64], D[96]);
  return 0;
}
