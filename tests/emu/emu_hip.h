// emu_hip.h — TEST INFRASTRUCTURE ONLY.
//
// A tiny SIMT simulator that lets the gfx950 kernel sources under mangatranslator_amd/csrc be
// compiled for the host (clang++ -x c++ -DMTX_EMU) and executed on the CPU: every HIP thread is
// a fiber (own stack, a hand-written register switch on x86-64 — glibc's swapcontext costs two
// signal-mask system calls per switch — ucontext elsewhere), a workgroup is a round-robin scheduler over its fibers, __syncthreads() and
// the wave-level primitives (MFMA, shuffles) are rendezvous points.  It exists so the index
// arithmetic of the kernels (LDS swizzles, halo tiles, MFMA fragment maps, epilogues) can be
// checked in the CPU-only test tier (`pytest -m "not gpu"`), where no MI355X is present.
//
// It is NOT a fallback: the product loader (mangatranslator_amd/hip/lib.py) only ever opens
// libmtx_hip.so and raises ModelError when that is missing; the simulator library
// (tests/emu/libmtx_emu.so) is opened by tests alone, through an explicit test-only entry point.
#pragma once
#if !defined(__x86_64__)
#include <ucontext.h>
#endif
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <atomic>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

namespace emu {

constexpr int kWave = 64;
constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 96 * 1024;

#if defined(__x86_64__)
struct Ctx { void* sp = nullptr; };                 // everything else of a suspended fiber lives on its stack
extern "C" void emu_switch(Ctx* from, Ctx* to);     // emu_hip.cpp: callee-saved registers + control words, no signal mask
inline void ctx_switch(Ctx& from, Ctx& to) { emu_switch(&from, &to); }
#else
typedef ucontext_t Ctx;
inline void ctx_switch(Ctx& from, Ctx& to) { swapcontext(&from, &to); }
#endif

struct Fiber {
  Ctx ctx;
  dim3 tid;
  int lin = 0;
  bool done = false;
  char* stack = nullptr;
};

struct WaveState {
  int count = 0;
  unsigned gen = 0;
  alignas(16) unsigned char xa[kWave][64];   // per-lane exchange slot A (up to 64 bytes)
  alignas(16) unsigned char xb[kWave][64];   // per-lane exchange slot B
};

struct Block {
  dim3 bid, bdim, gdim;
  int nthreads = 0;
  int cur = 0;
  int bar_count = 0;
  unsigned bar_gen = 0;
  Ctx main_ctx;
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  char* dyn_smem = nullptr;
  const std::function<void()>* body = nullptr;
};

extern thread_local Block* B;

inline void yield() { ctx_switch(B->fibers[B->cur].ctx, B->main_ctx); }

inline void syncthreads() {
  unsigned g = B->bar_gen;
  if (++B->bar_count == B->nthreads) { B->bar_count = 0; B->bar_gen++; }
  else while (B->bar_gen == g) yield();
}

inline int lane_id() { return B->fibers[B->cur].lin & (kWave - 1); }
inline WaveState& wave() { return B->waves[B->fibers[B->cur].lin / kWave]; }
inline int wave_width() {
  int w = B->fibers[B->cur].lin / kWave;
  int rem = B->nthreads - w * kWave;
  return rem < kWave ? rem : kWave;
}
inline void wave_sync() {
  WaveState& w = wave();
  unsigned g = w.gen;
  if (++w.count == wave_width()) { w.count = 0; w.gen++; }
  else while (w.gen == g) yield();
}

void fiber_entry();
void run_block(Block& blk);
void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body);

}  // namespace emu

#define threadIdx (emu::B->fibers[emu::B->cur].tid)
#define blockIdx (emu::B->bid)
#define blockDim (emu::B->bdim)
#define gridDim (emu::B->gdim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__
inline void __syncthreads() { emu::syncthreads(); }

// ---- vector types (clang ext vectors work on the host too) --------------------------------
typedef __attribute__((ext_vector_type(4))) float emu_f32x4;

// ---- wave primitives -----------------------------------------------------------------------
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  memcpy(w.xa[l], &v, sizeof(T));
  emu::wave_sync();
  T r;
  memcpy(&r, w.xa[(l ^ mask) & 63], sizeof(T));
  emu::wave_sync();
  return r;
}
template <typename T>
inline T __shfl_down(T v, int delta, int width = 64) {
  (void)width;
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  memcpy(w.xa[l], &v, sizeof(T));
  emu::wave_sync();
  T r;
  int src = l + delta;
  if (src > 63) src = l;
  memcpy(&r, w.xa[src], sizeof(T));
  emu::wave_sync();
  return r;
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  memcpy(w.xa[l], &v, sizeof(T));
  emu::wave_sync();
  T r;
  memcpy(&r, w.xa[src & 63], sizeof(T));
  emu::wave_sync();
  return r;
}

// D = A(16x32) * B(32x16) + C ; lane l holds A[l&15][8*(l>>4)..+8], B[8*(l>>4)..+8][l&15],
// C/D: col = l&15, row = 4*(l>>4) + r   (cdna_hip_programming.md §3 fragment layout)
template <typename V8>
inline emu_f32x4 emu_mfma_16x16x32(V8 a, V8 b, emu_f32x4 c) {
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
  memcpy(w.xa[l], fa, 32);
  memcpy(w.xb[l], fb, 32);
  emu::wave_sync();
  emu_f32x4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    float s = d[r];
    for (int kb = 0; kb < 4; ++kb) {
      const float* pa = reinterpret_cast<const float*>(w.xa[kb * 16 + row]);
      const float* pb = reinterpret_cast<const float*>(w.xb[kb * 16 + col]);
      for (int i = 0; i < 8; ++i) s += pa[i] * pb[i];
    }
    d[r] = s;
  }
  emu::wave_sync();
  return d;
}
// D = A(32x16) * B(16x32) + C ; lane l holds A[l&31][8*(l>>5)..+8], B[8*(l>>5)..+8][l&31],
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)   (checked on gfx950 with tools/probes/tr_probe.hip)
typedef __attribute__((ext_vector_type(16))) float emu_f32x16;
template <typename V8>
inline emu_f32x16 emu_mfma_32x32x16(V8 a, V8 b, emu_f32x16 c) {
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
  memcpy(w.xa[l], fa, 32);
  memcpy(w.xb[l], fb, 32);
  emu::wave_sync();
  emu_f32x16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float s = d[r];
    for (int hb = 0; hb < 2; ++hb) {
      const float* pa = reinterpret_cast<const float*>(w.xa[hb * 32 + row]);
      const float* pb = reinterpret_cast<const float*>(w.xb[hb * 32 + col]);
      for (int i = 0; i < 8; ++i) s += pa[i] * pb[i];
    }
    d[r] = s;
  }
  emu::wave_sync();
  return d;
}
// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, fp32 throughout; lane l holds A[l & 31][l >> 5] and B[l >> 5][l & 31]; C / D as the
// other 32x32 forms (column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5))
inline emu_f32x16 emu_mfma_32x32x2_f32(float a, float b, emu_f32x16 c) {
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  memcpy(w.xa[l], &a, 4);
  memcpy(w.xb[l], &b, 4);
  emu::wave_sync();
  emu_f32x16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float s = d[r];
    for (int k = 0; k < 2; ++k) {
      float pa, pb;
      memcpy(&pa, w.xa[k * 32 + row], 4);
      memcpy(&pb, w.xb[k * 32 + col], 4);
      s = fmaf(pa, pb, s);
    }
    d[r] = s;
  }
  emu::wave_sync();
  return d;
}
// ---- OCP fp8 e4m3fn (bias 7, no inf, 0x7f = NaN, max 448) --------------------------------------------------
inline float emu_e4m3_to_f32(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return s ? -v : v;
}
inline unsigned char emu_f32_to_e4m3(float f) {          // round to nearest even, saturating
  const unsigned char s = std::signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (std::isnan(a)) return s | 0x7f;
  if (a >= 448.f) return s | 0x7e;
  int e;
  (void)frexpf(a, &e);                                   // a = mant * 2^e, mant in [0.5, 1)
  e -= 1;                                                // a in [2^e, 2^(e+1))
  if (a == 0.f || e < -6) return s | (unsigned char)nearbyintf(ldexpf(a, 9));       // subnormals: steps of 2^-9 (8 = the first normal)
  int m = (int)nearbyintf((ldexpf(a, -e) - 1.0f) * 8.0f);
  if (m == 8) { m = 0; ++e; }
  return s | (unsigned char)(((e + 7) << 3) | m);
}
// D = A(32x64) * B(64x32) + C, MX block scales (v_mfma_scale_f32_32x32x64_f8f6f4, cbsz = blgp = 0), as measured on gfx950 with
// tools/probes/mx_probe.hip: lane l = (row l&31, half h = l>>5) holds e4m3 bytes A[row][16 h + 0..15] in its first 16 bytes and
// A[row][32 + 16 h + 0..15] in its second 16 bytes (B alike, column l&31); the 32 k of block b = 0, 1 are scaled by 2^(byte OPSEL of the
// scale word of lane (row, b) - 127).
typedef __attribute__((ext_vector_type(8))) int emu_i32x8;
typedef __attribute__((ext_vector_type(16))) float emu_f32x16_;
inline emu_f32x16_ emu_mfma_scale_32x32x64_f8(emu_i32x8 a, emu_i32x8 b, emu_f32x16_ c, int opa, int sa, int opb, int sb) {
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  memcpy(w.xa[l], &a, 32); w.xa[l][32] = (unsigned char)(((unsigned)sa >> (8 * opa)) & 0xff);
  memcpy(w.xb[l], &b, 32); w.xb[l][32] = (unsigned char)(((unsigned)sb >> (8 * opb)) & 0xff);
  emu::wave_sync();
  emu_f32x16_ d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double s = 0.0;
    for (int blk = 0; blk < 2; ++blk) {                       // MX block: k = 32 blk .. 32 blk + 31, scales from lanes (row, blk), (col, blk)
      double part = 0.0;
      for (int hb = 0; hb < 2; ++hb) {                        // the lane half that holds k = 32 blk + 16 hb + 0..15, in its bytes 16 blk ..
        const unsigned char* pa = w.xa[hb * 32 + row] + 16 * blk;
        const unsigned char* pb = w.xb[hb * 32 + col] + 16 * blk;
        for (int i = 0; i < 16; ++i) part += (double)emu_e4m3_to_f32(pa[i]) * (double)emu_e4m3_to_f32(pb[i]);
      }
      s += ldexp(part, (int)w.xa[blk * 32 + row][32] - 127 + (int)w.xb[blk * 32 + col][32] - 127);
    }
    d[r] += (float)s;
  }
  emu::wave_sync();
  return d;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, cbsz, blgp, opa, sa, opb, sb) emu_mfma_scale_32x32x64_f8(a, b, c, opa, sa, opb, sb)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16(a, b, c)

// ds_read_b64_tr_b16: within each 16-lane group the lanes' 4-element reads form a 4x16 matrix
// (lane i supplies row i>>2, columns 4*(i&3)..+3); lane i receives column i (checked on gfx950).
template <typename V4>
inline V4 emu_ds_read_tr16_b64(const void* p) {
  emu::WaveState& w = emu::wave();
  int l = emu::lane_id();
  memcpy(w.xa[l], p, 8);
  emu::wave_sync();
  V4 r;
  const int g = l & ~15, i = l & 15;
  for (int j = 0; j < 4; ++j) memcpy(reinterpret_cast<unsigned char*>(&r) + 2 * j, w.xa[g + 4 * j + (i >> 2)] + 2 * (i & 3), 2);
  emu::wave_sync();
  return r;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_16x16x32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_16x16x32(a, b, c)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
inline int __any(int pred) {
  int r = pred ? 1 : 0;
  for (int m = 32; m >= 1; m >>= 1) r |= __shfl_xor(r, m, 64);
  return r;
}

inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }

// atomics (single OS thread per block; blocks of one launch may run on several OS threads)
inline float atomicAdd(float* p, float v) {
  auto* a = reinterpret_cast<std::atomic<float>*>(p);
  float old = a->load();
  while (!a->compare_exchange_weak(old, old + v)) {}
  return old;
}
inline int atomicAdd(int* p, int v) { return reinterpret_cast<std::atomic<int>*>(p)->fetch_add(v); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return reinterpret_cast<std::atomic<unsigned long long>*>(p)->fetch_add(v); }

#define MTX_LAUNCH(kernel, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define MTX_DYN_SMEM(name) char* name = emu::B->dyn_smem
