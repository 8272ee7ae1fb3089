// mtx_device.h — shared device-side definitions for the gfx950 kernels.
//
// Built two ways:
//   hipcc --offload-arch=gfx950            -> libmtx_hip.so (the product)
//   clang++ -x c++ -DMTX_EMU (tests only)  -> tests/emu/libmtx_emu.so, a CPU SIMT simulator used
//                                             by the CPU-only test tier to check kernel indexing.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef MTX_EMU
#include "emu_hip.h"
#else
#include <hip/hip_runtime.h>
#define MTX_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define MTX_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include "../../include/mtx_hip.h"

// Workgroup barrier that orders LDS traffic only.  hipcc's __syncthreads() carries a workgroup
// release fence, i.e. `s_waitcnt vmcnt(0)`: every outstanding global STORE (and LDS-DMA) must retire
// before the barrier.  Inside a persistent tile loop that serialises each tile's store tail
// (measured: 226 -> see profiles/ for the c64 conv).  Barriers that only guard LDS reuse wait on
// lgkmcnt alone and let the stores drain behind the next tile's work.
#ifdef MTX_EMU
#define MTX_LDS_BARRIER() __syncthreads()
#else
#define MTX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

namespace mtx {

// 16 zero bytes in device memory: the source of out-of-image halo chunks for LDS-DMA loads
#ifdef MTX_EMU
static unsigned char g_zero16[16] __attribute__((aligned(16))) = {0};
#else
__device__ __attribute__((aligned(16))) static unsigned char g_zero16[16];
#endif

// LDS-DMA: every ACTIVE lane copies 16 bytes from its own global address to
// lds_wave_base + lane*16 (the LDS base must be wave-uniform; it is read with readfirstlane into M0).
// Completion is tracked by vmcnt; nothing orders a later ds_read behind it except the issuing
// wave's `s_waitcnt vmcnt` followed by a barrier.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef MTX_EMU
  memcpy(reinterpret_cast<unsigned char*>(lds_wave_base) + emu::lane_id() * 16, gsrc, 16);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
#ifdef MTX_EMU
#define MTX_WAIT_VMEM() ((void)0)
#define MTX_WAIT_VMEM_BUT(n) ((void)0)
#else
#define MTX_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// gfx950 retires vector memory operations in issue order on one counter (loads AND stores), so
// "all but the youngest n" is how a wave waits for its loads without waiting for the n stores it
// issued after them.  n must be the EXACT number of operations issued since the last one waited for.
#define MTX_WAIT_VMEM_BUT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kWave = 64;

template <typename T> struct Traits;
template <> struct Traits<__bf16> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Traits<_Float16> {
  typedef f16x8 v8; typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

typedef __attribute__((ext_vector_type(16))) float f32x16;
template <typename T> struct Mma32;
template <> struct Mma32<__bf16> {
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<_Float16> {
  static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// 32x32x16 MFMA with the accumulator pinned to a register class: AGPRs for accumulators only the matrix pipe touches
// (attention O^T), VGPRs for accumulators the VALU reads next (attention S^T) — with 512 registers per lane the
// compiler otherwise parks S^T in AGPRs and pays a v_accvgpr_read per element.
template <typename T> struct Mma32Pinned;
#ifdef MTX_EMU
template <typename T> struct Mma32Pinned {
  typedef typename Traits<T>::v8 v8;
  static __device__ __forceinline__ void acc_agpr(f32x16& c, v8 a, v8 b) { c = Mma32<T>::mfma(a, b, c); }
  static __device__ __forceinline__ void acc_vgpr(f32x16& c, v8 a, v8 b) { c = Mma32<T>::mfma(a, b, c); }
  static __device__ __forceinline__ void set_vgpr(f32x16& c, v8 a, v8 b) { const f32x16 z = {0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f}; c = Mma32<T>::mfma(a, b, z); }
  static __device__ __forceinline__ void set_from(f32x16& d, v8 a, v8 b, const f32x16& c) { d = Mma32<T>::mfma(a, b, c); }
};
#else
template <> struct Mma32Pinned<__bf16> {
  static __device__ __forceinline__ void acc_agpr(f32x16& c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc_vgpr(f32x16& c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void set_vgpr(f32x16& c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b)); }
  // D = A B + C with C in a DIFFERENT register tuple than D (the compiler's VGPR form ties them and copies C first)
  static __device__ __forceinline__ void set_from(f32x16& d, bf16x8 a, bf16x8 b, const f32x16& c) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c)); }
};
template <> struct Mma32Pinned<_Float16> {
  static __device__ __forceinline__ void acc_agpr(f32x16& c, f16x8 a, f16x8 b) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc_vgpr(f32x16& c, f16x8 a, f16x8 b) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void set_vgpr(f32x16& c, f16x8 a, f16x8 b) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void set_from(f32x16& d, f16x8 a, f16x8 b, const f32x16& c) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c)); }
};
#endif

// LDS transpose read (ds_read_b64_tr_b16): per 16-lane group a 4 x 16 block of 16-bit elements, lane i
// addresses row i>>2, columns 4*(i&3)..+3 and receives column i (4 rows).
template <typename T>
__device__ __forceinline__ typename Traits<T>::v4 lds_read_tr16(const void* lds_ptr) {
#ifdef MTX_EMU
  return emu_ds_read_tr16_b64<typename Traits<T>::v4>(lds_ptr);
#else
  typedef __attribute__((ext_vector_type(4))) short s4;
  s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)lds_ptr);
  return __builtin_bit_cast(typename Traits<T>::v4, r);
#endif
}

// combine a value with the one held by lane ^ 32 (v_permlane32_swap: one VALU op, no LDS)
__device__ __forceinline__ float half_max(float v) {
#ifdef MTX_EMU
  const float o = __shfl_xor(v, 32, 64); return o > v ? o : v;
#else
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float a = __builtin_bit_cast(float, (unsigned)r[0]), b = __builtin_bit_cast(float, (unsigned)r[1]);
  return a > b ? a : b;
#endif
}
__device__ __forceinline__ float half_sum(float v) {
#ifdef MTX_EMU
  return v + __shfl_xor(v, 32, 64);
#else
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
#endif
}
__device__ __forceinline__ float fast_exp2(float v) {
#ifdef MTX_EMU
  return exp2f(v);
#else
  return __builtin_amdgcn_exp2f(v);
#endif
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }
// _Float16 saturates instead of overflowing to inf (activations can spike on random weights)
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) {
#ifdef MTX_EMU
  v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
#else
  v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);      // one v_med3_f32
#endif
  return (_Float16)v;
}

// x / d where d = 1 + e^t >= 1: one v_rcp_f32 (1 ulp) and a multiply instead of the IEEE division sequence the compiler emits for `/` (two
// v_div_scale, v_rcp, four FMAs, v_div_fmas, v_div_fixup: 10 instructions per element — 1 300 per wave in the epilogue of a 256 x 256 GELU tile).
// Used where it was measured to pay and the parity tests hold: the tanh-GELU epilogue (v * sigmoid(2u)) and the SwiGLU sites of the FLUX graphs
// (bf16; profiles/r05_visit_m_*.log: GELU GEMMs -4.3 ... -4.9 %).  NOT used for SiLU / sigmoid: the f16 detector graphs go through those, and
// YOLO11-L's decoded-box error doubled with the bare reciprocal (0.061 -> 0.138 of a DFL bin at stride 32, YOLO12x + 28 %, same visit, same sources built
// both ways): a 1-ulp error that leans one way is a bias, and a bias of 6e-8 per activation summed over a K = 4 608 convolution is 3e-4 — the size of an
// f16 rounding step.  The Newton step removes the lean for the FLUX sites; the detector sites keep exact division.  d = inf gives 0 either way.
// (-DMTX_EXACT_DIV builds the previous form everywhere, for A/Bs; the fp32 path of csrc/f32ops.hip keeps exact division.)
__device__ __forceinline__ float div_by_1p(float x, float d) {
#if defined(MTX_EMU) || defined(MTX_EXACT_DIV)
  return x / d;
#else
  d = fminf(d, 3.0e38f);                                      // e^t may have overflowed: inf * 0 in the Newton step would be NaN where x / inf is 0
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);      // one Newton step: the 1-ulp result of v_rcp_f32 errs to one side, and a one-sided error
  return x * r;                                                // adds up coherently over the thousands of products of the next reduction (see below)
#endif
}

// compile-time activation (hot epilogues): ACT < 0 falls back to the runtime switch
template <int ACT>
__device__ __forceinline__ float apply_act_t(float v, int act, float p);

__device__ __forceinline__ float apply_act(float v, int act, float p) {
  switch (act) {
    case MTX_ACT_RELU: return v > 0.f ? v : 0.f;
    case MTX_ACT_SILU: return v / (1.f + __expf(-v));
    case MTX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case MTX_ACT_GELU_TANH: {
      float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return 0.5f * v * (1.f + tanhf(u));
    }
    case MTX_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case MTX_ACT_LEAKY: return v > 0.f ? v : v * p;
    default: return v;
  }
}

template <int ACT>
__device__ __forceinline__ float apply_act_t(float v, int act, float p) {
  if (ACT == MTX_ACT_NONE) return v;
  if (ACT == MTX_ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == MTX_ACT_SILU) return v / (1.f + __expf(-v));
  if (ACT == MTX_ACT_GELU_TANH) {       // 0.5 v (1 + tanh u) == v * sigmoid(2u): one exp, one divide
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return div_by_1p(v, 1.f + __expf(-2.f * u));
  }
  return apply_act(v, act, p);
}

// 16-byte chunk (8 x 16-bit) helpers
template <typename T>
__device__ __forceinline__ void unpack8(const u32x4& raw, float (&f)[8]) {
  typename Traits<T>::v8 v = __builtin_bit_cast(typename Traits<T>::v8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  typename Traits<T>::v8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = from_f32<T>(f[i]);
  return __builtin_bit_cast(u32x4, v);
}

// the xor butterfly through the LDS crossbar (ds_bpermute per level): what `wave_sum` was through round 4; kept as the yardstick the
// register-only form below was held to on hardware (round 5); the simulator's form
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
#ifndef MTX_EMU
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
#endif
// The same butterfly, same level order (32, 16, 8, 4, 2, 1), without LDS traffic: v_permlane32_swap, v_permlane16_swap, then DPP inside the
// 16-lane rows — row_ror:8 is lane ^ 8; row_ror:4 reads lane (l + 4) % 16, which after the ^ 8 level holds what lane l ^ 4 holds; quad_perm
// for ^ 2 and ^ 1.  Every level adds the same two numbers as the shuffle form, so the bytes are identical (fp addition commutes).
__device__ __forceinline__ float wave_sum(float v) {
#ifdef MTX_EMU
  return wave_sum_shfl(v);
#else
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
  v += dpp_move<0x128>(v);      // row_ror:8
  v += dpp_move<0x124>(v);      // row_ror:4
  v += dpp_move<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v += dpp_move<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  return v;
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef MTX_EMU
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { float o = __shfl_xor(v, m, 64); v = v > o ? v : o; }
  return v;
#else
  float o;
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = __builtin_bit_cast(float, (unsigned)r[0]), b = __builtin_bit_cast(float, (unsigned)r[1]);
    v = a > b ? a : b;
  }
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float a = __builtin_bit_cast(float, (unsigned)r[0]), b = __builtin_bit_cast(float, (unsigned)r[1]);
    v = a > b ? a : b;
  }
  o = dpp_move<0x128>(v); v = v > o ? v : o;
  o = dpp_move<0x124>(v); v = v > o ? v : o;
  o = dpp_move<0x4E>(v); v = v > o ? v : o;
  o = dpp_move<0xB1>(v); v = v > o ? v : o;
  return v;
#endif
}

// ---- OCP fp8 e4m3 ------------------------------------------------------------------------------------------
// two fp32 -> two e4m3 bytes (RNE) in the low (HI = false) or high half of `old`; callers clamp to +-448 first
template <bool HI>
__device__ __forceinline__ unsigned cvt_pk_fp8(float a, float b, unsigned old) {
#ifdef MTX_EMU
  const unsigned pk = (unsigned)emu_f32_to_e4m3(a) | ((unsigned)emu_f32_to_e4m3(b) << 8);
  return HI ? ((old & 0x0000ffffu) | (pk << 16)) : ((old & 0xffff0000u) | pk);
#else
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, HI);
#endif
}

// One lane's share of an MX block quantisation (mtx_quant_args): the lane holds 8 consecutive k (chunk c8 of its row), lanes 4 g .. 4 g + 3
// of a wave hold one 32-k block, 16 adjacent lanes one uint32 of four E8M0 scale bytes.  Every lane of the wave must call this (shuffles);
// lanes without data pass zeros.  Out: the 8 e4m3 bytes (w0, w1) and the group's scale word (to be stored by the lane with (c8 & 15) == 0).
__device__ __forceinline__ void mx_quantize_chunk(float (&f)[8], long c8, unsigned& w0, unsigned& w1, unsigned& word) {
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float a = fabsf(f[e]); amax = a > amax ? a : amax; }
  { float o = __shfl_xor(amax, 1, 64); amax = o > amax ? o : amax; }
  { float o = __shfl_xor(amax, 2, 64); amax = o > amax ? o : amax; }
  // smallest power of two 2^(eb - 127) >= amax / 448
  const float r = amax * (1.0f / 448.0f);
  const unsigned u = __builtin_bit_cast(unsigned, r);
  int eb = (int)((u >> 23) & 0xff) + ((u & 0x7fffffu) ? 1 : 0);
  eb = amax == 0.f ? 127 : (eb < 1 ? 1 : (eb > 253 ? 253 : eb));
  const float inv = __builtin_bit_cast(float, (unsigned)(254 - eb) << 23);      // 2^(127 - eb)
#pragma unroll
  for (int e = 0; e < 8; ++e) { float v = f[e] * inv; v = v > 448.f ? 448.f : (v < -448.f ? -448.f : v); f[e] = v; }
  w0 = 0; w1 = 0;
  w0 = cvt_pk_fp8<false>(f[0], f[1], w0); w0 = cvt_pk_fp8<true>(f[2], f[3], w0);
  w1 = cvt_pk_fp8<false>(f[4], f[5], w1); w1 = cvt_pk_fp8<true>(f[6], f[7], w1);
  word = (unsigned)eb << (8 * (int)((c8 & 15) >> 2));
  word |= __shfl_xor(word, 4, 64);
  word |= __shfl_xor(word, 8, 64);
}

// XCD-aware, bijective remap of a linear workgroup id: workgroup b runs on XCD b%8 (observed),
// so give each XCD a contiguous range of tiles and neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  if (nwg < nx * 2) return bid;
  unsigned q = nwg / nx, r = nwg % nx, xcd = bid % nx, idx = bid / nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- hand-offs between workgroups of ONE launch (cdna_hip_programming.md Guideline 16) ------------------------
// The per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores.  Small records are
// handed over like this: the writer stores them WRITE-THROUGH (sc1: a relaxed agent-scope atomic store, 4 bytes per lane), every
// storing wave drains (vmcnt(0)), the workgroup meets at a barrier, ONE lane draws a ticket from a counter (relaxed agent-scope
// atomic); the reader — the workgroup that drew the last ticket — runs ONE agent-scope acquire (drops its stale lines), meets at a
// barrier and reads with plain loads.  Nobody waits for anybody, so nothing depends on dispatch order or residency.
// (A plain-store + agent RELEASE publish also works but writes back the whole XCD L2's dirty lines: measured 35-80 us per episode behind
// a GEMM's 256 KB fp32 slabs, profiles/r03_gemm_one_launch_streamk_ab.log.)
__device__ __forceinline__ void agent_store(unsigned* word, unsigned v) {
#ifdef MTX_EMU
  reinterpret_cast<std::atomic<unsigned>*>(word)->store(v);
#else
  __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// 16 bytes per lane, write-through like agent_store (global_store_dwordx4 ... sc1).  The compiler does not count this store: the caller
// drains it with an explicit `s_waitcnt vmcnt(0)` (agent_drain) before the ticket.
__device__ __forceinline__ void agent_store16(float* p, f32x4 v) {
#ifdef MTX_EMU
  *reinterpret_cast<f32x4*>(p) = v;
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void agent_drain() {
#ifndef MTX_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void agent_store(float* word, float v) { agent_store(reinterpret_cast<unsigned*>(word), __builtin_bit_cast(unsigned, v)); }
__device__ __forceinline__ unsigned agent_ticket(unsigned* counter) {          // returns the value before the increment
#ifdef MTX_EMU
  return reinterpret_cast<std::atomic<unsigned>*>(counter)->fetch_add(1u);
#else
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void agent_acquire() {
#ifndef MTX_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}

// 16 bytes through a buffer descriptor: byte offset = voff (per lane) + soff (wave-uniform); reads past
// `bytes` return zeros (the hardware range check), which is how rows >= sk become zero rows.
struct BufView {
#ifdef MTX_EMU
  const unsigned char* base; unsigned bytes;
#else
  __amdgpu_buffer_rsrc_t rsrc;
#endif
};
__device__ __forceinline__ BufView make_buf(const void* base, unsigned bytes) {
  BufView b;
#ifdef MTX_EMU
  b.base = reinterpret_cast<const unsigned char*>(base); b.bytes = bytes;
#else
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
#endif
  return b;
}
__device__ __forceinline__ u32x4 buf_load16(const BufView& b, unsigned voff, unsigned soff) {
#ifdef MTX_EMU
  u32x4 r = u32x4{0u, 0u, 0u, 0u};
  if ((unsigned long)voff + soff + 16 <= b.bytes) memcpy(&r, b.base + voff + soff, 16);
  return r;
#else
  return __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)voff, (int)soff, 0);
#endif
}


__device__ __forceinline__ u32x2 buf_load8(const BufView& b, unsigned voff, unsigned soff) {
#ifdef MTX_EMU
  u32x2 r = u32x2{0u, 0u};
  if ((unsigned long)voff + 8 <= b.bytes) memcpy(&r, b.base + voff + soff, 8);
  return r;
#else
  return __builtin_amdgcn_raw_buffer_load_b64(b.rsrc, (int)voff, (int)soff, 0);
#endif
}

// 8-byte store through a buffer descriptor.  A lane that must not write passes an offset >= the
// descriptor's size: the range check drops it, and the instruction is still ISSUED by the wave — the
// number of stores in flight does not depend on predicates (see MTX_WAIT_VMEM_BUT).
__device__ __forceinline__ void buf_store8(const BufView& b, unsigned voff, u32x2 v, unsigned soff = 0) {
#ifdef MTX_EMU
  if ((unsigned long)voff + 8 <= b.bytes) memcpy(const_cast<unsigned char*>(b.base) + voff + soff, &v, 8);   // the range check covers voff only
#else
  __builtin_amdgcn_raw_buffer_store_b64(v, b.rsrc, (int)voff, (int)soff, 0);
#endif
}

__device__ __forceinline__ void buf_store16(const BufView& b, unsigned voff, u32x4 v, unsigned soff = 0) {
#ifdef MTX_EMU
  if ((unsigned long)voff + 16 <= b.bytes) memcpy(const_cast<unsigned char*>(b.base) + voff + soff, &v, 16);
#else
  __builtin_amdgcn_raw_buffer_store_b128(v, b.rsrc, (int)voff, (int)soff, 0);
  // A 16-byte store reads its data registers over several cycles.  With an SGPR soffset the compiler assumes the hardware covers a
  // vector write to those registers in the very next instruction; on gfx950 it does not: the last lanes of each row were stored
  // from the NEXT values (measured: r02 visit N/P, lanes 12..15 of every row held the following v_pk_add_f32's fp32 bits).
  // Keeping the data live across two wait states keeps the registers from being reused that early.
  asm volatile("s_nop 1" : "+v"(v));
#endif
}

// v_permlane16_swap: the lanes of an ODD 16-lane row trade their `a` for the `b` of the lane 16 below them (even row).
// After it an even-row lane holds (its own a, its upper neighbour's a) and an odd-row lane (its lower neighbour's b, its own b).
__device__ __forceinline__ void row_pair_exchange(uint32_t& a, uint32_t& b) {
#ifdef MTX_EMU
  const uint32_t pa = __shfl_xor(a, 16, 64), pb = __shfl_xor(b, 16, 64);
  if ((emu::lane_id() >> 4) & 1) a = pb; else b = pa;
#else
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}

// the same, straight into LDS (LDS-DMA): lane l's 16 bytes land at lds_wave_base + 16 l
__device__ __forceinline__ void buf_load16_lds(const BufView& b, unsigned voff, unsigned soff, void* lds_wave_base) {
#ifdef MTX_EMU
  unsigned char* d = reinterpret_cast<unsigned char*>(lds_wave_base) + emu::lane_id() * 16;
  if ((unsigned long)voff + soff + 16 <= b.bytes) memcpy(d, b.base + voff + soff, 16); else memset(d, 0, 16);
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
#endif
}

// Zero fill as a KERNEL.  Launchers that need a cleared accumulator in front of an atomics pass (mask_select's pixel counts, GroupNorm's
// channel sums, the cleaning statistics) used hipMemsetAsync; captured into a hipGraph that became a memset node, and on ROCm 7.2 / gfx950
// replays of the SAM decoder graph left counts of a 184-byte fill (23 boxes x 2 ints — not a multiple of 16 bytes) uncleared now and then:
// the stability-based mask choice of a box changed between identical calls (round 4, tools/diag_sam_repeat.py; eager launches and a
// 224-byte fill were stable).  A kernel node has no such special cases.
static __global__ __launch_bounds__(256) void zero_words_kernel(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0u;
}
// byte fill of any length (MTX_OP_MEMSET): 4-byte stores over the aligned middle, single bytes at the ragged ends
static __global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char* p, long bytes, unsigned char v) {
  const long head = (4 - (long)(reinterpret_cast<unsigned long long>(p) & 3)) & 3;          // bytes before the first aligned word
  const long lead = head < bytes ? head : bytes;
  const long words = (bytes - lead) / 4;
  const unsigned pattern = 0x01010101u * v;
  unsigned* w = reinterpret_cast<unsigned*>(p + lead);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long)gridDim.x * 256) w[i] = pattern;
  if (blockIdx.x == 0 && threadIdx.x < 8) {
    const long t = threadIdx.x;
    if (t < lead) p[t] = v;                                                             // up to 3 leading bytes
    const long tail0 = lead + words * 4;
    if (t >= 4 && tail0 + (t - 4) < bytes) p[tail0 + (t - 4)] = v;                          // up to 3 trailing bytes
  }
}
static inline void fill_bytes_async(void* p, int value, size_t bytes, void* stream) {
  if (bytes == 0) return;
  long blocks = ((long)(bytes / 4) + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 2048) blocks = 2048;
  MTX_LAUNCH(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<unsigned char*>(p), (long)bytes, (unsigned char)value);
}
static inline void zero_words_async(void* p, size_t bytes, void* stream) {      // bytes: a multiple of 4
  const long n = (long)(bytes / 4);
  if (n <= 0) return;
  long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  MTX_LAUNCH(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<unsigned*>(p), n);
}

}  // namespace mtx
