// attention.hip — fused softmax(scale * Q K^T) V on the gfx950 matrix cores (flash-style,
// online softmax, nothing of size sq x sk ever reaches HBM).
//
// Serves the windowed / global / query-pooled attention of SAM-2.1's Hiera encoder and the
// two-way mask decoder (transformers Sam2Model, core/image/detection.py:505) and the joint
// text+image attention of the FLUX MMDiT (diffusers, core/image/inpainting.py:877-887).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); 64-key K/V tiles go through LDS
// (K as swizzled [key][d] rows, V TRANSPOSED as [d][key] so both MFMA operands are contiguous).
// Both products are computed transposed (S^T = K Q^T, O^T = V^T P^T): a lane then owns ONE query
// row (col = lane & 15) in S^T and in O^T, so the running max / sum / rescale are lane-local
// (two xor-shuffles across the four 16-lane quads finish a row reduction), and P goes from the
// S^T accumulators straight into the next MFMA's B operand without touching LDS.
#include "mtx_device.h"
#include <cstdlib>
#include <type_traits>

namespace mtx {

struct AttnParams {
  const unsigned char* q; const unsigned char* k; const unsigned char* v; unsigned char* o;
  long batch, heads, sq, sk, d;
  long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
  float scale_log2;
  unsigned qblocks;
  // key-split tail of the long-sequence kernel: workgroups [0, n_full) cover whole key ranges; the remaining query
  // blocks are cut into `split` key ranges each and merged from the partials (unnormalised O^T, running max, sum)
  unsigned n_full, split;
  float* part_o; float* part_ml;
  int prescaled;                 // q carries scale * log2(e): scale_log2 == 1
  // MX fp8 output of the long-sequence kernel (mtx_attn_args.q8): bytes [sq][ldq8] (head h at byte column h * d), scale words [heads * d / 128][lds_q8]
  unsigned char* q8; unsigned* q8_scale; long ldq8, lds_q8;
  // fp8 scores (mtx_attn_args.q_f8 / k_f8): plain e4m3 rows, strides in bytes, head h at byte column h * 128; logits = 2^qk_f8_exp * sum q k
  const unsigned char* qf8; const unsigned char* kf8; long qf8_ss, kf8_ss; int qk_f8_exp;
  // fp8 P V (mtx_attn_args.v_f8t): e4m3 V^T [head][128][vf8_ld], keys in accumulator order inside every 64-key tile
  const unsigned char* vf8; long vf8_ld;
};

constexpr int AT_KV = 64;      // keys per tile
constexpr int AT_QW = 32;      // query rows per wave (2 MFMA column fragments)
constexpr int AT_QB = 128;     // query rows per workgroup

// DP = head dim padded to a multiple of 32 (zero-filled); K row = SLOTS 16-byte slots.
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int KST = DP / 32;                 // k-steps of the S^T product
  constexpr int DF = DP / 16;                  // d fragments of O^T
  constexpr int SLOTS = DP <= 64 ? 8 : 16;
  constexpr int KROW = SLOTS * 16;             // bytes per K row in LDS
  constexpr int K_BYTES = AT_KV * KROW;
  constexpr int VT_BYTES = DP * 128;           // [DP rows][64 keys] T
  __shared__ __attribute__((aligned(16))) unsigned char smem[K_BYTES + VT_BYTES];
  unsigned char* Ks = smem;
  unsigned char* Vt = smem + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q4 = lane >> 4;
  const long bh = blockIdx.x / p.qblocks;
  const long qb = blockIdx.x % p.qblocks;
  const long b = bh / p.heads, h = bh % p.heads;
  const long q0 = qb * AT_QB + wv * AT_QW;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.q_hs;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs;
  T* O = reinterpret_cast<T*>(p.o) + b * p.o_bs + h * p.o_hs;
  const int dch = (int)(p.d / 8);              // valid 16-byte chunks per row

  // Q fragments (B operand: col = query row, 8 consecutive d per lane), kept in registers
  v8 qf[2][KST];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const long qr = q0 + f * 16 + l15;
      const int ch = ks * 4 + q4;
      u32x4 raw = u32x4{0u, 0u, 0u, 0u};
      if (qr < p.sq && ch < dch) raw = *reinterpret_cast<const u32x4*>(Q + qr * p.q_ss + ch * 8);
      qf[f][ks] = __builtin_bit_cast(v8, raw);
    }

  f32x4 oacc[2][DF];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int d = 0; d < DF; ++d) oacc[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {-1.0e30f, -1.0e30f};
  float lrun[2] = {0.f, 0.f};

  constexpr int KCH = AT_KV * (DP / 8);        // 16-byte chunks per K (or V) tile
  constexpr int NLD = (KCH + 255) / 256;
  u32x4 rk[NLD], rv[NLD];
  auto load_tile = [&](long k0) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int idx = tid + it * 256;
      const int ch = idx % (DP / 8), row = idx / (DP / 8);
      u32x4 a = u32x4{0u, 0u, 0u, 0u}, c = u32x4{0u, 0u, 0u, 0u};
      if (idx < KCH && k0 + row < p.sk && ch < dch) {
        a = *reinterpret_cast<const u32x4*>(K + (k0 + row) * p.k_ss + ch * 8);
        c = *reinterpret_cast<const u32x4*>(V + (k0 + row) * p.v_ss + ch * 8);
      }
      rk[it] = a; rv[it] = c;
    }
  };

  const long ntiles = (p.sk + AT_KV - 1) / AT_KV;
  load_tile(0);
  for (long t = 0; t < ntiles; ++t) {
    const long k0 = t * AT_KV;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int idx = tid + it * 256;
      if (idx < KCH) {
        const int ch = idx % (DP / 8), row = idx / (DP / 8);
        *reinterpret_cast<u32x4*>(Ks + row * KROW + ((ch ^ (row & (SLOTS - 1))) << 4)) = rk[it];
        // V transposed: Vt[d][key], 8-byte granules (4 keys) XOR-swizzled by (d & 15)
        const typename Traits<T>::v8 vv = __builtin_bit_cast(v8, rv[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int dr = ch * 8 + e;
          *reinterpret_cast<T*>(Vt + dr * 128 + ((((row >> 2) ^ (dr & 15))) << 3) + ((row & 3) << 1)) = vv[e];
        }
      }
    }
    __syncthreads();
    if (t + 1 < ntiles) load_tile(k0 + AT_KV);

    // ---- S^T = K Q^T : sacc[f][kf] holds keys kf*16 + q4*4 + r for query f*16 + l15 -----------
    f32x4 sacc[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) sacc[f][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const int ch = ks * 4 + q4;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const int row = kf * 16 + l15;
        const v8 kfr = *reinterpret_cast<const v8*>(Ks + row * KROW + ((ch ^ (row & (SLOTS - 1))) << 4));
#pragma unroll
        for (int f = 0; f < 2; ++f) sacc[f][kf] = Traits<T>::mfma(kfr, qf[f][ks], sacc[f][kf]);
      }
    }

    // ---- online softmax (base-2), lane-local per query row --------------------------------------
    v8 pb[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float tmax = -1.0e30f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long key = k0 + kf * 16 + q4 * 4 + r;
          float s = sacc[f][kf][r] * p.scale_log2;
          if (key >= p.sk) s = -1.0e30f;
          sacc[f][kf][r] = s;
          tmax = s > tmax ? s : tmax;
        }
      { float o1 = __shfl_xor(tmax, 16, 64); tmax = o1 > tmax ? o1 : tmax; }
      { float o2 = __shfl_xor(tmax, 32, 64); tmax = o2 > tmax ? o2 : tmax; }
      const float mnew = tmax > mrun[f] ? tmax : mrun[f];
      const float alpha = exp2f(mrun[f] - mnew);
      mrun[f] = mnew;
      float psum = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = exp2f(sacc[f][kf][r] - mnew);
          psum += pv;
          // B operand of O^T = V^T P^T: k-slot q4*8 + e  <->  key (kk*32) + (e<4 ? q4*4+e : 16+q4*4+e-4)
          pb[f][kf >> 1][(kf & 1) * 4 + r] = from_f32<T>(pv);
        }
      lrun[f] = lrun[f] * alpha + psum;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        oacc[f][d][0] *= alpha; oacc[f][d][1] *= alpha; oacc[f][d][2] *= alpha; oacc[f][d][3] *= alpha;
      }
    }

    // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int dr = d * 16 + l15;
        const int g0 = kk * 8 + q4;            // granule of keys kk*32 + q4*4 .. +3
        const int g1 = kk * 8 + 4 + q4;        // granule of keys kk*32 + 16 + q4*4 .. +3
        const v4 lo = *reinterpret_cast<const v4*>(Vt + dr * 128 + ((g0 ^ (dr & 15)) << 3));
        const v4 hi = *reinterpret_cast<const v4*>(Vt + dr * 128 + ((g1 ^ (dr & 15)) << 3));
        v8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
        for (int f = 0; f < 2; ++f) oacc[f][d] = Traits<T>::mfma(vf, pb[f][kk], oacc[f][d]);
      }
    }
  }

  // ---- finish: row sums across the 4 quads, normalise, store 4 consecutive d per lane ----------
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float l = lrun[f];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const long qr = q0 + f * 16 + l15;
    if (qr < p.sq) {
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int dc = d * 16 + q4 * 4;
        if (dc < p.d) {
          v4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(oacc[f][d][r] * inv);
          *reinterpret_cast<v4*>(O + qr * p.o_ss + dc) = o;
        }
      }
    }
  }
}

// =====================================================================================================
// Large-sequence kernel (FLUX MMDiT joint attention: d = 128, T ~ 8.6 k tokens, 24 heads).
//
// 8 waves x 32 query rows = 256 queries per workgroup, 64-key K/V tiles double-buffered in LDS (64 KB),
// v_mfma_f32_32x32x16.  Both products are computed transposed (S^T = K Q^T, O^T = V^T P^T): lanes l and
// l^32 together own one query row, so the softmax is register-local plus one v_permlane32_swap.
//   * K rows (256 B) are XOR-swizzled by (row & 15) in 16-byte chunks -> conflict-free ds_read_b128.
//   * V stays row-major [key][d]; the V^T operand comes from ds_read_b64_tr_b16 (hardware 4x16 transpose).
//     Chunks are swizzled by 4*(row & 3) so the 32 lanes of a half hit 32 distinct bank pairs.
//   * The PV k-slots are assigned to keys in the order the S^T accumulators hold them (lane half `hi` owns
//     keys 8j + 4*hi + 0..3), so P goes from the accumulators into the B operand with a plain
//     v_cvt_pk — no cross-lane exchange, no LDS round trip.
//   * The running max is only refreshed when some row's tile max exceeds it by more than 2^8 (wave-uniform
//     branch); otherwise the stale max is kept (P <= 256, exact in the final normalisation) and the
//     64-register O^T rescale is skipped.
//   * Next tile: global -> registers at the top of the iteration, registers -> LDS after PV (one barrier
//     per tile, LDS-only: s_waitcnt lgkmcnt(0); s_barrier).
constexpr int AB_KV = 64;
// scheduling fence for LDS reads and MFMAs only: VALU / SALU / VMEM may still move across it
#ifdef MTX_EMU
#define MTX_SCHED_FENCE() ((void)0)
#else
#define MTX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0x0476)
#endif
constexpr int AB_QB = 256;

// One K/V tile of the main loop.  STAGE is a compile-time constant so every LDS address is
// (loop-invariant VGPR) + (immediate offset).
template <typename T, int DP, int STAGE, bool RAGGED>
__device__ __forceinline__ void attn_mma32_tile(unsigned char* smem, const typename Traits<T>::v8 (&qf)[DP / 16], f32x16 (&oacc)[DP / 32],
                                                float& m_raw, float& lsum, const float c, const float thr,
                                                const int (&kaddr)[DP / 16], const int (&vaddr)[DP / 32], const long kvalid, const int hi) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int KS = DP / 16, DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  const unsigned char* Ks = smem + STAGE * 2 * TILE_B;
  const unsigned char* Vs = Ks + TILE_B;

  // ---- S^T = K Q^T: sacc[kb][r] = key 32*kb + (r&3) + 8*(r>>2) + 4*hi of this tile, query l31 ----------------
  f32x16 sacc[2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8 kf = *reinterpret_cast<const v8*>(Ks + kaddr[ks] + kb * 32 * ROWB);
      sacc[kb] = Mma32<T>::mfma(kf, qf[ks], ks == 0 ? zero : sacc[kb]);
    }

  if (RAGGED) {                                // last tile of a ragged sequence: keys >= kvalid do not exist
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kvalid) sacc[kb][r] = -1.0e30f;
  }
  float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
  for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
  tmax = half_max(tmax);
  if (__any(tmax > m_raw + thr)) {
    const float m_new = fmaxf(m_raw, tmax);
    const float alpha = fast_exp2((m_raw - m_new) * c);
    m_raw = m_new;
    lsum *= alpha;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
  }
  const float mc = m_raw * c;
  v8 pb[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = fast_exp2(__builtin_fmaf(sacc[kb][r], c, -mc));
      lsum += pv;
      pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
    }

  // ---- O^T += V^T P^T: k-slot (hi, j) of step (kb, s2) is key 32*kb + 16*s2 + 8*(j>>2) + 4*hi + (j&3) ------------
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const unsigned char* a = Vs + vaddr[d] + (kb * 32 + s2 * 16) * ROWB;
        const v4 lo = lds_read_tr16<T>(a);
        const v4 hv = lds_read_tr16<T>(a + 8 * ROWB);
        v8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = hv[0]; vf[5] = hv[1]; vf[6] = hv[2]; vf[7] = hv[3];
        oacc[d] = Mma32<T>::mfma(vf, pb[kb][s2], oacc[d]);
      }
}

// Bias variant of the tile step: Q arrives pre-multiplied by scale * log2(e) and the S^T accumulators START at minus the running
// maximum (`minit`, a persistent 16-register tuple that only changes when the maximum is refreshed), so the matrix pipe delivers
// s*c - M directly and the softmax needs no per-score fused multiply-add: exp2 straight from the accumulator (32 VALU issue slots
// per wave and tile saved out of ~135).  `M` is in log2 units; the first tile always refreshes (M starts at 0, not -inf, so the
// accumulation keeps its precision).
// NOMAX: the row maximum is not computed on the hot path at all.  With M folded into the accumulator the only thing a stale M can
// do is let exp2 grow large, and fp32 / bf16 keep their relative precision while it does; the tile's partial row sums (which are
// needed anyway) flag the danger zone (> 2^40, inf or nan), and only then — and on the first tile — the exact maximum is taken and
// the tile's probabilities are recomputed.
// largest partial row sum a stale maximum may produce before the exact path is taken: the probabilities must stay finite in T
template <typename T> struct AttnSumLimit { static constexpr float v = 1.0e12f; };       // bf16: fp32 exponent range, 2^40
template <> struct AttnSumLimit<_Float16> { static constexpr float v = 3.0e4f; };        // f16: max 65504
#ifdef MTX_EMU
#define MTX_SCHED_GROUP(mask, n) ((void)0)
#else
#define MTX_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
// DEEP (round 5, schedule 68; measured against the plain form in one process): the K fragments of four k-steps and the V^T fragments of four
// MFMAs are in flight ahead of the matrix pipe, the order pinned with sched_group_barrier (0x100 = LDS reads, 0x008 = MFMA)
template <typename T, int DP, int STAGE, bool RAGGED, bool NOMAX = false, int DEEP = 0>      // DEEP = 0, or K k-steps ahead | V MFMAs ahead << 4
__device__ __forceinline__ void attn_bias_tile(unsigned char* smem, const typename Traits<T>::v8 (&qf)[DP / 16], f32x16 (&oacc)[DP / 32],
                                               float& M, float& lsum, f32x16& minit, bool& first,
                                               const int (&kaddr)[DP / 16], const int (&vaddr)[DP / 32], const long kvalid, const int hi) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int KS = DP / 16, DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  const unsigned char* Ks = smem + STAGE * 2 * TILE_B;
  const unsigned char* Vs = Ks + TILE_B;
  f32x16 sacc[2];
  if constexpr (DEEP) {
    static_assert(KS == 8, "pipeline written for eight k-steps");
    v8 kfd[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kfd[ks][kb] = *reinterpret_cast<const v8*>(Ks + kaddr[ks] + kb * 32 * ROWB);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sacc[kb] = Mma32<T>::mfma(kfd[ks][kb], qf[ks], ks == 0 ? minit : sacc[kb]);
    constexpr int KD = DEEP & 15;                // k-steps (read pairs) in flight ahead of the matrix pipe
    static_assert(KD >= 1 && KD <= KS, "K prefetch depth");
    MTX_SCHED_GROUP(0x100, 2 * KD);
#pragma unroll
    for (int i = 0; i < KS - KD; ++i) { MTX_SCHED_GROUP(0x008, 2); MTX_SCHED_GROUP(0x100, 2); }
    MTX_SCHED_GROUP(0x008, 2 * KD);
  } else {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8 kf = *reinterpret_cast<const v8*>(Ks + kaddr[ks] + kb * 32 * ROWB);
      if (ks == 0) Mma32Pinned<T>::set_from(sacc[kb], kf, qf[ks], minit);
      else sacc[kb] = Mma32<T>::mfma(kf, qf[ks], sacc[kb]);
    }
  }
  if (RAGGED) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kvalid) sacc[kb][r] = -1.0e30f;
  }
  v8 pb[2][2];
  if (!NOMAX) {
    float tmax = fmaxf(sacc[0][0], sacc[1][0]);       // relative to M
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
    tmax = half_max(tmax);
    if (first || __any(tmax > 8.0f)) {
      const float delta = first ? tmax : fmaxf(tmax, 0.f);     // the first tile sets M to its own maximum (may be negative)
      const float alpha = fast_exp2(-delta);
      M += delta;
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] = -M;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
      first = false;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(sacc[kb][r]);
        lsum += pv;
        pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
      }
  } else {
    float tsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(sacc[kb][r]);
        tsum += pv;
        pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
      }
    if (first || __any(!(tsum < AttnSumLimit<T>::v))) {          // far from overflow of T / fp32, catches inf / nan as well
      float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
      tmax = half_max(tmax);
      const float delta = first ? tmax : fmaxf(tmax, 0.f);
      const float alpha = fast_exp2(-delta);
      M += delta;
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] = -M;
      tsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(sacc[kb][r] - delta);
          tsum += pv;
          pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
        }
      first = false;
    }
    lsum += tsum;
  }
  if constexpr (DEEP) {
    v8 vfd[4][DB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const unsigned char* a = Vs + vaddr[d] + ((g >> 1) * 32 + (g & 1) * 16) * ROWB;
        const v4 lo = lds_read_tr16<T>(a);
        const v4 hv = lds_read_tr16<T>(a + 8 * ROWB);
        vfd[g][d][0] = lo[0]; vfd[g][d][1] = lo[1]; vfd[g][d][2] = lo[2]; vfd[g][d][3] = lo[3];
        vfd[g][d][4] = hv[0]; vfd[g][d][5] = hv[1]; vfd[g][d][6] = hv[2]; vfd[g][d][7] = hv[3];
      }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < DB; ++d) oacc[d] = Mma32<T>::mfma(vfd[g][d], pb[g >> 1][g & 1], oacc[d]);
    constexpr int VD = DEEP >> 4;                // MFMAs whose fragments (two transpose reads each) are in flight ahead of the matrix pipe
    static_assert(VD >= 1 && VD <= 16, "V prefetch depth");
    MTX_SCHED_GROUP(0x100, 2 * VD);
#pragma unroll
    for (int i = 0; i < 16 - VD; ++i) { MTX_SCHED_GROUP(0x008, 1); MTX_SCHED_GROUP(0x100, 2); }
    MTX_SCHED_GROUP(0x008, VD);
    return;
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const unsigned char* a = Vs + vaddr[d] + (kb * 32 + s2 * 16) * ROWB;
        const v4 lo = lds_read_tr16<T>(a);
        const v4 hv = lds_read_tr16<T>(a + 8 * ROWB);
        v8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = hv[0]; vf[5] = hv[1]; vf[6] = hv[2]; vf[7] = hv[3];
        oacc[d] = Mma32<T>::mfma(vf, pb[kb][s2], oacc[d]);
      }
}
// The fp8-score form of attn_bias_tile<.., NOMAX = true, DEEP> (round 6): S^T = K Q^T from plain e4m3 rows on v_mfma_scale_f32_32x32x64_f8f6f4 —
// two k-steps of 64 instead of eight of 16, at twice the rate per k — block scales 2^0 (K) and 2^qk_f8_exp (Q).  Softmax and P V as above.
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ f32x16 mfma_f8_scores(i32x8 a, i32x8 b, f32x16 c, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, sb);
}
template <typename T, int DP, int STAGE, bool RAGGED, int DEEP>
__device__ __forceinline__ void attn_bias_tile_k8(unsigned char* smem, const i32x8 (&qf)[2], f32x16 (&oacc)[DP / 32],
                                                  float& M, float& lsum, f32x16& minit, bool& first,
                                                  const int (&kaddr)[4], const int (&vaddr)[DP / 32], const long kvalid, const int hi, const int sb) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  static_assert(DP == 128, "two k-steps of 64 bytes");
  const unsigned char* Ks = smem + STAGE * 2 * TILE_B;
  const unsigned char* Vs = Ks + TILE_B;
  f32x16 sacc[2];
  {
    i32x8 kf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(Ks + kaddr[2 * ks] + kb * 32 * 128);
        const u32x4 hv = *reinterpret_cast<const u32x4*>(Ks + kaddr[2 * ks + 1] + kb * 32 * 128);
        kf[ks][kb] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hv[0], (int)hv[1], (int)hv[2], (int)hv[3]};
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sacc[kb] = mfma_f8_scores(kf[ks][kb], qf[ks], ks == 0 ? minit : sacc[kb], sb);
    MTX_SCHED_GROUP(0x100, 8);
    MTX_SCHED_GROUP(0x008, 4);
  }
  if (RAGGED) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kvalid) sacc[kb][r] = -1.0e30f;
  }
  v8 pb[2][2];
  float tsum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = fast_exp2(sacc[kb][r]);
      tsum += pv;
      pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
    }
  if (first || __any(!(tsum < AttnSumLimit<T>::v))) {
    float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
    tmax = half_max(tmax);
    const float delta = first ? tmax : fmaxf(tmax, 0.f);
    const float alpha = fast_exp2(-delta);
    M += delta;
    lsum *= alpha;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = -M;
    tsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(sacc[kb][r] - delta);
        tsum += pv;
        pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
      }
    first = false;
  }
  lsum += tsum;
  v8 vfd[4][DB];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const unsigned char* a = Vs + vaddr[d] + ((g >> 1) * 32 + (g & 1) * 16) * ROWB;
      const v4 lo = lds_read_tr16<T>(a);
      const v4 hv = lds_read_tr16<T>(a + 8 * ROWB);
      vfd[g][d][0] = lo[0]; vfd[g][d][1] = lo[1]; vfd[g][d][2] = lo[2]; vfd[g][d][3] = lo[3];
      vfd[g][d][4] = hv[0]; vfd[g][d][5] = hv[1]; vfd[g][d][6] = hv[2]; vfd[g][d][7] = hv[3];
    }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int d = 0; d < DB; ++d) oacc[d] = Mma32<T>::mfma(vfd[g][d], pb[g >> 1][g & 1], oacc[d]);
  constexpr int VD = DEEP >> 4;
  static_assert(VD >= 1 && VD <= 16, "V prefetch depth");
  MTX_SCHED_GROUP(0x100, 2 * VD);
#pragma unroll
  for (int i = 0; i < 16 - VD; ++i) { MTX_SCHED_GROUP(0x008, 1); MTX_SCHED_GROUP(0x100, 2); }
  MTX_SCHED_GROUP(0x008, VD);
}
// fp8 scores AND fp8 P V (round 6): the tile's probabilities are rounded to e4m3 where they stand — lane (query, half) holds its 32 keys in exactly
// the order MTX_EW_V_F8T stores a V^T row's bytes — and O^T += V^T P runs as ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32-row block of d.
// A probability must fit e4m3 (448): the stale-maximum path is left as soon as a lane's partial sum reaches 240.
template <typename T, int DP, int STAGE, bool RAGGED>
__device__ __forceinline__ void attn_bias_tile_k8v8(unsigned char* smem, const i32x8 (&qf)[2], f32x16 (&oacc)[DP / 32],
                                                    float& M, float& lsum, f32x16& minit, bool& first,
                                                    const int (&kaddr)[4], const int (&vaddr)[2], const long kvalid, const int hi, const int sb) {
  constexpr int DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  static_assert(DP == 128, "two k-steps of 64 bytes");
  const unsigned char* Ks = smem + STAGE * 2 * TILE_B;
  const unsigned char* Vs = Ks + TILE_B;
  f32x16 sacc[2];
  {
    i32x8 kf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(Ks + kaddr[2 * ks] + kb * 32 * 128);
        const u32x4 hv = *reinterpret_cast<const u32x4*>(Ks + kaddr[2 * ks + 1] + kb * 32 * 128);
        kf[ks][kb] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hv[0], (int)hv[1], (int)hv[2], (int)hv[3]};
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sacc[kb] = mfma_f8_scores(kf[ks][kb], qf[ks], ks == 0 ? minit : sacc[kb], sb);
    MTX_SCHED_GROUP(0x100, 8);
    MTX_SCHED_GROUP(0x008, 4);
  }
  if (RAGGED) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kvalid) sacc[kb][r] = -1.0e30f;
  }
  float pv[2][16];
  float tsum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { pv[kb][r] = fast_exp2(sacc[kb][r]); tsum += pv[kb][r]; }
  if (first || __any(!(tsum < 240.0f))) {
    float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
    tmax = half_max(tmax);
    const float delta = first ? tmax : fmaxf(tmax, 0.f);
    const float alpha = fast_exp2(-delta);
    M += delta;
    lsum *= alpha;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = -M;
    tsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { pv[kb][r] = fast_exp2(sacc[kb][r] - delta); tsum += pv[kb][r]; }
    first = false;
  }
  lsum += tsum;
  i32x8 pb8;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned w = 0;
      w = cvt_pk_fp8<false>(pv[kb][4 * g], pv[kb][4 * g + 1], w);
      w = cvt_pk_fp8<true>(pv[kb][4 * g + 2], pv[kb][4 * g + 3], w);
      pb8[kb * 4 + g] = (int)w;
    }
  i32x8 vf[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(Vs + vaddr[0] + d * 32 * 64);
    const u32x4 hv = *reinterpret_cast<const u32x4*>(Vs + vaddr[1] + d * 32 * 64);
    vf[d] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hv[0], (int)hv[1], (int)hv[2], (int)hv[3]};
  }
#pragma unroll
  for (int d = 0; d < DB; ++d) oacc[d] = mfma_f8_scores(vf[d], pb8, oacc[d], 127);
}
// a and b of lanes l / l ^ 32: afterwards the lower lane holds (its a, the upper lane's a), the upper lane (the lower lane's b, its b)
__device__ __forceinline__ void half_pair_exchange(uint32_t& a, uint32_t& b) {
#ifdef MTX_EMU
  const uint32_t pa = __shfl_xor(a, 32, 64), pb = __shfl_xor(b, 32, 64);
  if (emu::lane_id() >> 5) a = pb; else b = pa;
#else
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}

#define ATTN_MMA32_NAME attn_mma32_kernel
#define ATTN_MMA32_Q8 0
#define ATTN_MMA32_WIDE 0
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#undef ATTN_MMA32_Q8
#define ATTN_MMA32_NAME attn_mma32_q8_kernel
#define ATTN_MMA32_Q8 1
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#define ATTN_MMA32_DEEP 0x44
#define ATTN_MMA32_NAME attn_mma32_q8d_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#undef ATTN_MMA32_Q8
#undef ATTN_MMA32_WIDE
#define ATTN_MMA32_Q8 0
#define ATTN_MMA32_WIDE 1
#define ATTN_MMA32_NAME attn_mma32_d_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#define ATTN_MMA32_K8 1
#define ATTN_MMA32_NAME attn_mma32_k8_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#undef ATTN_MMA32_WIDE
#undef ATTN_MMA32_Q8
#define ATTN_MMA32_Q8 1
#define ATTN_MMA32_NAME attn_mma32_k8q_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#define ATTN_MMA32_V8 1
#define ATTN_MMA32_NAME attn_mma32_k8v8q_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#undef ATTN_MMA32_V8
#undef ATTN_MMA32_K8
#undef ATTN_MMA32_DEEP
#undef ATTN_MMA32_Q8

// merges the `split` key-range partials of every tail query block: O = sum_i 2^((m_i - M) c) O_i / sum_i 2^((m_i - M) c) l_i
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnParams p) {
  constexpr unsigned BANDS = AB_QB / 32;
  const unsigned tail = blockIdx.x / BANDS, band = blockIdx.x % BANDS;      // tail query block, 32-row band
  const unsigned vb = p.n_full + tail;
  const long bh = vb / p.qblocks, qb = vb % p.qblocks;
  const long b = bh / p.heads, h = bh % p.heads;
  T* O = reinterpret_cast<T*>(p.o) + b * p.o_bs + h * p.o_hs;
  for (int idx = threadIdx.x; idx < 32 * (DP / 8); idx += 256) {
    const int row = band * 32 + idx / (DP / 8), ch = idx % (DP / 8);
    const long qr = qb * AB_QB + row;
    if (qr >= p.sq) continue;
    float M = -1.0e30f;
    for (unsigned s = 0; s < p.split; ++s) { const float m = p.part_ml[((size_t)(tail * p.split + s) * AB_QB + row) * 2]; M = m > M ? m : M; }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, L = 0.f;
    for (unsigned s = 0; s < p.split; ++s) {
      const size_t base = (size_t)(tail * p.split + s) * AB_QB + row;
      const float w = fast_exp2((p.part_ml[base * 2] - M) * p.scale_log2);
      L += w * p.part_ml[base * 2 + 1];
      const float* po = p.part_o + base * DP + ch * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w * po[e];
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    *reinterpret_cast<u32x4*>(O + qr * p.o_ss + ch * 8) = pack8<T>(acc);
  }
}

// the same merge for the MX fp8 output (mtx_attn_args.q8): a thread owns 8 consecutive head-dim values of a row, 4 adjacent lanes one
// 32-wide block, 16 adjacent lanes the head's 128 columns = one scale word — the lane roles of mx_quantize_chunk
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_merge_q8_kernel(AttnParams p) {
  constexpr unsigned BANDS = AB_QB / 32;
  static_assert(DP == 128, "one scale word per head and row");
  const unsigned tail = blockIdx.x / BANDS, band = blockIdx.x % BANDS;
  const unsigned vb = p.n_full + tail;
  const long bh = vb / p.qblocks, qb = vb % p.qblocks;
  const long h = bh % p.heads;
  for (int idx = threadIdx.x; idx < 32 * (DP / 8); idx += 256) {          // 512 items: two full passes, wave-uniform
    const int row = band * 32 + idx / (DP / 8), ch = idx % (DP / 8);
    const long qr = qb * AB_QB + row;
    const bool valid = qr < p.sq;
    float M = -1.0e30f;
    for (unsigned s = 0; s < p.split; ++s) { const float m = p.part_ml[((size_t)(tail * p.split + s) * AB_QB + row) * 2]; M = m > M ? m : M; }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, L = 0.f;
    for (unsigned s = 0; s < p.split; ++s) {
      const size_t base = (size_t)(tail * p.split + s) * AB_QB + row;
      const float w = fast_exp2((p.part_ml[base * 2] - M) * p.scale_log2);
      L += w * p.part_ml[base * 2 + 1];
      const float* po = p.part_o + base * DP + ch * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w * po[e];
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = valid ? to_f32(from_f32<T>(acc[e] * inv)) : 0.f;
    unsigned w0, w1, word;
    mx_quantize_chunk(acc, ch, w0, w1, word);
    if (valid) {
      *reinterpret_cast<u32x2*>(p.q8 + (size_t)qr * p.ldq8 + h * DP + ch * 8) = u32x2{w0, w1};
      if (ch == 0) p.q8_scale[(size_t)h * p.lds_q8 + qr] = word;
    }
  }
}

static int attn_num_cus() {
  static int cus = 0;
  if (cus == 0) {
#ifdef MTX_EMU
    cus = 3;
#else
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
#endif
  }
  return cus;
}

template <typename T>
static int launch_attn_t(const AttnParams& p0, void* stream) {
  AttnParams p = p0;
  if (p.d == 128 && p.sq >= 1024 && p.sk >= 256) {       // long sequences: the 8-wave 32x32x16 kernel
    p.qblocks = (unsigned)((p.sq + AB_QB - 1) / AB_QB);
    const unsigned total = (unsigned)(p.batch * p.heads) * p.qblocks;
    // one workgroup per CU at a time: a partial last wave of `rem` query blocks leaves most of the chip idle for a
    // whole block time, so cut those blocks into `split` key ranges (fp32 partials in the caller's scratch) + merge
    const unsigned cus = (unsigned)attn_num_cus(), rem = total % cus;
    const unsigned ntiles = (unsigned)((p.sk + AB_KV - 1) / AB_KV);
    unsigned split = (rem > 0 && total > cus) ? cus / rem : 1;
    if (split > 8) split = 8;
    if (split > ntiles / 2) split = ntiles / 2;          // at least two key tiles per part
    if (split < 2 || p.part_o == nullptr) { p.n_full = total; p.split = 1; }
    else { p.n_full = total - rem; p.split = split; }
    const unsigned g = p.n_full + (total - p.n_full) * p.split;
    // What was tried against these kernels and dropped (docs/experiments.md keeps the numbers, the sources are in the history of this file up to
    // round 5): half-tile staggered wave groups, an S^T-pipelined LDS-DMA ring, 4 waves x 64 rows, two 128-query workgroups per CU, K / V by
    // LDS-DMA (-17 %), row sums on the matrix pipe (-6.7 %), a half-tile pipelined softmax (does not fit 256 registers).  What stayed: fragment
    // reads four steps ahead of the MFMAs (order pinned with sched_group_barrier; +1.5 ... 2.5 %) and 16-byte row stores (+0.3 %), identical bytes.
    if (p.kf8 != nullptr) {                      // fp8 scores (validated in attn_launch: pre-scaled q; a 16-bit output needs 16-byte rows)
      if (p.vf8 != nullptr) MTX_LAUNCH((attn_mma32_k8v8q_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);      // (validated: comes with q8)
      else if (p.q8 != nullptr) MTX_LAUNCH((attn_mma32_k8q_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
      else MTX_LAUNCH((attn_mma32_k8_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
      if (p.split > 1) {
        if (p.q8 != nullptr) MTX_LAUNCH((attn_merge_q8_kernel<T, 128>), dim3((total - p.n_full) * 8), dim3(256), 0, stream, p);
        else MTX_LAUNCH((attn_merge_kernel<T, 128>), dim3((total - p.n_full) * 8), dim3(256), 0, stream, p);
      }
      return MTX_OK;
    }
    if (p.q8 != nullptr) {
      if (p.prescaled) MTX_LAUNCH((attn_mma32_q8d_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
      else MTX_LAUNCH((attn_mma32_q8_kernel<T, 128, false>), dim3(g), dim3(512), 0, stream, p);
      if (p.split > 1) MTX_LAUNCH((attn_merge_q8_kernel<T, 128>), dim3((total - p.n_full) * 8), dim3(256), 0, stream, p);
      return MTX_OK;
    }
    const bool wide_ok = p.o_ss % 8 == 0 && p.o_hs % 8 == 0 && p.o_bs % 8 == 0 && ((size_t)p.o & 15) == 0;
    if (p.prescaled && wide_ok) MTX_LAUNCH((attn_mma32_d_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);      // the FLUX graphs' form
    else if (p.prescaled) MTX_LAUNCH((attn_mma32_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
    else MTX_LAUNCH((attn_mma32_kernel<T, 128, false>), dim3(g), dim3(512), 0, stream, p);
    if (p.split > 1) MTX_LAUNCH((attn_merge_kernel<T, 128>), dim3((total - p.n_full) * 8), dim3(256), 0, stream, p);
    return MTX_OK;
  }
  if (p.q8 != nullptr || p.kf8 != nullptr) return MTX_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)(p.batch * p.heads) * p.qblocks;
  if (p.d <= 32) MTX_LAUNCH((attn_kernel<T, 32>), dim3(grid), dim3(256), 0, stream, p);
  else if (p.d <= 64) MTX_LAUNCH((attn_kernel<T, 64>), dim3(grid), dim3(256), 0, stream, p);
  else if (p.d <= 96) MTX_LAUNCH((attn_kernel<T, 96>), dim3(grid), dim3(256), 0, stream, p);
  else MTX_LAUNCH((attn_kernel<T, 128>), dim3(grid), dim3(256), 0, stream, p);
  return MTX_OK;
}

int attn_f32_launch(const mtx_attn_args* a, void* stream, const char** err);      // f32ops.hip
int attn_launch(const mtx_attn_args* a, void* stream, const char** err) {
  if (a->dtype == MTX_F32) return attn_f32_launch(a, stream, err);
  if (!a->q || !a->k || !a->v || (!a->o && !a->q8)) { *err = "attention: null operand"; return MTX_ERR_INVALID; }
  if (a->q8 != nullptr && (!a->q8_scale || a->batch != 1 || a->d != 128 || a->sq < 1024 || a->sk < 256 || a->ldq8 % 16 || ((size_t)a->q8 & 15) || a->lds_q8 < a->sq)) {
    *err = "attention (MX fp8 output): long-sequence kernel only (d = 128, sq >= 1024, sk >= 256, batch 1), ldq8 % 16 == 0, lds_q8 >= sq"; return MTX_ERR_INVALID; }
  if ((a->q_f8 != nullptr) != (a->k_f8 != nullptr)) { *err = "attention (fp8 scores): q_f8 and k_f8 come together"; return MTX_ERR_INVALID; }
  if (a->k_f8 != nullptr && (!(a->flags & MTX_ATTN_Q_PRESCALED) || a->batch != 1 || a->d != 128 || a->sq < 1024 || a->sk < 256 || a->qf8_ss % 16 || a->kf8_ss % 16 ||
                             (((size_t)a->q_f8 | (size_t)a->k_f8) & 15) || a->qf8_ss < a->heads * 128 || a->kf8_ss < a->heads * 128 || a->qk_f8_exp < -64 || a->qk_f8_exp > 64 ||
                             (a->q8 == nullptr && (a->o_ss % 8 || a->o_hs % 8 || ((size_t)a->o & 15))))) {
    *err = "attention (fp8 scores): long-sequence kernel with pre-scaled q only (d = 128, sq >= 1024, sk >= 256, batch 1), 16-byte aligned fp8 rows of >= heads * 128 bytes, 16-byte output rows";
    return MTX_ERR_INVALID; }
  if (a->v_f8t != nullptr && (a->k_f8 == nullptr || a->q8 == nullptr || a->vf8_ld % 64 || a->vf8_ld < (a->sk + 63) / 64 * 64 || ((size_t)a->v_f8t & 15))) {
    *err = "attention (fp8 P V): needs q_f8 / k_f8 and the MX fp8 output q8; v_f8t as MTX_EW_V_F8T writes it (vf8_ld a multiple of 64 that covers sk, 16-byte aligned)";
    return MTX_ERR_INVALID; }
  if (a->d < 8 || a->d > 128 || a->d % 8) { *err = "attention: head dim must be a multiple of 8, <= 128"; return MTX_ERR_INVALID; }
  if (a->d % 4 || a->q_ss % 8 || a->k_ss % 8 || a->v_ss % 8 || a->o_ss % 4 || a->q_hs % 8 || a->k_hs % 8 || a->v_hs % 8 || a->o_hs % 4 ||
      a->q_bs % 8 || a->k_bs % 8 || a->v_bs % 8 || a->o_bs % 4) { *err = "attention: strides must keep 16-byte alignment"; return MTX_ERR_INVALID; }
  if (a->batch < 1 || a->heads < 1 || a->sq < 1 || a->sk < 1) { *err = "attention: empty problem"; return MTX_ERR_INVALID; }
  AttnParams p;
  p.q = (const unsigned char*)a->q; p.k = (const unsigned char*)a->k; p.v = (const unsigned char*)a->v; p.o = (unsigned char*)a->o;
  p.batch = a->batch; p.heads = a->heads; p.sq = a->sq; p.sk = a->sk; p.d = a->d;
  p.q_bs = a->q_bs; p.q_ss = a->q_ss; p.q_hs = a->q_hs; p.k_bs = a->k_bs; p.k_ss = a->k_ss; p.k_hs = a->k_hs;
  p.v_bs = a->v_bs; p.v_ss = a->v_ss; p.v_hs = a->v_hs; p.o_bs = a->o_bs; p.o_ss = a->o_ss; p.o_hs = a->o_hs;
  p.prescaled = (a->flags & MTX_ATTN_Q_PRESCALED) ? 1 : 0;
  p.scale_log2 = p.prescaled ? 1.0f : a->scale * 1.4426950408889634f;
  p.qblocks = (unsigned)((a->sq + AT_QB - 1) / AT_QB);
  p.n_full = 0; p.split = 1; p.part_o = nullptr; p.part_ml = nullptr;
  p.vf8 = reinterpret_cast<const unsigned char*>(a->v_f8t); p.vf8_ld = a->vf8_ld;
  p.qf8 = reinterpret_cast<const unsigned char*>(a->q_f8); p.kf8 = reinterpret_cast<const unsigned char*>(a->k_f8); p.qf8_ss = a->qf8_ss; p.kf8_ss = a->kf8_ss; p.qk_f8_exp = a->qk_f8_exp;
  p.q8 = reinterpret_cast<unsigned char*>(a->q8); p.q8_scale = reinterpret_cast<unsigned*>(a->q8_scale); p.ldq8 = a->ldq8; p.lds_q8 = a->lds_q8;
  if (a->workspace && a->workspace_bytes >= (int64_t)MTX_ATTN_WORKSPACE_BYTES) {      // 256 slots of [256][128] fp32 + [256][2] fp32
    p.part_o = reinterpret_cast<float*>(a->workspace);
    p.part_ml = p.part_o + (size_t)256 * AB_QB * 128;
  }
  if (a->dtype == MTX_BF16) return launch_attn_t<__bf16>(p, stream);
  if (a->dtype == MTX_F16) return launch_attn_t<_Float16>(p, stream);
  *err = "attention: dtype must be bf16 or f16";
  return MTX_ERR_INVALID;
}

}  // namespace mtx
