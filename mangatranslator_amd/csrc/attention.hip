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
  int schedule;                  // MTX_ATTN_SCHEDULE bits of the flags: 0 = the default kernel, s > 0 = attn_x_kernel<VAR = s - 1>
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

// Tile step with the row sums on the matrix pipe (round 5): attn_bias_tile's NOMAX form without its 32 v_add per tile — an all-ones A
// fragment against the P^T operand the P V product reads anyway puts the row sum into every row of `lacc` (one more 32x32x16 MFMA per 16
// keys: + 12.5 % matrix work for - 1/3 of the loop's vector instructions).  The running maximum is the FIRST tile's; nothing on the hot
// path looks at the probabilities' size (bf16 keeps fp32's exponent range and its relative precision), the caller checks the finished row
// sums and redoes the block with the classic form when they left the safe range.  bf16 only (f16 probabilities would saturate unseen).
template <typename T, int DP, int STAGE, bool RAGGED>
__device__ __forceinline__ void attn_bias_tile_ms(unsigned char* smem, const typename Traits<T>::v8 (&qf)[DP / 16], f32x16 (&oacc)[DP / 32], f32x16& lacc,
                                                  float& M, f32x16& minit, bool& first,
                                                  const int (&kaddr)[DP / 16], const int (&vaddr)[DP / 32], const long kvalid, const int hi) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int KS = DP / 16, DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  const unsigned char* Ks = smem + STAGE * 2 * TILE_B;
  const unsigned char* Vs = Ks + TILE_B;
  f32x16 sacc[2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8 kf = *reinterpret_cast<const v8*>(Ks + kaddr[ks] + kb * 32 * ROWB);
      if (ks == 0) Mma32Pinned<T>::set_from(sacc[kb], kf, qf[ks], minit);
      else sacc[kb] = Mma32<T>::mfma(kf, qf[ks], sacc[kb]);
    }
  if (RAGGED) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kvalid) sacc[kb][r] = -1.0e30f;
  }
  if (__builtin_expect(first, 0)) {            // (wave-uniform) M <- the first tile's maximum; O^T and the sums are still zero
    float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
    tmax = half_max(tmax);
    M += tmax;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] -= tmax;
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = -M;
    first = false;
  }
  v8 pb[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) pb[kb][r >> 3][r & 7] = from_f32<T>(fast_exp2(sacc[kb][r]));
  v8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = from_f32<T>(1.0f);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
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
      lacc = Mma32<T>::mfma(ones, pb[kb][s2], lacc);
    }
}

#define ATTN_MMA32_NAME attn_mma32_kernel
#define ATTN_MMA32_Q8 0
#define ATTN_MMA32_WIDE 0
#define ATTN_MMA32_MATSUM 0
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
#undef ATTN_MMA32_DEEP
#undef ATTN_MMA32_Q8
#undef ATTN_MMA32_WIDE
#define ATTN_MMA32_Q8 0
#define ATTN_MMA32_WIDE 1
#define ATTN_MMA32_NAME attn_mma32_w_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#define ATTN_MMA32_DEEP 0x44
#define ATTN_MMA32_NAME attn_mma32_d_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#undef ATTN_MMA32_DEEP
#undef ATTN_MMA32_MATSUM
#define ATTN_MMA32_MATSUM 1
#define ATTN_MMA32_NAME attn_mma32_ms_kernel
#include "attn_mma32_body.inc"
#undef ATTN_MMA32_NAME
#undef ATTN_MMA32_MATSUM
#undef ATTN_MMA32_WIDE
#undef ATTN_MMA32_Q8

// =====================================================================================================
// Round 5: alternative schedules of the long-sequence kernel (mtx_attn_args.flags, MTX_ATTN_SCHEDULE bits), built to be measured
// against the kernel above in one process (tools/bench_kernels.py attnx).  Same tile shape, fragment layouts, swizzles, key-split tail
// and partial format; what changes:
//   * K / V tiles arrive by LDS-DMA (buffer_load ... lds, one 1 KB piece = four 256-byte rows per wave instruction, the swizzle applied
//     to the lane's SOURCE chunk) instead of global -> registers -> ds_write: no staging registers (44 VGPRs), no LDS write instructions;
//   * AX_STAGGER: the workgroup's two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run half a tile apart, two barriers per
//     tile: while one group issues the 16 S^T MFMAs of tile u, the other runs softmax + P V of tile u - 1 — one wave's exponentials under
//     its SIMD partner's matrix work instead of both waves' softmax at the same time (MI355X_MICROARCH.md "Two waves per SIMD");
//     K(u + 1) is issued in the first half-tile phase of tile u, V(u + 1) in the second, each landing a full phase later (vmcnt(2)
//     before every barrier: all but the two pieces issued in the phase that is ending);
//   * AX_MATSUM: the row sums come from the matrix pipe (an all-ones A fragment: one more 32x32x16 MFMA per 16 keys) instead of 32
//     v_add per tile; the running maximum is taken on the first tile only and the block is REDONE with the classic online softmax
//     when a row sum ends up non-finite or above 1e30 (bf16 probabilities keep fp32's exponent range, so a stale maximum costs no precision
//     before that); bf16 only;
//   * AX_WIDE: the 16-bit rows leave as 8 x 16-byte stores per lane (lanes l and l ^ 32 trade two dwords by v_permlane32_swap)
//     instead of 16 x 8-byte stores.
constexpr int AX_STAGGER = 1, AX_MATSUM = 2, AX_WIDE = 4, AX_STAGED = 8, AX_DEEP = 16;

#ifdef MTX_EMU
#define AX_BAR(N) __syncthreads()
#else
#define AX_BAR(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)      /* fenced on both sides: MFMAs are not memory operations and would drift across the asm */
#endif

#ifdef MTX_EMU
#define AX_PIN(x) ((void)0)
#else
#define AX_PIN(x) asm volatile("" : "+v"(x))
#endif

// base ^ C as its own instruction at the point of use (never hoisted, never kept): the fragment addresses of one operand differ by an XOR
// of a few swizzle bits, and eleven loop-invariant address registers were what this kernel spilled
template <int C>
__device__ __forceinline__ int xor_here(int base) {
  if (C == 0) return base;
#ifdef MTX_EMU
  return base ^ C;
#else
  int r;
  asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(base), "v"(C));
  return r;
#endif
}

#ifdef MTX_EMU
#define AX_GROUP(mask, n) ((void)0)
#else
#define AX_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
constexpr int AX_SG_MFMA = 0x008, AX_SG_DS_READ = 0x100;

template <typename T, int DP, int STAGE, bool DEEP = false>
__device__ __forceinline__ void attn_x_scores(const unsigned char* smem, const typename Traits<T>::v8 (&qf)[DP / 16], f32x16 (&sacc)[2],
                                              const f32x16& minit, const int kaddr0) {
  typedef typename Traits<T>::v8 v8;
  constexpr int KS = DP / 16, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  const unsigned char* Ks = smem + STAGE * 2 * TILE_B;
  if constexpr (DEEP) {
    // AX_DEEP: the K fragments of four k-steps (eight 16-byte reads) are in flight before the first MFMA and stay four steps ahead, so the
    // 16 MFMAs of a tile issue back to back (512 cycles) instead of each pair waiting one LDS round trip (two reads in flight: ~1 500
    // cycles when the wave has the SIMD's matrix pipe to itself, which is what the staggered schedule gives it).  The order is pinned
    // with sched_group_barrier; the seed MFMAs are builtins here (the scheduler cannot classify inline asm).
    static_assert(KS == 8, "pipeline written for eight k-steps");
    v8 kf[KS][2];
    int ka[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      ka[ks] = ks == 0 ? kaddr0 : ks == 1 ? xor_here<1 << 5>(kaddr0) : ks == 2 ? xor_here<2 << 5>(kaddr0) : ks == 3 ? xor_here<3 << 5>(kaddr0) :
               ks == 4 ? xor_here<4 << 5>(kaddr0) : ks == 5 ? xor_here<5 << 5>(kaddr0) : ks == 6 ? xor_here<6 << 5>(kaddr0) : xor_here<7 << 5>(kaddr0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kf[ks][kb] = *reinterpret_cast<const v8*>(Ks + ka[ks] + kb * 32 * ROWB);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sacc[kb] = Mma32<T>::mfma(kf[ks][kb], qf[ks], ks == 0 ? minit : sacc[kb]);
    AX_GROUP(AX_SG_DS_READ, 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) { AX_GROUP(AX_SG_MFMA, 2); AX_GROUP(AX_SG_DS_READ, 2); }
    AX_GROUP(AX_SG_MFMA, 8);
    return;
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ka = ks == 0 ? kaddr0 : ks == 1 ? xor_here<1 << 5>(kaddr0) : ks == 2 ? xor_here<2 << 5>(kaddr0) : ks == 3 ? xor_here<3 << 5>(kaddr0) :
                   ks == 4 ? xor_here<4 << 5>(kaddr0) : ks == 5 ? xor_here<5 << 5>(kaddr0) : ks == 6 ? xor_here<6 << 5>(kaddr0) : xor_here<7 << 5>(kaddr0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8 kf = *reinterpret_cast<const v8*>(Ks + ka + kb * 32 * ROWB);
      if (ks == 0) Mma32Pinned<T>::set_from(sacc[kb], kf, qf[ks], minit);
      else sacc[kb] = Mma32<T>::mfma(kf, qf[ks], sacc[kb]);
    }
  }
}

// softmax of the S^T accumulators (seeded with minus the running maximum M, log2 units) and O^T += V^T P^T.
// !MATSUM: attn_bias_tile's NOMAX form (partial row sums flag a stale maximum).  MATSUM && !EXACT: maximum on the first tile only, sums on
// the matrix pipe (lacc: every row of the block is the row sum).  MATSUM && EXACT: maximum every tile, refreshed when it grew by > 2^8.
template <typename T, int DP, int STAGE, bool RAGGED, bool MATSUM, bool EXACT, bool DEEP = false>
__device__ __forceinline__ void attn_x_softmax_pv(const unsigned char* smem, f32x16 (&sacc)[2], f32x16 (&oacc)[DP / 32], f32x16& lacc,
                                                  float& M, float& lsum, f32x16& minit, bool& first,
                                                  const int vaddr0, const long kvalid, const int hi) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  static_assert(DB == 4, "four 32-row blocks of O^T");
  const unsigned char* Vs = smem + STAGE * 2 * TILE_B + TILE_B;
  if (RAGGED) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kvalid) sacc[kb][r] = -1.0e30f;
  }
  auto row_max = [&]() {
    float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);
    return half_max(tmax);
  };
  auto refresh = [&](float tmax) {             // M moves to the tile's maximum (first tile) or up by the growth; everything accumulated so far shrinks with it
    const float delta = first ? tmax : fmaxf(tmax, 0.f);
    const float alpha = fast_exp2(-delta);
    M += delta;
    lsum *= alpha;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    if (MATSUM) {
#pragma unroll
      for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
    }
    first = false;
    return delta;
  };
  // the seed tuple follows M as the LAST step of a refresh, after the probabilities have been recomputed: its old registers are then live
  // through the whole cold path and the new value can only land in them (redefined early, the allocator reused them for the cold path's
  // temporaries and reconciled the two paths by spilling the tuple on the HOT path: 14 scratch stores + loads per two tiles)
  auto reseed = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = -M;
    AX_PIN(minit);                             // ONE 16-register value from here on (not 16 scalars gathered in front of every MFMA that reads it)
  };
  v8 pb[2][2];
  if (MATSUM) {
    if (EXACT || __builtin_expect(first, 0)) {                      // (wave-uniform)
      const float tmax = row_max();
      if (__builtin_expect(first || __any(tmax > 8.0f), 0)) {
        const float delta = refresh(tmax);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
        reseed();
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) pb[kb][r >> 3][r & 7] = from_f32<T>(fast_exp2(sacc[kb][r]));
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
    if (__builtin_expect(first || __any(!(tsum < AttnSumLimit<T>::v)), 0)) {      // cold: the allocator should spill here, not on the hot path
      const float delta = refresh(row_max());
      tsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(sacc[kb][r] - delta);
          tsum += pv;
          pb[kb][r >> 3][r & 7] = from_f32<T>(pv);
        }
      reseed();
    }
    lsum += tsum;
  }
  v8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = from_f32<T>(1.0f);
  if constexpr (DEEP && !MATSUM) {
    // AX_DEEP: the V^T fragments of four MFMAs (eight transpose reads) ahead of the matrix pipe; same order of the 16 MFMAs as below
    int va[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) va[d] = d == 0 ? vaddr0 : d == 1 ? xor_here<1 << 6>(vaddr0) : d == 2 ? xor_here<2 << 6>(vaddr0) : xor_here<3 << 6>(vaddr0);
    v8 vf[4][DB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const unsigned char* a = Vs + va[d] + ((g >> 1) * 32 + (g & 1) * 16) * ROWB;
        const v4 lo = lds_read_tr16<T>(a);
        const v4 hv = lds_read_tr16<T>(a + 8 * ROWB);
        vf[g][d][0] = lo[0]; vf[g][d][1] = lo[1]; vf[g][d][2] = lo[2]; vf[g][d][3] = lo[3];
        vf[g][d][4] = hv[0]; vf[g][d][5] = hv[1]; vf[g][d][6] = hv[2]; vf[g][d][7] = hv[3];
      }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < DB; ++d) oacc[d] = Mma32<T>::mfma(vf[g][d], pb[g >> 1][g & 1], oacc[d]);
    AX_GROUP(AX_SG_DS_READ, 8);
#pragma unroll
    for (int i = 0; i < 12; ++i) { AX_GROUP(AX_SG_MFMA, 1); AX_GROUP(AX_SG_DS_READ, 2); }
    AX_GROUP(AX_SG_MFMA, 4);
    return;
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const int va = d == 0 ? vaddr0 : d == 1 ? xor_here<1 << 6>(vaddr0) : d == 2 ? xor_here<2 << 6>(vaddr0) : xor_here<3 << 6>(vaddr0);
        const unsigned char* a = Vs + va + (kb * 32 + s2 * 16) * ROWB;
        const v4 lo = lds_read_tr16<T>(a);
        const v4 hv = lds_read_tr16<T>(a + 8 * ROWB);
        v8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = hv[0]; vf[5] = hv[1]; vf[6] = hv[2]; vf[7] = hv[3];
        oacc[d] = Mma32<T>::mfma(vf, pb[kb][s2], oacc[d]);
      }
      if (MATSUM) lacc = Mma32<T>::mfma(ones, pb[kb][s2], lacc);
    }
}

template <typename T, int DP, int VAR>
__global__ __launch_bounds__(512) void attn_x_kernel(AttnParams p) {
  constexpr bool STAGGER = (VAR & AX_STAGGER) != 0, MATSUM = (VAR & AX_MATSUM) != 0, WIDE = (VAR & AX_WIDE) != 0, STAGED = (VAR & AX_STAGED) != 0, DEEP = (VAR & AX_DEEP) != 0;
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  constexpr int KS = DP / 16, DB = DP / 32, ROWB = DP * 2, TILE_B = AB_KV * ROWB;
  static_assert(DP == 128, "swizzles and the DMA piece map are written for 256-byte rows");
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_B];
  __shared__ int redo_flag;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int wvs = wv;
#ifndef MTX_EMU
  wvs = __builtin_amdgcn_readfirstlane(wv);
#endif
  unsigned vb, part = 0, nparts = 1;
  if (blockIdx.x < p.n_full) vb = xcd_remap(blockIdx.x, p.n_full);
  else { const unsigned i = blockIdx.x - p.n_full; vb = p.n_full + i / p.split; part = i % p.split; nparts = p.split; }
  const long bh = vb / p.qblocks, qb = vb % p.qblocks;
  const long b = bh / p.heads, h = bh % p.heads;
  const long q0 = qb * AB_QB + wv * 32;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.q_hs;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs;
  T* O = reinterpret_cast<T*>(p.o) + b * p.o_bs + h * p.o_hs;

  v8 qf[KS];
  {
    const long qr = q0 + l31;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 raw = u32x4{0u, 0u, 0u, 0u};
      if (qr < p.sq) raw = *reinterpret_cast<const u32x4*>(Q + qr * p.q_ss + ks * 16 + hi * 8);
      qf[ks] = __builtin_bit_cast(v8, raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = from_f32<T>(to_f32(qf[ks][e]) * p.scale_log2);
    }
  }

  // fragment addresses of k-step 0 / O^T block 0; the others are this ^ (ks << 5) / ^ (d << 6): the swizzle terms only touch those bits
  int kaddr0 = l31 * ROWB + ((hi ^ (l31 & 15)) << 4);
  int vaddr0;
  {
    const int ti = lane & 15, g1 = (lane >> 4) & 1;
    const int vrow = hi * 4 + (ti >> 2);
    vaddr0 = vrow * ROWB + (((2 * g1 + ((ti & 3) >> 1)) ^ ((ti >> 2) << 2)) << 4) + (ti & 1) * 8;
  }

  const long ntiles_all = (p.sk + AB_KV - 1) / AB_KV;
  const long t_begin = part * ntiles_all / nparts, t_end = (part + 1) * ntiles_all / nparts;
  const long nt = t_end - t_begin;
  const long kv_last = t_end == ntiles_all ? p.sk - (ntiles_all - 1) * AB_KV : AB_KV;

  // ---- LDS-DMA: piece j of this wave = rows 4 (wv + 8 j) .. + 3 of a tile; lane l -> row (l >> 4) of the piece, LDS chunk l & 15, which
  // holds the row's chunk (l & 15) ^ swizzle(row).  The tile position travels in the lane offset (inside the descriptor's range check:
  // rows >= sk read as zeros).
  const BufView kbuf = make_buf(K, (unsigned)(((p.sk - 1) * p.k_ss + DP) * sizeof(T)));
  const BufView vbuf = make_buf(V, (unsigned)(((p.sk - 1) * p.v_ss + DP) * sizeof(T)));
  unsigned kofs[2], vofs[2];
  auto first_tile_offsets = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned row = 4u * (unsigned)(wv + 8 * j) + ((unsigned)lane >> 4), c = (unsigned)lane & 15u;
      kofs[j] = (((unsigned)t_begin * AB_KV + row) * (unsigned)p.k_ss + ((c ^ (row & 15u)) << 3)) * (unsigned)sizeof(T);
      vofs[j] = (((unsigned)t_begin * AB_KV + row) * (unsigned)p.v_ss + ((c ^ ((row & 3u) << 2)) << 3)) * (unsigned)sizeof(T);
#ifndef MTX_EMU
      // opaque 32-bit lane values from here on: left to itself the optimiser keeps 64-bit bases + a scalar induction variable, spills the
      // bases and reloads them in front of every DMA issue — with the vmcnt(0) such a reload brings (one exposed round trip per tile)
      asm volatile("" : "+v"(kofs[j]), "+v"(vofs[j]));
#endif
    }
  };
  const unsigned k_step = (unsigned)(AB_KV * p.k_ss * sizeof(T)), v_step = (unsigned)(AB_KV * p.v_ss * sizeof(T));
  auto dma_k = [&](int stage) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { buf_load16_lds(kbuf, kofs[j], 0u, smem + stage * 2 * TILE_B + (wvs + 8 * j) * 1024); kofs[j] += k_step; AX_PIN(kofs[j]); }
  };
  auto dma_v = [&](int stage) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { buf_load16_lds(vbuf, vofs[j], 0u, smem + stage * 2 * TILE_B + TILE_B + (wvs + 8 * j) * 1024); vofs[j] += v_step; AX_PIN(vofs[j]); }
  };

  // AX_STAGED: the default kernel's transport instead — global -> registers at the start of a tile (phase A), registers -> LDS at its end (phase B)
  constexpr int CPR = DP / 8, NLD = AB_KV * CPR / 512;
  unsigned kvoff[NLD], vvoff[NLD];
  int ksw[NLD], vsw[NLD];
  u32x4 rk[NLD], rv[NLD];
  unsigned tile_next = 0;
  if (STAGED) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int idx = tid + it * 512;
      const int ch = idx % CPR, row = idx / CPR;
      kvoff[it] = (unsigned)((row * p.k_ss + ch * 8) * sizeof(T));
      vvoff[it] = (unsigned)((row * p.v_ss + ch * 8) * sizeof(T));
      ksw[it] = row * ROWB + ((ch ^ (row & 15)) << 4);
      vsw[it] = TILE_B + row * ROWB + ((ch ^ ((row & 3) << 2)) << 4);
    }
  }
  auto load_tile = [&]() {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      rk[it] = buf_load16(kbuf, kvoff[it], tile_next * k_step);
      rv[it] = buf_load16(vbuf, vvoff[it], tile_next * v_step);
    }
    ++tile_next;
  };
  auto store_tile = [&](int stage) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      *reinterpret_cast<u32x4*>(smem + stage * 2 * TILE_B + ksw[it]) = rk[it];
      *reinterpret_cast<u32x4*>(smem + stage * 2 * TILE_B + vsw[it]) = rv[it];
    }
  };
  // transport hooks of the tile loop.  One barrier per tile: `tile_begin(next stage)` ... `tile_end(next stage)`; staggered: phase A
  // (`a_begin` ... `a_end`) and phase B (`b_begin` ... `b_end`) of the same tile
  auto tile_begin = [&](int st) { if (STAGED) load_tile(); else { dma_k(st); dma_v(st); } };
  auto lds_bar = [&]() {
#ifndef MTX_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
    MTX_LDS_BARRIER();
#ifndef MTX_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto tile_end = [&](int st) { if (STAGED) { store_tile(st); lds_bar(); } else AX_BAR(0); };
  auto a_begin = [&](int st) { if (STAGED) load_tile(); else dma_k(st); };
  auto a_end = [&]() { if (STAGED) lds_bar(); else AX_BAR(2); };
  auto b_begin = [&](int st) { if (!STAGED) dma_v(st); };
  auto b_end = [&](int st) { if (STAGED) { store_tile(st); lds_bar(); } else AX_BAR(2); };

  f32x16 oacc[DB], lacc, minit, sacc[2];
  float M, lsum;
  bool first;
  if (tid == 0) redo_flag = 0;

  auto run_block = [&](auto exact_c) {
    constexpr bool EXACT = decltype(exact_c)::value;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { if (MATSUM) lacc[r] = 0.f; minit[r] = 0.f; }
    AX_PIN(minit);
    M = 0.f; lsum = 0.f; first = true;
    first_tile_offsets();
#define AX_S(ST) attn_x_scores<T, DP, ST, DEEP>(smem, qf, sacc, minit, kaddr0)
#define AX_PV(ST, RAG, KV) attn_x_softmax_pv<T, DP, ST, RAG, MATSUM, EXACT, DEEP>(smem, sacc, oacc, lacc, M, lsum, minit, first, vaddr0, KV, hi)
#define AX_PV_LAST(ST) do { if (kv_last < AB_KV) AX_PV(ST, true, kv_last); else AX_PV(ST, false, AB_KV); } while (0)
    if (STAGED) { tile_next = (unsigned)t_begin; load_tile(); store_tile(0); }
    else { dma_k(0); dma_v(0); MTX_WAIT_VMEM(); }
    __syncthreads();
    if (!STAGGER) {
      // one barrier per tile: tile u + 1 flies into the other stage behind tile u's S^T, softmax and P V
      long u = 0;
      for (; u + 2 < nt; u += 2) {
        tile_begin(1); AX_S(0); AX_PV(0, false, AB_KV); tile_end(1);
        tile_begin(0); AX_S(1); AX_PV(1, false, AB_KV); tile_end(0);
      }
      if (nt - u == 2) {
        tile_begin(1); AX_S(0); AX_PV(0, false, AB_KV); tile_end(1);
        AX_S(1); AX_PV_LAST(1);
      } else {
        AX_S(0); AX_PV_LAST(0);
      }
    } else if (wvs < 4) {
      // group 0: phase A_u = S^T(u), phase B_u = softmax + P V (u)
      long u = 0;
      for (; u + 2 < nt; u += 2) {
        a_begin(1); AX_S(0); a_end();
        b_begin(1); AX_PV(0, false, AB_KV); b_end(1);
        a_begin(0); AX_S(1); a_end();
        b_begin(0); AX_PV(1, false, AB_KV); b_end(0);
      }
      if (nt - u == 2) {
        a_begin(1); AX_S(0); a_end();
        b_begin(1); AX_PV(0, false, AB_KV); b_end(1);
        a_begin(0); AX_S(1); a_end();
        b_begin(0); AX_PV_LAST(1); b_end(0);
      } else {
        a_begin(1); AX_S(0); a_end();
        b_begin(1); AX_PV_LAST(0); b_end(1);
      }
    } else {
      // group 1, half a tile behind: phase A_u = softmax + P V (u - 1), phase B_u = S^T(u); the last P V runs after the last barrier
      a_begin(1); a_end();
      b_begin(1); AX_S(0); b_end(1);
      long u = 1;
      for (; u + 1 < nt; u += 2) {
        a_begin(0); AX_PV(0, false, AB_KV); a_end();
        b_begin(0); AX_S(1); b_end(0);
        a_begin(1); AX_PV(1, false, AB_KV); a_end();
        b_begin(1); AX_S(0); b_end(1);
      }
      if (u < nt) {
        a_begin(0); AX_PV(0, false, AB_KV); a_end();
        b_begin(0); AX_S(1); b_end(0);
        AX_PV_LAST(1);
      } else {
        AX_PV_LAST(0);
      }
    }
    MTX_WAIT_VMEM();      // pieces issued past the last tile (they land in a stage nobody reads again) retire before the stages are refilled or the wave ends
#undef AX_S
#undef AX_PV
#undef AX_PV_LAST
  };

  run_block(std::false_type());
  float l = MATSUM ? lacc[0] : half_sum(lsum);
  if (MATSUM) {
    // a row sum that left the safe range means the first tile's maximum was too stale for this block: once more, classic online softmax
    const bool bad = !(l < 1.0e30f);
    if (__any(bad) && lane == 0) redo_flag = 1;
    __syncthreads();
    if (redo_flag) {
      run_block(std::true_type());
      l = lacc[0];
    }
  }

  if (nparts > 1) {
    const unsigned slot = blockIdx.x - p.n_full;
    const int row = wv * 32 + l31;
    float* PO = p.part_o + ((size_t)slot * AB_QB + row) * DP;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {oacc[d][g * 4 + 0], oacc[d][g * 4 + 1], oacc[d][g * 4 + 2], oacc[d][g * 4 + 3]};
        *reinterpret_cast<f32x4*>(PO + d * 32 + g * 8 + hi * 4) = o;
      }
    if (hi == 0) { p.part_ml[((size_t)slot * AB_QB + row) * 2] = M / p.scale_log2; p.part_ml[((size_t)slot * AB_QB + row) * 2 + 1] = l; }
    return;
  }

  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const long qr = q0 + l31;
  if (WIDE) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        v4 x, y;
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[r] = from_f32<T>(oacc[d][gp * 8 + r] * inv); y[r] = from_f32<T>(oacc[d][gp * 8 + 4 + r] * inv); }
        u32x2 xw = __builtin_bit_cast(u32x2, x), yw = __builtin_bit_cast(u32x2, y);
        uint32_t x0 = xw[0], x1 = xw[1], y0 = yw[0], y1 = yw[1];
        half_pair_exchange(x0, y0);
        half_pair_exchange(x1, y1);
        if (qr < p.sq) *reinterpret_cast<u32x4*>(O + qr * p.o_ss + d * 32 + gp * 16 + hi * 8) = u32x4{x0, x1, y0, y1};
      }
  } else if (qr < p.sq) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(oacc[d][g * 4 + r] * inv);
        *reinterpret_cast<v4*>(O + qr * p.o_ss + d * 32 + g * 8 + hi * 4) = o;
      }
  }
}

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
    // schedules tried against this one and dropped (DESIGN.md §9 keeps the numbers): half-tile staggered wave groups, fragment reads
    // pinned 2-4 k-steps ahead, an S^T-pipelined LDS-DMA ring, 4 waves x 64 rows, two 128-query workgroups per CU — all equal or slower
    if (p.q8 != nullptr) {
      // (pre-scaled q: fragment reads four steps ahead since round 5, identical bytes; schedule 67 = the round-4 loop, for A/Bs)
      if (p.prescaled && p.schedule != 67) MTX_LAUNCH((attn_mma32_q8d_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
      else if (p.prescaled) MTX_LAUNCH((attn_mma32_q8_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
      else MTX_LAUNCH((attn_mma32_q8_kernel<T, 128, false>), dim3(g), dim3(512), 0, stream, p);
      if (p.split > 1) MTX_LAUNCH((attn_merge_q8_kernel<T, 128>), dim3((total - p.n_full) * 8), dim3(256), 0, stream, p);
      return MTX_OK;
    }
    // Measured in one process on MI355X at T = 8812, 24 heads (round 5, profiles/r05_visit_b / _c / _d_attention_*.log; default 0.827-0.835 ms):
    //   16-byte row stores on this kernel (schedule 65)  0.832 vs 0.835 ms (+0.3 %, four rounds of four)      -> the default from here on;
    //   matrix-pipe row sums on this kernel (66)          0.890 (-6.7 %: 22 % fewer vector instructions, 12.5 % more MFMAs; the kernel is
    //                                                     bound by its LDS-read -> MFMA dependency chains, not by vector issue slots);
    //   attn_x_kernel (schedules 1..64: the same loop rebuilt with the transport / schedule as template switches)
    //     K / V by LDS-DMA  0.967 (-17 %: four 1 KB pieces per wave and tile cost more issue time than 8 loads + 8 ds_write_b128);
    //     half-tile stagger of the two wave groups  0.966 with LDS-DMA (no change), 1.222 with register staging;
    //     register staging  0.983; + matrix-pipe sums 0.903; + 16-byte stores 0.970; both 0.894.
    //   fragment reads four steps ahead of the MFMAs, order pinned with sched_group_barrier (68): +9 % on attn_x (0.905 vs 0.984 / 0.991), on this kernel
    //     0.802 vs 0.814 ms (+1.5 %; T = 13 312 +2.5 %, T = 4 096 +1.3 %; four rounds, profiles/r05_visit_j_*.log), identical bytes -> the default, also for the
    //     MX-fp8-output form.  With the stagger it loses (1.069 / 1.087).
    //     Other depths (K k-steps | V MFMAs ahead: 3 | 4, 6 | 4, 4 | 6, 4 | 8, 4 | 2) all land within 0.8 % of 4 | 4 (0.812 ... 0.819 ms, profiles/r05_visit_l_*.log).
    //   softmax of a tile's second 32 keys placed in the issue gaps of the P V MFMAs of its first 32 (half-tile checks of the row sums, order pinned):
    //     2.05 ms — it does not fit the 256 registers of two waves per SIMD next to the prefetched fragments; the q fragments spill into the S^T MFMA
    //     chain (profiles/r05_visit_k_*.log).  Removed again.
    // schedule 67 = this kernel with its round-4 loop and epilogue (8-byte stores), 65 = 16-byte stores only.
    const bool wide_ok = p.o_ss % 8 == 0 && p.o_hs % 8 == 0 && p.o_bs % 8 == 0 && ((size_t)p.o & 15) == 0;
    bool launched = false;
    if ((p.schedule == 0 || p.schedule == 68) && p.prescaled && wide_ok) {      // the default since round 5: 16-byte stores + fragment reads four steps ahead
      MTX_LAUNCH((attn_mma32_d_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p); launched = true;
    } else if ((p.schedule == 65 || p.schedule == 66) && p.prescaled && wide_ok) {
      if constexpr (std::is_same<T, __bf16>::value) {
        if (p.schedule == 66) { MTX_LAUNCH((attn_mma32_ms_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p); launched = true; }
      }
      if (!launched) { MTX_LAUNCH((attn_mma32_w_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p); launched = true; }
    } else if (p.schedule > 0 && p.schedule <= 64 && p.prescaled) {
      if constexpr (std::is_same<T, __bf16>::value) {      // (f16 probabilities saturate: matrix-pipe sums would not notice a stale maximum; the measurement kernels are bf16)
        int var = p.schedule - 1;
        if (!wide_ok) var &= ~AX_WIDE;
#define MTX_AX(V) case V: MTX_LAUNCH((attn_x_kernel<T, 128, V>), dim3(g), dim3(512), 0, stream, p); break
#ifdef AX_ONLY
        switch (var) { MTX_AX(AX_ONLY); default: return MTX_ERR_INVALID; }
#else
        switch (var) { MTX_AX(0); MTX_AX(1); MTX_AX(2); MTX_AX(3); MTX_AX(4); MTX_AX(5); MTX_AX(6); MTX_AX(7);
                       MTX_AX(8); MTX_AX(9); MTX_AX(10); MTX_AX(11); MTX_AX(12); MTX_AX(13); MTX_AX(14); MTX_AX(15);
                       MTX_AX(16); MTX_AX(17); MTX_AX(20); MTX_AX(21); MTX_AX(24); MTX_AX(25); MTX_AX(28); MTX_AX(29);      // + AX_DEEP (no matrix-pipe sums)
                       default: return MTX_ERR_INVALID; }
#endif
#undef MTX_AX
        launched = true;
      }
    }
    if (!launched) {
      if (p.prescaled) MTX_LAUNCH((attn_mma32_kernel<T, 128, true>), dim3(g), dim3(512), 0, stream, p);
      else MTX_LAUNCH((attn_mma32_kernel<T, 128, false>), dim3(g), dim3(512), 0, stream, p);
    }
    if (p.split > 1) MTX_LAUNCH((attn_merge_kernel<T, 128>), dim3((total - p.n_full) * 8), dim3(256), 0, stream, p);
    return MTX_OK;
  }
  if (p.q8 != nullptr) return MTX_ERR_UNSUPPORTED;
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
  p.schedule = (a->flags >> MTX_ATTN_SCHEDULE_SHIFT) & 127;
  p.scale_log2 = p.prescaled ? 1.0f : a->scale * 1.4426950408889634f;
  p.qblocks = (unsigned)((a->sq + AT_QB - 1) / AT_QB);
  p.n_full = 0; p.split = 1; p.part_o = nullptr; p.part_ml = nullptr;
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
