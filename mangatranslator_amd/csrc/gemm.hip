// gemm.hip — C[M,N] = epilogue(A[M,K] * W[N,K]^T) on the gfx950 matrix cores.
//
// Every nn.Linear / 1x1 projection of the transformer parts of the hot path goes through this
// kernel: SAM-2.1 Hiera blocks + mask decoder (transformers Sam2Model, called at
// core/image/detection.py:505) and the FLUX MMDiT (diffusers, core/image/inpainting.py:877-887).
//
// 128x128 output tile per 4-wave workgroup, K slices of 64 staged through LDS as 128-byte rows
// XOR-swizzled by (row & 7) (conflict-free ds_read_b128, same argument as conv.hip), the next
// slice's global loads in flight behind the current slice's 32 MFMAs per wave (issue-early /
// write-late staging).  Operands are swapped (A-operand = W rows) so a lane owns 4 consecutive
// output columns; the tile is transposed through LDS and leaves as 16-byte row chunks with bias,
// activation, per-sample gate (adaLN-Zero) and residual fused.
#include "mtx_device.h"
#include <type_traits>
#include <cstdlib>
#include <mutex>
#include <unordered_set>

namespace mtx {

struct GemmParams {
  const unsigned char* a; const unsigned char* w; const float* bias; const unsigned char* res;
  const unsigned char* gate; unsigned char* c;
  const unsigned char* w2; int res_f32;      // 128-tile kernel: W_lo of a [W_hi, W_lo] weight pair (or null); residual is fp32 (fp32 output only)
  long m, n, k, lda, ldw, ldc, ldres, ldgate, a_bs, w_bs, c_bs, res_bs;
  int gate_rows_per, act; float act_param, alpha;
  int out_f32;
  unsigned tiles_m, tiles_n;
  // whole tiles [0, n_full) go to the tile kernel, tiles [n_full, tiles) to the K-slice tail; fp32 partials in `part`
  unsigned n_full; float* part;
  // K-slice tail (round 4): tiles [n_full, tiles) x `slices` equal K ranges of `slice_len` iterations; piece (slice j, tail tile t) keeps
  // its fp32 partial in slot j * rem + t of `part`; `tickets[t]` counts the finished slices of tile t (zero between launches)
  unsigned slices, slice_len; unsigned* tickets;
  // fp8 path (in_dtype == MTX_F8): MX scale planes, one uint32 (4 E8M0 bytes) per row and 128 k
  const unsigned* a_scale; const unsigned* w_scale; long lds_a, lds_w;
  int bk;           // K elements per LDS stage row: 64 (16-bit operands) or 128 (fp8); a row is 128 bytes either way
  // SwiGLU + MX-fp8 epilogue of the fp8 kernel (mtx_gemm_args.glu_*): columns >= glu_col0 are [32 a | 32 b] spans
  unsigned char* glu_q; unsigned* glu_scale; long glu_ldq, glu_lds, glu_col0;
  unsigned strip_w;         // 256-tile kernels: tile columns per strip of the workgroup -> tile map (gemm256_tile_origin); >= tiles_n: one strip
};

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int G_SMEM = (GBM + GBN) * 128;   // 32 KB; the output tile (128 x 256 B) aliases it
constexpr int G_SMEM_DUAL = (GBM + 2 * GBN) * 128;   // 48 KB: a second W stage for the W_lo half of a [W_hi, W_lo] weight pair

// DUAL (round 6, SAM-2.1 `precision = "high"`): the weights come as a pair W = W_hi + W_lo of the storage type (p.w, p.w2); both stages are
// multiplied with the SAME A fragments in LDS and add into one accumulator tile — per k-step W_hi first, then W_lo.  Round 5 ran this as a
// GEMM over K' = 2K against an operand [x | x]: twice the A traffic, a copy pass per linear to build [x | x], twice the LDS A reads.
// Epilogue (round 6): memory requests in batches, like the 256-tile kernel's since round 5 — the bias as four 16-byte loads, all gate and
// residual chunks of a thread requested before the first is used (it used to be bias dword by dword under exec branches, then per row
// gate -> wait -> residual -> wait -> store: 216 full waits on 242 loads in the ISA, profiles/r05_isa_audit.txt); the fp32-output form
// writes 16-byte vectors and takes an fp32 residual (p.res_f32: SAM's fp32 residual stream is added inside the GEMM that closes a branch).
// Same arithmetic and rounding order as before on the 16-bit path (gate product and residual sum stay two roundings).
template <typename T, bool DUAL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[DUAL ? G_SMEM_DUAL : G_SMEM];
  unsigned char* As = smem;
  unsigned char* Ws = smem + GBM * 128;
  unsigned char* W2s = smem + (GBM + GBN) * 128;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;

  const unsigned nwg = p.tiles_m * p.tiles_n;
  unsigned lin = xcd_remap(blockIdx.x, nwg);
  const long m0 = (long)(lin / p.tiles_n) * GBM;
  const long n0 = (long)(lin % p.tiles_n) * GBN;
  const long bz = blockIdx.y;
  const unsigned char* A = p.a + (size_t)bz * p.a_bs * sizeof(T);
  const unsigned char* W = p.w + (size_t)bz * p.w_bs * sizeof(T);
  const unsigned char* W2 = DUAL ? p.w2 + (size_t)bz * p.w_bs * sizeof(T) : nullptr;
  unsigned char* Cp = p.c + (size_t)bz * p.c_bs * (p.out_f32 ? 4 : sizeof(T));

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lc = tid & 7;          // 16-byte chunk within the 64-wide K slice
  const int lr = tid >> 3;         // row within a 32-row pass
  u32x4 ra[4], rw[4], rw2[DUAL ? 4 : 1];

  auto load_slice = [&](long k0) {
    const long kk = k0 + lc * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long r = lr + it * 32;
      u32x4 va = u32x4{0u, 0u, 0u, 0u}, vw = u32x4{0u, 0u, 0u, 0u}, vw2 = u32x4{0u, 0u, 0u, 0u};
      if (m0 + r < p.m && kk < p.k) va = *reinterpret_cast<const u32x4*>(A + ((size_t)(m0 + r) * p.lda + kk) * sizeof(T));
      if (n0 + r < p.n && kk < p.k) {
        vw = *reinterpret_cast<const u32x4*>(W + ((size_t)(n0 + r) * p.ldw + kk) * sizeof(T));
        if (DUAL) vw2 = *reinterpret_cast<const u32x4*>(W2 + ((size_t)(n0 + r) * p.ldw + kk) * sizeof(T));
      }
      ra[it] = va; rw[it] = vw;
      if (DUAL) rw2[it] = vw2;
    }
  };

  const long nk = (p.k + GBK - 1) / GBK;
  load_slice(0);
  for (long kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = lr + it * 32;
      *reinterpret_cast<u32x4*>(As + r * 128 + ((lc ^ (r & 7)) << 4)) = ra[it];
      *reinterpret_cast<u32x4*>(Ws + r * 128 + ((lc ^ (r & 7)) << 4)) = rw[it];
      if (DUAL) *reinterpret_cast<u32x4*>(W2s + r * 128 + ((lc ^ (r & 7)) << 4)) = rw2[it];
    }
    __syncthreads();
    if (kt + 1 < nk) load_slice((kt + 1) * GBK);
    const long rem = p.k - kt * GBK;
    const int nks = rem >= 64 ? 2 : (int)((rem + 31) / 32);
    for (int ks = 0; ks < nks; ++ks) {
      const int cch = ks * 4 + q;
      v8 wf[4], af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + l15;
        wf[j] = *reinterpret_cast<const v8*>(Ws + r * 128 + ((cch ^ (l15 & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + l15;
        af[i] = *reinterpret_cast<const v8*>(As + r * 128 + ((cch ^ (l15 & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Traits<T>::mfma(wf[j], af[i], acc[i][j]);
      if (DUAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = wn * 64 + j * 16 + l15;
          wf[j] = *reinterpret_cast<const v8*>(W2s + r * 128 + ((cch ^ (l15 & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Traits<T>::mfma(wf[j], af[i], acc[i][j]);
      }
    }
  }

  // lane holds C[m = m0 + wm*64 + i*16 + l15][n = n0 + wn*64 + j*16 + q*4 + r]
  // bias of this lane's 16 columns: four 16-byte loads in one batch when the vector is aligned and N a multiple of 4 (then a quad is valid or not as a whole)
  f32x4 bv[4];
  const bool bias_vec = p.bias != nullptr && ((size_t)p.bias & 15) == 0 && (p.n & 3) == 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long nb = n0 + wn * 64 + j * 16 + q * 4;
    if (bias_vec) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.bias + (nb < p.n ? nb : 0));
      bv[j] = nb < p.n ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (p.bias != nullptr && nb + r < p.n) ? p.bias[nb + r] : 0.f;
    }
  }
  if (p.out_f32) {
    float* Cf = reinterpret_cast<float*>(Cp);
    const bool vec = (p.n & 3) == 0 && (p.ldc & 3) == 0 && ((size_t)Cf & 15) == 0 && p.gate == nullptr &&
                     (p.res == nullptr || (p.res_f32 && (p.ldres & 3) == 0 && (p.res_bs & 3) == 0 && ((size_t)p.res & 15) == 0));
    if (vec) {
      // 16-byte stores (a lane's four consecutive columns); the fp32 residual of all sixteen quads requested before the first is used
      const float* Rf = reinterpret_cast<const float*>(p.res);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long m = m0 + wm * 64 + i * 16 + l15;
        const long mc = m < p.m ? m : p.m - 1;
        f32x4 rq4[4];
        if (Rf != nullptr) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const long n = n0 + wn * 64 + j * 16 + q * 4;
            rq4[j] = *reinterpret_cast<const f32x4*>(Rf + (size_t)bz * p.res_bs + (size_t)mc * p.ldres + (n < p.n ? n : 0));
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long n = n0 + wn * 64 + j * 16 + q * 4;
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = apply_act(acc[i][j][r] * p.alpha + bv[j][r], p.act, p.act_param);
            if (Rf != nullptr) v[r] += rq4[j][r];
          }
          if (m < p.m && n < p.n) *reinterpret_cast<f32x4*>(Cf + (size_t)m * p.ldc + n) = v;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + wm * 64 + i * 16 + l15;
      if (m >= p.m) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long n = n0 + wn * 64 + j * 16 + q * 4 + r;
          if (n < p.n) {
            float v = acc[i][j][r] * p.alpha + bv[j][r];
            v = apply_act(v, p.act, p.act_param);
            if (p.gate) v *= to_f32(reinterpret_cast<const T*>(p.gate)[(size_t)(m / p.gate_rows_per) * p.ldgate + n]);
            if (p.res) v += p.res_f32 ? reinterpret_cast<const float*>(p.res)[(size_t)bz * p.res_bs + (size_t)m * p.ldres + n]
                                      : to_f32(reinterpret_cast<const T*>(p.res)[(size_t)bz * p.res_bs + (size_t)m * p.ldres + n]);
            Cf[(size_t)m * p.ldc + n] = v;
          }
        }
      }
    }
    return;
  }

  __syncthreads();
  unsigned char* outs = smem;    // [128 rows][16 chunks of 16 B], chunk ^= (row & 15)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int chunk = wn * 8 + j * 2 + (q >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + l15;
      v4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(apply_act(acc[i][j][r] * p.alpha + bv[j][r], p.act, p.act_param));
      *reinterpret_cast<v4*>(outs + row * 256 + ((chunk ^ (row & 15)) << 4) + ((q & 1) << 3)) = o;
    }
  }
  const int oc = tid & 15;
  const bool vec_ok = (p.ldc % 8 == 0) && (!p.res || p.ldres % 8 == 0) && (!p.gate || p.ldgate % 8 == 0);
  const long n = n0 + oc * 8;
  const bool full = vec_ok && (n + 8 <= p.n);
  const T* G = reinterpret_cast<const T*>(p.gate);
  const T* R = reinterpret_cast<const T*>(p.res);
  const long gper = p.gate_rows_per > 0 ? p.gate_rows_per : 1;
  // the thread's eight gate and eight residual chunks, requested before the barrier (clamped addresses, no branches): one round trip
  u32x4 gq[8], rq[8];
  if (full && (G != nullptr || R != nullptr)) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const long m = m0 + (tid >> 4) + it * 16;
      const long mc = m < p.m ? m : p.m - 1;
      if (G != nullptr) gq[it] = *reinterpret_cast<const u32x4*>(G + (size_t)(mc / gper) * p.ldgate + n);
      if (R != nullptr) rq[it] = *reinterpret_cast<const u32x4*>(R + (size_t)bz * p.res_bs + (size_t)mc * p.ldres + n);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = (tid >> 4) + it * 16;
    const long m = m0 + row;
    if (m >= p.m || n >= p.n) continue;
    u32x4 raw = *reinterpret_cast<const u32x4*>(outs + row * 256 + ((oc ^ (row & 15)) << 4));
    if (G != nullptr || R != nullptr || !full) {
#pragma clang fp contract(off)
      float f[8];
      unpack8<T>(raw, f);
      if (full) {
        if (G) { float g8[8]; unpack8<T>(gq[it], g8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = f[e] * g8[e]; }
        if (R) { float r8[8]; unpack8<T>(rq[it], r8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = f[e] + r8[e]; }
        raw = pack8<T>(f);
      } else {
        const size_t goff = (size_t)(m / gper) * p.ldgate + n;
        const size_t roff = (size_t)bz * p.res_bs + (size_t)m * p.ldres + n;
        T* Co = reinterpret_cast<T*>(Cp);
        for (int e = 0; e < 8 && n + e < p.n; ++e) {
          float v = f[e];
          if (G) v = v * to_f32(G[goff + e]);
          if (R) v = v + to_f32(R[roff + e]);
          Co[(size_t)m * p.ldc + n + e] = from_f32<T>(v);
        }
        continue;
      }
    }
    *reinterpret_cast<u32x4*>(Cp + ((size_t)m * p.ldc + n) * sizeof(T)) = raw;
  }
}

// =====================================================================================================
// Large-problem kernel: 256 x 256 x 64 tiles, 8 waves (2 per SIMD), v_mfma_f32_32x32x16, operands brought in
// by LDS-DMA (global_load_lds, 16 B per lane) into two 64 KB stages — no staging registers, one barrier per
// K tile, the next tile's DMA in flight behind the current tile's 32 MFMAs per wave.
//   * LDS image of a stage: A rows then W rows, 128 B (64 k) per row, linear in DMA order; the XOR swizzle
//     (chunk ^ ((row >> 1) & 7), conflict-free for the 32-row ds_read_b128 pattern of the 32x32 MFMA) is
//     applied to the SOURCE address of each lane.
//   * wave (wm, wn) of a 2 x 4 grid owns 128 rows (m) x 64 columns (n): 4 x 2 accumulator blocks of 32 x 32.
//     Operands are swapped (A-operand = W rows), so a lane ends up with 4 consecutive n of one row m.
//   * epilogue: bias / activation in registers -> the wave's own 16 KB LDS region (swizzled) -> 16-byte row
//     chunks, gate (.) and residual (+) fused, 128-byte row segments to HBM.
//   * workgroup -> tile map: XCD-contiguous, then groups of 4 tile rows x 8 tile columns so the 32 workgroups
//     resident on one XCD share A and W panels through that XCD's L2.
constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_STAGE = (G2_BM + G2_BN) * 128;      // 64 KB

// Epilogue of the 256-tile kernels, run by the 8 MFMA waves (wv = 0..7).
// acc[i][j][r]: m = m0 + wm*128 + i*32 + l31, n = n0 + wn*64 + j*32 + 8*(r>>2) + 4*hi + (r&3)
// Round 5 (late): memory requests in batches.  Read as ISA, the form above asked for its bias values one dword at a time under per-element
// exec branches, eight groups each behind an `s_waitcnt vmcnt(0)`, and for the gate and residual chunks of a row inside that row's
// iteration — gate, wait, residual, wait, store, sixteen times per wave: ~40 dependent memory round trips per tile on a CU that has
// nothing else resident (one 128 KB workgroup).  Here the bias leaves as eight 16-byte loads in one batch, and — the accumulators being
// dead once staged — all sixteen gate and residual chunks of a lane are requested (clamped addresses, no branches) before the first is
// used.  Same arithmetic, same rounding order (the gate product and the residual sum stay two roundings): identical bytes to the one-at-a-time
// form of rounds 1-5a (compared on hardware in round 5, profiles/r05_visit_q_...log; that form is in this file's history).
template <typename T, int ACT>
__device__ __forceinline__ void gemm256_epilogue(const GemmParams& p, f32x16 (&acc)[4][2], unsigned char* smem, T* Cp,
                                                 long m0, long n0, long bz, int wv, int lane) {
  typedef typename Traits<T>::v4 v4;
  const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 2, wn = wv & 3;
  unsigned char* outs = smem + wv * 16384;          // [128 rows m][8 chunks of 16 B], chunk ^= (row & 7)
  f32x4 bv[2][4];
  if (p.bias != nullptr && ((size_t)p.bias & 15) == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long n = n0 + wn * 64 + j * 32 + g * 8 + hi * 4;          // N % 8 == 0: four valid values or none
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.bias + (n < p.n ? n : 0));
        bv[j][g] = n < p.n ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long n = n0 + wn * 64 + j * 32 + g * 8 + hi * 4 + r;
          bv[j][g][r] = (p.bias != nullptr && n < p.n) ? p.bias[n] : 0.f;
        }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = j * 32 + g * 8 + hi * 4;         // local n of this lane's 4 values
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + l31;
        v4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(apply_act_t<ACT>(acc[i][j][g * 4 + r] * p.alpha + bv[j][g][r], p.act, p.act_param));
        *reinterpret_cast<v4*>(outs + row * 128 + ((((nl >> 3)) ^ (row & 7)) << 4) + ((nl & 4) << 1)) = o;
      }
    }
  const T* G = reinterpret_cast<const T*>(p.gate);
  const T* R = reinterpret_cast<const T*>(p.res);
  const int oc = lane & 7;
  const long nn = n0 + wn * 64 + oc * 8;
  const bool n_ok = nn < p.n;
  const long nc = n_ok ? nn : 0;
  u32x4 gq[16], rq[16];
  if (G != nullptr) {
    // gate row = m / rows_per: one division per wave when a 128-row span crosses at most one boundary (FLUX: rows_per = a stream's length)
    // (the span's first row clamped like the rows themselves: a wave whose whole span lies beyond M must not compute a gate row past the last one)
    const unsigned per = (unsigned)p.gate_rows_per, mb = (unsigned)(m0 + wm * 128 < p.m ? m0 + wm * 128 : p.m - 1);
    const unsigned q0 = mb / per, next = (q0 + 1) * per;
    const bool one_step = per >= 128;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const long m = m0 + wm * 128 + it * 8 + (lane >> 3);
      const unsigned mc = (unsigned)(m < p.m ? m : p.m - 1);
      const unsigned gr = one_step ? q0 + (mc >= next ? 1u : 0u) : mc / per;
      gq[it] = *reinterpret_cast<const u32x4*>(G + (size_t)gr * p.ldgate + nc);
    }
  }
  if (R != nullptr) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const long m = m0 + wm * 128 + it * 8 + (lane >> 3);
      const long mc = m < p.m ? m : p.m - 1;
      rq[it] = *reinterpret_cast<const u32x4*>(R + (size_t)bz * p.res_bs + (size_t)mc * p.ldres + nc);
    }
  }
  // a wave only re-reads its own region: no workgroup barrier needed, just its own LDS writes
#ifdef MTX_EMU
  emu::wave_sync();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + (lane >> 3);
    const long m = m0 + wm * 128 + row;
    u32x4 raw = *reinterpret_cast<const u32x4*>(outs + row * 128 + ((oc ^ (row & 7)) << 4));
    if (G != nullptr || R != nullptr) {
#pragma clang fp contract(off)
      float f[8];
      unpack8<T>(raw, f);
      if (G != nullptr) { float g8[8]; unpack8<T>(gq[it], g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * g8[e]; }
      if (R != nullptr) { float r8[8]; unpack8<T>(rq[it], r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] + r8[e]; }
      raw = pack8<T>(f);
    }
    if (m < p.m && n_ok) *reinterpret_cast<u32x4*>(Cp + (size_t)m * p.ldc + nn) = raw;
  }
}

// tile `lin` of the grouped order -> its origin.  The tile plane is cut into STRIPS of `strip_w` tile columns; inside a strip the order
// is groups of 4 tile rows x the strip's columns, column-major inside a group (the 32 workgroups resident on an XCD work on ~8 columns
// x 4 rows that walk K in step and share 12 panels through that XCD's L2).  The XCD-contiguous workgroup order (xcd_remap) hands each
// XCD a run of tiles/8 consecutive tiles, i.e. a block of about (tiles/8 / strip_w) rows x strip_w columns (one strip = rounds 1-5: 4.4
// rows x all columns).  strip_w is chosen per launch (gemm256_choose_strip: what strips buy, and what they do not).
__device__ __forceinline__ void gemm256_tile_origin(const GemmParams& p, unsigned lin, long& m0, long& n0) {
  const unsigned GM = 4;
  const unsigned per_strip = p.tiles_m * p.strip_w;
  const unsigned strip = lin / per_strip, col0 = strip * p.strip_w;
  const unsigned sw = (p.tiles_n - col0) < p.strip_w ? (p.tiles_n - col0) : p.strip_w;      // the last strip may be narrower
  const unsigned rest = lin - strip * per_strip;
  const unsigned per_group = GM * sw;
  const unsigned group = rest / per_group, first_m = group * GM;
  const unsigned gsz = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
  m0 = (long)(first_m + (rest % per_group) % gsz) * G2_BM;
  n0 = (long)(col0 + (rest % per_group) / gsz) * G2_BN;
}

#ifdef MTX_EMU
#define G2_BAR() __syncthreads()
#else
#define G2_BAR() asm volatile("s_barrier" ::: "memory")
#endif

// The ping-pong K loop with descriptor DMA over iterations [kbeg, kend) of one 256 x 256 tile (used by the full-tile kernel and by
// the K-slice tail): barrier timeline B0, B1, ...: group 0 (waves 0-3) runs  L(k) B C(k) B  per k-step, group 1 the same one
// barrier later, so one group's load segment (fragment reads + DMA issue) coincides with the other's 8 MFMAs.
//   * DMA: per piece a loop-invariant 32-bit lane offset into a buffer descriptor over the valid bytes of A / W (rows past M / N
//     are zero-filled by the range check), the K position as the wave-uniform soffset, the LDS base (M0) from scalar arithmetic:
//     no VALU instruction per piece.
//   * the eight pieces of tile kt+1 are issued 3 / 3 / 2 / 0 over the load segments of tile kt; a wave waits for its own fragment
//     reads BEFORE it joins the barrier that ends a load segment, so the stage tile kt-1 occupied is free from L(kt, 0) on;
//   * every wave drains its DMA (vmcnt(0)) before the barrier that precedes group 0's first reads of tile kt+1: group 0 at the end
//     of C(kt, 3), group 1 at the end of L(kt, 3).
// The caller provides a workgroup barrier between two calls (the stages are reused).
template <typename T>
__device__ __forceinline__ void gemm256_pp_buf_loop(const GemmParams& p, unsigned char* smem, const T* A, const T* W, long m0, long n0,
                                                    long kbeg, long kend, f32x16 (&acc)[4][2]) {
  typedef typename Traits<T>::v8 v8;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3, grp = wv >> 2;
  int wvs = wv;
#ifndef MTX_EMU
  wvs = __builtin_amdgcn_readfirstlane(wv);
#endif
  // descriptors are TILE-relative (base = first row of the tile, records = the tile's valid rows): lane offsets stay far below
  // 4 GB whatever the matrix size, and rows past M / N fall outside the records (zero fill)
  const long mrows = p.m - m0 < G2_BM ? p.m - m0 : G2_BM, nrows = p.n - n0 < G2_BN ? p.n - n0 : G2_BN;
  const BufView abuf = make_buf(A + (size_t)m0 * p.lda, (unsigned)((((size_t)mrows - 1) * p.lda + p.k) * sizeof(T)));
  const BufView wbuf = make_buf(W + (size_t)n0 * p.ldw, (unsigned)((((size_t)nrows - 1) * p.ldw + p.k) * sizeof(T)));
  unsigned voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {           // piece i of wave wv fills LDS rows (i*8 + wv)*8 .. +7 of a stage; rows 0..255 are A, 256..511 W
    const int row = (i * 8 + wv) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[i] = i < 4 ? (unsigned)(((size_t)row * p.lda + c * 8) * sizeof(T)) : (unsigned)(((size_t)(row - G2_BM) * p.ldw + c * 8) * sizeof(T));
  }
  auto piece = [&](int i, int stage, long k0) {
    buf_load16_lds(i < 4 ? abuf : wbuf, voff[i], (unsigned)(k0 * sizeof(T)), smem + stage * G2_STAGE + (i * 8 + wvs) * 1024);
  };
  // fragment addresses: one VGPR per (stage, k-step, operand) — rows i*32 / j*32 further down share the swizzle term, so they are
  // immediate offsets (4096 per 32 rows) and the loop carries no address arithmetic (every VALU instruction of a load segment
  // takes an issue slot from the other group's MFMAs)
  const int ar0 = wm * 128 + l31, wr0 = G2_BM + wn * 64 + l31;
  int aaddr[2][4], waddr[2][4];
#pragma unroll
  for (int st_ = 0; st_ < 2; ++st_)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aaddr[st_][ks] = st_ * G2_STAGE + ar0 * 128 + (((2 * ks + hi) ^ ((ar0 >> 1) & 7)) << 4);
      waddr[st_][ks] = st_ * G2_STAGE + wr0 * 128 + (((2 * ks + hi) ^ ((wr0 >> 1) & 7)) << 4);
    }

#pragma unroll
  for (int i = 0; i < 8; ++i) piece(i, (int)(kbeg & 1), kbeg * G2_BK);
  MTX_WAIT_VMEM();
  __syncthreads();
  if (grp == 1) G2_BAR();
  auto tile = [&](auto stage_c, long kt) {
    constexpr int S = decltype(stage_c)::value;
    const bool more = kt + 1 < kend;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v8 af[4], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const v8*>(smem + waddr[S][ks] + j * 4096);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8*>(smem + aaddr[S][ks] + i * 4096);
      if (more && ks < 3) {
        const long k0 = (kt + 1) * G2_BK;
        const int first = ks * 3, cnt = ks < 2 ? 3 : 2;       // 3/3/2/0 (4/4/0/0 measured equal, 2/3/3/0 2-6 % slower)
#pragma unroll
        for (int i = 0; i < cnt; ++i) piece(first + i, 1 - S, k0);
      }
#ifndef MTX_EMU
      // the tile's LAST fragment reads finish before the barrier (from the next segment on the other group's DMA overwrites this
      // stage); the others are waited for after it, where the wait does not delay the other group's load segment (measured equal)
      if (ks == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      if (ks == 3 && grp == 1) MTX_WAIT_VMEM();
      G2_BAR();
#ifndef MTX_EMU
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma32<T>::mfma(wf[j], af[i], acc[i][j]);
#ifndef MTX_EMU
      __builtin_amdgcn_s_setprio(0);
#endif
      if (ks == 3 && grp == 0) MTX_WAIT_VMEM();
      G2_BAR();
    }
  };
  typedef std::integral_constant<int, 0> St0;
  typedef std::integral_constant<int, 1> St1;
  long kt = kbeg;
  if (kt & 1) { tile(St1(), kt); ++kt; }                      // tile kt lives in stage kt & 1
  for (; kt + 1 < kend; kt += 2) { tile(St0(), kt); tile(St1(), kt + 1); }
  if (kt < kend) tile(St0(), kt);
  if (grp == 0) G2_BAR();
}

// 256-tile kernel over whole tiles: the K loop above, then the epilogue.
template <typename T, int ACT>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the launch covers tiles [0, gridDim.x) of the grouped order (all of them, or the whole waves when the rest goes
  // to the K-slice tail kernel)
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  long m0, n0;
  gemm256_tile_origin(p, lin, m0, n0);
  const long bz = blockIdx.y;
  const T* A = reinterpret_cast<const T*>(p.a) + (size_t)bz * p.a_bs;
  const T* W = reinterpret_cast<const T*>(p.w) + (size_t)bz * p.w_bs;
  T* Cp = reinterpret_cast<T*>(p.c) + (size_t)bz * p.c_bs;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gemm256_pp_buf_loop<T>(p, smem, A, W, m0, n0, 0, p.k / G2_BK, acc);
  __syncthreads();
  gemm256_epilogue<T, ACT>(p, acc, smem, Cp, m0, n0, bz, wv, lane);
}

// =====================================================================================================
// fp8 path (BASELINE.json config 5, "CDNA4 fp8 MFMA path"): the same 256 x 256 tile, LDS image, DMA plan, ping-pong barrier
// timeline and epilogue, with OCP e4m3 operands and MX block scales on v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per
// instruction, twice the bf16 rate).  A 128-byte LDS row now holds 128 k, so a stage is one K = 128 tile = two k-steps; each
// k-step is split into two segments of 4 MFMAs (64 cycles each) so a segment still occupies the matrix pipe for 256 cycles and
// the bytes the DMA and the fragment reads move per pipe cycle are those of the 16-bit kernel.
//   * operand layout of the instruction, measured with tools/probes/mx_probe.hip (profiles/r02_mx_probe.txt): lane (row, half h) holds
//     k = 16 h + 0..15 in its first 16 bytes and k = 32 + 16 h + 0..15 in its second 16 bytes; the 32 k of MX block b (b = 0, 1) take
//     their scale from lane (row, b).  So lane (row l31, half hi) of k-step ks reads logical chunks 4 ks + hi (block 2 ks) and
//     4 ks + 2 + hi (block 2 ks + 1) and SUPPLIES the scale of block 2 ks + hi: byte 2 ks of (scale word >> 8 hi).
//   * the scale words of tile kt+1 (4 A rows + 2 W rows per lane) are fetched with plain global loads in the first load segment
//     of tile kt and first touched after the vmcnt(0) that ends the tile.
typedef __attribute__((ext_vector_type(8))) int i32x8;

// in place (acc += A B), as `asm volatile`: the builtin form is free to move, and the optimiser sinks the MFMAs of several segments
// past the barriers and load segments (all their fragments stay live: 150-300 spilled registers).  Every operand is in registers long
// before the instruction (fragments behind an lgkmcnt(0), scales shifted at the top of the tile); consecutive MFMAs use different
// accumulators, and a tile's second k-step re-enters an accumulator 3 x 64 cycles after the first wrote it.
template <int OPSEL>
__device__ __forceinline__ void mfma_mx_f8(f32x16& c, i32x8 a, i32x8 b, unsigned sa, unsigned sb) {
#ifdef MTX_EMU
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, (int)sa, OPSEL, (int)sb);
#else
  if (OPSEL == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
  else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,1,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
#endif
}

__device__ __forceinline__ void gemm256_f8_loop(const GemmParams& p, unsigned char* smem, const unsigned char* A, const unsigned char* W,
                                                long m0, long n0, long kbeg, long kend, f32x16 (&acc)[4][2]) {
  constexpr int BK = 128;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3, grp = wv >> 2;
  int wvs = wv;
#ifndef MTX_EMU
  wvs = __builtin_amdgcn_readfirstlane(wv);
#endif
  const long mrows = p.m - m0 < G2_BM ? p.m - m0 : G2_BM, nrows = p.n - n0 < G2_BN ? p.n - n0 : G2_BN;
  const BufView abuf = make_buf(A + (size_t)m0 * p.lda, (unsigned)(((size_t)mrows - 1) * p.lda + p.k));
  const BufView wbuf = make_buf(W + (size_t)n0 * p.ldw, (unsigned)(((size_t)nrows - 1) * p.ldw + p.k));
  unsigned voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (i * 8 + wv) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[i] = i < 4 ? (unsigned)((size_t)row * p.lda + c * 16) : (unsigned)((size_t)(row - G2_BM) * p.ldw + c * 16);
  }
  auto piece = [&](int i, int stage, long k0) {
    buf_load16_lds(i < 4 ? abuf : wbuf, voff[i], (unsigned)k0, smem + stage * G2_STAGE + (i * 8 + wvs) * 1024);
  };
  const int ar0 = wm * 128 + l31, wr0 = G2_BM + wn * 64 + l31;
  int aaddr[2][2][2], waddr[2][2][2];
#pragma unroll
  for (int st_ = 0; st_ < 2; ++st_)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ch = 4 * ks + 2 * e + hi;
        aaddr[st_][ks][e] = st_ * G2_STAGE + ar0 * 128 + ((ch ^ ((ar0 >> 1) & 7)) << 4);
        waddr[st_][ks][e] = st_ * G2_STAGE + wr0 * 128 + ((ch ^ ((wr0 >> 1) & 7)) << 4);
      }
  // scale rows of this lane (clamped: rows past M / N hold zero data, any finite scale will do)
  const unsigned* sap[4];
  const unsigned* swp[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { long r = m0 + wm * 128 + i * 32 + l31; r = r < p.m ? r : p.m - 1; sap[i] = p.a_scale + r; }
#pragma unroll
  for (int j = 0; j < 2; ++j) { long r = n0 + wn * 64 + j * 32 + l31; r = r < p.n ? r : p.n - 1; swp[j] = p.w_scale + r; }
  unsigned sa_raw[4], sw_raw[2];
  auto load_scales = [&](long kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sa_raw[i] = sap[i][kt * p.lds_a];
#pragma unroll
    for (int j = 0; j < 2; ++j) sw_raw[j] = swp[j][kt * p.lds_w];
  };
  load_scales(kbeg);
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(i, (int)(kbeg & 1), kbeg * BK);
  MTX_WAIT_VMEM();
  __syncthreads();
  if (grp == 1) G2_BAR();
  auto tile = [&](auto stage_c, long kt) {
    constexpr int S = decltype(stage_c)::value;
    const bool more = kt + 1 < kend;
    unsigned sa[4], sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) sa[i] = sa_raw[i] >> (8 * hi);
#pragma unroll
    for (int j = 0; j < 2; ++j) sw[j] = sw_raw[j] >> (8 * hi);
#ifndef MTX_EMU
    // materialise the shifted words here, in the load segment: they must not trail into the slots right in front of the asm MFMAs
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(sa[i]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(sw[j]));
#endif
    i32x8 wf[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = s >> 1, ih = s & 1;
      i32x8 af[2];
      if (ih == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + waddr[S][ks][0] + j * 4096);
          const u32x4 hi4 = *reinterpret_cast<const u32x4*>(smem + waddr[S][ks][1] + j * 4096);
          wf[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        }
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * ih + ii;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + aaddr[S][ks][0] + i * 4096);
        const u32x4 hi4 = *reinterpret_cast<const u32x4*>(smem + aaddr[S][ks][1] + i * 4096);
        af[ii] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
      }
      if (more && s == 0) load_scales(kt + 1);
      if (more && s < 3) {
        const long k0 = (kt + 1) * BK;
        const int first = s * 3, cnt = s < 2 ? 3 : 2;
#pragma unroll
        for (int i = 0; i < cnt; ++i) piece(first + i, 1 - S, k0);
      }
#ifndef MTX_EMU
      if (s == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      if (s == 3 && grp == 1) MTX_WAIT_VMEM();
      G2_BAR();
#ifndef MTX_EMU
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);      // the scheduler otherwise sinks the MFMAs of several segments past the barriers (fragments stay live: spills)
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 2 * ih + ii;
          if (ks == 0) mfma_mx_f8<0>(acc[i][j], wf[j], af[ii], sw[j], sa[i]); else mfma_mx_f8<2>(acc[i][j], wf[j], af[ii], sw[j], sa[i]);
        }
#ifndef MTX_EMU
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (s == 3 && grp == 0) MTX_WAIT_VMEM();
      G2_BAR();
    }
  };
  typedef std::integral_constant<int, 0> St0;
  typedef std::integral_constant<int, 1> St1;
  long kt = kbeg;
  if (kt & 1) { tile(St1(), kt); ++kt; }
  for (; kt + 1 < kend; kt += 2) { tile(St0(), kt); tile(St1(), kt + 1); }
  if (kt < kend) tile(St0(), kt);
#ifndef MTX_EMU
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results are read by VALU code next (the asm form hides them from the hazard recogniser)
#endif
  if (grp == 0) G2_BAR();
}

template <typename T, int ACT>
__global__ __launch_bounds__(512) void gemm256_f8_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  long m0, n0;
  gemm256_tile_origin(p, lin, m0, n0);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gemm256_f8_loop(p, smem, p.a, p.w, m0, n0, 0, p.k / 128, acc);
  __syncthreads();
  gemm256_epilogue<T, ACT>(p, acc, smem, reinterpret_cast<T*>(p.c), m0, n0, 0, wv, lane);
}

// The gated-MLP form of the same kernel (mtx_gemm_args.glu_*): tiles whose columns lie at or beyond glu_col0 hold, per wave, the "a"
// half of 32 outputs in acc[.][0] and the "b" half in acc[.][1] (the caller interleaved W's rows in 32-row runs), so silu(a) * b and its
// MX quantisation happen in registers: a lane owns 16 of a row's 32 outputs, lane ^ 32 the other 16 — one xor-shuffle for the block's
// maximum, one exchange of two dwords so that each lane ends up with 16 contiguous bytes — and the 16-bit [T, 2 * hidden] projection
// (FLUX.2-Klein: 314 MB written and read back per linear) never exists.  Arithmetic as in quant.hip's MTX_QUANT_SWIGLU on the rounded
// 16-bit projection: bit-identical to GEMM -> SwiGLU-quantiser.
template <typename T>
__global__ __launch_bounds__(512) void gemm256_f8_glu_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  long m0, n0;
  gemm256_tile_origin(p, lin, m0, n0);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gemm256_f8_loop(p, smem, p.a, p.w, m0, n0, 0, p.k / 128, acc);
  __syncthreads();
  if (n0 < p.glu_col0) {                                  // (workgroup-uniform: glu_col0 is a multiple of the tile width)
    gemm256_epilogue<T, MTX_ACT_NONE>(p, acc, smem, reinterpret_cast<T*>(p.c), m0, n0, 0, wv, lane);
    return;
  }
  const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 2, wn = wv & 3;
  const long span = (n0 - p.glu_col0) / 64 + wn;          // 32 outputs: columns 32 * span .. of the quantised matrix
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + wm * 128 + i * 32 + l31;
    float h[16];
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float a = to_f32(from_f32<T>(acc[i][0][e] * p.alpha)), b = to_f32(from_f32<T>(acc[i][1][e] * p.alpha));
      h[e] = to_f32(from_f32<T>(div_by_1p(a, 1.f + __expf(-a)) * b));
      const float v = fabsf(h[e]);
      amax = v > amax ? v : amax;
    }
    { const float o = __shfl_xor(amax, 32, 64); amax = o > amax ? o : amax; }
    const float r = amax * (1.0f / 448.0f);
    const unsigned u = __builtin_bit_cast(unsigned, r);
    int eb = (int)((u >> 23) & 0xff) + ((u & 0x7fffffu) ? 1 : 0);
    eb = amax == 0.f ? 127 : (eb < 1 ? 1 : (eb > 253 ? 253 : eb));
    const float inv = __builtin_bit_cast(float, (unsigned)(254 - eb) << 23);
    unsigned w[4];                                         // w[g]: outputs 8 g + 4 hi + 0..3 of the span
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float q4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { float v = h[g * 4 + e] * inv; q4[e] = v > 448.f ? 448.f : (v < -448.f ? -448.f : v); }
      w[g] = 0;
      w[g] = cvt_pk_fp8<false>(q4[0], q4[1], w[g]); w[g] = cvt_pk_fp8<true>(q4[2], q4[3], w[g]);
    }
    // lane (hi = 0) keeps bytes 0..15 of the span (g = 0, 1), lane ^ 32 bytes 16..31 (g = 2, 3): swap the two dwords the other needs
    const unsigned s0 = hi ? w[0] : w[2], s1 = hi ? w[1] : w[3];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 32, 64), r1 = (unsigned)__shfl_xor((int)s1, 32, 64);
    const u32x4 out = hi ? u32x4{r0, w[2], r1, w[3]} : u32x4{w[0], r0, w[1], r1};
    if (m < p.m) {
      *reinterpret_cast<u32x4*>(p.glu_q + (size_t)m * p.glu_ldq + span * 32 + hi * 16) = out;
      if (hi == 0) reinterpret_cast<unsigned char*>(p.glu_scale + (size_t)(span >> 2) * p.glu_lds + m)[span & 3] = (unsigned char)eb;
    }
  }
}


// ---- K-slice tail with a last-arriver fix-up (round 4) ---------------------------------------------------------------------
// With one 256 x 256 tile per CU at a time, `rem = tiles % CUs` left-over tiles keep rem CUs busy for a whole tile time while the
// others idle (FLUX proj_out: 420 tiles on 256 CUs = 1.64 waves billed as 2).  Rounds 2-3 dealt the left-over tiles' iterations out
// evenly (stream-K) and summed the pieces in a second launch: no two units ever read the same operand rows at the same time — 2.45 GB
// fetched per 8812 x 3072 x 15360 launch at 9 TB/s (PMC, profiles/r03_pmc_traffic.json), the tail ran at the fabric's limit, not the
// matrix pipe's (MFMA busy 0.36) — and the whole construction was SLOWER than not splitting at all (0.768 vs 0.725 ms, same process).
// Here every left-over tile is cut at the SAME K positions into `slices` pieces; piece p = (slice p / rem, tile p % rem), and the
// XCD-contiguous workgroup order gives an XCD a run of neighbouring tiles of ONE slice, which walk K in step and share their A / W
// panels through that XCD's L2 like the whole-tile kernel's workgroups do.  A piece publishes its fp32 partial write-through (sc1) in
// the lane-contiguous order of its accumulators (1 KB per wave store), drains, and draws a ticket of its tile; the piece that draws the
// last one acquires, adds the tile's partials IN SLICE ORDER (the sum does not depend on who came last) and runs the whole-tile
// epilogue.  No merge launch; tickets return to zero.  Measured (same process, profiles/r04_visit_c_slice_sweep.log): 8812x3072x15360
// 0.700 ms (stream-K + merge 0.768, unsplit 0.725), 8300x3072x12288 0.533 (0.572 / 0.561), 512x3072x12288 0.068 (0.081 / 0.198).
template <typename T, bool F8, int ACT>
__global__ __launch_bounds__(512) void gemm256_slice_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned tiles = p.tiles_m * p.tiles_n, rem = tiles - p.n_full;
  const unsigned piece = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned sl = piece / rem, ti = piece % rem;
  const long nk = p.k / p.bk;
  const long kbeg = (long)sl * p.slice_len;
  const long kend = kbeg + p.slice_len < nk ? kbeg + p.slice_len : nk;
  long m0, n0;
  gemm256_tile_origin(p, p.n_full + ti, m0, n0);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (kend > kbeg) {
    if (F8) gemm256_f8_loop(p, smem, p.a, p.w, m0, n0, kbeg, kend, acc);
    else gemm256_pp_buf_loop<T>(p, smem, reinterpret_cast<const T*>(p.a), reinterpret_cast<const T*>(p.w), m0, n0, kbeg, kend, acc);
  }
  // slot layout: [wave][block = (i, j, g)][lane] x 16 bytes
  const size_t slot_floats = (size_t)G2_BM * G2_BN;
  const size_t my_off = ((size_t)wv * 32 * 64 + lane) * 4;
  float* mine = p.part + (size_t)piece * slot_floats + my_off;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
        agent_store16(mine + (size_t)((i * 2 + j) * 4 + g) * 256, v);
      }
  agent_drain();
  __syncthreads();
  if (tid == 0) last_flag = agent_ticket(p.tickets + ti) + 1 == p.slices;
  __syncthreads();
  if (!last_flag) return;
  if (tid == 0) { agent_acquire(); agent_store(p.tickets + ti, 0u); }      // zero again for the next launch (a launch boundary away)
  __syncthreads();
  // sum of the tile's partials in slice order — ALL of them from memory, this piece's own included (it was just published; the
  // accumulators are dead from here on, which leaves the registers for 16 blocks x 16 bytes per lane in flight per slice: the first
  // version kept its own piece in registers and fetched block by block, 32 dependent round trips, ~60 us per fix-up on hardware)
  const float* base = p.part + (size_t)ti * slot_floats + my_off;
  const size_t slice_stride = (size_t)rem * slot_floats;
  constexpr int CH = 16;
#pragma unroll
  for (int c0 = 0; c0 < 32; c0 += CH) {
    f32x4 run[CH];
#pragma unroll
    for (int b = 0; b < CH; ++b) run[b] = *reinterpret_cast<const f32x4*>(base + (size_t)(c0 + b) * 256);
    for (unsigned s2 = 1; s2 < p.slices; ++s2) {
      f32x4 v[CH];
#pragma unroll
      for (int b = 0; b < CH; ++b) v[b] = *reinterpret_cast<const f32x4*>(base + (size_t)s2 * slice_stride + (size_t)(c0 + b) * 256);
#pragma unroll
      for (int b = 0; b < CH; ++b) run[b] += v[b];
    }
#pragma unroll
    for (int b = 0; b < CH; ++b) {
      const int blk = c0 + b, i = blk >> 3, j = (blk >> 2) & 1, g = blk & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][g * 4 + e] = run[b][e];
    }
  }
  __syncthreads();
  gemm256_epilogue<T, ACT>(p, acc, smem, reinterpret_cast<T*>(p.c), m0, n0, 0, wv, lane);
}

static int gemm_num_cus() {
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

template <typename T, bool F8>
static void launch_gemm256_tiles(const GemmParams& p, dim3 grid, void* stream) {
#define MTX_G256(ACTV) do { if (F8) MTX_LAUNCH((gemm256_f8_kernel<T, ACTV>), grid, dim3(512), 0, stream, p); \
                            else MTX_LAUNCH((gemm256_kernel<T, ACTV>), grid, dim3(512), 0, stream, p); } while (0)
  switch (p.act) {
    case MTX_ACT_NONE: MTX_G256(MTX_ACT_NONE); break;
    case MTX_ACT_SILU: MTX_G256(MTX_ACT_SILU); break;
    case MTX_ACT_GELU_TANH: MTX_G256(MTX_ACT_GELU_TANH); break;
    default: MTX_G256(-1); break;
  }
#undef MTX_G256
}

// measured on MI355X (tools/bench_kernels.py, FLUX shapes, random data): the ping-pong loop with descriptor DMA and the 3/3/2/0
// piece spread runs 1119-1341 TFLOP/s in bf16; the schedules it replaced (flat-address ping-pong, one-barrier, K = 32 ring, wave
// specialised DMA) were 2-15 % behind on every shape and are gone (docs/experiments.md keeps the numbers).
static thread_local int g_last_split[3] = {0, 0, 0};           // whole tiles, K slices, tail pieces of this thread's last 256-tile launch
void gemm_last_split(int* out) { out[0] = g_last_split[0]; out[1] = g_last_split[1]; out[2] = g_last_split[2]; }

constexpr long G2_TICKET_BYTES = 4096;                                                   // 1024 tickets at the end of the workspace
constexpr long G2_MAX_PIECES = ((long)MTX_GEMM_WORKSPACE_BYTES - G2_TICKET_BYTES) / ((long)G2_BM * G2_BN * 4);

// K slices for `r` tiles of `nk` iterations on `cus` CUs, in units of one K iteration of the main loop (~1.5 us bf16, ~1.4 us fp8).
// Calibrated on MI355X with forced slice counts (tools/bench_kernels.py gemmsN, profiles/r04_visit_c_slice_sweep.log): a round of pieces
// costs its iterations + ~6 (pipeline fill, publishing the partial, ticket); the partials' write + read-back costs ~0.07 per piece
// (512 KB of traffic each); the last arriver's epilogue and its serial reads ~14 + 0.4 s^2.  Measured / modelled tails: 8812x3072x15360
// s = 3: 223 / 224 iterations (s = 2: 281 / 291; one more whole wave: 240); 8300x3072x12288 s = 3: 173 / 187, s = 4: 224 / 221 (whole
// wave: 192); 512x3072x12288 over s = 2, 3, 4, 6, 8, 10: 120, 88, 75, 66, 71, 80 / 121, 93, 81, 77, 83, 97 (unsplit: 192).
// Returns 0 when no slicing beats `limit` (the cost of the alternative).
static unsigned gemm256_choose_slices(unsigned r, long nk, unsigned cus, double limit) {
  unsigned best = 0; double best_cost = limit;
  for (unsigned s = 2; s <= 16; ++s) {
    if ((long)r * s > G2_MAX_PIECES) break;
    const long len = (nk + s - 1) / s;
    if (len < 4) break;
    if ((nk + len - 1) / len != (long)s) continue;           // a slice would be empty
    const long rounds = ((long)r * s + cus - 1) / cus;
    const double cost = (double)rounds * (double)(len + 6) + 0.07 * (double)r * s + 14.0 + 0.4 * (double)s * s;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

// The K-slice tail counts arrivals in `tickets` and relies on finding zeros (the last arriver of a tile puts its ticket back to zero).  A
// workspace the library has not seen before is cleared here, in front of its first slice launch on that stream — callers that allocate the
// workspace themselves (hipMalloc does not clear) no longer have to know (include/mtx_hip.h still asks for a zeroed workspace:
// an address that is freed and handed out again with other contents is not seen as new).
static void gemm256_tickets_ready(unsigned* tickets, void* stream) {
  static std::mutex mu;
  static std::unordered_set<const void*> seen;
  bool fresh;
  { std::lock_guard<std::mutex> lock(mu); fresh = seen.insert(tickets).second; }
  if (fresh) zero_words_async(tickets, (size_t)G2_TICKET_BYTES, stream);
}

template <typename T, bool F8>
static void launch_gemm256_slices(const GemmParams& p, unsigned pieces, void* stream) {
  gemm256_tickets_ready(p.tickets, stream);
#define MTX_G256S(ACTV) MTX_LAUNCH((gemm256_slice_kernel<T, F8, ACTV>), dim3(pieces), dim3(512), 0, stream, p)
  switch (p.act) {
    case MTX_ACT_NONE: MTX_G256S(MTX_ACT_NONE); break;
    case MTX_ACT_SILU: MTX_G256S(MTX_ACT_SILU); break;
    case MTX_ACT_GELU_TANH: MTX_G256S(MTX_ACT_GELU_TANH); break;
    default: MTX_G256S(-1); break;
  }
#undef MTX_G256S
}

// measured on MI355X (tools/bench_kernels.py, FLUX shapes, random data): the ping-pong loop with descriptor DMA and the 3/3/2/0
// piece spread runs 1119-1341 TFLOP/s in bf16; the schedules it replaced (flat-address ping-pong, one-barrier, K = 32 ring, wave
// specialised DMA) were 2-15 % behind on every shape and are gone (docs/experiments.md keeps the numbers).
// Strip width of the tile map.  Measured on MI355X (same process, identical bytes: profiles/r06_visit_a / _b logs): with ONE strip every
// XCD walks all tile columns from column 0 at the same time, i.e. eight L2s pull the same W panel over the fabric at once; two strips put
// XCDs 0-3 and 4-7 on different halves of W.  The MX-fp8 kernel (half the time per byte of the 16-bit one) gains on the wide problems —
// 8512 x 27648 x 3072 0.635 -> 0.606 ms, 8000 x 18432 x 3072 0.420 -> 0.371 — and is flat up to N = 9216; the bf16 kernel is flat
// everywhere (+- 0.5 %) and its K-sliced shapes lose 2 % (their tail tiles move); four or eight strips and a per-group column rotation
// were within noise of two.  (The fabric-side read volume goes UP with strips — 1.26 -> 1.58 GB per launch, A panels re-fetched per
// strip: it was never the volume, it was eight XCDs asking for the same lines.)  MTX_GEMM_STRIPS = n forces a count (tests, tools/bench_kernels.py).
static unsigned gemm256_choose_strip(unsigned tiles_n, bool f8) {
  const char* e = getenv("MTX_GEMM_STRIPS");
  const int want = e ? atoi(e) : 0;
  const unsigned xc = want > 0 ? (unsigned)want : ((f8 && tiles_n >= 64) ? 2u : 1u);
  const unsigned w = (tiles_n + xc - 1) / xc;
  return w < 1 ? 1 : w;
}

template <typename T, bool F8>
static void launch_gemm256(const GemmParams& p0, dim3 grid, void* stream, bool force, bool nosplit, unsigned forced_slices) {
  GemmParams p = p0;
  p.strip_w = gemm256_choose_strip(p.tiles_n, F8);
  const unsigned tiles = p.tiles_m * p.tiles_n, cus = (unsigned)gemm_num_cus(), rem = tiles % cus;
  const long nk = p.k / p.bk;
  const bool can = p.part != nullptr && cus <= 320 && grid.y == 1 && !nosplit;
  // left-over tiles of the last wave: K slices when they finish clearly before a whole extra tile time would
  if (can && tiles > cus && rem > 0 && nk >= (force ? 4 : (F8 ? 32 : 64))) {
    unsigned s = gemm256_choose_slices(rem, nk, cus, 0.97 * (double)nk);
    if (forced_slices >= 2 && (long)rem * forced_slices <= G2_MAX_PIECES && (long)forced_slices * 2 <= nk) s = forced_slices;
    if (s) {
      p.n_full = tiles - rem; grid.x = p.n_full;
      launch_gemm256_tiles<T, F8>(p, grid, stream);
      p.slices = s; p.slice_len = (unsigned)((nk + s - 1) / s);
      launch_gemm256_slices<T, F8>(p, rem * s, stream);
      g_last_split[0] = (int)p.n_full; g_last_split[1] = (int)s; g_last_split[2] = (int)(rem * s);
      return;
    }
  }
  // few tiles but a long K (FLUX text-stream ff2: 24 tiles x 192 iterations): slices over the whole problem
  if (can && tiles * 2 <= cus && nk >= (F8 ? 64 : 128)) {
    unsigned s = gemm256_choose_slices(tiles, nk, cus, 0.9 * (double)nk);
    if (forced_slices >= 2 && (long)tiles * forced_slices <= G2_MAX_PIECES && (long)forced_slices * 2 <= nk) s = forced_slices;
    if (s) {
      p.n_full = 0; p.slices = s; p.slice_len = (unsigned)((nk + s - 1) / s);
      launch_gemm256_slices<T, F8>(p, tiles * s, stream);
      g_last_split[0] = 0; g_last_split[1] = (int)s; g_last_split[2] = (int)(tiles * s);
      return;
    }
  }
  launch_gemm256_tiles<T, F8>(p, grid, stream);
  g_last_split[0] = (int)tiles; g_last_split[1] = 1; g_last_split[2] = 0;
}

int gemm_f32_launch(const mtx_gemm_args* a, void* stream, const char** err);      // f32ops.hip
int gemm_launch(const mtx_gemm_args* a, void* stream, const char** err) {
  if (!a->a || !a->w || !a->c) { *err = "gemm: null operand"; return MTX_ERR_INVALID; }
  if (a->m < 1 || a->n < 1 || a->k < 1) { *err = "gemm: empty problem"; return MTX_ERR_INVALID; }
  if (a->dtype == MTX_F32) return gemm_f32_launch(a, stream, err);      // fp32 operands: the vector-ALU path (SAM's high-precision mask decoder)
  const bool f8 = a->in_dtype == MTX_F8;
  if (a->glu_q != nullptr && !f8) { *err = "gemm: the SwiGLU epilogue exists on the fp8 kernel only"; return MTX_ERR_INVALID; }
  if (a->in_dtype != 0 && !f8 && a->in_dtype != a->dtype) { *err = "gemm: in_dtype must be 0, dtype or MTX_F8"; return MTX_ERR_INVALID; }
  if (a->k % 8 || a->lda % 8 || a->ldw % 8) { *err = "gemm: K, lda, ldw must be multiples of 8 (16-byte chunks)"; return MTX_ERR_INVALID; }
  // every kernel below takes its operand rows as 16-byte vector loads / LDS-DMA pieces and stores 16-byte row chunks: a base pointer that
  // is only element-aligned (a view into a packed buffer) is refused here instead of faulting or tearing there
  if (((size_t)a->a | (size_t)a->w) & 15) { *err = "gemm: a and w must be 16-byte aligned"; return MTX_ERR_INVALID; }
  if ((((size_t)a->c | (size_t)a->res | (size_t)a->gate) & 15) && a->n % 8 == 0 && a->ldc % 8 == 0) { *err = "gemm: c, res and gate must be 16-byte aligned when rows are stored as 16-byte chunks"; return MTX_ERR_INVALID; }
  if (a->batch > 1 && (a->a_bstride % 8 || a->w_bstride % 8 || a->c_bstride % 8 || a->res_bstride % 8)) { *err = "gemm: batch strides must be multiples of 8"; return MTX_ERR_INVALID; }
  if (a->out_dtype != a->dtype && a->out_dtype != MTX_F32) { *err = "gemm: out_dtype must equal dtype or be f32"; return MTX_ERR_INVALID; }
  if (a->res_dtype != 0 && a->res_dtype != a->dtype && !(a->res_dtype == MTX_F32 && a->out_dtype == MTX_F32)) { *err = "gemm: res_dtype must be 0, dtype, or f32 together with an f32 output"; return MTX_ERR_INVALID; }
  if (a->w_lo != nullptr && (f8 || ((size_t)a->w_lo & 15))) { *err = "gemm: w_lo (the low half of a weight pair) needs 16-bit operands and a 16-byte aligned pointer"; return MTX_ERR_INVALID; }
  GemmParams p;
  p.a = (const unsigned char*)a->a; p.w = (const unsigned char*)a->w; p.bias = a->bias;
  p.res = (const unsigned char*)a->res; p.gate = (const unsigned char*)a->gate; p.c = (unsigned char*)a->c;
  p.w2 = (const unsigned char*)a->w_lo; p.res_f32 = (a->res != nullptr && a->res_dtype == MTX_F32) ? 1 : 0;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.ldres = a->ldres; p.ldgate = a->ldgate;
  p.a_bs = a->a_bstride; p.w_bs = a->w_bstride; p.c_bs = a->c_bstride; p.res_bs = a->res_bstride;
  p.gate_rows_per = a->gate_rows_per > 0 ? a->gate_rows_per : 1;
  p.act = a->act; p.act_param = a->act_param; p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
  p.out_f32 = a->out_dtype == MTX_F32 && a->dtype != MTX_F32;
  p.n_full = 0; p.slices = 1; p.slice_len = 0;
  p.part = (a->workspace && a->workspace_bytes >= (int64_t)MTX_GEMM_WORKSPACE_BYTES) ? reinterpret_cast<float*>(a->workspace) : nullptr;
  p.tickets = p.part ? reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(a->workspace) + MTX_GEMM_WORKSPACE_BYTES - G2_TICKET_BYTES) : nullptr;
  p.a_scale = reinterpret_cast<const unsigned*>(a->a_scale); p.w_scale = reinterpret_cast<const unsigned*>(a->w_scale);
  p.lds_a = a->lds_a; p.lds_w = a->lds_w;
  p.glu_q = nullptr; p.glu_scale = nullptr; p.glu_ldq = p.glu_lds = p.glu_col0 = 0;
  p.strip_w = 0x7fffffffu;
  p.bk = f8 ? 128 : G2_BK;
  p.tiles_m = (unsigned)((a->m + GBM - 1) / GBM);
  p.tiles_n = (unsigned)((a->n + GBN - 1) / GBN);
  const long batch = a->batch > 0 ? a->batch : 1;
  const bool force = (a->flags & MTX_GEMM_FORCE_TILE256) != 0, nosplit = (a->flags & MTX_GEMM_NO_SPLIT) != 0;
  const unsigned forced_slices = ((unsigned)a->flags >> 8) & 0xffu;
  // large, aligned problems: the 256 x 256 LDS-DMA kernel (needs whole K tiles and 16-byte rows everywhere)
  const long t256 = ((a->m + G2_BM - 1) / G2_BM) * ((a->n + G2_BN - 1) / G2_BN) * batch;
  const bool vec = a->n % 8 == 0 && a->ldc % 8 == 0 && (!a->res || a->ldres % 8 == 0) && (!a->gate || a->ldgate % 8 == 0) && a->c_bstride % 8 == 0;
  const size_t esz = f8 ? 1 : 2;
  const bool fits32 = ((size_t)G2_BM * a->lda + a->k) * esz < (1ull << 32) && ((size_t)G2_BN * a->ldw + a->k) * esz < (1ull << 32);      // one tile's rows under a descriptor
  if (f8) {
    if (a->dtype != MTX_BF16 && a->dtype != MTX_F16) { *err = "gemm(fp8): dtype (epilogue / output type) must be bf16 or f16"; return MTX_ERR_INVALID; }
    if (!a->a_scale || !a->w_scale || a->lds_a < a->m || a->lds_w < a->n) { *err = "gemm(fp8): scale planes missing or lds_a / lds_w shorter than the row count"; return MTX_ERR_INVALID; }
    if (a->k % 128 || a->lda % 16 || a->ldw % 16 || !vec || p.out_f32 || batch != 1 || !fits32) { *err = "gemm(fp8): needs K % 128 == 0, lda / ldw % 16 == 0, N / ldc % 8 == 0, 16-bit output, batch 1"; return MTX_ERR_INVALID; }
    p.tiles_m = (unsigned)((a->m + G2_BM - 1) / G2_BM);
    p.tiles_n = (unsigned)((a->n + G2_BN - 1) / G2_BN);
    dim3 g2(p.tiles_m * p.tiles_n, 1);
    if (a->glu_q != nullptr) {
      if (!a->glu_scale || a->glu_col0 < 0 || a->glu_col0 % G2_BN || a->glu_col0 >= a->n || (a->n - a->glu_col0) % G2_BN || a->glu_ldq % 16 ||
          ((size_t)a->glu_q & 15) || a->glu_lds < a->m || a->bias || a->gate || a->res || a->act != MTX_ACT_NONE) {
        *err = "gemm(fp8, SwiGLU epilogue): needs glu_col0 and n - glu_col0 multiples of 256, glu_ldq % 16 == 0, glu_lds >= m, and no bias / gate / res / act";
        return MTX_ERR_INVALID;
      }
      p.glu_q = reinterpret_cast<unsigned char*>(a->glu_q); p.glu_scale = reinterpret_cast<unsigned*>(a->glu_scale);
      p.glu_ldq = a->glu_ldq; p.glu_lds = a->glu_lds; p.glu_col0 = a->glu_col0;
      p.strip_w = gemm256_choose_strip(p.tiles_n, true);
      if (a->dtype == MTX_BF16) MTX_LAUNCH((gemm256_f8_glu_kernel<__bf16>), g2, dim3(512), 0, stream, p);
      else MTX_LAUNCH((gemm256_f8_glu_kernel<_Float16>), g2, dim3(512), 0, stream, p);
      return MTX_OK;
    }
    if (a->dtype == MTX_BF16) launch_gemm256<__bf16, true>(p, g2, stream, force, nosplit, forced_slices); else launch_gemm256<_Float16, true>(p, g2, stream, force, nosplit, forced_slices);
    return MTX_OK;
  }
  // with the descriptor-DMA loop the 256-tile kernel wins from ~24 tiles up even though most CUs idle (512x9216x3072: 55 vs 68 us,
  // 1024x4608x1152: 24.5 vs 32.7 us); below that the 128-tile kernel's extra parallelism pays.
  // (short K — SAM's 576-wide stage — keeps the old threshold: the tile prologue / epilogue dominates there, encoder 12.5 vs 11.8 ms)
  // and so do shapes that would pad a 256-tile row or column by more than 10 % (SAM's N = 576: 3 columns for 2.25)
  const long t256m = (a->m + G2_BM - 1) / G2_BM, t256n = (a->n + G2_BN - 1) / G2_BN;
  const bool snug = a->m * 10 >= t256m * G2_BM * 9 && a->n * 10 >= t256n * G2_BN * 9;
  const long min_tiles = force ? 1 : ((a->k >= 1024 && snug) ? 24 : 160);
  const bool few_long = a->workspace != nullptr && batch == 1 && t256 * 2 <= gemm_num_cus() && a->k / G2_BK >= 128 && a->m >= 256;
  if (!p.out_f32 && a->w_lo == nullptr && a->k % G2_BK == 0 && vec && fits32 && (t256 >= min_tiles || few_long) && (a->dtype == MTX_BF16 || a->dtype == MTX_F16)) {
    p.tiles_m = (unsigned)((a->m + G2_BM - 1) / G2_BM);
    p.tiles_n = (unsigned)((a->n + G2_BN - 1) / G2_BN);
    dim3 g2(p.tiles_m * p.tiles_n, (unsigned)batch);
    if (a->dtype == MTX_BF16) launch_gemm256<__bf16, false>(p, g2, stream, force, nosplit, forced_slices); else launch_gemm256<_Float16, false>(p, g2, stream, force, nosplit, forced_slices);
    return MTX_OK;
  }
  dim3 grid(p.tiles_m * p.tiles_n, (unsigned)batch);
  if (a->dtype != MTX_BF16 && a->dtype != MTX_F16) { *err = "gemm: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  if (p.w2 != nullptr) {
    if (a->dtype == MTX_BF16) MTX_LAUNCH((gemm_kernel<__bf16, true>), grid, dim3(256), 0, stream, p);
    else MTX_LAUNCH((gemm_kernel<_Float16, true>), grid, dim3(256), 0, stream, p);
  } else if (a->dtype == MTX_BF16) MTX_LAUNCH((gemm_kernel<__bf16, false>), grid, dim3(256), 0, stream, p);
  else MTX_LAUNCH((gemm_kernel<_Float16, false>), grid, dim3(256), 0, stream, p);
  return MTX_OK;
}

}  // namespace mtx
