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

namespace mtx {

struct GemmParams {
  const unsigned char* a; const unsigned char* w; const float* bias; const unsigned char* res;
  const unsigned char* gate; unsigned char* c;
  long m, n, k, lda, ldw, ldc, ldres, ldgate, a_bs, w_bs, c_bs, res_bs;
  int gate_rows_per, act; float act_param, alpha;
  int out_f32;
  unsigned tiles_m, tiles_n;
  // stream-K tail: tiles [n_full, tiles) are cut into `units` equal runs of K iterations; fp32 partials in `part`
  unsigned n_full, units; float* part;
  int abl;          // MTX_GEMM_ABL: timing ablations of the 256-tile kernel (1 = no DMA after the first tile, 2 = DMA burst after the barrier, 3 = two pieces per k-step)
};

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int G_SMEM = (GBM + GBN) * 128;   // 32 KB; the output tile (128 x 256 B) aliases it

template <typename T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G_SMEM];
  unsigned char* As = smem;
  unsigned char* Ws = smem + GBM * 128;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;

  const unsigned nwg = p.tiles_m * p.tiles_n;
  unsigned lin = xcd_remap(blockIdx.x, nwg);
  const long m0 = (long)(lin / p.tiles_n) * GBM;
  const long n0 = (long)(lin % p.tiles_n) * GBN;
  const long bz = blockIdx.y;
  const unsigned char* A = p.a + (size_t)bz * p.a_bs * sizeof(T);
  const unsigned char* W = p.w + (size_t)bz * p.w_bs * sizeof(T);
  unsigned char* Cp = p.c + (size_t)bz * p.c_bs * (p.out_f32 ? 4 : sizeof(T));

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lc = tid & 7;          // 16-byte chunk within the 64-wide K slice
  const int lr = tid >> 3;         // row within a 32-row pass
  u32x4 ra[4], rw[4];

  auto load_slice = [&](long k0) {
    const long kk = k0 + lc * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long r = lr + it * 32;
      u32x4 va = u32x4{0u, 0u, 0u, 0u}, vw = u32x4{0u, 0u, 0u, 0u};
      if (m0 + r < p.m && kk < p.k) va = *reinterpret_cast<const u32x4*>(A + ((size_t)(m0 + r) * p.lda + kk) * sizeof(T));
      if (n0 + r < p.n && kk < p.k) vw = *reinterpret_cast<const u32x4*>(W + ((size_t)(n0 + r) * p.ldw + kk) * sizeof(T));
      ra[it] = va; rw[it] = vw;
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
    }
  }

  // lane holds C[m = m0 + wm*64 + i*16 + l15][n = n0 + wn*64 + j*16 + q*4 + r]
  if (p.out_f32) {
    float* Cf = reinterpret_cast<float*>(Cp);
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
            float v = acc[i][j][r] * p.alpha + (p.bias ? p.bias[n] : 0.f);
            v = apply_act(v, p.act, p.act_param);
            if (p.gate) v *= to_f32(reinterpret_cast<const T*>(p.gate)[(size_t)(m / p.gate_rows_per) * p.ldgate + n]);
            if (p.res) v += to_f32(reinterpret_cast<const T*>(p.res)[(size_t)bz * p.res_bs + (size_t)m * p.ldres + n]);
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
    const long nb = n0 + wn * 64 + j * 16 + q * 4;
    float b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (p.bias != nullptr && nb + r < p.n) ? p.bias[nb + r] : 0.f;
    const int chunk = wn * 8 + j * 2 + (q >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + l15;
      v4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(apply_act(acc[i][j][r] * p.alpha + b[r], p.act, p.act_param));
      *reinterpret_cast<v4*>(outs + row * 256 + ((chunk ^ (row & 15)) << 4) + ((q & 1) << 3)) = o;
    }
  }
  __syncthreads();
  const int oc = tid & 15;
  const bool vec_ok = (p.ldc % 8 == 0) && (!p.res || p.ldres % 8 == 0) && (!p.gate || p.ldgate % 8 == 0);
  for (int it = 0; it < 8; ++it) {
    const int row = (tid >> 4) + it * 16;
    const long m = m0 + row, n = n0 + oc * 8;
    if (m >= p.m || n >= p.n) continue;
    u32x4 raw = *reinterpret_cast<const u32x4*>(outs + row * 256 + ((oc ^ (row & 15)) << 4));
    const bool full = vec_ok && (n + 8 <= p.n);
    if (p.gate != nullptr || p.res != nullptr || !full) {
      float f[8];
      unpack8<T>(raw, f);
      const T* G = reinterpret_cast<const T*>(p.gate);
      const T* R = reinterpret_cast<const T*>(p.res);
      const size_t goff = (size_t)(m / (p.gate_rows_per > 0 ? p.gate_rows_per : 1)) * p.ldgate + n;
      const size_t roff = (size_t)bz * p.res_bs + (size_t)m * p.ldres + n;
      if (full) {
        if (G) { float g8[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(G + goff), g8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= g8[e]; }
        if (R) { float r8[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(R + roff), r8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += r8[e]; }
        raw = pack8<T>(f);
      } else {
        T* Co = reinterpret_cast<T*>(Cp);
        for (int e = 0; e < 8 && n + e < p.n; ++e) {
          float v = f[e];
          if (G) v *= to_f32(G[goff + e]);
          if (R) v += to_f32(R[roff + e]);
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
template <typename T, int ACT>
__device__ __forceinline__ void gemm256_epilogue(const GemmParams& p, f32x16 (&acc)[4][2], unsigned char* smem, T* Cp,
                                                 long m0, long n0, long bz, int wv, int lane) {
  typedef typename Traits<T>::v4 v4;
  const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 2, wn = wv & 3;
  unsigned char* outs = smem + wv * 16384;          // [128 rows m][8 chunks of 16 B], chunk ^= (row & 7)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = j * 32 + g * 8 + hi * 4;         // local n of this lane's 4 values
      float b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long n = n0 + wn * 64 + nl + r;
        b[r] = (p.bias != nullptr && n < p.n) ? p.bias[n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + l31;
        v4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(apply_act_t<ACT>(acc[i][j][g * 4 + r] * p.alpha + b[r], p.act, p.act_param));
        *reinterpret_cast<v4*>(outs + row * 128 + ((((nl >> 3)) ^ (row & 7)) << 4) + ((nl & 4) << 1)) = o;
      }
    }
  // a wave only re-reads its own region: no workgroup barrier needed, just its own LDS writes
#ifdef MTX_EMU
  emu::wave_sync();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  const T* G = reinterpret_cast<const T*>(p.gate);
  const T* R = reinterpret_cast<const T*>(p.res);
  const int oc = lane & 7;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + (lane >> 3);
    const long m = m0 + wm * 128 + row, n = n0 + wn * 64 + oc * 8;
    if (m >= p.m || n >= p.n) continue;
    u32x4 raw = *reinterpret_cast<const u32x4*>(outs + row * 128 + ((oc ^ (row & 7)) << 4));
    if (G != nullptr || R != nullptr) {
      float f[8];
      unpack8<T>(raw, f);
      if (G) { float g8[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(G + (size_t)(m / p.gate_rows_per) * p.ldgate + n), g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= g8[e]; }
      if (R) { float r8[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(R + (size_t)bz * p.res_bs + (size_t)m * p.ldres + n), r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += r8[e]; }
      raw = pack8<T>(f);
    }
    *reinterpret_cast<u32x4*>(Cp + (size_t)m * p.ldc + n) = raw;
  }
}

// tile `lin` of the grouped order -> its origin: groups of 4 tile rows x all tile columns, column-major inside a group
__device__ __forceinline__ void gemm256_tile_origin(const GemmParams& p, unsigned lin, long& m0, long& n0) {
  const unsigned GM = 4;
  const unsigned per_group = GM * p.tiles_n;
  const unsigned group = lin / per_group, first_m = group * GM;
  const unsigned gsz = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
  m0 = (long)(first_m + (lin % per_group) % gsz) * G2_BM;
  n0 = (long)((lin % per_group) / gsz) * G2_BN;
}

#ifdef MTX_EMU
#define G2_BAR() __syncthreads()
#else
#define G2_BAR() asm volatile("s_barrier" ::: "memory")
#endif

// The ping-pong K loop with descriptor DMA over iterations [kbeg, kend) of one 256 x 256 tile (used by the full-tile kernel and by
// the stream-K tail): barrier timeline B0, B1, ...: group 0 (waves 0-3) runs  L(k) B C(k) B  per k-step, group 1 the same one
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

// PP = ping-pong schedule: the two waves of every SIMD (waves w and w+4) run one barrier apart, so while one
// issues its fragment reads / DMA for a k-step the other owns the matrix pipe for its 8 MFMAs.
// CLAMP: rows past M / N are read from the last valid row instead of a zero block — they only feed outputs the epilogue never
// stores — so the loop carries no per-piece select and no scalar load of the zero block's address (whose s_waitcnt lgkmcnt(0)
// also drained the fragment reads in front of every DMA burst).
// BUF: the K loop is gemm256_pp_buf_loop (descriptor DMA, 3/3/2/0 piece spread) — the default; PP / !PP are the earlier loops.
template <typename T, int ACT, bool PP, bool CLAMP = true, bool BUF = false>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3;

  // the launch covers tiles [0, gridDim.x) of the grouped order (all of them, or the whole waves when the rest goes
  // to the stream-K tail kernel)
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  long m0, n0;
  gemm256_tile_origin(p, lin, m0, n0);
  const long bz = blockIdx.y;
  const T* A = reinterpret_cast<const T*>(p.a) + (size_t)bz * p.a_bs;
  const T* W = reinterpret_cast<const T*>(p.w) + (size_t)bz * p.w_bs;
  T* Cp = reinterpret_cast<T*>(p.c) + (size_t)bz * p.c_bs;

  // ---- DMA plan: instruction i of wave wv fills LDS rows (i*8 + wv)*8 .. +7 of a stage (1 KB); rows 0..255
  // are A, 256..511 are W.  This lane supplies slot (row = base + lane/8, position lane%8).
  const T* src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (i * 8 + wv) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    if (row < G2_BM) src[i] = (m0 + row < p.m) ? A + (size_t)(m0 + row) * p.lda + c * 8 : (CLAMP ? A + (size_t)(p.m - 1) * p.lda + c * 8 : nullptr);
    else src[i] = (n0 + row - G2_BM < p.n) ? W + (size_t)(n0 + row - G2_BM) * p.ldw + c * 8 : (CLAMP ? W + (size_t)(p.n - 1) * p.ldw + c * 8 : nullptr);
  }
  auto srcp = [&](int i, long k0) -> const void* {
    if (CLAMP) return (const void*)(src[i] + k0);
    return src[i] ? (const void*)(src[i] + k0) : (const void*)g_zero16;
  };
  auto issue = [&](int stage, long k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(srcp(i, k0), smem + stage * G2_STAGE + (i * 8 + wv) * 1024);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows of this lane and their swizzle term
  int arow[4], wrow[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = wm * 128 + i * 32 + l31;
#pragma unroll
  for (int j = 0; j < 2; ++j) wrow[j] = G2_BM + wn * 64 + j * 32 + l31;

  const long nk = p.k / G2_BK;
  if (BUF) {
    gemm256_pp_buf_loop<T>(p, smem, A, W, m0, n0, 0, nk, acc);
  } else if (!PP) {
    issue(0, 0);
    for (long kt = 0; kt < nk; ++kt) {
      MTX_WAIT_VMEM();
      __syncthreads();
      const bool more = kt + 1 < nk;
      if (more && p.abl == 2) issue((int)((kt + 1) & 1), (kt + 1) * G2_BK);      // ablation: all eight pieces right after the barrier
      const int nst = (int)((kt + 1) & 1);
      const long nk0 = p.abl == 5 ? 0 : (kt + 1) * G2_BK;            // ablation 5: always re-read the first K tile (L2-resident source)
      auto piece = [&](int i) {                 // one DMA instruction (1 KB) of the next tile
        glds16(srcp(i, nk0), smem + nst * G2_STAGE + (i * 8 + wv) * 1024);
      };
      const unsigned char* st = smem + (kt & 1) * G2_STAGE;
      // fragment reads run one k-step ahead of the MFMAs that consume them (two register sets)
      v8 af[2][4], wf[2][2];
      auto read_frags = [&](int ks, int set) {
        const int ch = 2 * ks + hi;
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[set][j] = *reinterpret_cast<const v8*>(st + wrow[j] * 128 + ((ch ^ ((wrow[j] >> 1) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) af[set][i] = *reinterpret_cast<const v8*>(st + arow[i] * 128 + ((ch ^ ((arow[i] >> 1) & 7)) << 4));
      };
      read_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) read_frags(ks + 1, (ks + 1) & 1);
#ifndef MTX_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (p.abl == 3 && more) {               // two pieces ahead of each k-step's MFMAs
          piece(2 * ks); piece(2 * ks + 1);
#ifndef MTX_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = Mma32<T>::mfma(wf[ks & 1][j], af[ks & 1][i], acc[i][j]);
            if ((p.abl == 0 || p.abl == 5) && more) {           // the next tile's 8 DMA pieces threaded between the MFMAs: 3, 3, 2, 0 per k-step (+3 % vs a burst)
              const int m = i * 2 + j;
              const int first = ks == 0 ? 0 : (ks == 1 ? 3 : 6), cnt = ks < 2 ? 3 : (ks == 2 ? 2 : 0);
              if (m == 1 && cnt > 0) piece(first);
              if (m == 3 && cnt > 1) piece(first + 1);
              if (m == 5 && cnt > 2) piece(first + 2);
#ifndef MTX_EMU
              __builtin_amdgcn_sched_barrier(0);
#endif
            }
          }
#ifndef MTX_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
  } else {
    // Barrier timeline B0, B1, ...: group 0 runs  L(k) B C(k) B  per k-step, group 1 the same one barrier later,
    // so group 1's load segment L coincides with group 0's compute segment C and vice versa.
    //  * tile kt+1 is DMA'd into the other stage from the load segments of k-steps 1 and 2 of tile kt: by then
    //    both groups have finished (lgkmcnt(0)) their reads of tile kt-1, which used that stage;
    //  * every wave drains its own DMA (vmcnt(0)) before the barrier that precedes group 0's first reads of
    //    tile kt+1: group 0 at the end of C(kt,3), group 1 at the end of L(kt,3).
    const int grp = wv >> 2;
    issue(0, 0);
    MTX_WAIT_VMEM();
    __syncthreads();
    if (grp == 1) G2_BAR();
    for (long kt = 0; kt < nk; ++kt) {
      const unsigned char* st = smem + (kt & 1) * G2_STAGE;
      const bool more = kt + 1 < nk;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 2 * ks + hi;
        v8 af[4], wf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const v8*>(st + wrow[j] * 128 + ((ch ^ ((wrow[j] >> 1) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8*>(st + arow[i] * 128 + ((ch ^ ((arow[i] >> 1) & 7)) << 4));
        if (more && (ks == 1 || ks == 2)) {
          const int stage = (int)((kt + 1) & 1);
          const long k0 = (kt + 1) * G2_BK;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int pi = (ks - 1) * 4 + i;
            glds16(srcp(pi, k0), smem + stage * G2_STAGE + (pi * 8 + wv) * 1024);
          }
        }
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
    }
    if (grp == 0) G2_BAR();
  }

  __syncthreads();
  gemm256_epilogue<T, ACT>(p, acc, smem, Cp, m0, n0, bz, wv, lane);
}

// Ring schedule: the same 8-wave ping-pong, but the 128 KB of LDS is a ring of four K = 32 slices (64-byte rows,
// chunk ^ ((row >> 2) & 3): 16 consecutive rows of one logical chunk cover a 256-byte bank row exactly once) instead
// of two K = 64 stages.  A slice's slot is released after its two k-steps, so slice s+3 is DMA'd while slice s is
// consumed and the wait before the barrier that publishes slice s+1 is a COUNTED vmcnt(8): the two youngest slices
// stay in flight and every piece has two slice-times (~2k cycles) to land instead of the ~0.5-1k of the two-stage
// loops, whose vmcnt(0) per K tile exposes HBM latency (ablation: no DMA 1460-1567 vs 1141 TFLOP/s).
//   WAR: slot (s+3)&3 held slice s-1; both groups finished those reads (lgkmcnt(0) at the start of their C(s-1, 1))
//        at least one barrier before anyone's L(s, 1), where the refill is issued.
//   RAW: every wave counts its own pieces of slice s+1 down before the barrier that precedes group 0's L(s+1, 0):
//        group 0 at the end of C(s, 1), group 1 at the end of L(s, 1) (the same physical barrier).
constexpr int G2R_SLOT = (G2_BM + G2_BN) * 64;      // 32 KB
#ifdef MTX_EMU
#define MTX_WAIT_VMEM_N(n) ((void)0)
#else
#define MTX_WAIT_VMEM_N(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif
template <typename T, int ACT>
__global__ __launch_bounds__(512) void gemm256r_kernel(GemmParams p) {
  typedef typename Traits<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * G2R_SLOT];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3, grp = wv >> 2;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  long m0, n0;
  gemm256_tile_origin(p, lin, m0, n0);
  const long bz = blockIdx.y;
  const T* A = reinterpret_cast<const T*>(p.a) + (size_t)bz * p.a_bs;
  const T* W = reinterpret_cast<const T*>(p.w) + (size_t)bz * p.w_bs;
  T* Cp = reinterpret_cast<T*>(p.c) + (size_t)bz * p.c_bs;

  // DMA plan: piece i (0..3) of wave wv fills slot rows (i*8 + wv)*16 .. +15 (1 KB); lane -> row base + lane/4, position lane%4
  const T* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 8 + wv) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    // rows past M / N are clamped to the last valid row: they only feed outputs the epilogue never stores
    if (row < G2_BM) { const long m = m0 + row < p.m ? m0 + row : p.m - 1; src[i] = A + (size_t)m * p.lda + c * 8; }
    else { const long n = n0 + row - G2_BM < p.n ? n0 + row - G2_BM : p.n - 1; src[i] = W + (size_t)n * p.ldw + c * 8; }
  }
  auto issue = [&](long s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(src[i] + s * 32, smem + (int)(s & 3) * G2R_SLOT + (i * 8 + wv) * 1024);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // byte offsets of this lane's fragment rows inside a slot, for chunk 0; the chunk term is XORed in per k-step
  int aoff[4], woff[2], asw[4], wsw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + l31; aoff[i] = r * 64; asw[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = G2_BM + wn * 64 + j * 32 + l31; woff[j] = r * 64; wsw[j] = (r >> 2) & 3; }

  const long ns = p.k / 32;
  // the wait that makes slice `nxt` visible: everything but the pieces of the slices after it (at most two) has landed
  auto wait_slice = [&](long nxt) {
    if (nxt + 2 < ns) MTX_WAIT_VMEM_N(8);
    else if (nxt + 1 < ns) MTX_WAIT_VMEM_N(4);
    else MTX_WAIT_VMEM();
  };
  issue(0);
  if (ns > 1) issue(1);
  if (ns > 2) issue(2);
  wait_slice(0);
  __syncthreads();
  if (grp == 1) G2_BAR();
  for (long s = 0; s < ns; ++s) {
    const unsigned char* st = smem + (int)(s & 3) * G2R_SLOT;
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      const int ch = 2 * ksl + hi;
      v8 af[4], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const v8*>(st + woff[j] + ((ch ^ wsw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8*>(st + aoff[i] + ((ch ^ asw[i]) << 4));
      if (ksl == 1 && s + 3 < ns) issue(s + 3);
      if (ksl == 1 && grp == 1) wait_slice(s + 1);
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
      if (ksl == 1 && grp == 0) wait_slice(s + 1);
      G2_BAR();
    }
  }
  if (grp == 0) G2_BAR();

  __syncthreads();
  gemm256_epilogue<T, ACT>(p, acc, smem, Cp, m0, n0, bz, wv, lane);
}
template <typename T>
static void launch_gemm256r(const GemmParams& p, dim3 grid, void* stream) {
  switch (p.act) {
    case MTX_ACT_NONE: MTX_LAUNCH((gemm256r_kernel<T, MTX_ACT_NONE>), grid, dim3(512), 0, stream, p); break;
    case MTX_ACT_SILU: MTX_LAUNCH((gemm256r_kernel<T, MTX_ACT_SILU>), grid, dim3(512), 0, stream, p); break;
    case MTX_ACT_GELU_TANH: MTX_LAUNCH((gemm256r_kernel<T, MTX_ACT_GELU_TANH>), grid, dim3(512), 0, stream, p); break;
    default: MTX_LAUNCH((gemm256r_kernel<T, -1>), grid, dim3(512), 0, stream, p); break;
  }
}

// Wave-specialised variant: 8 MFMA waves (never touch VMEM) + 4 DMA waves (one per SIMD) that stream the next
// K tile into the other LDS stage while the MFMA waves work.  An LDS-DMA instruction costs its issuing wave
// 60-180 cycles; in the kernel above that time is taken from the matrix pipe (both waves of a SIMD issue their
// 8 pieces right after the barrier: no-DMA ablation 1460 vs 1150 TFLOP/s).  768 threads = 3 waves per SIMD,
// so a wave has 168 registers: 128 accumulators + one fragment set.
template <typename T, int ACT>
__global__ __launch_bounds__(768) void gemm256ws_kernel(GemmParams p) {
  typedef typename Traits<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;

  const unsigned nwg = p.tiles_m * p.tiles_n;
  const unsigned lin = xcd_remap(blockIdx.x, nwg);
  const unsigned GM = 4;
  const unsigned per_group = GM * p.tiles_n;
  const unsigned group = lin / per_group, first_m = group * GM;
  const unsigned gsz = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
  const long m0 = (long)(first_m + (lin % per_group) % gsz) * G2_BM;
  const long n0 = (long)((lin % per_group) / gsz) * G2_BN;
  const long bz = blockIdx.y;
  const T* A = reinterpret_cast<const T*>(p.a) + (size_t)bz * p.a_bs;
  const T* W = reinterpret_cast<const T*>(p.w) + (size_t)bz * p.w_bs;
  T* Cp = reinterpret_cast<T*>(p.c) + (size_t)bz * p.c_bs;
  const long nk = p.k / G2_BK;

  if (wv >= 8) {
    // ---- DMA wave L: LDS regions (1 KB = 8 rows of a stage) L*16 .. L*16+15; rows 0..255 are A, 256..511 W
    const int L = wv - 8;
    const T* src[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (L * 16 + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      if (row < G2_BM) src[i] = (m0 + row < p.m) ? A + (size_t)(m0 + row) * p.lda + c * 8 : nullptr;
      else src[i] = (n0 + row - G2_BM < p.n) ? W + (size_t)(n0 + row - G2_BM) * p.ldw + c * 8 : nullptr;
    }
    auto issue = [&](int stage, long k0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const void* g = src[i] ? (const void*)(src[i] + k0) : (const void*)g_zero16;
        glds16(g, smem + stage * G2_STAGE + (L * 16 + i) * 1024);
      }
    };
    issue(0, 0);
    for (long kt = 0; kt < nk; ++kt) {
      MTX_WAIT_VMEM();                         // tile kt has landed (this wave's part)
      G2_BAR();                                // barrier kt: everyone's part landed; stage (kt+1)&1 is free again
      if (kt + 1 < nk) issue((int)((kt + 1) & 1), (kt + 1) * G2_BK);
    }
    __syncthreads();                           // matches the MFMA waves' barrier before the epilogue
    return;
  }

  const int wm = wv >> 2, wn = wv & 3;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aoff[4], woff[2];                        // byte offset of this lane's fragment row, k-step 0, hi-half folded in
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + l31; aoff[i] = r * 128 + ((hi ^ ((r >> 1) & 1)) << 4); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = G2_BM + wn * 64 + j * 32 + l31; woff[j] = r * 128 + ((hi ^ ((r >> 1) & 1)) << 4); }
  // chunk (2*ks + hi) ^ ((r>>1)&7) = ((ks ^ ((r>>2)&3)) << 1) | (hi ^ ((r>>1)&1)): the ks-dependent part is the same
  // for all rows of a lane's fragments up to bits 2..3 of r, i.e. of l31 (row bases are multiples of 32)
  const int kx = (l31 >> 2) & 3;
  for (long kt = 0; kt < nk; ++kt) {
    G2_BAR();                                  // barrier kt (the DMA waves waited for tile kt before arriving)
    const unsigned char* st = smem + (kt & 1) * G2_STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ko = (ks ^ kx) << 5;
      v8 af[4], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const v8*>(st + woff[j] + ko);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8*>(st + aoff[i] + ko);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma32<T>::mfma(wf[j], af[i], acc[i][j]);
    }
#ifndef MTX_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  }
  __syncthreads();
  gemm256_epilogue<T, ACT>(p, acc, smem, Cp, m0, n0, bz, wv, lane);
}

template <typename T>
static void launch_gemm256ws(const GemmParams& p, dim3 grid, void* stream) {
  switch (p.act) {
    case MTX_ACT_NONE: MTX_LAUNCH((gemm256ws_kernel<T, MTX_ACT_NONE>), grid, dim3(768), 0, stream, p); break;
    case MTX_ACT_SILU: MTX_LAUNCH((gemm256ws_kernel<T, MTX_ACT_SILU>), grid, dim3(768), 0, stream, p); break;
    case MTX_ACT_GELU_TANH: MTX_LAUNCH((gemm256ws_kernel<T, MTX_ACT_GELU_TANH>), grid, dim3(768), 0, stream, p); break;
    default: MTX_LAUNCH((gemm256ws_kernel<T, -1>), grid, dim3(768), 0, stream, p); break;
  }
}

// ---- stream-K tail ------------------------------------------------------------------------------------------
// With one 256 x 256 tile per CU at a time, `rem = tiles % CUs` left-over tiles keep rem CUs busy for a whole tile
// time while the others idle (FLUX proj_out: 408 tiles on 256 CUs = 1.59 waves billed as 2).  The left-over tiles'
// K iterations are instead dealt out evenly: unit u takes iterations [u*I/units, (u+1)*I/units) of the rem * nk
// iterations, i.e. the end of one tile and possibly the start of the next, and leaves an fp32 partial per piece
// (slot 2u, 2u+1); the merge kernel adds a tile's pieces in K order and applies the usual epilogue.
template <typename T>
__global__ __launch_bounds__(512) void gemm256_tail_kernel(GemmParams p) {
  typedef typename Traits<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3;
  const long nk = p.k / G2_BK;
  const unsigned tiles = p.tiles_m * p.tiles_n, rem = tiles - p.n_full;
  const long I = (long)rem * nk;
  const unsigned u = blockIdx.x;
  const long it0 = u * I / p.units, it1 = (u + 1) * I / p.units;
  const T* A = reinterpret_cast<const T*>(p.a);
  const T* W = reinterpret_cast<const T*>(p.w);
  int arow[4], wrow[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = wm * 128 + i * 32 + l31;
#pragma unroll
  for (int j = 0; j < 2; ++j) wrow[j] = G2_BM + wn * 64 + j * 32 + l31;

  for (int seg = 0; seg < 2; ++seg) {
    const long t0 = it0 / nk;                                   // tail tile of the first piece
    const long kbeg = seg == 0 ? it0 - t0 * nk : 0;
    const long kend = seg == 0 ? (it1 < (t0 + 1) * nk ? it1 - t0 * nk : nk) : it1 - (t0 + 1) * nk;
    if (kend <= kbeg) continue;                                 // (wave-uniform)
    long m0, n0;
    gemm256_tile_origin(p, (unsigned)(p.n_full + t0 + seg), m0, n0);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();                                            // the previous piece is done with the LDS stages
    gemm256_pp_buf_loop<T>(p, smem, A, W, m0, n0, kbeg, kend, acc);
    // fp32 partial of this piece: [256 m][256 n], a lane stores 4 consecutive n
    float* P = p.part + (size_t)(2 * u + seg) * G2_BM * G2_BN;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = wm * 128 + i * 32 + l31, n = wn * 64 + j * 32 + g * 8 + hi * 4;
          f32x4 v = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
          *reinterpret_cast<f32x4*>(P + (size_t)m * G2_BN + n) = v;
        }
  }
}

// one workgroup per (tail tile, 32-row band): sum the pieces in K order, epilogue, 16-byte stores
template <typename T>
__global__ __launch_bounds__(256) void gemm256_merge_kernel(GemmParams p) {
  const long nk = p.k / G2_BK;
  const unsigned tiles = p.tiles_m * p.tiles_n, rem = tiles - p.n_full;
  const long I = (long)rem * nk;
  const unsigned ti = blockIdx.x / 8, band = blockIdx.x % 8;
  long m0, n0;
  gemm256_tile_origin(p, p.n_full + ti, m0, n0);
  // units whose run intersects iterations [ti*nk, (ti+1)*nk)
  long u_lo = (long)ti * nk * p.units / I; if (u_lo > 0) --u_lo;
  long u_hi = ((long)(ti + 1) * nk * p.units + I - 1) / I + 1; if (u_hi > p.units) u_hi = p.units;
  T* Cp = reinterpret_cast<T*>(p.c);
  const T* G = reinterpret_cast<const T*>(p.gate);
  const T* R = reinterpret_cast<const T*>(p.res);
  for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
    const int row = band * 32 + idx / 32, ch = idx % 32;
    const long m = m0 + row, n = n0 + ch * 8;
    if (m >= p.m || n >= p.n) continue;
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long u = u_lo; u < u_hi; ++u) {
      const long it0 = u * I / p.units, it1 = (u + 1) * I / p.units;
      const long t0 = it0 / nk;
      int seg = -1;
      if (t0 == ti && it1 > it0) seg = 0;
      else if (t0 + 1 == ti && it1 > (t0 + 1) * nk) seg = 1;
      if (seg < 0) continue;
      const float* P = p.part + ((size_t)(2 * u + seg) * G2_BM + row) * G2_BN + ch * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += P[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = apply_act(f[e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f), p.act, p.act_param);
    if (G) { float g8[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(G + (size_t)(m / p.gate_rows_per) * p.ldgate + n), g8);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= g8[e]; }
    if (R) { float r8[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(R + (size_t)m * p.ldres + n), r8);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r8[e]; }
    *reinterpret_cast<u32x4*>(Cp + (size_t)m * p.ldc + n) = pack8<T>(f);
  }
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

template <typename T, bool PP, bool CLAMP>
static void launch_gemm256_ppc(const GemmParams& p, dim3 grid, void* stream) {
  switch (p.act) {
    case MTX_ACT_NONE: MTX_LAUNCH((gemm256_kernel<T, MTX_ACT_NONE, PP, CLAMP>), grid, dim3(512), 0, stream, p); break;
    case MTX_ACT_SILU: MTX_LAUNCH((gemm256_kernel<T, MTX_ACT_SILU, PP, CLAMP>), grid, dim3(512), 0, stream, p); break;
    case MTX_ACT_GELU_TANH: MTX_LAUNCH((gemm256_kernel<T, MTX_ACT_GELU_TANH, PP, CLAMP>), grid, dim3(512), 0, stream, p); break;
    default: MTX_LAUNCH((gemm256_kernel<T, -1, PP, CLAMP>), grid, dim3(512), 0, stream, p); break;
  }
}
template <typename T>
static void launch_gemm256_buf(const GemmParams& p, dim3 grid, void* stream) {
  switch (p.act) {
    case MTX_ACT_NONE: MTX_LAUNCH((gemm256_kernel<T, MTX_ACT_NONE, true, true, true>), grid, dim3(512), 0, stream, p); break;
    case MTX_ACT_SILU: MTX_LAUNCH((gemm256_kernel<T, MTX_ACT_SILU, true, true, true>), grid, dim3(512), 0, stream, p); break;
    case MTX_ACT_GELU_TANH: MTX_LAUNCH((gemm256_kernel<T, MTX_ACT_GELU_TANH, true, true, true>), grid, dim3(512), 0, stream, p); break;
    default: MTX_LAUNCH((gemm256_kernel<T, -1, true, true, true>), grid, dim3(512), 0, stream, p); break;
  }
}
template <typename T, bool PP>
static void launch_gemm256_pp(const GemmParams& p, dim3 grid, void* stream) {
  const char* e = getenv("MTX_GEMM_CLAMP");              // A/B switch: "0" = the zero-block select of the first version
  if (e && e[0] == '0') launch_gemm256_ppc<T, PP, false>(p, grid, stream);
  else launch_gemm256_ppc<T, PP, true>(p, grid, stream);
}
template <typename T>
static void launch_gemm256(const GemmParams& p0, dim3 grid, void* stream) {
  GemmParams p = p0;
  const char* e = getenv("MTX_GEMM256_SCHED");          // A/B switch: "buf" (default), "pingpong", "lockstep", "ring", "ws"
  // measured on MI355X (tools/bench_kernels.py ab, FLUX shapes, random data): the ping-pong loop with descriptor DMA and the
  // 3/3/2/0 piece spread runs 1119-1341 TFLOP/s, 12-15 % ahead of the flat-address ping-pong (wins up to K = 4096) and one-barrier
  // (wins above) loops it replaces; the ring is 2-5 % and the wave-specialised variant 10-15 % behind those
  const bool fits32 = ((size_t)G2_BM * p.lda + p.k) * sizeof(T) < (1ull << 32) && ((size_t)G2_BN * p.ldw + p.k) * sizeof(T) < (1ull << 32);      // one tile's rows under a descriptor
  const char mode = e ? e[0] : (fits32 ? 'b' : (p.k <= 4096 ? 'p' : 'l'));
  // stream-K tail when the last wave of tiles would fill less than ~70 % of the chip
  const unsigned tiles = p.tiles_m * p.tiles_n, cus = (unsigned)gemm_num_cus(), rem = tiles % cus;
  const char* ns = getenv("MTX_GEMM_NOSPLIT");
  const bool tail = fits32 && p.part != nullptr && cus <= 320 && grid.y == 1 && tiles > cus && rem > 0 && rem * 10 < cus * 7 && p.k / G2_BK >= (getenv("MTX_GEMM256_MIN_TILES") ? 8 : 64) && !(ns && ns[0] == '1');
  // few tiles but a long K (FLUX text-stream ff2: 24 tiles x 192 iterations): stream-K over the whole problem
  const bool allk = fits32 && p.part != nullptr && cus <= 320 && grid.y == 1 && tiles * 2 <= cus && p.k / G2_BK >= 128 && !(ns && ns[0] == '1');
  if (allk) { p.n_full = 0; p.units = cus; }
  else if (tail) { p.n_full = tiles - rem; p.units = cus; grid.x = p.n_full; }
  if (!allk) {
    if (mode == 'l') launch_gemm256_pp<T, false>(p, grid, stream);
    else if (mode == 'p') launch_gemm256_pp<T, true>(p, grid, stream);
    else if (mode == 'r') launch_gemm256r<T>(p, grid, stream);
    else if (mode == 'b' && fits32) launch_gemm256_buf<T>(p, grid, stream);
    else if (mode == 'b') launch_gemm256_pp<T, true>(p, grid, stream);          // a descriptor addresses 4 GB
    else launch_gemm256ws<T>(p, grid, stream);
  }
  if (tail || allk) {
    MTX_LAUNCH((gemm256_tail_kernel<T>), dim3(p.units), dim3(512), 0, stream, p);
    MTX_LAUNCH((gemm256_merge_kernel<T>), dim3((tiles - p.n_full) * 8), dim3(256), 0, stream, p);
  }
}

int gemm_launch(const mtx_gemm_args* a, void* stream, const char** err) {
  if (!a->a || !a->w || !a->c) { *err = "gemm: null operand"; return MTX_ERR_INVALID; }
  if (a->m < 1 || a->n < 1 || a->k < 1) { *err = "gemm: empty problem"; return MTX_ERR_INVALID; }
  if (a->k % 8 || a->lda % 8 || a->ldw % 8) { *err = "gemm: K, lda, ldw must be multiples of 8 (16-byte chunks)"; return MTX_ERR_INVALID; }
  if (a->batch > 1 && (a->a_bstride % 8 || a->w_bstride % 8 || a->c_bstride % 8 || a->res_bstride % 8)) { *err = "gemm: batch strides must be multiples of 8"; return MTX_ERR_INVALID; }
  if (a->out_dtype != a->dtype && a->out_dtype != MTX_F32) { *err = "gemm: out_dtype must equal dtype or be f32"; return MTX_ERR_INVALID; }
  GemmParams p;
  p.a = (const unsigned char*)a->a; p.w = (const unsigned char*)a->w; p.bias = a->bias;
  p.res = (const unsigned char*)a->res; p.gate = (const unsigned char*)a->gate; p.c = (unsigned char*)a->c;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.ldres = a->ldres; p.ldgate = a->ldgate;
  p.a_bs = a->a_bstride; p.w_bs = a->w_bstride; p.c_bs = a->c_bstride; p.res_bs = a->res_bstride;
  p.gate_rows_per = a->gate_rows_per > 0 ? a->gate_rows_per : 1;
  p.act = a->act; p.act_param = a->act_param; p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
  p.out_f32 = a->out_dtype == MTX_F32 && a->dtype != MTX_F32;
  p.abl = getenv("MTX_GEMM_ABL") ? atoi(getenv("MTX_GEMM_ABL")) : 0;
  p.n_full = 0; p.units = 0;
  p.part = (a->workspace && a->workspace_bytes >= (int64_t)MTX_GEMM_WORKSPACE_BYTES) ? reinterpret_cast<float*>(a->workspace) : nullptr;
  p.tiles_m = (unsigned)((a->m + GBM - 1) / GBM);
  p.tiles_n = (unsigned)((a->n + GBN - 1) / GBN);
  const long batch = a->batch > 0 ? a->batch : 1;
  // large, aligned problems: the 256 x 256 LDS-DMA kernel (needs whole K tiles and 16-byte rows everywhere)
  const long t256 = ((a->m + G2_BM - 1) / G2_BM) * ((a->n + G2_BN - 1) / G2_BN) * batch;
  const bool vec = a->n % 8 == 0 && a->ldc % 8 == 0 && (!a->res || a->ldres % 8 == 0) && (!a->gate || a->ldgate % 8 == 0) && a->c_bstride % 8 == 0;
  // with the descriptor-DMA loop the 256-tile kernel wins from ~24 tiles up even though most CUs idle (512x9216x3072: 55 vs 68 us,
  // 1024x4608x1152: 24.5 vs 32.7 us); below that the 128-tile kernel's extra parallelism pays.  Tests lower it.
  // (short K — SAM's 576-wide stage — keeps the old threshold: the tile prologue / epilogue dominates there, encoder 12.5 vs 11.8 ms)
  // and so do shapes that would pad a 256-tile row or column by more than 10 % (SAM's N = 576: 3 columns for 2.25)
  const long t256m = (a->m + G2_BM - 1) / G2_BM, t256n = (a->n + G2_BN - 1) / G2_BN;
  const bool snug = a->m * 10 >= t256m * G2_BM * 9 && a->n * 10 >= t256n * G2_BN * 9;
  const long min_tiles = getenv("MTX_GEMM256_MIN_TILES") ? atol(getenv("MTX_GEMM256_MIN_TILES")) : ((a->k >= 1024 && snug) ? 24 : 160);
  const bool few_long = a->workspace != nullptr && batch == 1 && t256 * 2 <= gemm_num_cus() && a->k / G2_BK >= 128 && a->m >= 256;
  if (!p.out_f32 && a->k % G2_BK == 0 && vec && (t256 >= min_tiles || few_long) && (a->dtype == MTX_BF16 || a->dtype == MTX_F16)) {
    p.tiles_m = (unsigned)((a->m + G2_BM - 1) / G2_BM);
    p.tiles_n = (unsigned)((a->n + G2_BN - 1) / G2_BN);
    dim3 g2(p.tiles_m * p.tiles_n, (unsigned)batch);
    if (a->dtype == MTX_BF16) launch_gemm256<__bf16>(p, g2, stream); else launch_gemm256<_Float16>(p, g2, stream);
    return MTX_OK;
  }
  dim3 grid(p.tiles_m * p.tiles_n, (unsigned)batch);
  if (a->dtype == MTX_BF16) MTX_LAUNCH((gemm_kernel<__bf16>), grid, dim3(256), 0, stream, p);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((gemm_kernel<_Float16>), grid, dim3(256), 0, stream, p);
  else { *err = "gemm: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

}  // namespace mtx
