// conv_c64.hip — persistent 3x3 / stride-1 convolution for Cin, Cout <= 64: the RCAN body conv
// (400 of the 404 convolutions of 2x-AnimeSharpV4_RCAN; spandrel model called at reference
// core/image/image_utils.py:369-374) and the narrow YOLO stem layers.
//
// At 64 channels the whole filter bank is 9 x 64 x 64 x 2 B = 72 KiB: it fits in LDS next to the
// input halos, so the kernel is PERSISTENT — one 8-wave workgroup per CU loads the filters once and
// walks 16x16-pixel output tiles in XCD-contiguous order (neighbouring halos hit the same L2).
//
// The 8 waves form TWO GROUPS of 4 that run half a period out of phase (LDS: filters 72 KiB +
// one 18x18-pixel halo per group, 2 x 40.5 KiB):
//
//      slot 2k     : group 0  MFMA(tile k)          | group 1  memory phase of its tile k-1
//      slot 2k + 1 : group 0  memory phase (tile k) | group 1  MFMA(tile k)
//
//   memory phase = { halo of the NEXT tile: prefetched registers -> LDS ; bias/act/residual epilogue
//                    and NHWC stores of the finished tile ; issue the global loads of the tile after }
//
// so the matrix pipe always has one group's 288 MFMAs per wave to run while the other group's
// loads, LDS writes and stores are in flight; one LDS-only barrier (s_waitcnt lgkmcnt(0); s_barrier
// — never vmcnt, see MTX_LDS_BARRIER) separates the slots.  Inside the memory phase the order is
// chosen for gfx950's in-order vmcnt: the halo registers are consumed BEFORE the tile's stores are
// issued, and no global load that is needed soon is ever issued behind a store.
// Activations cross HBM once in (x1.27 halo overlap, L2-served) and once out; filters never again.
// Same LDS swizzle and MFMA operand swap as conv.hip.
#include "mtx_device.h"
#include <cstdlib>

namespace mtx {

struct ConvC64Params {
  const unsigned char* x; const unsigned char* w; const float* bias; const unsigned char* res;
  unsigned char* y; float* chan_sum;
  int n, h, w_in, cin, cout;
  int ldx, ldy, ldres;
  int act; float act_param; float res_scale;
  int ps;
  int res_bcast;
  int tiles_x, tiles_y;
  const int* valid_hw;   // device {valid_h, valid_w} or null: outputs beyond are zero and left out of the channel sums
  unsigned y_bytes;      // extent of the output tensor (buffer-descriptor stores; < 4 GiB, see conv_c64_applicable)
  unsigned x_bytes;      // extent of the input tensor (buffer-descriptor halo DMA of interior tiles)
  const float* out_scale;   // device [n][cout] factor on act(conv + bias), before the residual (or null)
  unsigned res_bytes;    // extent of the residual tensor (0: no descriptor path for it)
};

constexpr int C64_T = 16;                                  // tile edge (pixels)
constexpr int C64_HW = C64_T + 2, C64_HPIX = C64_HW * C64_HW;   // 18 x 18 = 324 halo pixels
constexpr int C64_W_BYTES = 9 * 64 * 128;
constexpr int C64_HALO_BYTES = (C64_HPIX + 4) * 128;          // + 4 rows: the last DMA instruction of a halo covers rows 320 .. 327
constexpr int C64_BIAS_BYTES = 64 * 4 + 2 * 64 * 4;       // bias + one per-channel output-scale row per group
constexpr int C64_NDMA_C = (C64_HPIX * 8 + 63) / 64;              // 41 wave-instructions of 1 KiB per halo
constexpr int C64_SMEM = C64_W_BYTES + 2 * C64_HALO_BYTES + C64_BIAS_BYTES;

// MFMA in place as `asm volatile`: issue order = program order (the builtin form is free to move, and with one
// wave per SIMD on the matrix pipe the order of reads and MFMAs is the whole schedule).  C64_FENCE pins a plain LDS
// read between two MFMAs; the compiler still places (and counts) the s_waitcnt in front of each consumer.
#ifdef MTX_EMU
template <typename T> __device__ __forceinline__ void mfma_inplace(f32x4& c, const typename Traits<T>::v8& a, const typename Traits<T>::v8& b) {
  c = Traits<T>::mfma(a, b, c);
}
#define C64_FENCE() ((void)0)
#define C64_MFMA_DRAIN() ((void)0)
#else
template <typename T> __device__ __forceinline__ void mfma_inplace(f32x4& c, const typename Traits<T>::v8& a, const typename Traits<T>::v8& b);
template <> __device__ __forceinline__ void mfma_inplace<_Float16>(f32x4& c, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_inplace<__bf16>(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define C64_FENCE() asm volatile("" ::: "memory")
// the accumulators are read by vector ALU code next; the asm form hides the MFMAs from the hazard recogniser
#define C64_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")
#endif

// bias already added: activation + conversion to the storage type of four consecutive channels.  The f16 forms convert first and
// apply ReLU / saturation on the packed halves (ReLU and saturation commute with round-to-nearest): 1.5 instead of 3 vector
// instructions per value — the memory slot of this kernel is bound by one wave's instruction issue, not by bytes.
template <typename T, int ACT> __device__ __forceinline__ typename Traits<T>::v4 epi_pack(f32x4 v, int act, float act_param) {
  typename Traits<T>::v4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(apply_act_t<ACT>(v[r], act, act_param));
  return o;
}
#ifndef MTX_EMU
template <int ACT> __device__ __forceinline__ f16x4 epi_pack_f16(f32x4 v) {
  f16x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];                       // v_cvt_pk_f16_f32; overflow -> +-inf
  const f16x4 hi = {(_Float16)65504.f, (_Float16)65504.f, (_Float16)65504.f, (_Float16)65504.f};
  const f16x4 lo = ACT == MTX_ACT_RELU ? f16x4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f} : -hi;
  return __builtin_elementwise_max(__builtin_elementwise_min(o, hi), lo);
}
template <> __device__ __forceinline__ f16x4 epi_pack<_Float16, MTX_ACT_NONE>(f32x4 v, int, float) { return epi_pack_f16<MTX_ACT_NONE>(v); }
template <> __device__ __forceinline__ f16x4 epi_pack<_Float16, MTX_ACT_RELU>(f32x4 v, int, float) { return epi_pack_f16<MTX_ACT_RELU>(v); }
#endif

__device__ __forceinline__ int grp_of(unsigned tid) { return (int)(tid >> 8); }

// ABL: timing-only ablations for tools/probes/conv_probe.hip; 0 = the real kernel.  1: no MFMA loop, 3: no halo DMA, 4: no epilogue / stores,
// 7: shader-clock stamps at the phase boundaries, 8: the generic per-tile address / bounds paths on interior tiles too.  (The what-ifs that
// settled design questions — LDS reads or MFMAs alone, s_nop behind every MFMA, stores before the DMA, no wait at all, 8-byte stores — are
// recorded with their numbers in DESIGN.md section 5 and were taken out of the source.)
// SUM: fused channel sums (p.chan_sum); RES: residual input (p.res) — separate variants because each keeps 16 - 36 registers alive across
// the slot barrier; a launch that wants both goes to the generic kernel (conv_c64_applicable)
template <typename T, int ABL, int ACT, bool SUM, bool RES = false>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(ConvC64Params p) {
  constexpr int nks_c = 2;                         // k-steps of 32 input channels (narrower inputs read zero chunks: the 3 -> 64 head conv is one launch per page)
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C64_SMEM];
  unsigned char* wts = smem;
  // bias lives in LDS: a global load inside the tile loop would sit behind earlier stores on the
  // in-order vmcnt counter and expose their latency
  float* bias_s = reinterpret_cast<float*>(smem + C64_W_BYTES + 2 * C64_HALO_BYTES);
  float* scale_s = bias_s + 64 + grp_of(threadIdx.x) * 64;      // this group's out_scale row (refreshed per tile in the MFMA slot)

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);           // wave-uniform group id
  const int gt = tid & 255;                                            // thread within the group
  const int wv = (tid >> 6) & 3;                                       // wave within the group
  unsigned char* halo = smem + C64_W_BYTES + grp * C64_HALO_BYTES;
  const unsigned tiles_per_img = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned total = tiles_per_img * (unsigned)p.n;
  const unsigned npairs = (total + 1) / 2;

  // ---- filters + bias: once per workgroup ---------------------------------------------------------
  for (int idx = tid; idx < 9 * 64 * 8; idx += 512) {
    const int c = idx & 7, row = idx >> 3;            // row = tap*64 + co
    const int tap = row >> 6, co = row & 63;
    const int ch = c * 8;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (co < p.cout && ch < p.cin)
      v = *reinterpret_cast<const u32x4*>(p.w + (((size_t)co * 9 + tap) * (size_t)p.cin + ch) * sizeof(T));
    *reinterpret_cast<u32x4*>(wts + row * 128 + ((c ^ (co & 7)) << 4)) = v;
  }
  if (tid < 64) bias_s[tid] = (p.bias != nullptr && tid < p.cout) ? p.bias[tid] : 0.f;
  if (p.out_scale != nullptr && tid >= 64 && tid < 192)       // one image: its output factors once, for both groups (several images: per tile, below)
    bias_s[tid] = ((tid & 63) < p.cout) ? p.out_scale[tid & 63] : 0.f;
  const BufView ybuf = make_buf(p.y, p.y_bytes);
  const BufView xbuf = make_buf(p.x, p.x_bytes);
  // INTERIOR tiles (halo and outputs inside the image; the common case by far): everything per-lane is tile-invariant and
  // computed once — the halo DMA is 11 instructions with a scalar tile base and no address arithmetic or bounds tests, the
  // 16 stores are base register + immediate.  A wave's memory slot is bound by its own instruction issue (~1300
  // instructions on the generic path, stamped at 7 000 cycles against 5 500 for the MFMA slot of the other group).
  unsigned dma_off[(C64_NDMA_C + 3) / 4];
#pragma unroll
  for (int it = 0; it < (C64_NDMA_C + 3) / 4; ++it) {
    const int slot = (wv + it * 4) * 64 + lane;
    const int hp = slot >> 3, hy = hp / C64_HW, hx = hp - hy * C64_HW, ch = (((slot & 7) ^ (hx & 7))) * 8;
    dma_off[it] = (hp < C64_HPIX && ch < p.cin) ? (unsigned)(((hy * p.w_in + hx) * p.ldx + ch) * (int)sizeof(T)) : p.x_bytes;   // out of range: zeros
  }
  // stores of the fast path: 16 bytes per lane.  A lane's accumulators hold channel quads 16j + 4q (8-byte units scattered over the
  // pixel's 128-byte line); one v_permlane16_swap per dword between the lanes q and q ^ 1 of a pixel turns two units into one
  // contiguous 16-byte chunk, so a store instruction writes a 64-byte half line per pixel and a tile takes 8 of them, not 16
  unsigned st_off[4];
  const unsigned chunk_off = (q & 1) ? 32u + 8u * (unsigned)(q - 1) : 8u * (unsigned)q;
#pragma unroll
  for (int i = 0; i < 4; ++i) st_off[i] = (unsigned)(((wv * 4 + i) * p.w_in + l15) * p.ldy * (int)sizeof(T)) + chunk_off;
  const bool fast_ok = p.ps == 0 && p.cout == 64 && p.valid_hw == nullptr && ABL != 8 && (!RES || (!p.res_bcast && p.res_bytes != 0));
  constexpr bool FAST_ACT = ACT == MTX_ACT_NONE || ACT == MTX_ACT_RELU;       // the activations the interior-tile paths are written for
  const bool has_post = RES || p.out_scale != nullptr;        // fp32 scale / residual after the activation
  const BufView rbuf = make_buf(RES ? p.res : p.x, p.res_bytes);
  unsigned res_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) res_off[i] = (unsigned)(((wv * 4 + i) * p.w_in + l15) * p.ldres * (int)sizeof(T)) + chunk_off;     // 16-byte chunks, like the stores
  // ABL 7 (tools/probes/conv_probe.hip): wave 0 of each group of workgroups 0 and 97 writes the shader clock at the phase
  // boundaries of every slot into chan_sum, viewed as uint64 [2 workgroups][2 groups][64 slots][8 events]
  auto stamp = [&](unsigned s_, int ev) {
#ifndef MTX_EMU
    if (ABL == 7 && (blockIdx.x == 0 || blockIdx.x == 97) && wv == 0 && lane == 0 && s_ < 64)
      reinterpret_cast<unsigned long long*>(p.chan_sum)[((((blockIdx.x ? 1 : 0) * 2 + grp) * 64 + s_) * 8) + ev] = __builtin_readcyclecounter();
#endif
  };
  // Fragment addresses of the MFMA loop, computed ONCE: with the halo swizzled by column, the address of
  // (tile row i, tap (ky, kx), k-step ks) is xa[kx][ks] + (i + ky) * 18 * 128 and a filter fragment is
  // wa[ks] + tap * 8 KiB + j * 2 KiB — every read of the loop is base register + immediate, and the loop
  // carries no vector ALU instruction at all (MFMA issue shares the VALU port with them).
  const unsigned char* wa[2];
  const unsigned char* xa[3][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    wa[ks] = wts + l15 * 128 + (((ks * 4 + q) ^ (l15 & 7)) << 4);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int hx = l15 + kx;
      xa[kx][ks] = halo + ((wv * 4) * C64_HW + hx) * 128 + (((ks * 4 + q) ^ (hx & 7)) << 4);
    }
  }

  // tile owned by this group in pair-iteration k (or ~0u when past the end)
  auto tile_of = [&](unsigned k) -> unsigned {
    const unsigned pid = blockIdx.x + k * gridDim.x;
    if (pid >= npairs) return ~0u;
    // XCD-contiguous order only when the grid is a multiple of 8: then a workgroup's pairs — hence its images — come in increasing order
    // (its XCD is fixed and the index inside the XCD's run grows with k), which the channel-sum rows below rely on
    const unsigned vp = (gridDim.x % 8u == 0u) ? xcd_remap(pid, npairs) : pid;
    const unsigned lin = 2u * vp + (unsigned)grp;
    return lin < total ? lin : ~0u;
  };

  // halo staging by LDS-DMA (no staging registers): DMA instruction m of a group fills LDS rows
  // 8m .. 8m+7 of the group's halo; lane l -> row 8m + l/8, 16-byte slot l%8, which must hold chunk
  // (slot ^ (halo column & 7)) of that pixel — the swizzle is applied on the per-lane SOURCE address.
  // Out-of-image pixels read the zero page.
  constexpr int C64_NDMA = (C64_HPIX * 8 + 63) / 64;      // 41 wave-instructions per halo
  auto dma_halo = [&](unsigned lin_in) {
#ifdef MTX_EMU
    const unsigned lin = lin_in;
#else
    const unsigned lin = (unsigned)__builtin_amdgcn_readfirstlane((int)lin_in);
#endif
    const int img = (int)(lin / tiles_per_img);
    const int tile = (int)(lin % tiles_per_img);
    const int iy0 = (tile / p.tiles_x) * C64_T - 1, ix0 = (tile % p.tiles_x) * C64_T - 1;
    const size_t img_off = (size_t)img * p.h * p.w_in;
    if (ABL != 8 && iy0 >= 0 && ix0 >= 0 && iy0 + C64_HW <= p.h && ix0 + C64_HW <= p.w_in) {      // interior: scalar base + invariant lane offsets
      const unsigned sbase = (unsigned)(((img_off + (size_t)iy0 * p.w_in + ix0) * (size_t)p.ldx) * sizeof(T));
#pragma unroll
      for (int it = 0; it < (C64_NDMA + 3) / 4; ++it) {
        const int m = __builtin_amdgcn_readfirstlane(wv + it * 4);
        if (m < C64_NDMA) buf_load16_lds(xbuf, dma_off[it], sbase, halo + m * 1024);
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < (C64_NDMA + 3) / 4; ++it) {
      const int m = __builtin_amdgcn_readfirstlane(wv + it * 4);
      if (m < C64_NDMA) {
        const int slot = m * 64 + lane;
        const int hp = slot >> 3;
        const int hy = hp / C64_HW, hx = hp - hy * C64_HW;
        const int c = (slot & 7) ^ (hx & 7);              // swizzle by halo COLUMN: a row step is a constant LDS offset for the readers
        const int gy = iy0 + hy, gx = ix0 + hx, ch = c * 8;
        const unsigned char* src = g_zero16;
        if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_in && ch < p.cin)
          src = p.x + ((img_off + (size_t)gy * p.w_in + gx) * (size_t)p.ldx + ch) * sizeof(T);
        if (hp < C64_HPIX) glds16(src, halo + m * 1024);
      }
    }
  };

  // per-wave channel sums (fused global average pool): lane (l15, q) owns channels 16j + 4q + r;
  // flushed per image as one partial row per WAVE: chan_sum[img][blockIdx.x * 8 + wave][C]
  float csum[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[j][r] = 0.f;
  // Every wave owns one chan_sum row per image and WRITES each of them exactly once (its images come in increasing order): the sums of
  // an image when it moves on to the next, zeros for the images it never touched — no memset in front of the launch.
  int sum_img = -1;
  auto sum_row = [&](int img_) -> float* {
    return p.chan_sum + ((size_t)img_ * (gridDim.x * 8) + blockIdx.x * 8 + (tid >> 6)) * p.cout;
  };
  auto zero_rows = [&](int from, int to) {            // images [from, to)
    for (int im = from; im < to; ++im)
      if (l15 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j * 16 + q * 4 + r < p.cout) sum_row(im)[j * 16 + q * 4 + r] = 0.f;
      }
  };
  auto flush_sums = [&](int img_) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = csum[j][r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (l15 == 0 && j * 16 + q * 4 + r < p.cout) sum_row(img_)[j * 16 + q * 4 + r] = v;
        csum[j][r] = 0.f;
      }
  };

  // ---- prologue: tile 0 -> LDS -------------------------------------------------
  unsigned K = 0;                                   // pair-iterations of this workgroup
  for (unsigned pid = blockIdx.x; pid < npairs; pid += gridDim.x) ++K;
  {
    const unsigned t0 = tile_of(0);
    if (t0 != ~0u) dma_halo(t0);
  }
  MTX_WAIT_VMEM();
  MTX_LDS_BARRIER();

  f32x4 acc[4][4];
  u32x2_t rv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; rv[i][j] = u32x2_t{0u, 0u}; }

  for (unsigned s = 0; s < 2 * K + 1; ++s) {
    const bool mfma_slot = (int)(s & 1u) == grp;
    stamp(s, 0);
    if (mfma_slot) {
      // ================= MFMA slot: tile k of this group ============================================
      const unsigned k = (s - (unsigned)grp) >> 1;
      const unsigned lin = k < K ? tile_of(k) : ~0u;
      if (lin != ~0u) {
        const int img = (int)(lin / tiles_per_img);
        const int tile = (int)(lin % tiles_per_img);
        const int ty0 = (tile / p.tiles_x) * C64_T, tx0 = (tile % p.tiles_x) * C64_T;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // wave wv owns tile rows 4wv .. 4wv+3 (4 fragments of 16 px) x 64 couts (4 fragments).
        // Step order (kx, ks) outer, ky inner: the six halo-row fragments of one (kx, ks) serve all three ky
        // (tile row i under tap row ky reads halo row i + ky), so a tile costs 36 + 72 fragment reads for 288 MFMAs.
        auto load_w = [&](int st, v8 (&wf)[4]) {           // st = (kx * NKS + ks) * 3 + ky
          const int g = st / 3, ky = st - g * 3, kx = g / nks_c, ks = g - kx * nks_c, tap = ky * 3 + kx;
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const v8*>(wa[ks] + tap * 8192 + j * 2048);
        };
        auto load_x2 = [&](int g, int r0, v8 (&xr)[6]) {    // halo rows r0, r0 + 1 of group g
          const int kx = g / nks_c, ks = g - kx * nks_c;
#pragma unroll
          for (int r = r0; r < r0 + 2; ++r) xr[r] = *reinterpret_cast<const v8*>(xa[kx][ks] + r * (C64_HW * 128));
        };
        if (ABL != 1) {
          // ONE wave per SIMD runs this loop (its SIMD-mate is in the memory slot), so nothing but the wave's own
          // instruction order hides the LDS round trip.  The loop is fully unrolled; every read is base register +
          // immediate (no vector ALU instruction in the loop); the MFMAs are `asm volatile` and the reads sit between
          // compiler fences, so the order below IS the issue order: the six reads that feed LATER steps (next step's
          // four filter fragments, two halo rows of the next (kx, ks)) go out behind the first six MFMAs of the
          // current step, ten MFMAs (160 cycles) ahead of the wait in front of the next step.
          v8 wf[2][4], xr[2][6];
          constexpr int NG = 3 * nks_c, NST = NG * 3;
          load_x2(0, 0, xr[0]); load_x2(0, 2, xr[0]); load_x2(0, 4, xr[0]);
          load_w(0, wf[0]);
#pragma unroll
          for (int st = 0; st < NST; ++st) {
            const int g = st / 3, ky = st - g * 3;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                mfma_inplace<T>(acc[i][j], wf[st & 1][j], xr[g & 1][i + ky]);
                const int idx = j * 4 + i;
                if (idx < 6 && st + 1 < NST) {
                  const int rd = idx;                            // which read goes out behind this MFMA
                  if (rd < 4) {
                    C64_FENCE();
                    const int s1 = st + 1, g1 = s1 / 3, ky1 = s1 - g1 * 3, kx1 = g1 / nks_c, ks1 = g1 - kx1 * nks_c, tap1 = ky1 * 3 + kx1;
                    wf[s1 & 1][rd] = *reinterpret_cast<const v8*>(wa[ks1] + tap1 * 8192 + rd * 2048);
                    C64_FENCE();
                  } else if (rd < 6 && g + 1 < NG) {
                    C64_FENCE();
                    const int g1 = g + 1, kx1 = g1 / nks_c, ks1 = g1 - kx1 * nks_c, r = 2 * ky + (rd - 4);
                    xr[g1 & 1][r] = *reinterpret_cast<const v8*>(xa[kx1][ks1] + r * (C64_HW * 128));
                    C64_FENCE();
                  }
                }
              }
          }
          C64_MFMA_DRAIN();
        }
        stamp(s, 1);
        if (p.out_scale != nullptr && p.n > 1 && gt < 64)       // several images: this tile's per-channel output factors -> the group's LDS row (read after the slot barrier)
          scale_s[gt] = gt < p.cout ? p.out_scale[(size_t)img * p.cout + gt] : 0.f;
        if (RES && FAST_ACT && fast_ok && ty0 + C64_T <= p.h && tx0 + C64_T <= p.w_in) {
          // residual of a tile that lies inside the image: 8 descriptor loads of 16 bytes with tile-invariant lane offsets, in the
          // lane-exchanged layout of the stores (64-byte half lines per pixel); the epilogue's exchange puts the quads back
          const unsigned sbase = (unsigned)((((size_t)img * p.h + ty0) * p.w_in + tx0) * (size_t)p.ldres * sizeof(T));
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
              const u32x4 r4 = buf_load16(rbuf, res_off[i] + (unsigned)(jp * 32 * sizeof(T)), sbase);
              rv[i][2 * jp] = u32x2_t{r4[0], r4[1]}; rv[i][2 * jp + 1] = u32x2_t{r4[2], r4[3]};
            }
        } else
        if (RES) {   // this tile's residual, issued after the MFMA loop (its registers are not live inside it);
          // the latency hides behind the slot barrier and the next halo's LDS writes
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int oy = ty0 + wv * 4 + i, ox = tx0 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int co = j * 16 + q * 4;
              u32x2_t v = u32x2_t{0u, 0u};
              if (oy < p.h && ox < p.w_in && co < p.cout) {
                size_t opix; int oc = co;
                const int rimg = p.res_bcast ? 0 : img;
                if (p.ps == 2) {
                  const int cps = p.cout >> 2; const int g = co / cps; oc = co - g * cps;
                  opix = ((size_t)rimg * (2 * p.h) + (2 * oy + (g >> 1))) * (size_t)(2 * p.w_in) + (2 * ox + (g & 1));
                } else {
                  opix = ((size_t)rimg * p.h + oy) * (size_t)p.w_in + ox;
                }
                v = *reinterpret_cast<const u32x2_t*>(p.res + (opix * (size_t)p.ldres + oc) * sizeof(T));
              }
              rv[i][j] = v;
            }
          }
        }
      }
    } else if (s > (unsigned)grp) {
      // ================= memory slot: finish tile k, stage tile k+1, fetch tile k+2 ===================
      const unsigned k = (s - (unsigned)grp - 1) >> 1;
      const unsigned lin = k < K ? tile_of(k) : ~0u;
      const unsigned lin1 = k + 1 < K ? tile_of(k + 1) : ~0u;
      int stored_n = 0;                       // store instructions this wave issued in this slot (exact: see the wait below)
      // the next tile's halo goes in flight FIRST (this group's halo buffer is idle from the slot barrier on), so its latency runs
      // behind the epilogue below (whole RCAN graph 92.4 vs 96.5 ms with it issued after the stores)
      if (ABL != 3 && lin1 != ~0u) dma_halo(lin1);
      stamp(s, 2);
      // (1) epilogue straight from the accumulators: bias, activation, residual, 8-byte NHWC stores
      if (lin != ~0u) {
        const int img = (int)(lin / tiles_per_img);
        const int tile = (int)(lin % tiles_per_img);
        const int ty0 = (tile / p.tiles_x) * C64_T, tx0 = (tile % p.tiles_x) * C64_T;
        const bool want_sum = SUM;
        if (want_sum && img != sum_img) {
          if (sum_img >= 0) flush_sums(sum_img);
          zero_rows(sum_img + 1, img);                   // images this wave skipped (or everything before its first one)
          sum_img = img;
        }
        const bool fast = fast_ok && FAST_ACT && ty0 + C64_T <= p.h && tx0 + C64_T <= p.w_in;
        if (ABL != 4 && fast) {
          const unsigned sbase = (unsigned)((((size_t)img * p.h + ty0) * p.w_in + tx0) * (size_t)p.ldy * sizeof(T));
          f32x4 b4[4], sc4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            b4[j] = *reinterpret_cast<const f32x4*>(bias_s + j * 16 + q * 4);
            sc4[j] = p.out_scale != nullptr ? *reinterpret_cast<const f32x4*>(scale_s + j * 16 + q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            u32x2 o[4];
            if (RES) {               // the residual arrived as 16-byte chunks in the exchanged layout: the same exchange restores this lane's quads
#pragma unroll
              for (int jp = 0; jp < 2; ++jp) {
                uint32_t a0 = rv[i][2 * jp][0], a1 = rv[i][2 * jp][1], c0 = rv[i][2 * jp + 1][0], c1 = rv[i][2 * jp + 1][1];
                row_pair_exchange(a0, c0);
                row_pair_exchange(a1, c1);
                rv[i][2 * jp] = u32x2_t{a0, a1}; rv[i][2 * jp + 1] = u32x2_t{c0, c1};
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v4 ov;
              if (!has_post) {
                ov = epi_pack<T, ACT>(acc[i][j] + b4[j], p.act, p.act_param);
                if (want_sum) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) csum[j][r] += to_f32(ov[r]);
                }
              } else {       // y = out_scale * act(conv + bias) + res_scale * res, in fp32 with one rounding
                f32x4 v = acc[i][j] + b4[j];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(v[r], p.act, p.act_param);
                if (want_sum) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) csum[j][r] += to_f32(from_f32<T>(v[r]));
                }
                if (p.out_scale != nullptr) v = v * sc4[j];
                if (RES) {
                  const v4 g4 = __builtin_bit_cast(v4, rv[i][j]);
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] += p.res_scale * to_f32(g4[r]);
                }
                ov = epi_pack<T, MTX_ACT_NONE>(v, 0, 0.f);
              }
              o[j] = __builtin_bit_cast(u32x2, ov);
            }
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
              uint32_t a0 = o[2 * jp][0], a1 = o[2 * jp][1], c0 = o[2 * jp + 1][0], c1 = o[2 * jp + 1][1];
              row_pair_exchange(a0, c0);
              row_pair_exchange(a1, c1);
              buf_store16(ybuf, st_off[i] + (unsigned)(jp * 32 * sizeof(T)), u32x4{a0, a1, c0, c1}, sbase);
            }
          }
          stored_n = 8;
        } else
        if (ABL == 4) {   // keep EVERY accumulator live (an ablation must not let the MFMAs be DCE'd)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
        } else {
          // stores go through a buffer descriptor: lanes outside the image or past cout pass an out-of-range
          // offset (dropped by the range check), so every wave issues exactly 16 store instructions per tile
          unsigned pix_off[4];
          bool pix_ok[4], pix_in[4];
          const int vh = p.valid_hw ? p.valid_hw[0] : p.h, vw = p.valid_hw ? p.valid_hw[1] : p.w_in;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int oy = ty0 + wv * 4 + i, ox = tx0 + l15;
            pix_ok[i] = oy < p.h && ox < p.w_in;
            pix_in[i] = oy < vh && ox < vw;                    // inside the image (bucket plans: the canvas is larger)
            pix_off[i] = p.ps == 2 ? ((unsigned)img * (2u * p.h) + 2u * oy) * (2u * p.w_in) + 2u * ox
                                   : ((unsigned)img * p.h + oy) * (unsigned)p.w_in + ox;
          }
          const int cps = p.cout >> 2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = j * 16 + q * 4;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + co);
            int oc = co;
            unsigned sub = 0;
            if (p.ps == 2) { const int g = co / cps; oc = co - g * cps; sub = (unsigned)(g >> 1) * (2u * p.w_in) + (g & 1); }
            const bool ch_ok = co < p.cout;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float f[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) f[r] = apply_act_t<ACT>(acc[i][j][r] + b4[r], p.act, p.act_param);
              if (want_sum && pix_ok[i] && pix_in[i] && ch_ok) {   // pooled statistics of the values as stored (rounded to T)
#pragma unroll
                for (int r = 0; r < 4; ++r) { f[r] = to_f32(from_f32<T>(f[r])); csum[j][r] += f[r]; }
              }
              if (p.out_scale != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] *= scale_s[co + r];
              }
              if (RES) {
                const v4 g4 = __builtin_bit_cast(v4, rv[i][j]);
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] += p.res_scale * to_f32(g4[r]);
              }
              v4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(pix_in[i] ? f[r] : 0.f);
              const unsigned voff = (pix_ok[i] && ch_ok) ? ((pix_off[i] + sub) * (unsigned)p.ldy + (unsigned)oc) * (unsigned)sizeof(T)
                                                         : p.y_bytes;
              buf_store8(ybuf, voff, __builtin_bit_cast(u32x2, o));
            }
          }
          stored_n = 16;
        }
      }
      // (2) next tile's halo by LDS-DMA into this group's (now idle) halo buffer, then drain: the
      //     DMA and this tile's stores must have landed before the barrier that opens our MFMA slot.
      //     The drain overlaps the other group's MFMA slot.
      //     The tile's 8 (fast path) or 16 stores are the youngest operations on the counter and are NOT waited for: they retire under the next slots.
      stamp(s, 3);
      if (stored_n == 0) MTX_WAIT_VMEM();
      else if (stored_n == 8) MTX_WAIT_VMEM_BUT(8);
      else MTX_WAIT_VMEM_BUT(16);
      stamp(s, 4);
    }
    stamp(s, 5);
    MTX_LDS_BARRIER();
    stamp(s, 6);
  }
  if (SUM) {
    if (sum_img >= 0) flush_sums(sum_img);
    zero_rows(sum_img + 1, p.n);
  }
}

static int g_num_cus = 0;

static int c64_num_cus(const char** err) {
  if (g_num_cus == 0) {
#ifdef MTX_EMU
    g_num_cus = 3;
#else
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { if (err) *err = "conv2d: device query failed"; return -1; }
    g_num_cus = prop.multiProcessorCount;
#endif
  }
  return g_num_cus;
}

static unsigned c64_grid(int n, int h, int w) {
  const long total = (long)((w + C64_T - 1) / C64_T) * ((h + C64_T - 1) / C64_T) * n;
  const long npairs = (total + 1) / 2;
  const int cus = c64_num_cus(nullptr);
  return (unsigned)(npairs < cus ? npairs : cus);
}

// rows of the chan_sum buffer per image = one partial per wave of the persistent launch
int conv_c64_tiles(int n, int h, int w) {
  if (c64_num_cus(nullptr) < 0) return -1;
  return (int)c64_grid(n, h, w) * 8;
}

static unsigned long long conv_c64_out_bytes(const mtx_conv2d_args* a) {
  const unsigned long long px = (unsigned long long)a->n * a->h * a->w_in * (a->pixel_shuffle == 2 ? 4 : 1);
  return px * (unsigned long long)a->ldy * (a->dtype == MTX_F32 ? 4 : 2);
}

bool conv_c64_applicable(const mtx_conv2d_args* a) {
  if (a->act_after_res) return false;
  if (a->chan_sum != nullptr && a->res != nullptr) return false;        // no variant carries both register sets
  if (conv_c64_out_bytes(a) >= 0xFFFFFFF0ull) return false;   // 32-bit store offsets
  if ((unsigned long long)a->n * a->h * a->w_in * a->ldx * 2 >= 0xFFFFFFF0ull) return false;
  return a->ksize == 3 && a->stride == 1 && a->cin <= 64 && a->cout <= 64;
}

int conv_c64_launch(const mtx_conv2d_args* a, void* stream, const char** err) {
  ConvC64Params p;
  p.x = (const unsigned char*)a->x; p.w = (const unsigned char*)a->w; p.bias = a->bias;
  p.res = (const unsigned char*)a->res; p.y = (unsigned char*)a->y; p.chan_sum = a->chan_sum;
  p.n = a->n; p.h = a->h; p.w_in = a->w_in; p.cin = a->cin; p.cout = a->cout;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldres = a->ldres;
  p.act = a->act; p.act_param = a->act_param; p.res_scale = a->res_scale; p.ps = a->pixel_shuffle; p.res_bcast = a->res_broadcast_n;
  p.valid_hw = a->valid_hw;
  p.y_bytes = (unsigned)conv_c64_out_bytes(a);
  p.x_bytes = (unsigned)((unsigned long long)a->n * a->h * a->w_in * a->ldx * 2);
  p.out_scale = a->out_scale;
  {
    const unsigned long long rb = a->res ? (unsigned long long)(a->res_broadcast_n ? 1 : a->n) * a->h * a->w_in * (a->pixel_shuffle == 2 ? 4 : 1) * a->ldres * 2 : 0;
    p.res_bytes = rb < 0xFFFFFFF0ull ? (unsigned)rb : 0u;
  }
  p.tiles_x = (a->w_in + C64_T - 1) / C64_T;
  p.tiles_y = (a->h + C64_T - 1) / C64_T;
  if (c64_num_cus(err) < 0) return MTX_ERR_HIP;
  const unsigned grid = c64_grid(a->n, a->h, a->w_in);
#define C64_GO(TT, AB, AC, SM, RS) MTX_LAUNCH((conv3x3_c64_kernel<TT, AB, AC, SM, RS>), dim3(grid), dim3(512), 0, stream, p)
#define C64_ACT(TT, SM, RS) do { if (a->act == MTX_ACT_NONE) C64_GO(TT, 0, MTX_ACT_NONE, SM, RS); else if (a->act == MTX_ACT_RELU) C64_GO(TT, 0, MTX_ACT_RELU, SM, RS); \
                                 else C64_GO(TT, 0, -1, SM, RS); } while (0)
#define C64_VAR(TT) do { if (sum) C64_ACT(TT, true, false); else if (a->res != nullptr) C64_ACT(TT, false, true); else C64_ACT(TT, false, false); } while (0)
  const bool sum = a->chan_sum != nullptr;
  if (a->dtype == MTX_BF16) C64_VAR(__bf16);
  else if (a->dtype == MTX_F16) C64_VAR(_Float16);
  else { *err = "conv2d: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

}  // namespace mtx
