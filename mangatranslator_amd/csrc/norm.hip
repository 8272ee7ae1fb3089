// norm.hip — wave-reduced row normalisations (HBM-streaming: one read, one write).
//
//   LayerNorm / RMSNorm over [rows, C], one 64-lane wave per row, 16-byte chunks per lane held in
//   registers between the statistics and the normalise pass (two-pass variance, fp32), with the
//   adaLN modulation y*(1+scale)+shift of the FLUX DiT fused (diffusers AdaLayerNormZero, reached
//   from core/image/inpainting.py:877-887) — also Hiera's LayerNorms (SAM-2.1,
//   core/image/detection.py:505).
//   GroupNorm(+SiLU) over NHWC for the FLUX VAE: per-channel partial sums -> per-group statistics
//   -> apply, so activations are read twice and written once.
#include "mtx_device.h"

namespace mtx {

constexpr int NORM_MAXCH = 12;   // chunks of 8 per lane -> C <= 6144

// (rounds 1-4 kept the row as fp32 values in NCH x 8 registers — `norm_kernel`, in this file's history; round 5's packed-row kernel below gives
// the same bytes in 19.0 instead of 30.1 us on the adaLN rows of a FLUX block: profiles/r05_visit_o / _p logs)
// FULL: every lane owns NCH valid chunks (C == 512 NCH), no gamma / beta / activation — the adaLN rows of the FLUX blocks: straight-line
// code, no per-chunk exec masks, no per-element affine loads.  PRE (with FULL): the modulation rows are requested together with x instead of
// after the two reductions (their L2 latency leaves the critical path at the price of 8 NCH more registers).
// Round 5: the row kept PACKED between the passes (NCH x 4 registers instead of NCH x 8 fp32 values) and unpacked again where each pass needs
// it — a 16-bit -> fp32 unpack is one shift / convert per element, and the kernel has nothing but latency to hide: a 3072-wide row takes 74
// VGPRs instead of 105; with the modulation rows requested up front (PRE) 98 — every memory request of a row is in flight at once, which is
// what pays (19.0 against 21.3 us).  Additions and products run in the order of the rounds-1-4 kernel and nothing is fused (identical bytes,
// compared on hardware in round 5).  NORM_PIN keeps the optimiser from carrying the unpacked values across the passes.
#ifdef MTX_EMU
#define NORM_PIN(x) do { } while (0)
#else
#define NORM_PIN(x) asm volatile("" : "+v"(x))
#endif
template <typename T, int NCH, bool FULL, bool PRE>
__global__ __launch_bounds__(256) void norm_packed_kernel(mtx_norm_args p) {
#pragma clang fp contract(off)      // as in norm_kernel: no fused multiply-adds, so both kernels round alike (hardware visit o: the fused variance differed in the last bit of some rows)
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const long nch = p.c / 8;
  const T* X = reinterpret_cast<const T*>(p.x) + row * p.ldx;
  T* Y = reinterpret_cast<T*>(p.y) + row * p.ldy;
  u32x4 raw[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const long ch = lane + (long)i * 64;
    raw[i] = (FULL || ch < nch) ? *reinterpret_cast<const u32x4*>(X + ch * 8) : u32x4{0u, 0u, 0u, 0u};
  }
  const T* MS = reinterpret_cast<const T*>(p.mod_scale);
  const T* MH = reinterpret_cast<const T*>(p.mod_shift);
  const long mrow = p.rows_per > 0 ? row / p.rows_per : 0;
  u32x4 gs[PRE ? NCH : 1], gh[PRE ? NCH : 1];
  if (PRE) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const long ch = lane + (long)i * 64;
      gs[i] = MS ? *reinterpret_cast<const u32x4*>(MS + mrow * p.ldmod + ch * 8) : u32x4{0u, 0u, 0u, 0u};
      gh[i] = MH ? *reinterpret_cast<const u32x4*>(MH + mrow * p.ldmod + ch * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const long ch = lane + (long)i * 64;
    if (FULL || ch < nch) {
      float f[8];
      unpack8<T>(raw[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[e];
    }
  }
  float mean = 0.f;
  if (p.kind == 0) { s = wave_sum(s); mean = s / (float)p.c; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) NORM_PIN(raw[i]);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const long ch = lane + (long)i * 64;
    if (FULL || ch < nch) {
      float f[8];
      unpack8<T>(raw[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; ss += d * d; }
    }
  }
  ss = wave_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / (float)p.c + p.eps);
  const bool affine_vec = (((size_t)p.gamma | (size_t)p.beta) & 15) == 0;      // fp32 vectors, chunk offsets are multiples of 32 bytes
#pragma unroll
  for (int i = 0; i < NCH; ++i) NORM_PIN(raw[i]);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const long ch = lane + (long)i * 64;
    if (FULL || ch < nch) {
      float f[8], o[8];
      unpack8<T>(raw[i], f);
      // affine parameters of the chunk as one batch of 16-byte loads (element by element each load sat behind its own full wait: sixteen
      // dependent L2 round trips per chunk in the ISA); same products and sums, so the bytes do not change
      float ga[8], be[8];
      if (!FULL) {
        if (p.gamma != nullptr && affine_vec) {
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { ga[e] = g0[e]; ga[4 + e] = g1[e]; }
        } else if (p.gamma != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ga[e] = p.gamma[ch * 8 + e];
        }
        if (p.beta != nullptr && affine_vec) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { be[e] = b0[e]; be[4 + e] = b1[e]; }
        } else if (p.beta != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) be[e] = p.beta[ch * 8 + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = (f[e] - mean) * rstd;
        if (!FULL) {
          if (p.gamma) t *= ga[e];
          if (p.beta) t += be[e];
        }
        o[e] = t;
      }
      if (MS) { float g[8]; unpack8<T>(PRE ? gs[i] : *reinterpret_cast<const u32x4*>(MS + mrow * p.ldmod + ch * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= (1.f + g[e]); }
      if (MH) { float g[8]; unpack8<T>(PRE ? gh[i] : *reinterpret_cast<const u32x4*>(MH + mrow * p.ldmod + ch * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += g[e]; }
      if (!FULL && p.act != MTX_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = apply_act(o[e], p.act, 0.f);
      }
      raw[i] = pack8<T>(o);                                   // the result as rounded to T: what y holds and what the quantiser reads
      if (p.y != nullptr) *reinterpret_cast<u32x4*>(Y + ch * 8) = raw[i];
    }
  }
  if (p.q != nullptr) {
    unsigned char* Q = reinterpret_cast<unsigned char*>(p.q) + row * p.ldq;
    unsigned* S = reinterpret_cast<unsigned*>(p.q_scale);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const long ch = lane + (long)i * 64;
      if (FULL || (long)i * 64 < nch) {             // wave-uniform
        float f[8];
        unpack8<T>(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (FULL || ch < nch) ? f[e] : 0.f;
        unsigned w0, w1, word;
        mx_quantize_chunk(f, ch, w0, w1, word);
        if (FULL || ch < nch) {
          *reinterpret_cast<u32x2*>(Q + ch * 8) = u32x2{w0, w1};
          if ((ch & 15) == 0) S[(ch >> 4) * p.lds_q + row] = word;
        }
      }
    }
  }
}

int norm_f32_launch(const mtx_norm_args* a, void* stream, const char** err);      // f32ops.hip
int norm_launch(const mtx_norm_args* a, void* stream, const char** err) {
  if (a->dtype == MTX_F32) return norm_f32_launch(a, stream, err);
  if (!a->x || (!a->y && !a->q)) { *err = "norm: null operand"; return MTX_ERR_INVALID; }
  if (a->q && (!a->q_scale || a->c % 128 || a->ldq % 8 || a->lds_q < a->rows)) { *err = "norm (fp8 twin): needs q_scale, C % 128 == 0, ldq % 8 == 0, lds_q >= rows"; return MTX_ERR_INVALID; }
  if (a->c % 8 || a->ldx % 8 || a->ldy % 8 || a->c > NORM_MAXCH * 64 * 8 || a->c < 8) { *err = "norm: C must be a multiple of 8, <= 6144"; return MTX_ERR_INVALID; }
  if ((a->mod_scale || a->mod_shift) && (a->ldmod % 8 || a->rows_per < 1)) { *err = "norm: bad modulation layout"; return MTX_ERR_INVALID; }
  if (a->kind != 0 && a->kind != 1) { *err = "norm: kind must be 0 (LayerNorm) or 1 (RMSNorm)"; return MTX_ERR_INVALID; }
  if (a->rows < 1) return MTX_OK;
  const unsigned blocks = (unsigned)((a->rows + 3) / 4);
  if (a->dtype != MTX_BF16 && a->dtype != MTX_F16) { *err = "norm: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  const long per_lane = (a->c / 8 + 63) / 64;
  // straight-line form (rows of whole 512-chunks without affine / activation: the adaLN rows of the FLUX blocks, with the modulation rows requested
  // together with x) or the general one.  Round 5 measured two more forms (modulation after the reductions: 21.6 against 18.8 us; held to 96
  // registers: 23.1) — dropped, profiles/r05_visit_o_*.log.
  const bool full = a->c % 512 == 0 && (per_lane == 2 || per_lane == 4 || per_lane == 6 || per_lane == NORM_MAXCH) && !a->gamma && !a->beta && a->act == MTX_ACT_NONE;
#define MTX_NORM_T(TT, N) do { if (!full) MTX_LAUNCH((norm_packed_kernel<TT, N, false, false>), dim3(blocks), dim3(256), 0, stream, *a); \
                               else MTX_LAUNCH((norm_packed_kernel<TT, N, true, true>), dim3(blocks), dim3(256), 0, stream, *a); } while (0)
#define MTX_NORM(N) do { if (a->dtype == MTX_BF16) MTX_NORM_T(__bf16, N); else MTX_NORM_T(_Float16, N); } while (0)
  if (per_lane <= 2) MTX_NORM(2);
  else if (per_lane <= 4) MTX_NORM(4);
  else if (per_lane <= 6) MTX_NORM(6);
  else MTX_NORM(NORM_MAXCH);
#undef MTX_NORM_T
#undef MTX_NORM
  return MTX_OK;
}

// ---- GroupNorm -----------------------------------------------------------------------------------
// workspace layout (fp32, MTX_GROUPNORM_WS_FLOATS): [N][G][2] {mean, rstd}  followed by  [N][BX][C][2] per-block per-channel {sum, sumsq},
// BX = ceil(HW / MTX_GN_PIX_PER_BLOCK).  Round 4: the per-channel sums used to be accumulated with fp32 atomicAdd from every block — the
// order of those additions, hence the last bits of mean / rstd and of every VAE output, changed from run to run.  Now each block stores
// its partial sums and one workgroup per (image, group) adds them in a fixed order (double accumulators): identical calls give identical
// bytes (tests/test_determinism_gpu.py), and nothing has to be cleared first.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(mtx_groupnorm_args p, int pix_per_block) {
  __shared__ float red[256 * 16];
  const int C8 = (int)(p.c / 8);
  const int tid = threadIdx.x;
  const int col = tid % C8, sub = tid / C8, nsub = 256 / C8;
  const long n = blockIdx.y;
  const long p0 = (long)blockIdx.x * pix_per_block;
  long p1 = p0 + pix_per_block; if (p1 > p.hw) p1 = p.hw;
  const T* X = reinterpret_cast<const T*>(p.x) + n * p.hw * p.c;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  for (long px = p0 + sub; px < p1; px += nsub) {
    float f[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(X + px * p.c + col * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s[e]; red[tid * 16 + 8 + e] = ss[e]; }
  __syncthreads();
  if (sub == 0) {
    float* part = p.workspace + p.n * p.groups * 2 + ((n * gridDim.x + blockIdx.x) * p.c + col * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f, b = 0.f;
      for (int k = 0; k < nsub; ++k) { a += red[(k * C8 + col) * 16 + e]; b += red[(k * C8 + col) * 16 + 8 + e]; }
      part[e * 2] = a;
      part[e * 2 + 1] = b;
    }
  }
}

// one workgroup per (image, group): thread t adds the (block, channel) pairs t, t + 256, ... of the group, then a fixed binary tree over the
// 256 threads — the same order on every run
__global__ __launch_bounds__(256) void gn_finalize_kernel(mtx_groupnorm_args p, int nblocks) {
  __shared__ double rs[256], rq[256];
  const long n = blockIdx.x / p.groups, g = blockIdx.x % p.groups;
  const long cg = p.c / p.groups;
  const float* part = p.workspace + p.n * p.groups * 2 + n * (long)nblocks * p.c * 2;
  double s = 0.0, ss = 0.0;
  const long pairs = (long)nblocks * cg;
  for (long i = threadIdx.x; i < pairs; i += 256) {
    const long b = i / cg, c = g * cg + i % cg;
    s += (double)part[(b * p.c + c) * 2];
    ss += (double)part[(b * p.c + c) * 2 + 1];
  }
  rs[threadIdx.x] = s; rq[threadIdx.x] = ss;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { rs[threadIdx.x] += rs[threadIdx.x + w]; rq[threadIdx.x] += rq[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cnt = (double)(cg * p.hw);
    const double mean = rs[0] / cnt;
    double var = rq[0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    float* out = p.workspace + (n * p.groups + g) * 2;
    out[0] = (float)mean;
    out[1] = 1.0f / sqrtf((float)var + p.eps);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(mtx_groupnorm_args p) {
  const long C8 = p.c / 8;
  const long total = p.n * p.hw * C8;
  const long cg = p.c / p.groups;
  const float* st = p.workspace;
  const T* X = reinterpret_cast<const T*>(p.x);
  T* Y = reinterpret_cast<T*>(p.y);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long c0 = (idx % C8) * 8;
    const long pix = idx / C8;
    const long n = pix / p.hw;
    float f[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(X + pix * p.c + c0), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long g = (c0 + e) / cg;
      const float mean = st[(n * p.groups + g) * 2], rstd = st[(n * p.groups + g) * 2 + 1];
      float t = (f[e] - mean) * rstd;
      if (p.gamma) t *= p.gamma[c0 + e];
      if (p.beta) t += p.beta[c0 + e];
      f[e] = apply_act(t, p.act, 0.f);
    }
    *reinterpret_cast<u32x4*>(Y + pix * p.c + c0) = pack8<T>(f);
  }
}

int groupnorm_launch(const mtx_groupnorm_args* a, void* stream, const char** err) {
  if (!a->x || !a->y || !a->workspace) { *err = "groupnorm: null operand"; return MTX_ERR_INVALID; }
  const long C8 = a->c / 8;
  if (a->c % 8 || C8 < 1 || C8 > 256 || (256 % C8) != 0 || a->groups < 1 || a->c % a->groups) {
    *err = "groupnorm: C/8 must divide 256 and groups must divide C"; return MTX_ERR_INVALID;
  }
  if (a->n < 1 || a->hw < 1) return MTX_OK;
  const int ppb = MTX_GN_PIX_PER_BLOCK;
  const int nb = (int)((a->hw + ppb - 1) / ppb);
  dim3 g1((unsigned)nb, (unsigned)a->n);
  long tot = a->n * a->hw * C8;
  long blocks = (tot + 255) / 256; if (blocks > 8192) blocks = 8192;
  if (a->dtype == MTX_BF16) {
    MTX_LAUNCH((gn_stats_kernel<__bf16>), g1, dim3(256), 0, stream, *a, ppb);
    MTX_LAUNCH(gn_finalize_kernel, dim3((unsigned)(a->n * a->groups)), dim3(256), 0, stream, *a, nb);
    MTX_LAUNCH((gn_apply_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  } else if (a->dtype == MTX_F16) {
    MTX_LAUNCH((gn_stats_kernel<_Float16>), g1, dim3(256), 0, stream, *a, ppb);
    MTX_LAUNCH(gn_finalize_kernel, dim3((unsigned)(a->n * a->groups)), dim3(256), 0, stream, *a, nb);
    MTX_LAUNCH((gn_apply_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  } else { *err = "groupnorm: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

}  // namespace mtx
