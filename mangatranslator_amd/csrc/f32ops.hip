// f32ops.hip — fp32 instantiations of the ops SAM-2.1's prompt encoder / two-way transformer / mask head are made of (MTX_F32 in the
// `dtype` field of mtx_gemm / mtx_attention / mtx_norm / mtx_elementwise): plain fp32 operands, fp32 arithmetic on the vector ALUs,
// accurate expf / erff.  Round 4, for `Sam2Hip(precision="high")` (reference core/image/detection.py:494-510: the `> 0` masks are what the
// page flow keeps; DESIGN.md §3's error budget: with 16-bit storage the mask decoder contributes 0.0067 + 0.0053 of 0.0130 logit units).
// These are small problems — 28 GFLOP per page at eight boxes against the trunk's 1.6 TFLOP — so the kernels are written to be obviously
// right: an LDS-tiled FMA GEMM, a wave-per-query attention, a wave-per-row LayerNorm, element-wise maps.  First hardware run: round 5
// (tests/test_ops_gpu.py::test_f32_ops, profiles/r05_visit_a_sam_high_first_run.log).
#include "mtx_device.h"
#include <math.h>

namespace mtx {

__device__ __forceinline__ float act_f32(float v, int act, float p) {
  switch (act) {
    case MTX_ACT_RELU: return v > 0.f ? v : 0.f;
    case MTX_ACT_SILU: return v / (1.f + expf(-v));
    case MTX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case MTX_ACT_GELU_TANH: return 0.5f * v * (1.f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    case MTX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case MTX_ACT_LEAKY: return v > 0.f ? v : v * p;
    default: return v;
  }
}

// ---- GEMM: C[m, n] = act(alpha * sum_k A[m, k] W[n, k] + bias[n]) + res[m, n]; strided batches in blockIdx.z ------------------------
constexpr int F_TM = 128, F_TN = 64, F_TK = 16;
__global__ __launch_bounds__(256) void gemm_f32_kernel(mtx_gemm_args p) {
  __shared__ float As[F_TK][F_TM + 4];
  __shared__ float Ws[F_TK][F_TN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;          // micro-tile: rows ty*8..+7, columns tx*4..+3
  const long m0 = (long)blockIdx.y * F_TM, n0 = (long)blockIdx.x * F_TN, z = blockIdx.z;
  const float* A = reinterpret_cast<const float*>(p.a) + z * p.a_bstride;
  const float* W = reinterpret_cast<const float*>(p.w) + z * p.w_bstride;
  float acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (long k0 = 0; k0 < p.k; k0 += F_TK) {
    // A tile: 128 rows x 16 k = 2048 values, 8 per thread (row = tid / 2, k = (tid & 1) * 8 ..+7); W tile: 64 x 16, 4 per thread
    {
      const long r = m0 + (tid >> 1);
      const int kq = (tid & 1) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long k = k0 + kq + i;
        As[kq + i][tid >> 1] = (r < p.m && k < p.k) ? A[r * p.lda + k] : 0.f;
      }
      const long c = n0 + (tid >> 2);
      const int kw = (tid & 3) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long k = k0 + kw + i;
        Ws[kw + i][tid >> 2] = (c < p.n && k < p.k) ? W[c * p.ldw + k] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < F_TK; ++k) {
      float a[8], w[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = As[k][ty * 8 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = Ws[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* Cz = reinterpret_cast<float*>(p.c) + z * p.c_bstride;
  const float* R = p.res ? reinterpret_cast<const float*>(p.res) + z * p.res_bstride : nullptr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long r = m0 + ty * 8 + i;
    if (r >= p.m) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long c = n0 + tx * 4 + j;
      if (c >= p.n) continue;
      float v = acc[i][j] * p.alpha + (p.bias ? p.bias[c] : 0.f);
      v = act_f32(v, p.act, p.act_param);
      if (R) v += R[r * p.ldres + c];
      Cz[r * p.ldc + c] = v;
    }
  }
}

// ---- the same GEMM on the matrix pipe: v_mfma_f32_32x32x2_f32 — fp32 operands, fp32 products and sums (exact fp32, no TF32 on gfx950), at
// the vector ALUs' peak rate but without their LDS-operand traffic (MI355X_MICROARCH.md: 155 TFLOP/s measured).  Round 5: with precision
// "high" SAM's mask decoder spends its time in the image-side projections (8 boxes x 4 096 tokens x 256 -> 128 / 256: 28 GFLOP per
// page) and ran 4.5 ms against 1.6 ms in 16-bit storage (profiles/r05_bench_config2_sam_high.json).
// 128 x 64 tile, 4 waves in a 2 x 2 grid of 64 x 32 wave tiles (two 32 x 32 accumulator blocks each); the k-major LDS images of the FMA
// kernel; lane (l31, hi) feeds A[row l31][k = hi] and B[k = hi][column l31] of a k pair, and holds C rows (r & 3) + 8 (r >> 2) + 4 hi of
// column l31 — so a wave's stores are 128-byte row segments.
__device__ __forceinline__ f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
#ifdef MTX_EMU
  return emu_mfma_32x32x2_f32(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(mtx_gemm_args p) {
  __shared__ float As[F_TK][F_TM + 4];
  __shared__ float Ws[F_TK][F_TN + 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wv >> 1, wn = wv & 1;
  const long m0 = (long)blockIdx.y * F_TM, n0 = (long)blockIdx.x * F_TN, z = blockIdx.z;
  const float* A = reinterpret_cast<const float*>(p.a) + z * p.a_bstride;
  const float* W = reinterpret_cast<const float*>(p.w) + z * p.w_bstride;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // the next K step's operands are requested into registers before this step's MFMAs and written to LDS after them: with two workgroups per
  // CU (SAM's 32 768 x 128 projections) the global-load latency was the whole step (round 5: four such launches 0.20 -> 0.16 ms, profiles/r05_visit_p_*.log)
  const long ar = m0 + (tid >> 1), wc = n0 + (tid >> 2);
  const int kq = (tid & 1) * 8, kw = (tid & 3) * 4;
  const bool row_ok = ar < p.m, col_ok = wc < p.n;
  const float* Arow = A + (row_ok ? ar : 0) * p.lda;
  const float* Wrow = W + (col_ok ? wc : 0) * p.ldw;
  const bool a_vec = (p.lda & 3) == 0 && ((size_t)Arow & 15) == 0, w_vec = (p.ldw & 3) == 0 && ((size_t)Wrow & 15) == 0;
  float ra[8], rw[4];
  auto fetch = [&](long k0) {
    if (row_ok && a_vec && k0 + kq + 8 <= p.k) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(Arow + k0 + kq), v1 = *reinterpret_cast<const f32x4*>(Arow + k0 + kq + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra[i] = v0[i]; ra[4 + i] = v1[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) ra[i] = (row_ok && k0 + kq + i < p.k) ? Arow[k0 + kq + i] : 0.f;
    }
    if (col_ok && w_vec && k0 + kw + 4 <= p.k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wrow + k0 + kw);
#pragma unroll
      for (int i = 0; i < 4; ++i) rw[i] = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) rw[i] = (col_ok && k0 + kw + i < p.k) ? Wrow[k0 + kw + i] : 0.f;
    }
  };
  fetch(0);
  for (long k0 = 0; k0 < p.k; k0 += F_TK) {
#pragma unroll
    for (int i = 0; i < 8; ++i) As[kq + i][tid >> 1] = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) Ws[kw + i][tid >> 2] = rw[i];
    __syncthreads();
    if (k0 + F_TK < p.k) fetch(k0 + F_TK);
#pragma unroll
    for (int kk = 0; kk < F_TK; kk += 2) {
      const float b = Ws[kk + hi][wn * 32 + l31];
      const float a0 = As[kk + hi][wm * 64 + l31], a1 = As[kk + hi][wm * 64 + 32 + l31];
      acc[0] = mfma_f32_32x32x2(a0, b, acc[0]);
      acc[1] = mfma_f32_32x32x2(a1, b, acc[1]);
    }
    __syncthreads();
  }
  float* Cz = reinterpret_cast<float*>(p.c) + z * p.c_bstride;
  const float* R = p.res ? reinterpret_cast<const float*>(p.res) + z * p.res_bstride : nullptr;
  const long c = n0 + wn * 32 + l31;
  if (c >= p.n) return;
  const float bias = p.bias ? p.bias[c] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row >= p.m) continue;
      float v = acc[i][r] * p.alpha + bias;
      v = act_f32(v, p.act, p.act_param);
      if (R) v += R[row * p.ldres + c];
      Cz[row * p.ldc + c] = v;
    }
}

// ---- few rows, long K (the token-side MLP-out of SAM's two-way blocks: 72 x 256 from K = 2 048): the tiled kernels above give that
// problem four workgroups walking 128 K steps each (0.27 ms per launch on MI355X).  Here a wave owns one output column and SK_R rows:
// its lanes stride over K in 16-byte pieces (coalesced), each keeps SK_R partial dot products, one wave reduction per row at the end —
// rows / SK_R x n waves (2 304 for the shape above), A re-read from the L2.
constexpr int SK_R = 8;
__global__ __launch_bounds__(256) void gemm_f32_skinny_kernel(mtx_gemm_args p) {
  const int lane = threadIdx.x & 63;
  const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6), r0 = (long)blockIdx.y * SK_R, z = blockIdx.z;
  if (c >= p.n) return;                                                   // wave-uniform
  const float* A = reinterpret_cast<const float*>(p.a) + z * p.a_bstride;
  const float* W = reinterpret_cast<const float*>(p.w) + z * p.w_bstride + c * p.ldw;
  float acc[SK_R];
#pragma unroll
  for (int i = 0; i < SK_R; ++i) acc[i] = 0.f;
  for (long k = (long)lane * 4; k < p.k; k += 256) {                      // K % 4 == 0 (launcher)
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + k);
#pragma unroll
    for (int i = 0; i < SK_R; ++i) {
      const long r = r0 + i < p.m ? r0 + i : p.m - 1;
      const f32x4 a = *reinterpret_cast<const f32x4*>(A + r * p.lda + k);
      acc[i] = fmaf(a[0], w[0], fmaf(a[1], w[1], fmaf(a[2], w[2], fmaf(a[3], w[3], acc[i]))));
    }
  }
  float* Cz = reinterpret_cast<float*>(p.c) + z * p.c_bstride;
  const float* R = p.res ? reinterpret_cast<const float*>(p.res) + z * p.res_bstride : nullptr;
  const float bias = p.bias ? p.bias[c] : 0.f;
#pragma unroll
  for (int i = 0; i < SK_R; ++i) {
    const float t = wave_sum(acc[i]);
    if (lane == i && r0 + i < p.m) {
      float v = act_f32(t * p.alpha + bias, p.act, p.act_param);
      if (R) v += R[(r0 + i) * p.ldres + c];
      Cz[(r0 + i) * p.ldc + c] = v;
    }
  }
}

int gemm_f32_launch(const mtx_gemm_args* a, void* stream, const char** err) {
  if (!a->a || !a->w || !a->c) { *err = "gemm f32: null operand"; return MTX_ERR_INVALID; }
  if (a->gate || a->glu_q || a->in_dtype == MTX_F8 || (a->out_dtype != MTX_F32 && a->out_dtype != a->dtype)) {
    *err = "gemm f32: gate / glu / fp8 operands and 16-bit outputs are not part of the fp32 path"; return MTX_ERR_UNSUPPORTED;
  }
  if (a->m < 1 || a->n < 1 || a->batch < 1) return MTX_OK;
  if (a->batch > 65535) { *err = "gemm f32: batch > 65535"; return MTX_ERR_INVALID; }
  dim3 grid((unsigned)((a->n + F_TN - 1) / F_TN), (unsigned)((a->m + F_TM - 1) / F_TM), (unsigned)a->batch);
  if (grid.y > 65535) { *err = "gemm f32: more than 65535 row tiles"; return MTX_ERR_INVALID; }
  if (a->m <= 128 && a->k >= 1024 && a->k % 4 == 0 && a->lda % 4 == 0 && a->ldw % 4 == 0 && a->a_bstride % 4 == 0 && a->w_bstride % 4 == 0 &&
      (((size_t)a->a | (size_t)a->w) & 15) == 0 && !(a->flags & MTX_GEMM_FORCE_TILE256)) {
    const dim3 sgrid((unsigned)((a->n + 3) / 4), (unsigned)((a->m + SK_R - 1) / SK_R), (unsigned)a->batch);
    MTX_LAUNCH(gemm_f32_skinny_kernel, sgrid, dim3(256), 0, stream, *a);
    return MTX_OK;
  }
  // from a few row tiles up the matrix pipe wins; the token-side GEMMs (72 rows) stay on the vector ALUs (MTX_GEMM_FORCE_TILE256 forces the
  // matrix kernel: tests)
  if (a->m >= 256 || (a->flags & MTX_GEMM_FORCE_TILE256)) MTX_LAUNCH(gemm_f32_mfma_kernel, grid, dim3(256), 0, stream, *a);
  else MTX_LAUNCH(gemm_f32_kernel, grid, dim3(256), 0, stream, *a);
  return MTX_OK;
}

// ---- attention: one wave per (batch, head, query); its 64 lanes share the keys ------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_f32_kernel(mtx_attn_args p) {
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long total = p.batch * p.heads * p.sq;
  if (unit >= total) return;
  const long qi = unit % p.sq, h = (unit / p.sq) % p.heads, b = unit / (p.sq * p.heads);
  const float* Q = reinterpret_cast<const float*>(p.q) + b * p.q_bs + qi * p.q_ss + h * p.q_hs;
  const float* K = reinterpret_cast<const float*>(p.k) + b * p.k_bs + h * p.k_hs;
  const float* V = reinterpret_cast<const float*>(p.v) + b * p.v_bs + h * p.v_hs;
  float q[D], acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) { q[c] = Q[c] * p.scale; acc[c] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (long j = lane; j < p.sk; j += 64) {
    const float* kj = K + j * p.k_ss;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s = fmaf(q[c], kj[c], s);
    const float mn = s > m ? s : m;
    const float corr = expf(m - mn), e = expf(s - mn);          // first key of a lane: expf(-inf) = 0
    l = l * corr + e;
    const float* vj = V + j * p.v_ss;
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = acc[c] * corr + e * vj[c];
    m = mn;
  }
  const float M = wave_max(m);
  const float f = (m == -INFINITY) ? 0.f : expf(m - M);          // lanes without a key
  const float L = wave_sum(l * f);
  float* O = reinterpret_cast<float*>(p.o) + b * p.o_bs + qi * p.o_ss + h * p.o_hs;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float o = wave_sum(acc[c] * f);
    if (lane == 0) O[c] = o / L;
  }
}

// Few keys, many queries (SAM's image-to-token attention: 4 096 queries per box and head against 9 tokens): with a wave per query only
// `sk` of the 64 lanes have a key.  Here a LANE owns a query — q and the accumulator in its registers, the keys walked in order with the
// same running-max update — and the K / V rows, shared by all queries of a (batch, head), are broadcast loads.
template <int D>
__global__ __launch_bounds__(256) void attn_f32_rows_kernel(mtx_attn_args p) {
  const long unit = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = p.batch * p.heads * p.sq;
  if (unit >= total) return;
  const long qi = unit % p.sq, h = (unit / p.sq) % p.heads, b = unit / (p.sq * p.heads);
  const float* Q = reinterpret_cast<const float*>(p.q) + b * p.q_bs + qi * p.q_ss + h * p.q_hs;
  const float* K = reinterpret_cast<const float*>(p.k) + b * p.k_bs + h * p.k_hs;
  const float* V = reinterpret_cast<const float*>(p.v) + b * p.v_bs + h * p.v_hs;
  float q[D], acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) { q[c] = Q[c] * p.scale; acc[c] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (long j = 0; j < p.sk; ++j) {
    const float* kj = K + j * p.k_ss;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s = fmaf(q[c], kj[c], s);
    const float mn = s > m ? s : m;
    const float corr = expf(m - mn), e = expf(s - mn);          // first key: expf(-inf) = 0
    l = l * corr + e;
    const float* vj = V + j * p.v_ss;
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = acc[c] * corr + e * vj[c];
    m = mn;
  }
  float* O = reinterpret_cast<float*>(p.o) + b * p.o_bs + qi * p.o_ss + h * p.o_hs;
#pragma unroll
  for (int c = 0; c < D; ++c) O[c] = acc[c] / l;
}

int attn_f32_launch(const mtx_attn_args* a, void* stream, const char** err) {
  if (!a->q || !a->k || !a->v || !a->o) { *err = "attention f32: null operand"; return MTX_ERR_INVALID; }
  if (a->q8 || (a->flags & MTX_ATTN_Q_PRESCALED)) { *err = "attention f32: fp8 output / pre-scaled q are not part of the fp32 path"; return MTX_ERR_UNSUPPORTED; }
  if (a->sk < 1) { *err = "attention f32: no keys"; return MTX_ERR_INVALID; }
  const long total = a->batch * a->heads * a->sq;
  if (total < 1) return MTX_OK;
  if (a->sk <= 32 && a->sq >= 64 && a->d <= 32) {          // a lane per query
    const dim3 rgrid((unsigned)((total + 255) / 256));
    switch (a->d) {
      case 8: MTX_LAUNCH(attn_f32_rows_kernel<8>, rgrid, dim3(256), 0, stream, *a); return MTX_OK;
      case 16: MTX_LAUNCH(attn_f32_rows_kernel<16>, rgrid, dim3(256), 0, stream, *a); return MTX_OK;
      case 32: MTX_LAUNCH(attn_f32_rows_kernel<32>, rgrid, dim3(256), 0, stream, *a); return MTX_OK;
      default: break;
    }
  }
  const dim3 grid((unsigned)((total + 3) / 4));
  switch (a->d) {
    case 8: MTX_LAUNCH(attn_f32_kernel<8>, grid, dim3(256), 0, stream, *a); break;
    case 16: MTX_LAUNCH(attn_f32_kernel<16>, grid, dim3(256), 0, stream, *a); break;
    case 32: MTX_LAUNCH(attn_f32_kernel<32>, grid, dim3(256), 0, stream, *a); break;
    case 64: MTX_LAUNCH(attn_f32_kernel<64>, grid, dim3(256), 0, stream, *a); break;
    default: *err = "attention f32: head dim must be 8, 16, 32 or 64"; return MTX_ERR_UNSUPPORTED;
  }
  return MTX_OK;
}

// ---- LayerNorm over the last dim, one wave per row (two passes over the row: mean, then centred variance) ------------------------------------
// TO = float, or (round 6, mtx_norm_args.out_dtype) the 16-bit type of the linear that follows: the fp32 residual stream of SAM's precision
// "high" is normalised in fp32 and rounded ONCE here — no fp32 result, no conversion pass.  Rows of up to 64 x 4 x NCH values stay in registers
// between the passes (16-byte loads); wider or unaligned rows take the element loop.
template <typename TO> __device__ __forceinline__ TO norm_out(float v);
template <> __device__ __forceinline__ float norm_out<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 norm_out<__bf16>(float v) { return from_f32<__bf16>(v); }
template <> __device__ __forceinline__ _Float16 norm_out<_Float16>(float v) { return from_f32<_Float16>(v); }

template <typename TO>
__global__ __launch_bounds__(256) void norm_f32_kernel(mtx_norm_args p) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= p.rows) return;
  const float* X = reinterpret_cast<const float*>(p.x) + row * p.ldx;
  TO* Y = reinterpret_cast<TO*>(p.y) + row * p.ldy;
  constexpr int NCH = 5;                                   // 5 x 256 = 1280 values per row in registers (Hiera-L: 144 .. 1152)
  const bool fits = (p.c & 3) == 0 && p.c <= 256 * NCH && (p.ldx & 3) == 0 && ((size_t)p.x & 15) == 0;
  if (fits) {
    f32x4 xv[NCH];
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const long c = (long)(ch * 64 + lane) * 4;
      xv[ch] = c < p.c ? *reinterpret_cast<const f32x4*>(X + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (xv[ch][0] + xv[ch][1]) + (xv[ch][2] + xv[ch][3]);
    }
    const float mean = wave_sum(s) / (float)p.c;
    float v = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const long c = (long)(ch * 64 + lane) * 4;
      if (c < p.c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[ch][e] - mean; v = fmaf(d, d, v); }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)p.c + p.eps);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const long c = (long)(ch * 64 + lane) * 4;
      if (c < p.c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (xv[ch][e] - mean) * rstd;
          if (p.gamma) t *= p.gamma[c + e];
          if (p.beta) t += p.beta[c + e];
          Y[c + e] = norm_out<TO>(act_f32(t, p.act, 0.f));
        }
      }
    }
    return;
  }
  float s = 0.f;
  for (long c = lane; c < p.c; c += 64) s += X[c];
  const float mean = wave_sum(s) / (float)p.c;
  float v = 0.f;
  for (long c = lane; c < p.c; c += 64) { const float d = X[c] - mean; v = fmaf(d, d, v); }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)p.c + p.eps);
  for (long c = lane; c < p.c; c += 64) {
    float t = (X[c] - mean) * rstd;
    if (p.gamma) t *= p.gamma[c];
    if (p.beta) t += p.beta[c];
    Y[c] = norm_out<TO>(act_f32(t, p.act, 0.f));
  }
}

int norm_f32_launch(const mtx_norm_args* a, void* stream, const char** err) {
  if (!a->x || !a->y) { *err = "norm f32: null operand"; return MTX_ERR_INVALID; }
  if (a->kind != 0 || a->mod_scale || a->mod_shift || a->q) { *err = "norm f32: LayerNorm without modulation / fp8 twin only"; return MTX_ERR_UNSUPPORTED; }
  if (a->rows < 1 || a->c < 1) return MTX_OK;
  const dim3 grid((unsigned)((a->rows + 3) / 4));
  if (a->out_dtype == MTX_F32) MTX_LAUNCH(norm_f32_kernel<float>, grid, dim3(256), 0, stream, *a);
  else if (a->out_dtype == MTX_BF16) MTX_LAUNCH(norm_f32_kernel<__bf16>, grid, dim3(256), 0, stream, *a);
  else if (a->out_dtype == MTX_F16) MTX_LAUNCH(norm_f32_kernel<_Float16>, grid, dim3(256), 0, stream, *a);
  else { *err = "norm f32: out_dtype must be f32, bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

// ---- element-wise maps over [n, h, w, c] with per-pixel strides --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ew_f32_kernel(mtx_ew_args p) {
  const long pixels = p.n * p.h * p.w, total = pixels * p.c;
  const float* A = reinterpret_cast<const float*>(p.a);
  const float* B = reinterpret_cast<const float*>(p.b);
  float* Y = reinterpret_cast<float*>(p.y);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long c = idx % p.c, px = idx / p.c;
    switch (p.kind) {
      case MTX_EW_ADD: Y[px * p.ldy + c] = A[px * p.lda + c] + B[px * p.ldb + c]; break;
      case MTX_EW_MUL: Y[px * p.ldy + c] = A[px * p.lda + c] * B[px * p.ldb + c]; break;
      case MTX_EW_ACT: Y[px * p.ldy + c] = act_f32(A[px * p.lda + c], p.act, p.act_param); break;
      case MTX_EW_COPY: Y[px * p.ldy + c] = A[px * p.lda + c]; break;
      case MTX_EW_ROW_GATHER: Y[px * p.ldy + c] = A[(long)reinterpret_cast<const int32_t*>(p.s)[px] * p.lda + c]; break;
      case MTX_EW_SHUFFLE2_ADD: {
        // a holds, per input pixel (n, y, x), the four output pixels' channels: column (dy * 2 + dx) * c + ch; b (optional) is the skip
        // feature at output resolution, one image for every n when lds == 0 (lds = its per-sample stride in elements otherwise)
        const long x = px % p.w, y = (px / p.w) % p.h, n = px / (p.w * p.h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long oy = 2 * y + (q >> 1), ox = 2 * x + (q & 1);
          const long opix = (n * 2 * p.h + oy) * 2 * p.w + ox;
          float v = A[px * p.lda + q * p.c + c];
          if (B) v += B[n * p.lds + (oy * 2 * p.w + ox) * p.ldb + c];
          Y[opix * p.ldy + c] = v;
        }
        break;
      }
      default: break;
    }
  }
}
// 16-bit storage -> fp32 (i0 = the source's mtx_dtype)
template <typename T>
__global__ __launch_bounds__(256) void cvt_f32_kernel(mtx_ew_args p) {
  const long pixels = p.n * p.h * p.w, total = pixels * p.c;
  const T* A = reinterpret_cast<const T*>(p.a);
  float* Y = reinterpret_cast<float*>(p.y);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long c = idx % p.c, px = idx / p.c;
    Y[px * p.ldy + c] = (float)A[px * p.lda + c];
  }
}

// fp32 -> 16-bit storage (i0 = the destination's mtx_dtype), i1 copies side by side: y[px][j * c + ch] for j < i1 — the [x | x] operand of a
// linear whose weights are [W_hi | W_lo] pairs, written by the conversion itself
template <typename T>
__global__ __launch_bounds__(256) void cvt_16_kernel(mtx_ew_args p) {
  const long pixels = p.n * p.h * p.w, total = pixels * p.c;
  const float* A = reinterpret_cast<const float*>(p.a);
  T* Y = reinterpret_cast<T*>(p.y);
  const int copies = p.i1 < 1 ? 1 : p.i1;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long c = idx % p.c, px = idx / p.c;
    const T v = from_f32<T>(A[px * p.lda + c]);
    for (int j = 0; j < copies; ++j) Y[px * p.ldy + j * p.c + c] = v;
  }
}

int ew_f32_launch(const mtx_ew_args* a, void* stream, const char** err) {
  if (!a->a || !a->y) { *err = "elementwise f32: null operand"; return MTX_ERR_INVALID; }
  const long total = a->n * a->h * a->w * a->c;
  if (total < 1) return MTX_OK;
  long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
  const dim3 grid((unsigned)blocks);
  switch (a->kind) {
    case MTX_EW_ADD: case MTX_EW_MUL:
      if (!a->b) { *err = "elementwise f32: second operand missing"; return MTX_ERR_INVALID; }
      break;
    case MTX_EW_ROW_GATHER:
      if (!a->s) { *err = "elementwise f32: row index missing"; return MTX_ERR_INVALID; }
      break;
    case MTX_EW_ACT: case MTX_EW_COPY: case MTX_EW_SHUFFLE2_ADD: break;
    case MTX_EW_CVT_F32:
      if (a->i0 == MTX_F16) MTX_LAUNCH(cvt_f32_kernel<_Float16>, grid, dim3(256), 0, stream, *a);
      else if (a->i0 == MTX_BF16) MTX_LAUNCH(cvt_f32_kernel<__bf16>, grid, dim3(256), 0, stream, *a);
      else { *err = "elementwise f32: CVT source must be f16 or bf16"; return MTX_ERR_INVALID; }
      return MTX_OK;
    case MTX_EW_CVT_16:
      if (a->i1 > 4 || a->ldy < (a->i1 < 1 ? 1 : a->i1) * a->c) { *err = "elementwise f32: CVT_16 writes at most four copies and needs ldy >= copies * c"; return MTX_ERR_INVALID; }
      if (a->i0 == MTX_F16) MTX_LAUNCH(cvt_16_kernel<_Float16>, grid, dim3(256), 0, stream, *a);
      else if (a->i0 == MTX_BF16) MTX_LAUNCH(cvt_16_kernel<__bf16>, grid, dim3(256), 0, stream, *a);
      else { *err = "elementwise f32: CVT_16 destination must be f16 or bf16"; return MTX_ERR_INVALID; }
      return MTX_OK;
    default: *err = "elementwise f32: op kind is not part of the fp32 path"; return MTX_ERR_UNSUPPORTED;
  }
  MTX_LAUNCH(ew_f32_kernel, grid, dim3(256), 0, stream, *a);
  return MTX_OK;
}

}  // namespace mtx
