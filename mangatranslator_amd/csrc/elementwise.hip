// elementwise.hip — HBM-streaming NHWC kernels: 16 bytes per lane, grid-stride, fp32 math.
//
//   MTX_EW_SCALE_RES   RCAN channel-attention scale + skip (RCAB tail; spandrel RCAN, called
//                      at core/image/image_utils.py:369-374)
//   MTX_EW_UPSAMPLE2X  nn.Upsample(2, 'nearest') of the YOLO neck / SAM FPN top-down path
//   MTX_EW_MAXPOOL     SPPF 5x5 max-pool (YOLO) and Hiera 2x2 query pooling (SAM-2.1)
//   MTX_EW_GATE_RES    x + gate * y of the FLUX DiT blocks
//   channel attention MLP, image<->tensor conversions of image_to_tensor / tensor_to_image
//   (core/image/image_utils.py:351-366) and the bilinear-resize + threshold of
//   Sam2ImageProcessor.post_process_masks (core/image/detection.py:507-510).
#include "mtx_device.h"

namespace mtx {

template <typename T>
__global__ __launch_bounds__(256) void ew_kernel(mtx_ew_args p) {
  const long C8 = p.c / 8;
  const int kind = p.kind;
  long oh = p.h, ow = p.w;
  if (kind == MTX_EW_UPSAMPLE2X) { oh = 2 * p.h; ow = 2 * p.w; }
  if (kind == MTX_EW_MAXPOOL) { const int k = p.i0, s = p.i1, pd = (k & 1) ? k / 2 : 0; oh = (p.h + 2 * pd - k) / s + 1; ow = (p.w + 2 * pd - k) / s + 1; }
  if (kind == MTX_EW_AVGPOOL2) { oh = (p.h + 1) / 2; ow = (p.w + 1) / 2; }
  if (kind == MTX_EW_IM2COL) { const int k = p.i0, s = p.i1, pd = k / 2; oh = (p.h + 2 * pd - k) / s + 1; ow = (p.w + 2 * pd - k) / s + 1; }
  const T* A = reinterpret_cast<const T*>(p.a);
  if (kind == MTX_EW_IM2COL) {
    // one 16-byte chunk per (output row, tap, 8 channels)
    const int k = p.i0, st = p.i1, pd = k / 2;
    const long per_row = (long)k * k * C8;
    const long rows = p.n * oh * ow;
    const int32_t* map = reinterpret_cast<const int32_t*>(p.s);
    T* Yo = reinterpret_cast<T*>(p.y);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < rows * per_row; idx += (long)gridDim.x * 256) {
      const long r = idx / per_row, rem = idx % per_row;
      const long tap = rem / C8, c = (rem % C8) * 8;
      const long opix = map ? (long)map[r] : r;
      const long ox = opix % ow, oy = (opix / ow) % oh, n = opix / (ow * oh);
      const long iy = oy * st - pd + tap / k, ix = ox * st - pd + tap % k;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) v = *reinterpret_cast<const u32x4*>(A + ((n * p.h + iy) * p.w + ix) * p.lda + c);
      *reinterpret_cast<u32x4*>(Yo + r * p.ldy + tap * p.c + c) = v;
    }
    return;
  }
  if (kind == MTX_EW_ROW_GATHER) {
    const long rows = p.n * p.h * p.w;
    const int32_t* map = reinterpret_cast<const int32_t*>(p.s);
    T* Yo = reinterpret_cast<T*>(p.y);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < rows * C8; idx += (long)gridDim.x * 256) {
      const long r = idx / C8, c = (idx % C8) * 8;
      *reinterpret_cast<u32x4*>(Yo + r * p.ldy + c) = *reinterpret_cast<const u32x4*>(A + (long)map[r] * p.lda + c);
    }
    return;
  }
  const long total = p.n * oh * ow * C8;
  const T* Bp = reinterpret_cast<const T*>(p.b);
  const T* Sp = reinterpret_cast<const T*>(p.s);
  T* Y = reinterpret_cast<T*>(p.y);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long c = (idx % C8) * 8;
    const long pix = idx / C8;              // output pixel, n-major
    const long x = pix % ow, y = (pix / ow) % oh, n = pix / (ow * oh);
    float f[8];
    if (kind == MTX_EW_UPSAMPLE2X) {
      const long ip = (n * p.h + (y >> 1)) * p.w + (x >> 1);
      unpack8<T>(*reinterpret_cast<const u32x4*>(A + ip * p.lda + c), f);
      if (Bp) { float g[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += g[e]; }
    } else if (kind == MTX_EW_DWCONV) {
      const int k = p.i0, pd = k / 2;
      const float* bias = reinterpret_cast<const float*>(p.b);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = bias ? bias[c + e] : 0.f;
      for (int dy = 0; dy < k; ++dy) {
        const long iy = y - pd + dy;
        if (iy < 0 || iy >= p.h) continue;
        for (int dx = 0; dx < k; ++dx) {
          const long ix = x - pd + dx;
          if (ix < 0 || ix >= p.w) continue;
          float g[8], wv[8];
          unpack8<T>(*reinterpret_cast<const u32x4*>(A + ((n * p.h + iy) * p.w + ix) * p.lda + c), g);
          unpack8<T>(*reinterpret_cast<const u32x4*>(Sp + (long)(dy * k + dx) * p.c + c), wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += g[e] * wv[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = apply_act(f[e], p.act, p.act_param);
    } else if (kind == MTX_EW_AVGPOOL2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
      int taps = 0;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const long iy = y * 2 + dy, ix = x * 2 + dx;
          if (iy >= p.h || ix >= p.w) continue;
          float g[8];
          unpack8<T>(*reinterpret_cast<const u32x4*>(A + ((n * p.h + iy) * p.w + ix) * p.lda + c), g);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += g[e];
          ++taps;
        }
      const float inv = 1.0f / (float)taps;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= inv;
    } else if (kind == MTX_EW_MAXPOOL) {
      const int k = p.i0, s = p.i1, pd = (k & 1) ? k / 2 : 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = -3.0e38f;
      for (int dy = 0; dy < k; ++dy) {
        const long iy = y * s - pd + dy;
        if (iy < 0 || iy >= p.h) continue;
        for (int dx = 0; dx < k; ++dx) {
          const long ix = x * s - pd + dx;
          if (ix < 0 || ix >= p.w) continue;
          float g[8];
          unpack8<T>(*reinterpret_cast<const u32x4*>(A + ((n * p.h + iy) * p.w + ix) * p.lda + c), g);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = g[e] > f[e] ? g[e] : f[e];
        }
      }
    } else {
      unpack8<T>(*reinterpret_cast<const u32x4*>(A + pix * p.lda + c), f);
      if (kind == MTX_EW_SCALE_RES) {
        const float* S32 = reinterpret_cast<const float*>(p.s);     // fp32 [N][C] from the CA MLP
        float g[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * S32[n * p.lds + c + e] + g[e];
      } else if (kind == MTX_EW_SWIGLU) {
        float g[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = div_by_1p(f[e], 1.f + __expf(-f[e])) * g[e];
      } else if (kind == MTX_EW_ADD || kind == MTX_EW_MUL || kind == MTX_EW_SUB) {
        float g[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = kind == MTX_EW_ADD ? f[e] + g[e] : kind == MTX_EW_SUB ? f[e] - g[e] : f[e] * g[e];
      } else if (kind == MTX_EW_GATE_RES) {
        float g[8], s8[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), g);
        unpack8<T>(*reinterpret_cast<const u32x4*>(Sp + n * p.lds + c), s8);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = g[e] + f[e] * s8[e];
      }
      if (kind == MTX_EW_ACT || p.act != MTX_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = apply_act(f[e], p.act, p.act_param);
      }
    }
    *reinterpret_cast<u32x4*>(Y + pix * p.ldy + c) = pack8<T>(f);
  }
}

// ---- MTX_EW_RESIDUAL_DIST: the probe of the first-block cache (reference core/ml/model_manager.py:1159-1162, nunchaku's
// apply_cache_on_pipe: "is this step's first-block residual close to the last computed step's?") --------------------------------------
// r = a - b rounded to T (the residual as a 16-bit tensor), prev = s.  Workgroup g of MTX_RESDIST_PARTS leaves y[2 g] = sum |prev - r| and
// y[2 g + 1] = sum |prev| over its grid-strided share, accumulated in a fixed order (lane-local, wave tree, waves in order): the caller adds
// the parts in index order, so the decision does not depend on scheduling.  No atomics, no clearing pass.
template <typename T>
__global__ __launch_bounds__(256) void residual_dist_kernel(mtx_ew_args p) {
  __shared__ float red[2][4];
  const long C8 = p.c / 8, total = p.n * p.h * p.w * C8;
  const T* A = reinterpret_cast<const T*>(p.a);
  const T* Bp = reinterpret_cast<const T*>(p.b);
  const T* Sp = reinterpret_cast<const T*>(p.s);
  float d = 0.f, m = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long c = (idx % C8) * 8, pix = idx / C8;
    float fa[8], fb[8], fs[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(A + pix * p.lda + c), fa);
    unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), fb);
    unpack8<T>(*reinterpret_cast<const u32x4*>(Sp + pix * p.lds + c), fs);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float r = to_f32(from_f32<T>(fa[e] - fb[e]));
      d += fabsf(fs[e] - r);
      m += fabsf(fs[e]);
    }
  }
  d = wave_sum(d); m = wave_sum(m);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = d; red[1][wv] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* Y = reinterpret_cast<float*>(p.y);
    Y[2 * blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    Y[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}

// ---- row softmax (one workgroup per row, any length; fp32 math) ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(mtx_ew_args p) {
  __shared__ float red[8];
  const long row = blockIdx.x;
  const T* X = reinterpret_cast<const T*>(p.a) + row * p.lda;
  T* Y = reinterpret_cast<T*>(p.y) + row * p.ldy;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long nch = p.c / 8;
  const long valid = p.i0 > 0 ? p.i0 : p.c;          // columns >= valid are padding: no weight, output 0
  const float sc = p.act_param * 1.4426950408889634f;
  float m = -3.0e38f;
  for (long ch = tid; ch < nch; ch += 256) {
    float f[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(X + ch * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = (ch * 8 + e < valid && f[e] > m) ? f[e] : m;
  }
  m = wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (long ch = tid; ch < nch; ch += 256) {
    float f[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(X + ch * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += ch * 8 + e < valid ? exp2f((f[e] - m) * sc) : 0.f;
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wv] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (long ch = tid; ch < nch; ch += 256) {
    float f[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(X + ch * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = ch * 8 + e < valid ? exp2f((f[e] - m) * sc) * inv : 0.f;
    *reinterpret_cast<u32x4*>(Y + ch * 8) = pack8<T>(f);
  }
}

// ---- 2-D transpose through a padded LDS tile --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(mtx_ew_args p) {
  __shared__ T tile[64][66];
  const long rows = p.h * p.w, cols = p.c;
  const long n = blockIdx.z;
  const T* X = reinterpret_cast<const T*>(p.a) + n * rows * p.lda;
  T* Y = reinterpret_cast<T*>(p.y) + n * cols * p.ldy;
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? X[(r0 + r) * p.lda + c0 + c] : from_f32<T>(0.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) Y[(c0 + c) * p.ldy + r0 + r] = tile[r][c];
  }
}

// ---- FLUX q/k prep: per-head RMSNorm(d) * gamma, then rotary on interleaved pairs ----------------------
// Latency-bound if a thread has one 16-byte load in flight (measured 1.4 TB/s): every thread keeps QR_U rows of the
// same column chunk in flight, all loads issued before the first reduction.
constexpr int QR_U = 4;
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(mtx_ew_args p) {
  const int d = p.i0, lpr = d / 8;                                   // lanes per (token, head) row: 8 or 16
  const unsigned cpr = (unsigned)(p.c / 8);                          // 16-byte chunks per token row
  const unsigned rows = (unsigned)(p.n * p.h * p.w);
  const float* gamma0 = reinterpret_cast<const float*>(p.s);
  const int split_at = p.i1;          // > 0: heads >= split_at use the second gamma vector (fused q|k slices)
  const float* cs = reinterpret_cast<const float*>(p.b);
  const T* X = reinterpret_cast<const T*>(p.a);
  T* Y = reinterpret_cast<T*>(p.y);
  // thread -> column chunk `ch` (fixed) and QR_U consecutive token rows per step; blockDim.x covers whole heads
  const unsigned ch = blockIdx.x * 256u + threadIdx.x;               // grid.x = ceil(cpr / 256)
  const bool col_ok = ch < cpr;
  const unsigned chc = col_ok ? ch : cpr - lpr + (threadIdx.x % lpr);
  const int part = (int)(chc & (unsigned)(lpr - 1));
  const int hd = (int)(chc / (unsigned)lpr);
  const float* gamma = (gamma0 != nullptr && split_at > 0 && hd >= split_at) ? gamma0 + d : gamma0;
  if (split_at > 0 && p.ldb > 0 && hd < split_at) cs += p.ldb;       // q heads: the pre-scaled copy of the rotary table
  float gm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gm[e] = gamma ? gamma[part * 8 + e] : 1.f;
  const float y8m = (split_at > 0 && hd < split_at) ? p.y8_mul : 1.0f;
  for (unsigned r0 = blockIdx.y * QR_U; r0 < rows; r0 += gridDim.y * QR_U) {
    u32x4 raw[QR_U];
#pragma unroll
    for (int u = 0; u < QR_U; ++u) {
      const unsigned r = r0 + u < rows ? r0 + u : rows - 1;
      raw[u] = *reinterpret_cast<const u32x4*>(X + (size_t)r * p.lda + chc * 8);
    }
#pragma unroll
    for (int u = 0; u < QR_U; ++u) {
      const unsigned r = r0 + u < rows ? r0 + u : rows - 1;
      float f[8];
      unpack8<T>(raw[u], f);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
      for (int m = 1; m < lpr; m <<= 1) ss += __shfl_xor(ss, m, 64);
      const float rs = 1.0f / sqrtf(ss / (float)d + p.act_param);
      const float* c_ = cs + (size_t)r * d + part * 4;                // [rows][2][d/2]
      const float* s_ = c_ + d / 2;
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // the reference rounds the normalised value to the model dtype before the rotary product
        const float x0 = to_f32(from_f32<T>(f[2 * k] * rs * gm[2 * k]));
        const float x1 = to_f32(from_f32<T>(f[2 * k + 1] * rs * gm[2 * k + 1]));
        o[2 * k] = x0 * c_[k] - x1 * s_[k];
        o[2 * k + 1] = x1 * c_[k] + x0 * s_[k];
      }
      if (col_ok && r0 + u < rows) *reinterpret_cast<u32x4*>(Y + (size_t)r * p.ldy + chc * 8) = pack8<T>(o);
      if (p.y8 != nullptr) {                     // e4m3 twin of the values as stored (rounded to T first), q heads times y8_mul: the fp8 score operands of the attention
        float f8[8];
        unpack8<T>(pack8<T>(o), f8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = f8[e] * y8m; f8[e] = v > 448.f ? 448.f : (v < -448.f ? -448.f : v); }
        unsigned w0 = 0, w1 = 0;
        w0 = cvt_pk_fp8<false>(f8[0], f8[1], w0); w0 = cvt_pk_fp8<true>(f8[2], f8[3], w0);
        w1 = cvt_pk_fp8<false>(f8[4], f8[5], w1); w1 = cvt_pk_fp8<true>(f8[6], f8[7], w1);
        if (col_ok && r0 + u < rows) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.y8) + (size_t)r * p.ldy8 + chc * 8) = u32x2{w0, w1};
      }
    }
  }
}

// ---- MTX_EW_V_F8T: v rows -> e4m3 V^T of one head and one 64-key tile per workgroup ---------------------------------------------------
// in: 64 keys x 128 d (16-bit) of head blockIdx.y, tile blockIdx.x; out: 128 d-rows x 64 bytes, byte j of a row = key
// 32 (j >> 5) + (j & 3) + 8 ((j & 15) >> 2) + 4 ((j >> 4) & 1) — the order in which lane (query, half) of the S^T accumulators holds its 32 keys.
template <typename T>
__global__ __launch_bounds__(256) void v_f8t_kernel(mtx_ew_args p) {
  __shared__ unsigned char tile[64][128 + 16];                   // [key][d] e4m3, rows padded against bank conflicts of the column gathers
  const long rows = p.n * p.h * p.w;
  const long k0 = (long)blockIdx.x * 64, head = blockIdx.y;
  const T* X = reinterpret_cast<const T*>(p.a) + head * 128;
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {              // 16 chunks of 8 values per key row
    const int key = i >> 4, ch = i & 15;
    float f[8];
    if (k0 + key < rows) unpack8<T>(*reinterpret_cast<const u32x4*>(X + (size_t)(k0 + key) * p.lda + ch * 8), f);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = f[e] > 448.f ? 448.f : (f[e] < -448.f ? -448.f : f[e]);
    unsigned w0 = 0, w1 = 0;
    w0 = cvt_pk_fp8<false>(f[0], f[1], w0); w0 = cvt_pk_fp8<true>(f[2], f[3], w0);
    w1 = cvt_pk_fp8<false>(f[4], f[5], w1); w1 = cvt_pk_fp8<true>(f[6], f[7], w1);
    *reinterpret_cast<u32x2*>(&tile[key][ch * 8]) = u32x2{w0, w1};
  }
  __syncthreads();
  unsigned char* Y = reinterpret_cast<unsigned char*>(p.y8) + ((size_t)head * 128) * p.ldy8 + k0;
  for (int i = threadIdx.x; i < 128 * 4; i += 256) {              // 4 chunks of 16 keys per d-row
    const int d = i >> 2, c = i & 3;
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int j = c * 16 + b;
      const int key = 32 * (j >> 5) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * ((j >> 4) & 1);
      w[b >> 2] |= (unsigned)tile[key][d] << (8 * (b & 3));
    }
    *reinterpret_cast<u32x4*>(Y + (size_t)d * p.ldy8 + c * 16) = u32x4{w[0], w[1], w[2], w[3]};
  }
}

int ew_f32_launch(const mtx_ew_args* a, void* stream, const char** err);      // f32ops.hip
int ew_launch(const mtx_ew_args* a, void* stream, const char** err) {
  if (a->dtype == MTX_F32) return ew_f32_launch(a, stream, err);
  if (a->kind == MTX_EW_SOFTMAX_ROWS) {
    if (!a->a || !a->y || a->c % 8 || a->lda % 8 || a->ldy % 8) { *err = "softmax_rows: bad layout"; return MTX_ERR_INVALID; }
    const long rows = a->n * a->h * a->w;
    if (rows < 1) return MTX_OK;
    if (a->dtype == MTX_BF16) MTX_LAUNCH((softmax_rows_kernel<__bf16>), dim3((unsigned)rows), dim3(256), 0, stream, *a);
    else if (a->dtype == MTX_F16) MTX_LAUNCH((softmax_rows_kernel<_Float16>), dim3((unsigned)rows), dim3(256), 0, stream, *a);
    else { *err = "softmax_rows: dtype"; return MTX_ERR_INVALID; }
    return MTX_OK;
  }
  if (a->kind == MTX_EW_RESIDUAL_DIST) {
    if (!a->a || !a->b || !a->s || !a->y || a->c % 8 || a->lda % 8 || a->ldb % 8 || a->lds % 8) { *err = "residual_dist: needs a, b, s (16-bit, strides % 8 == 0) and y (fp32 [2 * MTX_RESDIST_PARTS])"; return MTX_ERR_INVALID; }
    if (a->dtype == MTX_BF16) MTX_LAUNCH((residual_dist_kernel<__bf16>), dim3(MTX_RESDIST_PARTS), dim3(256), 0, stream, *a);
    else if (a->dtype == MTX_F16) MTX_LAUNCH((residual_dist_kernel<_Float16>), dim3(MTX_RESDIST_PARTS), dim3(256), 0, stream, *a);
    else { *err = "residual_dist: dtype"; return MTX_ERR_INVALID; }
    return MTX_OK;
  }
  if (a->kind == MTX_EW_TRANSPOSE) {
    if (!a->a || !a->y) { *err = "transpose: null operand"; return MTX_ERR_INVALID; }
    const long rows = a->h * a->w;
    dim3 grid((unsigned)((a->c + 63) / 64), (unsigned)((rows + 63) / 64), (unsigned)a->n);
    if (a->dtype == MTX_BF16) MTX_LAUNCH((transpose_kernel<__bf16>), grid, dim3(256), 0, stream, *a);
    else if (a->dtype == MTX_F16) MTX_LAUNCH((transpose_kernel<_Float16>), grid, dim3(256), 0, stream, *a);
    else { *err = "transpose: dtype"; return MTX_ERR_INVALID; }
    return MTX_OK;
  }
  if (a->kind == MTX_EW_V_F8T) {
    const long rows = a->n * a->h * a->w;
    if (!a->a || !a->y8 || a->c % 128 || a->lda % 8 || a->ldy8 % 64 || a->ldy8 < (rows + 63) / 64 * 64 || ((size_t)a->y8 & 15)) {
      *err = "v_f8t: c must be heads * 128, lda % 8 == 0, ldy8 a multiple of 64 that covers the rows, y8 16-byte aligned"; return MTX_ERR_INVALID; }
    if (rows < 1) return MTX_OK;
    const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)(a->c / 128));
    if (a->dtype == MTX_BF16) MTX_LAUNCH((v_f8t_kernel<__bf16>), grid, dim3(256), 0, stream, *a);
    else if (a->dtype == MTX_F16) MTX_LAUNCH((v_f8t_kernel<_Float16>), grid, dim3(256), 0, stream, *a);
    else { *err = "v_f8t: dtype"; return MTX_ERR_INVALID; }
    return MTX_OK;
  }
  if (a->kind == MTX_EW_QK_NORM_ROPE) {
    const int d = a->i0;
    if (!a->a || !a->y || !a->b || (d != 64 && d != 128) || a->c % d || a->lda % 8 || a->ldy % 8) { *err = "qk_norm_rope: head dim must be 64 or 128"; return MTX_ERR_INVALID; }
    if (a->y8 != nullptr && (a->ldy8 % 16 || ((size_t)a->y8 & 15) || a->ldy8 < a->c)) { *err = "qk_norm_rope: y8 needs 16-byte aligned rows of at least c bytes"; return MTX_ERR_INVALID; }
    const long rows = a->n * a->h * a->w;
    if (rows < 1) return MTX_OK;
    if (rows >= (1L << 31) || a->c / 8 >= (1L << 24)) { *err = "qk_norm_rope: problem too large"; return MTX_ERR_INVALID; }
    const unsigned gx = (unsigned)((a->c / 8 + 255) / 256);
    long gy = (rows + QR_U - 1) / QR_U; if (gy > 4096) gy = 4096;
    const dim3 grid(gx, (unsigned)gy);
    if (a->dtype == MTX_BF16) MTX_LAUNCH((qk_norm_rope_kernel<__bf16>), grid, dim3(256), 0, stream, *a);
    else if (a->dtype == MTX_F16) MTX_LAUNCH((qk_norm_rope_kernel<_Float16>), grid, dim3(256), 0, stream, *a);
    else { *err = "qk_norm_rope: dtype"; return MTX_ERR_INVALID; }
    return MTX_OK;
  }
  if (!a->a || !a->y) { *err = "elementwise: null operand"; return MTX_ERR_INVALID; }
  if (a->c % 8 || a->lda % 8 || a->ldy % 8 || (a->b && a->ldb % 8 && a->kind != MTX_EW_DWCONV)) { *err = "elementwise: C and pixel strides must be multiples of 8"; return MTX_ERR_INVALID; }
  if ((a->kind == MTX_EW_SCALE_RES || a->kind == MTX_EW_ADD || a->kind == MTX_EW_MUL || a->kind == MTX_EW_SUB || a->kind == MTX_EW_GATE_RES || a->kind == MTX_EW_SWIGLU) && !a->b) { *err = "elementwise: missing operand b"; return MTX_ERR_INVALID; }
  if ((a->kind == MTX_EW_SCALE_RES || a->kind == MTX_EW_GATE_RES) && !a->s) { *err = "elementwise: missing operand s"; return MTX_ERR_INVALID; }
  if (a->kind == MTX_EW_MAXPOOL && (a->i0 < 1 || a->i1 < 1)) { *err = "elementwise: maxpool needs kernel/stride"; return MTX_ERR_INVALID; }
  if (a->kind == MTX_EW_DWCONV && (!a->s || (a->i0 & 1) == 0 || a->i0 < 1 || a->i0 > 7)) { *err = "elementwise: dwconv needs weights in s and an odd kernel <= 7"; return MTX_ERR_INVALID; }
  long oh = a->h, ow = a->w;
  if (a->kind == MTX_EW_UPSAMPLE2X) { oh *= 2; ow *= 2; }
  if (a->kind == MTX_EW_MAXPOOL) { const int pd = (a->i0 & 1) ? a->i0 / 2 : 0; oh = (a->h + 2 * pd - a->i0) / a->i1 + 1; ow = (a->w + 2 * pd - a->i0) / a->i1 + 1; }
  if (a->kind == MTX_EW_AVGPOOL2) { oh = (a->h + 1) / 2; ow = (a->w + 1) / 2; }
  long total = a->n * oh * ow * (a->c / 8);
  if (a->kind == MTX_EW_IM2COL) {
    if (a->i0 < 1 || a->i1 < 1 || a->ldy < (long)a->i0 * a->i0 * a->c) { *err = "elementwise: im2col needs k, stride and ldy >= k*k*C"; return MTX_ERR_INVALID; }
    const int pd = a->i0 / 2;
    total = a->n * ((a->h + 2 * pd - a->i0) / a->i1 + 1) * ((a->w + 2 * pd - a->i0) / a->i1 + 1) * (long)a->i0 * a->i0 * (a->c / 8);
  }
  if (a->kind == MTX_EW_ROW_GATHER && !a->s) { *err = "elementwise: row_gather needs the index map in s"; return MTX_ERR_INVALID; }
  if (total <= 0) return MTX_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 2048 * 4) blocks = 2048 * 4;
  if (a->dtype == MTX_BF16) MTX_LAUNCH((ew_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((ew_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else { *err = "elementwise: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

// ---- RCAN channel attention squeeze/excite: one workgroup per image ---------------------------
// Two forms.  Classic: chan_sum holds the channel sums of the tensor to pool.  "Pool before the conv" (p.t != null): chan_sum holds the
// sums of t, the INPUT of a 3x3 conv, and the pooled vector is mean(conv(t) + b) by linearity —
//   sum_p conv(t)[p][co] = sum_tap sum_ci W[co][tap][ci] * S_tap[ci],   S_tap = total - (border row the tap never reaches) - (border column) + (corner)
// — so the attention factors exist before the conv runs and the conv can write x + s * conv(t) itself (mtx_conv2d_args.out_scale).
// pool-before-conv, second half: from the channel totals of t (mean[]) and its four border-strip sums to mean(conv3x3(t) + b), left in
// mean[].  Run by all 1024 threads of a workgroup; `red` is LDS scratch of at least 9 * 8 * 64 floats.
template <typename T>
__device__ __forceinline__ void ca_conv_mean(const mtx_ca_args& p, int n, float* mean, float (*strip)[64], float* red, float inv_hw, int H, int W) {
  __shared__ float corner[4][64];                       // (0,0) (0,W-1) (H-1,0) (H-1,W-1)
  __shared__ float stap[9][64];
  const int tid = threadIdx.x;
  const T* tb = reinterpret_cast<const T*>(p.t) + (size_t)n * p.h * p.w * p.ldt;
  if (tid < 4 * 64) {
    const int k = tid >> 6, c = tid & 63;
    const int y = (k & 2) ? H - 1 : 0, x = (k & 1) ? W - 1 : 0;
    corner[k][c] = c < p.c ? to_f32(tb[((size_t)y * p.w + x) * p.ldt + c]) : 0.f;
  }
  __syncthreads();
  if (tid < 9 * 64) {
    const int tap = tid >> 6, c = tid & 63, dy = tap / 3 - 1, dx = tap % 3 - 1;
    float s = mean[c];
    if (dy == 1) s -= strip[0][c]; else if (dy == -1) s -= strip[1][c];      // a tap one row DOWN never reads row 0 for an in-image output, ...
    if (dx == 1) s -= strip[2][c]; else if (dx == -1) s -= strip[3][c];
    if (dy != 0 && dx != 0) s += corner[(dy == -1 ? 2 : 0) + (dx == -1 ? 1 : 0)][c];
    stap[tap][c] = s;
  }
  __syncthreads();
  // mean of conv(t) + b: one 16-byte filter chunk (8 input channels of one tap of one output channel) per item, 4.5 items per thread
  const int chunks = p.c / 8, per_co = 9 * chunks, items = p.c * per_co;
  for (int it = tid; it < items; it += 1024) {
    const int co = it / per_co, rem = it - co * per_co, tap = rem / chunks, c8 = rem - tap * chunks;
    const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.conv_w) + ((size_t)co * 9 + tap) * p.c + c8 * 8);
    float f[8];
    unpack8<T>(raw, f);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += f[e] * stap[tap][c8 * 8 + e];
    red[it] = s;
  }
  __syncthreads();
  if (tid < p.c) {
    float s = 0.f;
    for (int k = 0; k < per_co; ++k) s += red[tid * per_co + k];
    mean[tid] = (p.conv_b ? p.conv_b[tid] : 0.f) + s * inv_hw;
  }
  __syncthreads();
}

// the squeeze / excite MLP on the pooled vector in mean[]: s = sigmoid(W2 relu(W1 mean + b1) + b2)
__device__ __forceinline__ void ca_mlp(const mtx_ca_args& p, int n, const float* mean, float* hid) {
  const int tid = threadIdx.x;
  for (int r = tid; r < p.cr; r += 1024) {
    float s = p.b1 ? p.b1[r] : 0.f;
    for (int c = 0; c < p.c; ++c) s += p.w1[r * p.c + c] * mean[c];
    hid[r] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < p.c; c += 1024) {
    float s = p.b2 ? p.b2[c] : 0.f;
    for (int r = 0; r < p.cr; ++r) s += p.w2[c * p.cr + r] * hid[r];
    p.s[(size_t)n * p.c + c] = 1.f / (1.f + __expf(-s));
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void ca_kernel(mtx_ca_args p) {
  __shared__ float mean[512];
  __shared__ float hid[128];
  __shared__ float part[1024];
  const int n = blockIdx.x, tid = threadIdx.x;
  // partial-sum rows are reduced by all 1024 threads
  int per;                                                // row subsets
  if (p.c % 4 == 0 && p.c <= 256) {
    // thread -> (row subset, 4-channel group): 16-byte loads, eight in flight (the rows come from L2 one round trip apiece; a chain of
    // dependent adds around single 4-byte loads made this kernel 33 us for 2 048 rows — RCAN runs it 200 times per page)
    const int c4n = p.c / 4;
    per = 1024 / c4n;
    const int c4 = tid % c4n, sub = tid / c4n;
    f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (sub < per) {
      const float* src = p.chan_sum + (size_t)n * p.tiles * p.c + c4 * 4;
      int t = sub;
      for (; t + 15 * per < p.tiles; t += 16 * per) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)(t + k * per) * p.c);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc4 += v[k];
      }
      for (; t < p.tiles; t += per) acc4 += *reinterpret_cast<const f32x4*>(src + (size_t)t * p.c);
    }
    __shared__ float part4[1024 * 4];
#pragma unroll
    for (int e = 0; e < 4; ++e) part4[(sub * c4n + c4) * 4 + e] = acc4[e];
    __syncthreads();
    for (int c = tid; c < p.c; c += 1024) {
      float s = 0.f;
      for (int k = 0; k < per; ++k) s += part4[k * p.c + c];
      part[c] = s;                                        // the channel totals
    }
    __syncthreads();
    per = 1;
  } else {
    per = 1024 / p.c > 0 ? 1024 / p.c : 1;
    const int c = tid % p.c, sub = tid / p.c;
    float s = 0.f;
    if (sub < per) {
      const float* src = p.chan_sum + (size_t)n * p.tiles * p.c + c;
      for (int t = sub; t < p.tiles; t += per) s += src[(size_t)t * p.c];
    }
    part[tid] = s;
    __syncthreads();
  }
  const float inv_hw = p.inv_hw_dev ? *p.inv_hw_dev : p.inv_hw;
  for (int c = tid; c < p.c; c += 1024) {
    float s = 0.f;
    for (int k = 0; k < per; ++k) s += part[k * p.c + c];
    mean[c] = p.t != nullptr ? s : s * inv_hw;          // pool-before-conv: the TOTAL of t, turned into the conv's mean below
  }
  __syncthreads();
  if (p.t != nullptr) {                                 // C <= 64 (checked by the launcher)
    __shared__ float red[4 * 32 * 64];                  // strip partials [4 strips][32 pixel subsets][64 ch]; then the 9 * 8 * C dot partials
    __shared__ float strip[4][64];                      // top row, bottom row, left column, right column
    float* redall = red;
    const int H = p.valid_hw ? p.valid_hw[0] : p.h, W = p.valid_hw ? p.valid_hw[1] : p.w;
    const T* tb = reinterpret_cast<const T*>(p.t) + (size_t)n * p.h * p.w * p.ldt;
    {   // the four border strips at once: 256 threads per strip = 8 channel chunks x 32 pixel subsets, four loads in flight per thread
      const int k = tid >> 8, c8 = tid & 7, sub = (tid >> 3) & 31;
      const int npx = k < 2 ? W : H;
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (c8 * 8 < p.c) {
        auto at = [&](int px) -> u32x4 {
          const int y = k == 0 ? 0 : (k == 1 ? H - 1 : px), x = k == 2 ? 0 : (k == 3 ? W - 1 : px);
          return *reinterpret_cast<const u32x4*>(tb + ((size_t)y * p.w + x) * p.ldt + c8 * 8);
        };
        int px = sub;
        for (; px + 480 < npx; px += 512) {             // sixteen pixels in flight per thread
          u32x4 r4[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) r4[j] = at(px + 32 * j);
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            float f[8];
            unpack8<T>(r4[j], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += f[e];
          }
        }
        for (; px + 224 < npx; px += 256) {             // eight
          u32x4 r4[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) r4[j] = at(px + 32 * j);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float f[8];
            unpack8<T>(r4[j], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += f[e];
          }
        }
        for (; px < npx; px += 32) {
          float f[8];
          unpack8<T>(at(px), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += f[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) redall[(k * 32 + sub) * 64 + c8 * 8 + e] = a[e];
    }
    __syncthreads();
    if (tid < 4 * 64) {
      const int k = tid >> 6, c = tid & 63;
      float s = 0.f;
      for (int j = 0; j < 32; ++j) s += redall[(k * 32 + j) * 64 + c];
      strip[k][c] = s;
    }
    __syncthreads();
    ca_conv_mean<T>(p, n, mean, strip, redall, inv_hw, H, W);
  }
  ca_mlp(p, n, mean, hid);
}

// Pool-before-conv with MTX_CA_SPLIT workgroups per image (mtx_ca_args.scratch).  Workgroup g of image n reduces its share of the
// sum rows (row t belongs to g when (t / 64) % SPLIT == g) and of the four border strips (pixel px when (px / 32) % SPLIT == g) — at
// 1024 x 1536 that is one 16-byte row chunk and one or two 16-byte pixel chunks per thread — and leaves a record
// {totals[64], strips[4][64]} in the scratch with write-through stores; then every wave drains, the workgroup meets and one lane draws
// a ticket.  Whoever draws the last ticket acquires, adds the SPLIT records up in the order g = 0, 1, ... (the result does not depend
// on who came last), runs the conv-mean and the MLP, and resets the counter.
// The launch sits between an RCAB's two convs and is pure latency, so EVERY workgroup issues all the loads the last arriver will
// need (its filter chunks, the MLP weights, the corner pixels) at the very start, beside its row and strip loads: the finishing
// workgroup then runs from registers instead of paying a memory round trip per stage (29 us single workgroup -> 15 us split -> this).
template <typename T>
__global__ __launch_bounds__(1024) void ca_split_kernel(mtx_ca_args p) {
  __shared__ float mean[64];
  __shared__ float hid[128];
  __shared__ float red[4 * 32 * 64];                    // row partials [64 subsets][64 ch] / strip partials [4][32][64]; later the dot partials
  __shared__ float strip[4][64];
  __shared__ float corner[4][64];
  __shared__ float stap[9][64];
  __shared__ unsigned last_flag;
  const int n = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int G = MTX_CA_SPLIT;
  float* rec = p.scratch + ((size_t)n * G + g) * MTX_CA_RECORD;
  unsigned* counter = reinterpret_cast<unsigned*>(p.scratch + (size_t)p.n * G * MTX_CA_RECORD) + n;
  const int H = p.valid_hw ? p.valid_hw[0] : p.h, W = p.valid_hw ? p.valid_hw[1] : p.w;
  const T* tb = reinterpret_cast<const T*>(p.t) + (size_t)n * p.h * p.w * p.ldt;
  // ---- everything anybody will need, issued before the first wait
  const int chunks = p.c / 8, per_co = 9 * chunks, items = p.c * per_co;           // filter chunks: <= 4608, at most 5 per thread
  u32x4 wq[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int it = tid + 1024 * j;
    wq[j] = u32x4{0u, 0u, 0u, 0u};
    if (it < items) {
      const int co = it / per_co, rem = it - co * per_co, tap = rem / chunks, c8 = rem - tap * chunks;
      wq[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.conv_w) + ((size_t)co * 9 + tap) * p.c + c8 * 8);
    }
  }
  float w1v = 0.f, w2v = 0.f, b1v = 0.f, b2v = 0.f, cbv = 0.f, cornerv = 0.f;
  if (tid < p.cr * p.c && tid < 1024) { w1v = p.w1[tid]; w2v = p.w2[tid]; }        // C * Cr <= 1024 (checked by the launcher)
  if (tid < p.cr && p.b1) b1v = p.b1[tid];
  if (tid < p.c) { if (p.b2) b2v = p.b2[tid]; if (p.conv_b) cbv = p.conv_b[tid]; }
  if (tid < 4 * 64 && (tid & 63) < p.c) {
    const int k = tid >> 6, y = (k & 2) ? H - 1 : 0, x = (k & 1) ? W - 1 : 0;
    cornerv = to_f32(tb[((size_t)y * p.w + x) * p.ldt + (tid & 63)]);
  }
  const float inv_hw = p.inv_hw_dev ? *p.inv_hw_dev : p.inv_hw;
  f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
  {   // sum rows: thread -> (row subset of 64, 4-channel group of 16)
    const int c4 = tid & 15, sub = tid >> 4;
    if (c4 * 4 < p.c) {
      const float* src = p.chan_sum + (size_t)n * p.tiles * p.c + c4 * 4;
      for (int t = sub + 64 * g; t < p.tiles; t += 64 * G) acc4 += *reinterpret_cast<const f32x4*>(src + (size_t)t * p.c);
    }
  }
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {   // border strips: 256 threads per strip = 8 channel chunks x 32 pixel subsets
    const int k = tid >> 8, c8 = tid & 7, sub = (tid >> 3) & 31;
    const int npx = k < 2 ? W : H;
    if (c8 * 8 < p.c) {
      for (int px = sub + 32 * g; px < npx; px += 32 * G) {
        const int y = k == 0 ? 0 : (k == 1 ? H - 1 : px), x = k == 2 ? 0 : (k == 3 ? W - 1 : px);
        float f[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(tb + ((size_t)y * p.w + x) * p.ldt + c8 * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += f[e];
      }
    }
  }
  // ---- this workgroup's record
  {
    const int c4 = tid & 15, sub = tid >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[sub * 64 + c4 * 4 + e] = acc4[e];
  }
  __syncthreads();
  if (tid < 64) {
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += red[k * 64 + tid];
    agent_store(rec + tid, s);
  }
  __syncthreads();
  {
    const int k = tid >> 8, c8 = tid & 7, sub = (tid >> 3) & 31;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(k * 32 + sub) * 64 + c8 * 8 + e] = a[e];
  }
  __syncthreads();
  if (tid < 4 * 64) {
    const int k = tid >> 6, c = tid & 63;
    float s = 0.f;
    for (int j = 0; j < 32; ++j) s += red[(k * 32 + j) * 64 + c];
    agent_store(rec + 64 + tid, s);
  }
  MTX_WAIT_VMEM();                                       // every storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) last_flag = agent_ticket(counter) + 1 == (unsigned)G;
  __syncthreads();
  if (!last_flag) return;                                // (workgroup-uniform)
  // ---- the last arriver: records -> totals and strips -> per-tap sums -> mean(conv(t) + b) -> MLP
  if (tid == 0) { agent_acquire(); agent_store(counter, 0u); }      // zero again for the next launch (a launch boundary away)
  __syncthreads();
  if (tid < MTX_CA_RECORD) {
    const float* r0 = p.scratch + (size_t)n * G * MTX_CA_RECORD + tid;
    float v[MTX_CA_SPLIT];
#pragma unroll
    for (int j = 0; j < G; ++j) v[j] = r0[j * MTX_CA_RECORD];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < G; ++j) s += v[j];
    if (tid < 64) mean[tid] = s; else strip[(tid - 64) >> 6][(tid - 64) & 63] = s;
  }
  if (tid < 4 * 64) corner[tid >> 6][tid & 63] = cornerv;
  __syncthreads();
  if (tid < 9 * 64) {
    const int tap = tid >> 6, c = tid & 63, dy = tap / 3 - 1, dx = tap % 3 - 1;
    float s = mean[c];
    if (dy == 1) s -= strip[0][c]; else if (dy == -1) s -= strip[1][c];      // a tap one row DOWN never reads row 0 for an in-image output, ...
    if (dx == 1) s -= strip[2][c]; else if (dx == -1) s -= strip[3][c];
    if (dy != 0 && dx != 0) s += corner[(dy == -1 ? 2 : 0) + (dx == -1 ? 1 : 0)][c];
    stap[tap][c] = s;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int it = tid + 1024 * j;
    if (it < items) {
      const int rem = it % per_co, tap = rem / chunks, c8 = rem - tap * chunks;
      float f[8];
      unpack8<T>(wq[j], f);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[e] * stap[tap][c8 * 8 + e];
      red[it] = s;
    }
  }
  __syncthreads();
  if (tid < p.c) {
    float s = 0.f;
    for (int k = 0; k < per_co; ++k) s += red[tid * per_co + k];
    mean[tid] = cbv + s * inv_hw;
  }
  __syncthreads();
  // MLP from the prefetched weights: thread r * C + c holds w1[r][c]; thread c * Cr + r holds w2[c][r]
  if (tid < p.cr * p.c) red[tid] = w1v * mean[tid % p.c];
  __syncthreads();
  if (tid < p.cr) {
    float s = b1v;
    for (int c = 0; c < p.c; ++c) s += red[tid * p.c + c];
    hid[tid] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  if (tid < p.cr * p.c) red[tid] = w2v * hid[tid % p.cr];
  __syncthreads();
  if (tid < p.c) {
    float s = b2v;
    for (int r = 0; r < p.cr; ++r) s += red[tid * p.cr + r];
    p.s[(size_t)n * p.c + tid] = 1.f / (1.f + __expf(-s));
  }
}

int ca_launch(const mtx_ca_args* a, void* stream, const char** err) {
  if (!a->chan_sum || !a->w1 || !a->w2 || !a->s) { *err = "channel_attention: null operand"; return MTX_ERR_INVALID; }
  if (a->c > 512 || a->cr > 128 || a->c < 1 || a->cr < 1) { *err = "channel_attention: C <= 512 and C/r <= 128"; return MTX_ERR_INVALID; }
  if (a->t != nullptr) {
    if (!a->conv_w || a->c > 64 || a->c % 8 || a->ldt % 8 || a->h < 1 || a->w < 1) { *err = "channel_attention (pool before the conv): needs conv_w, C <= 64 in multiples of 8, ldt % 8 == 0"; return MTX_ERR_INVALID; }
    if (a->scratch != nullptr && a->c * a->cr <= 1024) {
      const dim3 grid((unsigned)a->n, MTX_CA_SPLIT);
      if (a->dtype == MTX_BF16) MTX_LAUNCH(ca_split_kernel<__bf16>, grid, dim3(1024), 0, stream, *a);
      else if (a->dtype == MTX_F16) MTX_LAUNCH(ca_split_kernel<_Float16>, grid, dim3(1024), 0, stream, *a);
      else { *err = "channel_attention: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
      return MTX_OK;
    }
    if (a->dtype == MTX_BF16) MTX_LAUNCH(ca_kernel<__bf16>, dim3((unsigned)a->n), dim3(1024), 0, stream, *a);
    else if (a->dtype == MTX_F16) MTX_LAUNCH(ca_kernel<_Float16>, dim3((unsigned)a->n), dim3(1024), 0, stream, *a);
    else { *err = "channel_attention: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
    return MTX_OK;
  }
  MTX_LAUNCH(ca_kernel<_Float16>, dim3((unsigned)a->n), dim3(1024), 0, stream, *a);
  return MTX_OK;
}

// ---- page boundary conversions ----------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void img_kernel(mtx_img_args p) {
  const long tid0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
  if (p.kind == MTX_IMG_NCHW_F32_TO_NHWC || p.kind == MTX_IMG_HWC_U8_TO_NHWC) {
    // one thread per OUTPUT pixel: gathers u*u*3 source samples, writes c_pad channels
    const int u = p.unshuffle;
    const long oh = p.h / u, ow = p.w / u;
    const long total = p.n * oh * ow;
    T* D = reinterpret_cast<T*>(p.dst);
    const long vh = p.valid_hw ? p.valid_hw[0] : p.h, vw = p.valid_hw ? p.valid_hw[1] : p.w;
    for (long idx = tid0; idx < total; idx += step) {
      const long x = idx % ow, y = (idx / ow) % oh, n = idx / (ow * oh);
      T* o = D + idx * p.c_pad;
      int oc = 0;
      if (y * u >= vh || x * u >= vw) {            // beyond the image inside a bucket canvas
        for (; oc < p.c_pad; ++oc) o[oc] = from_f32<T>(0.f);
        continue;
      }
      // torch.pixel_unshuffle channel order: c*u*u + dy*u + dx
      for (int c = 0; c < 3; ++c)
        for (int dy = 0; dy < u; ++dy)
          for (int dx = 0; dx < u; ++dx) {
            const long sy = y * u + dy, sx = x * u + dx;
            float v;
            if (p.kind == MTX_IMG_NCHW_F32_TO_NHWC) v = reinterpret_cast<const float*>(p.src)[((n * 3 + c) * p.h + sy) * p.w + sx];
            else v = (float)reinterpret_cast<const uint8_t*>(p.src)[((n * p.h + sy) * p.w + sx) * 3 + c] / 255.0f;
            o[oc++] = from_f32<T>(v * p.mul + p.add[c]);
          }
      for (; oc < p.c_pad; ++oc) o[oc] = from_f32<T>(0.f);
    }
  } else {
    const long total = p.n * p.h * p.w;
    const T* S = reinterpret_cast<const T*>(p.src);
    for (long idx = tid0; idx < total; idx += step) {
      const long x = idx % p.w, y = (idx / p.w) % p.h, n = idx / (p.w * p.h);
      for (int c = 0; c < 3; ++c) {
        float v = to_f32(S[idx * p.c_pad + c]) * p.mul + p.add[c];
        if (p.kind == MTX_IMG_NHWC_TO_NCHW_F32) {
          reinterpret_cast<float*>(p.dst)[((n * 3 + c) * p.h + y) * p.w + x] = v;
        } else {
          v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
          reinterpret_cast<uint8_t*>(p.dst)[idx * 3 + c] = (uint8_t)(v * 255.0f);   // truncation, as .astype(np.uint8)
        }
      }
    }
  }
}

int img_launch(const mtx_img_args* a, void* stream, const char** err) {
  if (!a->src || !a->dst) { *err = "image_convert: null operand"; return MTX_ERR_INVALID; }
  if (a->unshuffle != 1 && a->unshuffle != 2) { *err = "image_convert: unshuffle must be 1 or 2"; return MTX_ERR_INVALID; }
  if (a->c_pad < 3 * a->unshuffle * a->unshuffle && a->kind != MTX_IMG_NHWC_TO_NCHW_F32 && a->kind != MTX_IMG_NHWC_TO_HWC_U8) { *err = "image_convert: c_pad too small"; return MTX_ERR_INVALID; }
  if ((a->h % a->unshuffle) || (a->w % a->unshuffle)) { *err = "image_convert: H, W must divide by the unshuffle factor"; return MTX_ERR_INVALID; }
  long total = a->n * a->h * a->w / (a->unshuffle * a->unshuffle);
  long blocks = (total + 255) / 256;
  if (blocks < 1) return MTX_OK;
  if (blocks > 8192) blocks = 8192;
  if (a->dtype == MTX_BF16) MTX_LAUNCH((img_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((img_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else { *err = "image_convert: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

// ---- bilinear (align_corners=False) resize + threshold -> 0/1 bytes -----------------------------
template <typename T>
__global__ __launch_bounds__(256) void resize_thresh_kernel(mtx_resize_thresh_args p) {
  const long total = p.n * p.hd * p.wd;
  const long rh = p.roi_h > 0 ? p.roi_h : p.hs, rw = p.roi_h > 0 ? p.roi_w : p.ws;
  const long ry = p.roi_h > 0 ? p.roi_y : 0, rx = p.roi_h > 0 ? p.roi_x : 0;
  const float sy = (float)rh / (float)p.hd, sx = (float)rw / (float)p.wd;
  const T* S = reinterpret_cast<const T*>(p.src);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long x = idx % p.wd, y = (idx / p.wd) % p.hd, n = idx / (p.wd * p.hd);
    // torch area_pixel_compute_source_index: max(0, (dst + 0.5) * scale - 0.5)
    float fy = ((float)y + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = ((float)x + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
    long y0 = (long)fy, x0 = (long)fx;
    if (y0 > rh - 1) y0 = rh - 1;
    if (x0 > rw - 1) x0 = rw - 1;
    long y1 = y0 + (y0 < rh - 1 ? 1 : 0), x1 = x0 + (x0 < rw - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    y0 += ry; y1 += ry; x0 += rx; x1 += rx;
    const long ps = p.pix_stride > 0 ? p.pix_stride : 1;
    const long bs = p.batch_stride >= 0 ? p.batch_stride : p.hs * p.ws * ps;
    if (p.crop_xyxy) {
      const float* b = p.crop_xyxy + n * 4;
      if (!((float)x >= b[0] && (float)x < b[2] && (float)y >= b[1] && (float)y < b[3])) { p.dst[idx] = 0; continue; }
    }
    const T* base = S + n * bs + (p.sel ? p.sel[n] : 0);
    const float v00 = to_f32(base[(y0 * p.ws + x0) * ps]), v01 = to_f32(base[(y0 * p.ws + x1) * ps]);
    const float v10 = to_f32(base[(y1 * p.ws + x0) * ps]), v11 = to_f32(base[(y1 * p.ws + x1) * ps]);
    const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    p.dst[idx] = v > p.thresh ? 1 : 0;
  }
}
int resize_thresh_launch(const mtx_resize_thresh_args* a, void* stream, const char** err) {
  if (!a->src || !a->dst) { *err = "resize_threshold: null operand"; return MTX_ERR_INVALID; }
  const long total = a->n * a->hd * a->wd;
  long blocks = (total + 255) / 256;
  if (blocks < 1) return MTX_OK;
  if (blocks > 8192) blocks = 8192;
  if (a->dtype == MTX_F32) MTX_LAUNCH((resize_thresh_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_BF16) MTX_LAUNCH((resize_thresh_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((resize_thresh_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else { *err = "resize_threshold: bad dtype"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

// ---- SAM-2.1 single-mask selection by stability ----------------------------------------------------
__global__ __launch_bounds__(256) void mask_count_kernel(mtx_mask_select_args p) {
  const long n = blockIdx.y;
  const float* L = p.logits + n * p.pix * 4;
  int hi = 0, lo = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.pix; i += (long)gridDim.x * 256) {
    const float v = L[i * 4];
    hi += v > p.delta ? 1 : 0;
    lo += v > -p.delta ? 1 : 0;
  }
  float fh = wave_sum((float)hi), fl = wave_sum((float)lo);   // <= 2^24 per wave: exact in fp32
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(p.counts + n * 2, (int)fh);
    atomicAdd(p.counts + n * 2 + 1, (int)fl);
  }
}
__global__ __launch_bounds__(64) void mask_pick_kernel(mtx_mask_select_args p) {
  const long n = (long)blockIdx.x * 64 + threadIdx.x;
  if (n >= p.n) return;
  const float ai = (float)p.counts[n * 2], au = (float)p.counts[n * 2 + 1];
  const float stab = au > 0.f ? ai / au : 1.0f;
  int sel = 0;
  if (!(stab >= p.thresh)) {
    const float* io = p.iou + n * 4;
    sel = 1;
    float best = io[1];
    if (io[2] > best) { best = io[2]; sel = 2; }
    if (io[3] > best) { best = io[3]; sel = 3; }
  }
  p.sel[n] = sel;
}

int mask_select_launch(const mtx_mask_select_args* a, void* stream, const char** err) {
  if (!a->logits || !a->iou || !a->counts || !a->sel) { *err = "mask_select: null operand"; return MTX_ERR_INVALID; }
  if (a->n < 1 || a->pix < 1) return MTX_OK;
  zero_words_async(a->counts, (size_t)a->n * 2 * sizeof(int), stream);        // a kernel, not a memset node (mtx_device.h)
  long bx = (a->pix + 255) / 256; if (bx > 64) bx = 64;
  MTX_LAUNCH(mask_count_kernel, dim3((unsigned)bx, (unsigned)a->n), dim3(256), 0, stream, *a);
  MTX_LAUNCH(mask_pick_kernel, dim3((unsigned)((a->n + 63) / 64)), dim3(64), 0, stream, *a);
  return MTX_OK;
}

// ---- SAM pre-processing: antialiased bilinear resize (uint8 levels) + normalise -> NHWC T ---------
// separable triangle filter, torch upsample_bilinear2d_aa indexing (support = max(scale, 1))
template <typename T>
__global__ __launch_bounds__(256) void preproc_kernel(mtx_preproc_args p) {
  const long total = p.oh * p.ow;
  const float sy = (float)p.h / (float)p.oh, sx = (float)p.w / (float)p.ow;
  const float supy = sy >= 1.f ? sy : 1.f, supx = sx >= 1.f ? sx : 1.f;
  const float invy = sy >= 1.f ? 1.f / sy : 1.f, invx = sx >= 1.f ? 1.f / sx : 1.f;
  T* D = reinterpret_cast<T*>(p.dst);
  if (p.mode == 1) {
    const float ly = (float)p.h / (float)p.new_h, lx = (float)p.w / (float)p.new_w;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
      const long ox = idx % p.ow, oy = idx / p.ow;
      const long ry = oy - p.pad_top, rx = ox - p.pad_left;
      float v[3] = {p.pad_value, p.pad_value, p.pad_value};
      if (ry >= 0 && ry < p.new_h && rx >= 0 && rx < p.new_w) {
        float fy = ((float)ry + 0.5f) * ly - 0.5f, fx = ((float)rx + 0.5f) * lx - 0.5f;
        if (fy < 0.f) fy = 0.f;
        if (fx < 0.f) fx = 0.f;
        long y0 = (long)fy, x0 = (long)fx;
        if (y0 > p.h - 1) y0 = p.h - 1;
        if (x0 > p.w - 1) x0 = p.w - 1;
        const long y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        for (int c = 0; c < 3; ++c) {
          const float a = (float)p.src[(y0 * p.w + x0) * 3 + c], b = (float)p.src[(y0 * p.w + x1) * 3 + c];
          const float d = (float)p.src[(y1 * p.w + x0) * 3 + c], e = (float)p.src[(y1 * p.w + x1) * 3 + c];
          v[2 - c] = rintf((1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * d + wx * e));   // BGR -> RGB
        }
      }
      T* o = D + idx * p.c_pad;
      for (int c = 0; c < 3; ++c) o[c] = from_f32<T>(v[c] * (1.0f / 255.0f));
      for (int c = 3; c < p.c_pad; ++c) o[c] = from_f32<T>(0.f);
    }
    return;
  }
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long ox = idx % p.ow, oy = idx / p.ow;
    const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
    long y0 = (long)(cy - supy + 0.5f); if (y0 < 0) y0 = 0;
    long y1 = (long)(cy + supy + 0.5f); if (y1 > p.h) y1 = p.h;
    long x0 = (long)(cx - supx + 0.5f); if (x0 < 0) x0 = 0;
    long x1 = (long)(cx + supx + 0.5f); if (x1 > p.w) x1 = p.w;
    float wys = 0.f, wxs = 0.f;
    for (long y = y0; y < y1; ++y) { float t = fabsf(((float)y - cy + 0.5f) * invy); wys += t < 1.f ? 1.f - t : 0.f; }
    for (long x = x0; x < x1; ++x) { float t = fabsf(((float)x - cx + 0.5f) * invx); wxs += t < 1.f ? 1.f - t : 0.f; }
    float acc[3] = {0.f, 0.f, 0.f};
    for (long y = y0; y < y1; ++y) {
      float ty = fabsf(((float)y - cy + 0.5f) * invy);
      const float wy = (ty < 1.f ? 1.f - ty : 0.f) / wys;
      float row[3] = {0.f, 0.f, 0.f};
      for (long x = x0; x < x1; ++x) {
        float tx = fabsf(((float)x - cx + 0.5f) * invx);
        const float wx = (tx < 1.f ? 1.f - tx : 0.f) / wxs;
        const uint8_t* px = p.src + (y * p.w + x) * 3;
        row[0] += wx * (float)px[0]; row[1] += wx * (float)px[1]; row[2] += wx * (float)px[2];
      }
      acc[0] += wy * row[0]; acc[1] += wy * row[1]; acc[2] += wy * row[2];
    }
    T* o = D + idx * p.c_pad;
    for (int c = 0; c < 3; ++c) {
      float v = rintf(acc[c]);
      v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
      o[c] = from_f32<T>((v * (1.0f / 255.0f) - p.mean[c]) / p.std[c]);
    }
    for (int c = 3; c < p.c_pad; ++c) o[c] = from_f32<T>(0.f);
  }
}

int preproc_launch(const mtx_preproc_args* a, void* stream, const char** err) {
  if (!a->src || !a->dst) { *err = "preprocess: null operand"; return MTX_ERR_INVALID; }
  if (a->h < 1 || a->w < 1 || a->oh < 1 || a->ow < 1 || a->c_pad < 3) { *err = "preprocess: bad geometry"; return MTX_ERR_INVALID; }
  long blocks = (a->oh * a->ow + 255) / 256; if (blocks > 8192) blocks = 8192;
  if (a->dtype == MTX_BF16) MTX_LAUNCH((preproc_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((preproc_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else { *err = "preprocess: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

// ---- YOLOv8 head decode --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void yolo_decode_kernel(mtx_yolo_decode_args p, int total) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= total) return;
  int l = 0, off = a;
  while (l < p.n_levels - 1 && off >= p.lh[l] * p.lw[l]) { off -= p.lh[l] * p.lw[l]; ++l; }
  const int y = off / p.lw[l], x = off % p.lw[l];
  const T* src = reinterpret_cast<const T*>(p.level[l]) + (size_t)off * p.lld[l];
  float d4[4];
  const float* b32 = p.box_f32[l] != nullptr ? p.box_f32[l] + (size_t)off * 4 * p.reg_max : nullptr;      // fp32 DFL logits (round 6): see mtx_yolo_decode_args.box_f32
  for (int s = 0; s < 4; ++s) {          // DFL: expectation of softmax over reg_max bins
    float m = -1e30f;
    for (int k = 0; k < p.reg_max; ++k) { const float v = b32 ? b32[s * p.reg_max + k] : to_f32(src[s * p.reg_max + k]); m = v > m ? v : m; }
    float se = 0.f, sw = 0.f;
    for (int k = 0; k < p.reg_max; ++k) { const float e = expf((b32 ? b32[s * p.reg_max + k] : to_f32(src[s * p.reg_max + k])) - m); se += e; sw += e * (float)k; }
    d4[s] = sw / se;
  }
  const float ax = (float)x + 0.5f, ay = (float)y + 0.5f, st = (float)p.lstride[l];
  float* o = p.out + (size_t)a * (4 + p.nc + p.nm);
  o[0] = (ax - d4[0]) * st; o[1] = (ay - d4[1]) * st; o[2] = (ax + d4[2]) * st; o[3] = (ay + d4[3]) * st;
  const T* cls = src + (p.cls_off > 0 ? p.cls_off : 4 * p.reg_max);
  for (int c = 0; c < p.nc; ++c) o[4 + c] = 1.f / (1.f + __expf(-to_f32(cls[c])));
  const T* mc = p.mc_off > 0 ? src + p.mc_off : cls + p.nc;
  for (int c = 0; c < p.nm; ++c) o[4 + p.nc + c] = to_f32(mc[c]);
}

int yolo_decode_launch(const mtx_yolo_decode_args* a, void* stream, const char** err) {
  if (!a->out || a->n_levels < 1 || a->n_levels > 4 || a->reg_max < 1 || a->nc < 1) { *err = "yolo_decode: bad arguments"; return MTX_ERR_INVALID; }
  int total = 0;
  for (int l = 0; l < a->n_levels; ++l) { if (!a->level[l]) { *err = "yolo_decode: null level"; return MTX_ERR_INVALID; } total += a->lh[l] * a->lw[l]; }
  if (total == 0) return MTX_OK;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (a->dtype == MTX_BF16) MTX_LAUNCH((yolo_decode_kernel<__bf16>), dim3(blocks), dim3(256), 0, stream, *a, total);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((yolo_decode_kernel<_Float16>), dim3(blocks), dim3(256), 0, stream, *a, total);
  else { *err = "yolo_decode: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

}  // namespace mtx
