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
  if (kind == MTX_EW_MAXPOOL) { const int k = p.i0, s = p.i1, pd = k / 2; oh = (p.h + 2 * pd - k) / s + 1; ow = (p.w + 2 * pd - k) / s + 1; }
  const long total = p.n * oh * ow * C8;
  const T* A = reinterpret_cast<const T*>(p.a);
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
    } else if (kind == MTX_EW_MAXPOOL) {
      const int k = p.i0, s = p.i1, pd = k / 2;
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
      } else if (kind == MTX_EW_ADD || kind == MTX_EW_MUL) {
        float g[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(Bp + pix * p.ldb + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = kind == MTX_EW_ADD ? f[e] + g[e] : f[e] * g[e];
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

int ew_launch(const mtx_ew_args* a, void* stream, const char** err) {
  if (!a->a || !a->y) { *err = "elementwise: null operand"; return MTX_ERR_INVALID; }
  if (a->c % 8 || a->lda % 8 || a->ldy % 8 || (a->b && a->ldb % 8)) { *err = "elementwise: C and pixel strides must be multiples of 8"; return MTX_ERR_INVALID; }
  if ((a->kind == MTX_EW_SCALE_RES || a->kind == MTX_EW_ADD || a->kind == MTX_EW_MUL || a->kind == MTX_EW_GATE_RES) && !a->b) { *err = "elementwise: missing operand b"; return MTX_ERR_INVALID; }
  if ((a->kind == MTX_EW_SCALE_RES || a->kind == MTX_EW_GATE_RES) && !a->s) { *err = "elementwise: missing operand s"; return MTX_ERR_INVALID; }
  if (a->kind == MTX_EW_MAXPOOL && (a->i0 < 1 || a->i1 < 1)) { *err = "elementwise: maxpool needs kernel/stride"; return MTX_ERR_INVALID; }
  long oh = a->h, ow = a->w;
  if (a->kind == MTX_EW_UPSAMPLE2X) { oh *= 2; ow *= 2; }
  if (a->kind == MTX_EW_MAXPOOL) { const int pd = a->i0 / 2; oh = (a->h + 2 * pd - a->i0) / a->i1 + 1; ow = (a->w + 2 * pd - a->i0) / a->i1 + 1; }
  const long total = a->n * oh * ow * (a->c / 8);
  if (total <= 0) return MTX_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 2048 * 4) blocks = 2048 * 4;
  if (a->dtype == MTX_BF16) MTX_LAUNCH((ew_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((ew_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else { *err = "elementwise: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

// ---- RCAN channel attention squeeze/excite: one workgroup per image ---------------------------
__global__ __launch_bounds__(256) void ca_kernel(mtx_ca_args p) {
  __shared__ float mean[512];
  __shared__ float hid[128];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < p.c; c += 256) {
    float s = 0.f;
    const float* src = p.chan_sum + (size_t)n * p.tiles * p.c + c;
    for (int t = 0; t < p.tiles; ++t) s += src[(size_t)t * p.c];
    mean[c] = s * p.inv_hw;
  }
  __syncthreads();
  for (int r = tid; r < p.cr; r += 256) {
    float s = p.b1 ? p.b1[r] : 0.f;
    for (int c = 0; c < p.c; ++c) s += p.w1[r * p.c + c] * mean[c];
    hid[r] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < p.c; c += 256) {
    float s = p.b2 ? p.b2[c] : 0.f;
    for (int r = 0; r < p.cr; ++r) s += p.w2[c * p.cr + r] * hid[r];
    p.s[(size_t)n * p.c + c] = 1.f / (1.f + __expf(-s));
  }
}

int ca_launch(const mtx_ca_args* a, void* stream, const char** err) {
  if (!a->chan_sum || !a->w1 || !a->w2 || !a->s) { *err = "channel_attention: null operand"; return MTX_ERR_INVALID; }
  if (a->c > 512 || a->cr > 128 || a->c < 1 || a->cr < 1) { *err = "channel_attention: C <= 512 and C/r <= 128"; return MTX_ERR_INVALID; }
  MTX_LAUNCH(ca_kernel, dim3((unsigned)a->n), dim3(256), 0, stream, *a);
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
    for (long idx = tid0; idx < total; idx += step) {
      const long x = idx % ow, y = (idx / ow) % oh, n = idx / (ow * oh);
      T* o = D + idx * p.c_pad;
      int oc = 0;
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
  const float sy = (float)p.hs / (float)p.hd, sx = (float)p.ws / (float)p.wd;
  const T* S = reinterpret_cast<const T*>(p.src);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long x = idx % p.wd, y = (idx / p.wd) % p.hd, n = idx / (p.wd * p.hd);
    // torch area_pixel_compute_source_index: max(0, (dst + 0.5) * scale - 0.5)
    float fy = ((float)y + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = ((float)x + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
    long y0 = (long)fy, x0 = (long)fx;
    if (y0 > p.hs - 1) y0 = p.hs - 1;
    if (x0 > p.ws - 1) x0 = p.ws - 1;
    const long y1 = y0 + (y0 < p.hs - 1 ? 1 : 0), x1 = x0 + (x0 < p.ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const T* base = S + n * p.hs * p.ws;
    const float v00 = to_f32(base[y0 * p.ws + x0]), v01 = to_f32(base[y0 * p.ws + x1]);
    const float v10 = to_f32(base[y1 * p.ws + x0]), v11 = to_f32(base[y1 * p.ws + x1]);
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

}  // namespace mtx
