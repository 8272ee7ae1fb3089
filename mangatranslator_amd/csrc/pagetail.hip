// pagetail.hip — the byte-level tail of the inpainting stage on the device (include/mtx_hip.h mtx_tail_args).
//
// Around the diffusion pipeline the reference runs four pieces of image arithmetic on the host, per region
// (core/image/inpainting.py:1258-1313 / 1577-1665 Klein, :543-611 / 877-968 Kontext):
//     crop -> PIL LANCZOS to the inference size -> [pipeline] -> PIL LANCZOS back -> Lab luminance match -> alpha composite into the page
// 100-160 ms at 2048 x 3072 (tools/time_klein_host.py) next to 8 x 57 ms of denoising.  These kernels keep the whole chain in HBM:
//   RESAMPLE   Pillow's 8-bit resampling, bit for bit: per output coordinate a window [xmin, xmin + n) and fixed-point taps (22
//              fractional bits) — built on the host in float64 exactly as Pillow builds them (core/image/device_tail.py) — accumulated
//              in int32 from 2^21, shifted, clipped to 0..255; one pass per axis, horizontal first, the intermediate rounded to
//              uint8 like Pillow's two-pass resize.
//   COMPOSITE  patch * alpha + page * (1 - alpha) in fp32 with the reference's operation order and NO contraction into FMAs
//              (round-to-nearest intrinsics), truncated to uint8: bit-identical to the numpy expression.
//   LAB_STATS / LAB_REMAP  OpenCV's fixed-point 8-bit RGB -> Lab (integer tables: exact), the sums the luminance match needs as
//              exact integers, the affine L / a / b remap on the masked pixels and the float Lab -> RGB way back.
//   EDT_COLS / EDT_ROWS    the feathered composite weight: exact Euclidean distance to the mask inside a window of the blur radius.
// All HBM-bound byte work: one thread per output pixel, coalesced rows; nothing here is reshaped into a GEMM.
#include "mtx_device.h"

namespace mtx {

// ---- Pillow resample, one axis (coeff_bits = 0 / 22).  The same arithmetic with 16-bit taps of `coeff_bits` fractional bits is ATen's
// uint8 antialiased resize (aten/src/ATen/native/cpu/UpSampleKernelAVXAntialias.h), i.e. torchvision's `resize(antialias=True)` of a
// uint8 image — what HF's Sam2ImageProcessorFast runs on the page (core/image/device_tail.py aten_aa_bilinear_tables) ------------------
// axis 0 (horizontal): dst[y][xx][c] = clip8((2^21 + sum_k src[y + row0][xmin(xx) + k][c] * coeff[xx][k]) >> 22)
// axis 1 (vertical):   dst[yy][x][c] = clip8((2^21 + sum_k src[ymin(yy) + k][x][c] * coeff[yy][k]) >> 22)
__global__ __launch_bounds__(256) void tail_resample_kernel(mtx_tail_args p) {
  const long total = (long)p.out_h * p.out_w;
  const uint8_t* S = reinterpret_cast<const uint8_t*>(p.src);
  uint8_t* D = reinterpret_cast<uint8_t*>(p.dst);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int oy = (int)(idx / p.out_w), ox = (int)(idx % p.out_w);
    const int o = p.axis == 0 ? ox : oy;
    const int lo = p.bounds[2 * o], n = p.bounds[2 * o + 1];
    const int* k = p.coeff + (long)o * p.ksize;
    const int bits = p.coeff_bits > 0 ? p.coeff_bits : 22, half = 1 << (bits - 1);
    int acc[4] = {half, half, half, half};
    for (int t = 0; t < n; ++t) {
      const uint8_t* px = p.axis == 0 ? S + ((long)(oy + p.src_row0) * p.ld_src + (long)(lo + t) * p.c)
                                      : S + ((long)(lo + t) * p.ld_src + (long)ox * p.c);
      const int kv = k[t];
      for (int c = 0; c < p.c; ++c) acc[c] += (int)px[c] * kv;
    }
    uint8_t* out = D + ((long)oy * p.ld_dst + (long)ox * p.c);
    for (int c = 0; c < p.c; ++c) {
      const int v = acc[c] >> bits;
      out[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

// ---- alpha composite of a patch into the page (in place on the page) ----------------------------------------------------------
__device__ __forceinline__ float rn_mul(float a, float b) {
#ifdef MTX_EMU
  volatile float r = a * b; return r;
#else
  return __fmul_rn(a, b);
#endif
}
__device__ __forceinline__ float rn_add(float a, float b) {
#ifdef MTX_EMU
  volatile float r = a + b; return r;
#else
  return __fadd_rn(a, b);
#endif
}
__device__ __forceinline__ float rn_div(float a, float b) {
#ifdef MTX_EMU
  volatile float r = a / b; return r;
#else
  return __fdiv_rn(a, b);
#endif
}

__global__ __launch_bounds__(256) void tail_composite_kernel(mtx_tail_args p) {
  // the patch's [0, out_h) x [0, out_w) window (already clipped to the page by the launcher) lands at (y, x)
  const long total = (long)p.out_h * p.out_w;
  const uint8_t* S = reinterpret_cast<const uint8_t*>(p.src);
  uint8_t* D = reinterpret_cast<uint8_t*>(p.dst);
  const int src_c = p.src_c > 0 ? p.src_c : p.c;                               // a patch with more channels than the page blends its first c
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int py = (int)(idx / p.out_w), px = (int)(idx % p.out_w);
    const float a = p.alpha[(long)py * p.ld_alpha + px];
    const float one_minus = rn_add(1.0f, -a);
    const uint8_t* s = S + ((long)py * p.ld_src + (long)px * src_c);
    uint8_t* d = D + ((long)(p.y + py) * p.ld_dst + (long)(p.x + px) * p.page_c);
    for (int c = 0; c < p.page_c; ++c) {
      const float sv = c < p.c ? rn_div((float)s[c], 255.0f) : 1.0f;           // a page with more channels than the patch: opaque source alpha
      const float dv = rn_div((float)d[c], 255.0f);
      const float v = rn_mul(rn_add(rn_mul(sv, a), rn_mul(dv, one_minus)), 255.0f);
      d[c] = (uint8_t)v;                                                       // truncation, as .astype(np.uint8)
    }
  }
}

// ---- OpenCV 8-bit RGB -> Lab (fixed point), shared by the two Lab kernels ------------------------------------------------------
__device__ __forceinline__ void rgb_to_lab8(const mtx_tail_args& p, const uint8_t* px, int& L, int& A, int& B) {
  const int r = p.gamma_tab[px[0]], g = p.gamma_tab[px[1]], b = p.gamma_tab[px[2]];
  int f[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int xyz = (r * p.lab_coef[c * 3] + g * p.lab_coef[c * 3 + 1] + b * p.lab_coef[c * 3 + 2] + (1 << 11)) >> 12;
    xyz = xyz < 0 ? 0 : (xyz > p.cbrt_n - 1 ? p.cbrt_n - 1 : xyz);
    f[c] = p.cbrt_tab[xyz];
  }
  const int lshift = -((16 * 255 * (1 << 15) + 50) / 100), h2 = 1 << 14;
  L = (((116 * 255 + 50) / 100) * f[1] + lshift + h2) >> 15;
  A = (500 * (f[0] - f[1]) + 128 * (1 << 15) + h2) >> 15;
  B = (200 * (f[1] - f[2]) + 128 * (1 << 15) + h2) >> 15;
  L = L < 0 ? 0 : (L > 255 ? 255 : L); A = A < 0 ? 0 : (A > 255 ? 255 : A); B = B < 0 ? 0 : (B > 255 ? 255 : B);
}

// sums over the CONTEXT pixels (mask == 0) of both images: [n, L, L^2, a, b] of `src` (the generated patch) then [L, L^2, a, b] of
// `other` (the original crop), as exact integers (uint64 atomics; a crop is at most a few megapixels of 8-bit values)
__global__ __launch_bounds__(256) void tail_lab_stats_kernel(mtx_tail_args p) {
  __shared__ unsigned long long red[9];
  if (threadIdx.x < 9) red[threadIdx.x] = 0ull;
  __syncthreads();
  const long total = (long)p.out_h * p.out_w;
  const uint8_t* S = reinterpret_cast<const uint8_t*>(p.src);
  const uint8_t* O = reinterpret_cast<const uint8_t*>(p.other);
  unsigned long long acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int py = (int)(idx / p.out_w), px = (int)(idx % p.out_w);
    if (p.mask[(long)py * p.ld_mask + px]) continue;
    int L, A, B;
    rgb_to_lab8(p, S + ((long)py * p.ld_src + (long)px * p.c), L, A, B);
    acc[0] += 1; acc[1] += L; acc[2] += (unsigned long long)(L * L); acc[3] += A; acc[4] += B;
    rgb_to_lab8(p, O + ((long)py * p.ld_other + (long)px * p.c), L, A, B);
    acc[5] += L; acc[6] += (unsigned long long)(L * L); acc[7] += A; acc[8] += B;
  }
  for (int i = 0; i < 9; ++i) atomicAdd(&red[i], acc[i]);
  __syncthreads();
  if (threadIdx.x < 9) atomicAdd(p.sums + threadIdx.x, red[threadIdx.x]);
}

// params: {g_mean, gain, o_mean, shift_a, shift_b, use_a, use_b}: L' = clip((L - g_mean) * gain + o_mean), a' = clip(a + shift_a) ... on the
// masked pixels; every pixel then goes Lab -> RGB through the float formulation (as cv2.cvtColor(COLOR_LAB2RGB) does for the whole patch)
__global__ __launch_bounds__(256) void tail_lab_remap_kernel(mtx_tail_args p) {
  const long total = (long)p.out_h * p.out_w;
  const uint8_t* S = reinterpret_cast<const uint8_t*>(p.src);
  uint8_t* D = reinterpret_cast<uint8_t*>(p.dst);
  const float g_mean = p.params[0], gain = p.params[1], o_mean = p.params[2], sa = p.params[3], sb = p.params[4];
  const bool use_a = p.params[5] != 0.f, use_b = p.params[6] != 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int py = (int)(idx / p.out_w), px = (int)(idx % p.out_w);
    int Li, Ai, Bi;
    rgb_to_lab8(p, S + ((long)py * p.ld_src + (long)px * p.c), Li, Ai, Bi);
    float L = (float)Li, A = (float)Ai, B = (float)Bi;
    if (p.mask[(long)py * p.ld_mask + px]) {
      L = rn_add(rn_mul(rn_add(L, -g_mean), gain), o_mean);
      L = L < 0.f ? 0.f : (L > 255.f ? 255.f : L);
      if (use_a) { A = rn_add(A, sa); A = A < 0.f ? 0.f : (A > 255.f ? 255.f : A); }
      if (use_b) { B = rn_add(B, sb); B = B < 0.f ? 0.f : (B > 255.f ? 255.f : B); }
    }
    L = (float)(int)L; A = (float)(int)A; B = (float)(int)B;            // .astype(np.uint8): truncation
    // float Lab -> RGB (core/image/color.py lab_to_rgb_u8)
    const float Lf = rn_mul(L, 100.0f / 255.0f), Af = rn_add(A, -128.f), Bf = rn_add(B, -128.f);
    float y, fy;
    if (Lf <= (float)(0.008856 * 903.3)) { y = rn_div(Lf, 903.3f); fy = rn_add(rn_mul(7.787f, y), (float)(16.0 / 116.0)); }
    else { fy = rn_div(rn_add(Lf, 16.0f), 116.0f); y = rn_mul(rn_mul(fy, fy), fy); }
    const float thr = (float)(7.787 * 0.008856 + 16.0 / 116.0);
    float fx = rn_add(fy, rn_div(Af, 500.0f)), fz = rn_add(fy, -rn_div(Bf, 200.0f));
    float x = fx <= thr ? rn_div(rn_add(fx, -(float)(16.0 / 116.0)), 7.787f) : rn_mul(rn_mul(fx, fx), fx);
    float z = fz <= thr ? rn_div(rn_add(fz, -(float)(16.0 / 116.0)), 7.787f) : rn_mul(rn_mul(fz, fz), fz);
    x = rn_mul(x, 0.950456f); z = rn_mul(z, 1.088754f);
    const float M[9] = {3.240479f, -1.53715f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
    uint8_t* out = D + ((long)py * p.ld_dst + (long)px * p.c);
    for (int r = 0; r < 3; ++r) {
      float lin = rn_add(rn_add(rn_mul(M[r * 3], x), rn_mul(M[r * 3 + 1], y)), rn_mul(M[r * 3 + 2], z));
      lin = lin < 0.f ? 0.f : (lin > 1.f ? 1.f : lin);
      const float g = lin <= 0.0031308f ? rn_mul(lin, 12.92f) : rn_add(rn_mul(1.055f, powf(lin, 1.0f / 2.4f)), -0.055f);
      float v = rintf(rn_mul(g, 255.0f));
      v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
      out[r] = (uint8_t)v;
    }
  }
}

// ---- feather: exact Euclidean distance to the mask, up to `radius`, as the composite weight ----------------------------------------
// The reference's weight is 1 on the mask and clip(1 - d / blur, 0, 1) outside it, d = scipy's exact Euclidean distance transform of the
// crop (core/image/inpainting.py:1126-1163; blur <= MAX_BLUR_RADIUS = 10).  Only distances below `blur` matter, so the transform is
// evaluated in a window: pass 1 (columns) g[y][x] = min |dy| <= R with mask[y + dy][x] set (R + 1: none), pass 2 (rows)
// d^2 = min over |dx| <= R of dx^2 + g[y][x + dx]^2 — exact integers, the same squared distances the separable EDT produces wherever
// d < R; beyond that the weight is 0 either way.  The weight itself comes from a table indexed by d^2 (built on the host in float64
// and rounded to float32 exactly like the numpy expression), so no device sqrt / division rounding enters.
__global__ __launch_bounds__(256) void tail_edt_cols_kernel(mtx_tail_args p) {
  const long total = (long)p.out_h * p.out_w;
  const uint8_t* M = reinterpret_cast<const uint8_t*>(p.src);
  uint8_t* G = reinterpret_cast<uint8_t*>(p.dst);
  const int R = p.ksize;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int py = (int)(idx / p.out_w), px = (int)(idx % p.out_w);
    int g = R + 1;
    for (int dy = 0; dy <= R; ++dy) {
      const bool up = py - dy >= 0 && M[(long)(py - dy) * p.ld_src + px] != 0;
      const bool dn = py + dy < p.out_h && M[(long)(py + dy) * p.ld_src + px] != 0;
      if (up || dn) { g = dy; break; }
    }
    G[(long)py * p.ld_dst + px] = (uint8_t)g;
  }
}

// x, y, page_c, cbrt_n: the rectangle [x, page_c) x [y, cbrt_n) outside which the weight is 0 (composite_clip_bbox); axis != 0: strict
// (weight 0 off the mask); params: the table [R * R + 1]
__global__ __launch_bounds__(256) void tail_edt_rows_kernel(mtx_tail_args p) {
  const long total = (long)p.out_h * p.out_w;
  const uint8_t* G = reinterpret_cast<const uint8_t*>(p.src);
  float* A = reinterpret_cast<float*>(p.dst);
  const int R = p.ksize, R2 = R * R;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int py = (int)(idx / p.out_w), px = (int)(idx % p.out_w);
    const uint8_t* row = G + (long)py * p.ld_src;
    float a = 0.f;
    if (px >= p.x && px < p.page_c && py >= p.y && py < p.cbrt_n) {
      if (row[px] == 0) a = 1.f;
      else if (p.axis == 0) {
        int best = R2 + 1;
        for (int dx = -R; dx <= R; ++dx) {
          const int qx = px + dx;
          if (qx < 0 || qx >= p.out_w) continue;
          const int g = row[qx];
          if (g > R) continue;
          const int d2 = dx * dx + g * g;
          best = d2 < best ? d2 : best;
        }
        a = best <= R2 ? p.params[best] : 0.f;
      }
    }
    A[(long)py * p.ld_dst + px] = a;
  }
}

int tail_launch(const mtx_tail_args* a, void* stream, const char** err) {
  if (!a->src || !a->dst || a->out_h < 0 || a->out_w < 0) { *err = "page tail: null operand"; return MTX_ERR_INVALID; }
  const bool edt = a->kind == MTX_TAIL_EDT_COLS || a->kind == MTX_TAIL_EDT_ROWS;
  if (!edt && (a->c < 1 || a->c > 4)) { *err = "page tail: 1..4 channels"; return MTX_ERR_INVALID; }
  const long total = (long)a->out_h * a->out_w;
  if (total == 0) return MTX_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 grid((unsigned)blocks), block(256);
  switch (a->kind) {
    case MTX_TAIL_RESAMPLE:
      if (!a->bounds || !a->coeff || a->ksize < 1 || (a->axis != 0 && a->axis != 1)) { *err = "page tail (resample): bounds / coeff / axis"; return MTX_ERR_INVALID; }
      MTX_LAUNCH(tail_resample_kernel, grid, block, 0, stream, *a);
      return MTX_OK;
    case MTX_TAIL_COMPOSITE:
      if (!a->alpha || a->page_c < a->c || a->page_c > 4 || a->x < 0 || a->y < 0 || (a->src_c != 0 && a->src_c < a->c)) { *err = "page tail (composite): alpha / page channels / origin"; return MTX_ERR_INVALID; }
      MTX_LAUNCH(tail_composite_kernel, grid, block, 0, stream, *a);
      return MTX_OK;
    case MTX_TAIL_LAB_STATS:
      if (!a->gamma_tab || !a->cbrt_tab || !a->lab_coef || !a->mask || !a->other || !a->sums || a->c != 3) { *err = "page tail (Lab statistics): tables / mask / other image / sums"; return MTX_ERR_INVALID; }
      MTX_LAUNCH(tail_lab_stats_kernel, grid, block, 0, stream, *a);
      return MTX_OK;
    case MTX_TAIL_LAB_REMAP:
      if (!a->gamma_tab || !a->cbrt_tab || !a->lab_coef || !a->mask || !a->params || a->c != 3) { *err = "page tail (Lab remap): tables / mask / params"; return MTX_ERR_INVALID; }
      MTX_LAUNCH(tail_lab_remap_kernel, grid, block, 0, stream, *a);
      return MTX_OK;
    case MTX_TAIL_EDT_COLS:
      if (a->ksize < 1 || a->ksize > 254) { *err = "page tail (feather): radius 1..254"; return MTX_ERR_INVALID; }
      MTX_LAUNCH(tail_edt_cols_kernel, grid, block, 0, stream, *a);
      return MTX_OK;
    case MTX_TAIL_EDT_ROWS:
      if (a->ksize < 1 || a->ksize > 254 || !a->params) { *err = "page tail (feather): radius 1..254, weight table"; return MTX_ERR_INVALID; }
      MTX_LAUNCH(tail_edt_rows_kernel, grid, block, 0, stream, *a);
      return MTX_OK;
    default: *err = "page tail: unknown kind"; return MTX_ERR_INVALID;
  }
}

}  // namespace mtx
