// conv.hip — NHWC 2-D convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces the F.conv2d calls inside the third-party networks the reference drives:
//   RCAN upscaler  (spandrel, called at core/image/image_utils.py:369-374)
//   YOLO backbones (ultralytics, called at core/image/detection.py:1337-1345)
//   FLUX VAE       (diffusers, called inside core/image/inpainting.py:877-887)
//
// Design (MI355X-first, not an im2col GEMM):
//   * one workgroup = 4 waves = a 16-wide spatial tile of output pixels x 64 output channels;
//   * the INPUT HALO of that tile (18x18 px for 3x3/s1) is staged ONCE per 64-channel slice
//     into LDS as 128-byte pixel rows (one HBM cache line per pixel), and all nine filter taps
//     read their MFMA fragments from it with shifted addresses — activations cross HBM->LDS once
//     instead of nine times;
//   * LDS rows are XOR-swizzled in 16-byte chunks by (row & 7): a ds_read_b128 lane group always
//     covers 16 consecutive rows with chunk ids {c, c^1}, which land on 16 distinct 16-B slots of
//     the 256-B bank row (conflict-free);
//   * MFMA operands are swapped (A = weights, B = pixels) so each lane ends up holding 4
//     CONSECUTIVE output channels of one pixel: the epilogue packs them with one 8-byte LDS write,
//     and the tile leaves the CU as full 16-byte channel chunks (NHWC-coalesced), with bias,
//     activation, residual add, pixel-shuffle addressing and the per-tile channel sums of the
//     RCAN channel-attention pool fused in.
#include "mtx_device.h"

namespace mtx {

struct ConvParams {
  const unsigned char* x; const unsigned char* w; const float* bias; const unsigned char* res;
  unsigned char* y; float* chan_sum;
  int n, h, w_in, cin, cout, ho, wo;
  int ldx, ldy, ldres;
  int act; float act_param; float res_scale; int act_after;
  int ps;           // pixel shuffle factor (0 or 2)
  int res_bcast;
  int pad_lo;       // zero padding on the top/left side
  int tiles_x, tiles_y, nblk;
  const int* valid_hw;   // device {valid_h, valid_w} or null (stride 1 only): outputs beyond are zero
  const float* out_scale;   // device [n][cout] factor on act(conv + bias), before the residual (or null)
};

template <int KS, int S>
struct ConvCfg {
  static constexpr int TW = 16;
  static constexpr int TH = (S == 1) ? 16 : 8;
  static constexpr int NPIX = TW * TH;
  static constexpr int BN = 64;
  static constexpr int HW_ = (TW - 1) * S + KS;
  static constexpr int HH_ = (TH - 1) * S + KS;
  static constexpr int HWE = (HW_ + 1) / 2;           // even-column plane width (S == 2)
  static constexpr int HALO_PIX = HW_ * HH_;
  static constexpr int HALO_BYTES = HALO_PIX * 128;
  static constexpr int OUT_BYTES = NPIX * 128;
  static constexpr int A_BYTES = HALO_BYTES > OUT_BYTES ? HALO_BYTES : OUT_BYTES;
  static constexpr int WT_ROWS = KS * BN;             // one filter row (ky) at a time
  static constexpr int WT_BYTES = WT_ROWS * 128 > 64 * 32 * 4 ? WT_ROWS * 128 : 64 * 32 * 4;
  static constexpr int SMEM = A_BYTES + WT_BYTES;
  static constexpr int FR = TH / 4;                   // 16-pixel fragments (tile rows) per wave
  // LDS row index of halo pixel (hy, hx)
  static __device__ __forceinline__ int lds_pix(int hy, int hx) {
    if (S == 1) return hy * HW_ + hx;
    return hy * HW_ + (hx & 1) * HWE + (hx >> 1);     // de-interleave column parity for stride 2
  }
};

template <typename T, int KS, int S>
__global__ __launch_bounds__(256) void conv2d_nhwc_kernel(ConvParams p) {
  typedef ConvCfg<KS, S> C;
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::SMEM];
  unsigned char* halo = smem;
  unsigned char* outs = smem;                 // aliases the halo after the main loop
  unsigned char* wts = smem + C::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int l15 = lane & 15;
  const int q = lane >> 4;

  const unsigned tiles = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned nwg = tiles * (unsigned)p.nblk * (unsigned)p.n;
  unsigned lin = xcd_remap(blockIdx.x, nwg);
  const int nb = (int)(lin % (unsigned)p.nblk);
  lin /= (unsigned)p.nblk;
  const int tile = (int)(lin % tiles);
  const int img = (int)(lin / tiles);
  const int ty0 = (tile / p.tiles_x) * C::TH;
  const int tx0 = (tile % p.tiles_x) * C::TW;
  const int n0 = nb * C::BN;
  const int iy0 = ty0 * S - p.pad_lo;
  const int ix0 = tx0 * S - p.pad_lo;

  f32x4 acc[C::FR][4];
#pragma unroll
  for (int i = 0; i < C::FR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t img_off = (size_t)img * p.h * p.w_in;

  // Software pipeline (round 4): the halo slice and the filter rows used to be fetched global -> register -> LDS right where they were
  // needed — per 64-channel slice four exposed memory round trips against ~2.4 us of MFMAs.  Now every fetch is ISSUED one phase early
  // into registers (next filter row / next slice's halo + first row, before the current row's MFMAs) and only STORED to LDS where the
  // old code fetched: same LDS image, same barriers, the latencies run under the matrix work.
  constexpr int HN = (C::HALO_PIX * 8 + 255) / 256;     // 16-byte halo chunks per thread and slice
  constexpr int WN = (C::WT_ROWS * 8 + 255) / 256;      // 16-byte filter chunks per thread and row
  constexpr int HB = HN;                                // halo chunks in flight per batch: all of a slice (the stride-2 tile is LDS-limited to one workgroup per CU either way)
  u32x4 wreg[WN];
  auto stage_halo = [&](int kc0) {
#pragma unroll
    for (int b0 = 0; b0 < HN; b0 += HB) {
      u32x4 hreg[HB];
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int idx = tid + (b0 + j) * 256;
        const int c = idx & 7;
        const int hp = idx >> 3;
        const int hy = hp / C::HW_;
        const int hx = hp - hy * C::HW_;
        const int gy = iy0 + hy, gx = ix0 + hx;
        const int ch = kc0 + c * 8;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (b0 + j < HN && idx < C::HALO_PIX * 8 && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_in && ch < p.cin) {
          const size_t off = ((img_off + (size_t)gy * p.w_in + gx) * (size_t)p.ldx + ch) * sizeof(T);
          v = *reinterpret_cast<const u32x4*>(p.x + off);
        }
        hreg[j] = v;
      }
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int idx = tid + (b0 + j) * 256;
        if (b0 + j < HN && idx < C::HALO_PIX * 8) {
          const int c = idx & 7;
          const int hp = idx >> 3;
          const int hy = hp / C::HW_;
          const int hx = hp - hy * C::HW_;
          const int lp = C::lds_pix(hy, hx);
          *reinterpret_cast<u32x4*>(halo + lp * 128 + ((c ^ (lp & 7)) << 4)) = hreg[j];
        }
      }
    }
  };
  auto load_w = [&](int kc0, int ky) {
#pragma unroll
    for (int it = 0; it < WN; ++it) {
      const int idx = tid + it * 256;
      const int c = idx & 7;
      const int row = idx >> 3;
      const int kx = row / C::BN;
      const int co = row - kx * C::BN;
      const int gco = n0 + co;
      const int ch = kc0 + c * 8;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (idx < C::WT_ROWS * 8 && gco < p.cout && ch < p.cin) {
        const size_t off = (((size_t)gco * (KS * KS) + ky * KS + kx) * (size_t)p.cin + ch) * sizeof(T);
        v = *reinterpret_cast<const u32x4*>(p.w + off);
      }
      wreg[it] = v;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int it = 0; it < WN; ++it) {
      const int idx = tid + it * 256;
      if (idx < C::WT_ROWS * 8) {
        const int c = idx & 7;
        const int row = idx >> 3;
        const int co = row % C::BN;
        *reinterpret_cast<u32x4*>(wts + row * 128 + ((c ^ (co & 7)) << 4)) = wreg[it];
      }
    }
  };
  load_w(0, 0);
  for (int kc0 = 0; kc0 < p.cin; kc0 += 64) {
    __syncthreads();   // previous slice's fragment reads are done
    stage_halo(kc0);   // this slice's halo: [HALO_PIX] x 64 channels, zero outside the image
    const int rem = p.cin - kc0;
    const int nks = rem >= 64 ? 2 : (rem + 31) / 32;

    for (int ky = 0; ky < KS; ++ky) {
      if (ky > 0) __syncthreads();   // previous filter row's fragment reads are done
      store_w();                     // one filter row: [KS taps][64 couts] x 64 channels
      __syncthreads();
      if (ky + 1 < KS) load_w(kc0, ky + 1);
      else if (kc0 + 64 < p.cin) load_w(kc0 + 64, 0);

#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        for (int ks = 0; ks < nks; ++ks) {
          const int cch = ks * 4 + q;
          v8 wf[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = kx * C::BN + j * 16 + l15;
            wf[j] = *reinterpret_cast<const v8*>(wts + row * 128 + ((cch ^ (l15 & 7)) << 4));
          }
#pragma unroll
          for (int i = 0; i < C::FR; ++i) {
            const int r = wv * C::FR + i;
            const int lp = C::lds_pix(r * S + ky, l15 * S + kx);
            const v8 xf = *reinterpret_cast<const v8*>(halo + lp * 128 + ((cch ^ (lp & 7)) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Traits<T>::mfma(wf[j], xf, acc[i][j]);
          }
        }
      }
    }
  }

  // ---- epilogue 1: bias + activation, 4 consecutive channels per lane -> LDS tile -----------
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = n0 + j * 16 + q * 4;
    float b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (p.bias != nullptr && co + r < p.cout) ? p.bias[co + r] : 0.f;
    const int chunk = j * 2 + (q >> 1);
#pragma unroll
    for (int i = 0; i < C::FR; ++i) {
      const int pt = (wv * C::FR + i) * 16 + l15;
      v4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(p.act_after ? acc[i][j][r] + b[r] : apply_act(acc[i][j][r] + b[r], p.act, p.act_param));
      *reinterpret_cast<v4*>(outs + pt * 128 + ((chunk ^ (pt & 7)) << 4) + ((q & 1) << 3)) = o;
    }
  }
  __syncthreads();

  // ---- epilogue 2: 16-byte channel chunks out, residual / pixel-shuffle / channel sums -------
  const int c = tid & 7;           // constant per thread: idx += 256 keeps idx & 7
  const int co = n0 + c * 8;
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  const int vh = p.valid_hw ? p.valid_hw[0] : p.ho, vw = p.valid_hw ? p.valid_hw[1] : p.wo;
  for (int idx = tid; idx < C::NPIX * 8; idx += 256) {
    const int pt = idx >> 3;
    const int oy = ty0 + (pt >> 4), ox = tx0 + (pt & 15);
    if (oy < p.ho && ox < p.wo && co < p.cout) {
      u32x4 raw = *reinterpret_cast<const u32x4*>(outs + pt * 128 + ((c ^ (pt & 7)) << 4));
      const bool outside = oy >= vh || ox >= vw;          // beyond the image inside a bucket canvas: zero, no sums, no residual
      size_t opix;
      int oc = co;
      if (p.ps == 2) {
        const int cps = p.cout >> 2;
        const int g = co / cps;
        oc = co - g * cps;
        opix = ((size_t)img * (2 * p.ho) + (2 * oy + (g >> 1))) * (size_t)(2 * p.wo) + (2 * ox + (g & 1));
      } else {
        opix = ((size_t)img * p.ho + oy) * (size_t)p.wo + ox;
      }
      if (outside) raw = u32x4{0u, 0u, 0u, 0u};
      else if (p.chan_sum != nullptr || p.res != nullptr || p.out_scale != nullptr) {
        float f[8];
        unpack8<T>(raw, f);
        if (p.chan_sum != nullptr) {                       // the sums are those of act(conv + bias) as rounded to T, BEFORE out_scale (mtx_hip.h):
#pragma unroll                                          // the same quantity the c64 kernel sums, whichever kernel the dispatcher picks
          for (int e = 0; e < 8; ++e) csum[e] += f[e];
        }
        if (p.out_scale != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= p.out_scale[(size_t)img * p.cout + co + e];
          if (p.res == nullptr) raw = pack8<T>(f);
        }
        if (p.res != nullptr) {
          const size_t rpix = p.res_bcast ? opix - (size_t)img * (p.ps == 2 ? 4 : 1) * (size_t)p.ho * (size_t)p.wo : opix;
          const u32x4 rr = *reinterpret_cast<const u32x4*>(p.res + (rpix * (size_t)p.ldres + oc) * sizeof(T));
          float g8[8];
          unpack8<T>(rr, g8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += p.res_scale * g8[e];
          if (p.act_after) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = apply_act(f[e], p.act, p.act_param);
          }
          raw = pack8<T>(f);
        }
      }
      *reinterpret_cast<u32x4*>(p.y + (opix * (size_t)p.ldy + oc) * sizeof(T)) = raw;
    }
  }
  if (p.chan_sum != nullptr) {
    // reduce the 32 threads that share chunk c through LDS (filter buffer is free now)
    float* red = reinterpret_cast<float*>(wts);     // [32][64]
    const int slot = tid >> 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[slot * 64 + c * 8 + e] = csum[e];
    __syncthreads();
    if (tid < 64 && n0 + tid < p.cout) {
      float s = 0.f;
      for (int k = 0; k < 32; ++k) s += red[k * 64 + tid];
      p.chan_sum[((size_t)img * tiles + tile) * p.cout + n0 + tid] = s;
    }
  }
}

template <typename T>
static int launch_conv_t(const mtx_conv2d_args* a, const ConvParams& p, void* stream, int tiles) {
  const unsigned grid = (unsigned)tiles * (unsigned)p.nblk * (unsigned)p.n;
  if (a->ksize == 3 && a->stride == 1) MTX_LAUNCH((conv2d_nhwc_kernel<T, 3, 1>), dim3(grid), dim3(256), 0, stream, p);
  else if (a->ksize == 3 && a->stride == 2) MTX_LAUNCH((conv2d_nhwc_kernel<T, 3, 2>), dim3(grid), dim3(256), 0, stream, p);
  else if (a->ksize == 1 && a->stride == 1) MTX_LAUNCH((conv2d_nhwc_kernel<T, 1, 1>), dim3(grid), dim3(256), 0, stream, p);
  else return MTX_ERR_UNSUPPORTED;
  return MTX_OK;
}

static bool conv_geometry(const mtx_conv2d_args* a, ConvParams& p, int& tiles) {
  if (!((a->ksize == 3 && (a->stride == 1 || a->stride == 2)) || (a->ksize == 1 && a->stride == 1))) return false;
  const int pad = a->ksize / 2;
  const int pad_total = a->pad_mode == 1 ? 1 : 2 * pad;
  p.pad_lo = a->pad_mode == 1 ? 0 : pad;
  p.ho = (a->h + pad_total - a->ksize) / a->stride + 1;
  p.wo = (a->w_in + pad_total - a->ksize) / a->stride + 1;
  const int th = a->stride == 1 ? 16 : 8;
  p.tiles_x = (p.wo + 15) / 16;
  p.tiles_y = (p.ho + th - 1) / th;
  p.nblk = (a->cout + 63) / 64;
  tiles = p.tiles_x * p.tiles_y;
  return true;
}

int conv_c64_tiles(int n, int h, int w);
bool conv_c64_applicable(const mtx_conv2d_args* a);   // (never for pad_mode 1: stride 2)
int conv_c64_launch(const mtx_conv2d_args* a, void* stream, const char** err);

int conv2d_tiles(const mtx_conv2d_args* a) {
  if (conv_c64_applicable(a)) return conv_c64_tiles(a->n, a->h, a->w_in);
  ConvParams p; int tiles = 0;
  if (!conv_geometry(a, p, tiles)) return -1;
  return tiles;
}

int conv2d_launch(const mtx_conv2d_args* a, void* stream, const char** err) {
  ConvParams p;
  int tiles = 0;
  if (a->x == nullptr || a->w == nullptr || a->y == nullptr) { *err = "conv2d: null operand"; return MTX_ERR_INVALID; }
  if (!conv_geometry(a, p, tiles)) { *err = "conv2d: only 3x3 (stride 1/2) and 1x1 (stride 1) are built"; return MTX_ERR_UNSUPPORTED; }
  if (a->cin % 8 || a->cout % 8 || a->ldx % 8 || a->ldy % 8 || (a->res && a->ldres % 8)) {
    *err = "conv2d: channel counts and pixel strides must be multiples of 8 (16-byte chunks)"; return MTX_ERR_INVALID;
  }
  if (a->pixel_shuffle != 0 && (a->pixel_shuffle != 2 || (a->cout / 4) % 8)) { *err = "conv2d: pixel_shuffle must be 2 with Cout/4 % 8 == 0"; return MTX_ERR_INVALID; }
  if (a->n < 1 || a->h < 1 || a->w_in < 1) { *err = "conv2d: empty input"; return MTX_ERR_INVALID; }
  if (conv_c64_applicable(a)) return conv_c64_launch(a, stream, err);
  p.x = (const unsigned char*)a->x; p.w = (const unsigned char*)a->w; p.bias = a->bias;
  p.res = (const unsigned char*)a->res; p.y = (unsigned char*)a->y; p.chan_sum = a->chan_sum;
  p.n = a->n; p.h = a->h; p.w_in = a->w_in; p.cin = a->cin; p.cout = a->cout;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldres = a->ldres;
  p.act = a->act; p.act_param = a->act_param; p.res_scale = a->res_scale; p.act_after = (a->act_after_res && a->res != nullptr) ? 1 : 0; p.ps = a->pixel_shuffle; p.res_bcast = a->res_broadcast_n;
  p.valid_hw = a->stride == 1 ? a->valid_hw : nullptr;
  p.out_scale = a->out_scale;
  int rc;
  if (a->dtype == MTX_BF16) rc = launch_conv_t<__bf16>(a, p, stream, tiles);
  else if (a->dtype == MTX_F16) rc = launch_conv_t<_Float16>(a, p, stream, tiles);
  else { *err = "conv2d: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  if (rc != MTX_OK) *err = "conv2d: unsupported kernel/stride";
  return rc;
}

}  // namespace mtx
