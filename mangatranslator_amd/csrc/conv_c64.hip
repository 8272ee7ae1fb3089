// conv_c64.hip — persistent 3x3 / stride-1 convolution for Cin, Cout <= 64: the RCAN body conv
// (400 of the 404 convolutions of 2x-AnimeSharpV4_RCAN; spandrel model called at reference
// core/image/image_utils.py:369-374) and the narrow YOLO stem layers.
//
// At 64 channels the whole filter bank is 9 x 64 x 64 x 2 B = 72 KiB: it fits in LDS next to one
// input halo tile, so the kernel is PERSISTENT — one 8-wave workgroup per CU loads the filters
// once and then walks 32x16-pixel output tiles (XCD-contiguous tile order, so neighbouring halos
// hit the same L2):
//     LDS:  filters [9][64 co][64 ci] 72 KiB  +  halo 18x34 px x 128 B = 76.5 KiB  (148.5 of 160 KiB)
//     per tile:  prefetched halo registers -> LDS | issue next tile's halo loads (in flight behind
//                the MFMAs) | 9 taps x 2 k-steps x 16 MFMA per wave | tile -> LDS -> 16-byte NHWC
//                stores with bias / activation / residual / pixel-shuffle / channel sums fused.
// Activations cross HBM once in (x1.20 halo overlap, L2-served) and once out; filters never again.
// Same LDS swizzle, MFMA operand swap and epilogue as conv.hip.
#include "mtx_device.h"

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
};

constexpr int C64_TW = 32, C64_TH = 16, C64_NPIX = C64_TW * C64_TH;
constexpr int C64_HW = C64_TW + 2, C64_HH = C64_TH + 2, C64_HPIX = C64_HW * C64_HH;   // 34 x 18 = 612
constexpr int C64_W_BYTES = 9 * 64 * 128;
constexpr int C64_HALO_BYTES = C64_HPIX * 128;
constexpr int C64_SMEM = C64_W_BYTES + C64_HALO_BYTES;
constexpr int C64_NLD = (C64_HPIX * 8 + 511) / 512;     // halo chunks per thread (10)

template <typename T>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(ConvC64Params p) {
  typedef typename Traits<T>::v8 v8;
  typedef typename Traits<T>::v4 v4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C64_SMEM];
  unsigned char* wts = smem;
  unsigned char* halo = smem + C64_W_BYTES;
  unsigned char* outs = halo;                                   // 64 KiB, aliases the halo
  float* red = reinterpret_cast<float*>(halo + C64_NPIX * 128);  // [8 waves][64] after the tile

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const unsigned tiles_per_img = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned total = tiles_per_img * (unsigned)p.n;

  // ---- filters: once per workgroup ---------------------------------------------------------------
  for (int idx = tid; idx < 9 * 64 * 8; idx += 512) {
    const int c = idx & 7, row = idx >> 3;            // row = tap*64 + co
    const int tap = row >> 6, co = row & 63;
    const int ch = c * 8;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (co < p.cout && ch < p.cin)
      v = *reinterpret_cast<const u32x4*>(p.w + (((size_t)co * 9 + tap) * (size_t)p.cin + ch) * sizeof(T));
    *reinterpret_cast<u32x4*>(wts + row * 128 + ((c ^ (co & 7)) << 4)) = v;
  }
  const int nks = p.cin > 32 ? 2 : 1;

  u32x4 pre[C64_NLD];
  auto issue_halo = [&](unsigned t) {
    const unsigned lin = xcd_remap(t, total);
    const int img = (int)(lin / tiles_per_img);
    const int tile = (int)(lin % tiles_per_img);
    const int iy0 = (tile / p.tiles_x) * C64_TH - 1, ix0 = (tile % p.tiles_x) * C64_TW - 1;
    const size_t img_off = (size_t)img * p.h * p.w_in;
#pragma unroll
    for (int it = 0; it < C64_NLD; ++it) {
      const int idx = tid + it * 512;
      const int c = idx & 7, hp = idx >> 3;
      const int hy = hp / C64_HW, hx = hp - hy * C64_HW;
      const int gy = iy0 + hy, gx = ix0 + hx, ch = c * 8;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (hp < C64_HPIX && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_in && ch < p.cin)
        v = *reinterpret_cast<const u32x4*>(p.x + ((img_off + (size_t)gy * p.w_in + gx) * (size_t)p.ldx + ch) * sizeof(T));
      pre[it] = v;
    }
  };

  // channel sums (fused global average pool): accumulated per workgroup across its tiles and
  // flushed once per image -> chan_sum[img][blockIdx.x][C]  (rows = gridDim.x, zeroed by the launcher)
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  int sum_img = -1;
  auto flush_sums = [&](int img_) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = csum[e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      csum[e] = v;
    }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wv * 64 + lane * 8 + e] = csum[e];
    }
    __syncthreads();
    if (tid < 64 && tid < p.cout) {
      float s_ = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s_ += red[k * 64 + tid];
      p.chan_sum[((size_t)img_ * gridDim.x + blockIdx.x) * p.cout + tid] = s_;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  };

  unsigned t = blockIdx.x;
  if (t < total) issue_halo(t);
  for (; t < total; t += gridDim.x) {
    const unsigned lin = xcd_remap(t, total);
    const int img = (int)(lin / tiles_per_img);
    const int tile = (int)(lin % tiles_per_img);
    if (p.chan_sum != nullptr && img != sum_img) {
      if (sum_img >= 0) flush_sums(sum_img);
      sum_img = img;
    }
    const int ty0 = (tile / p.tiles_x) * C64_TH, tx0 = (tile % p.tiles_x) * C64_TW;

    // [A] prefetched halo -> LDS
#pragma unroll
    for (int it = 0; it < C64_NLD; ++it) {
      const int idx = tid + it * 512;
      const int c = idx & 7, hp = idx >> 3;
      if (hp < C64_HPIX) *reinterpret_cast<u32x4*>(halo + hp * 128 + ((c ^ (hp & 7)) << 4)) = pre[it];
    }
    __syncthreads();
    // [B] next tile's halo: global loads stay in flight behind the MFMAs
    if (t + gridDim.x < total) issue_halo(t + gridDim.x);

    // [C] 9 taps x nks k-steps; wave wv owns tile rows 2wv, 2wv+1 (4 fragments of 16 px)
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll 1
      for (int ks = 0; ks < nks; ++ks) {
        const int cch = ks * 4 + q;
        v8 wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = tap * 64 + j * 16 + l15;
          wf[j] = *reinterpret_cast<const v8*>(wts + row * 128 + ((cch ^ (l15 & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = wv * 2 + (i >> 1), x0 = (i & 1) * 16;
          const int lp = (r + ky) * C64_HW + x0 + l15 + kx;
          const v8 xf = *reinterpret_cast<const v8*>(halo + lp * 128 + ((cch ^ (lp & 7)) << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Traits<T>::mfma(wf[j], xf, acc[i][j]);
        }
      }
    }
    __syncthreads();   // [D] every wave is done with the halo

    // [E] bias + activation, 4 consecutive channels per lane -> LDS tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int chunk = j * 2 + (q >> 1);
      float b4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bc = j * 16 + q * 4 + r;
        b4[r] = (p.bias != nullptr && bc < p.cout) ? p.bias[bc] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pt = (wv * 2 + (i >> 1)) * C64_TW + (i & 1) * 16 + l15;
        v4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(apply_act(acc[i][j][r] + b4[r], p.act, p.act_param));
        *reinterpret_cast<v4*>(outs + pt * 128 + ((chunk ^ (pt & 7)) << 4) + ((q & 1) << 3)) = o;
      }
    }
    __syncthreads();   // [F]

    // [G] 16-byte channel chunks out
    const int c = tid & 7;
    const int co = c * 8;
#pragma unroll 2
    for (int idx = tid; idx < C64_NPIX * 8; idx += 512) {
      const int pt = idx >> 3;
      const int oy = ty0 + (pt >> 5), ox = tx0 + (pt & 31);
      if (oy < p.h && ox < p.w_in && co < p.cout) {
        u32x4 raw = *reinterpret_cast<const u32x4*>(outs + pt * 128 + ((c ^ (pt & 7)) << 4));
        size_t opix;
        int oc = co;
        if (p.ps == 2) {
          const int cps = p.cout >> 2;
          const int g = co / cps;
          oc = co - g * cps;
          opix = ((size_t)img * (2 * p.h) + (2 * oy + (g >> 1))) * (size_t)(2 * p.w_in) + (2 * ox + (g & 1));
        } else {
          opix = ((size_t)img * p.h + oy) * (size_t)p.w_in + ox;
        }
        if (p.chan_sum != nullptr || p.res != nullptr) {
          float f[8];
          unpack8<T>(raw, f);
          if (p.chan_sum != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[e] += f[e];
          }
          if (p.res != nullptr) {
            const size_t rpix = p.res_bcast ? opix - (size_t)img * (p.ps == 2 ? 4 : 1) * (size_t)p.h * (size_t)p.w_in : opix;
          const u32x4 rr = *reinterpret_cast<const u32x4*>(p.res + (rpix * (size_t)p.ldres + oc) * sizeof(T));
            float g8[8];
            unpack8<T>(rr, g8);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += p.res_scale * g8[e];
            raw = pack8<T>(f);
          }
        }
        *reinterpret_cast<u32x4*>(p.y + (opix * (size_t)p.ldy + oc) * sizeof(T)) = raw;
      }
    }
    __syncthreads();   // [H] tile buffer is free for the next halo
  }
  if (p.chan_sum != nullptr && sum_img >= 0) flush_sums(sum_img);
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

// rows of the chan_sum buffer per image = workgroups of the persistent launch
int conv_c64_tiles(int n, int h, int w) {
  const long total = (long)((w + C64_TW - 1) / C64_TW) * ((h + C64_TH - 1) / C64_TH) * n;
  const int cus = c64_num_cus(nullptr);
  if (cus < 0) return -1;
  return (int)(total < cus ? total : cus);
}

bool conv_c64_applicable(const mtx_conv2d_args* a) {
  return a->ksize == 3 && a->stride == 1 && a->cin <= 64 && a->cout <= 64;
}

int conv_c64_launch(const mtx_conv2d_args* a, void* stream, const char** err) {
  ConvC64Params p;
  p.x = (const unsigned char*)a->x; p.w = (const unsigned char*)a->w; p.bias = a->bias;
  p.res = (const unsigned char*)a->res; p.y = (unsigned char*)a->y; p.chan_sum = a->chan_sum;
  p.n = a->n; p.h = a->h; p.w_in = a->w_in; p.cin = a->cin; p.cout = a->cout;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldres = a->ldres;
  p.act = a->act; p.act_param = a->act_param; p.res_scale = a->res_scale; p.ps = a->pixel_shuffle; p.res_bcast = a->res_broadcast_n;
  p.tiles_x = (a->w_in + C64_TW - 1) / C64_TW;
  p.tiles_y = (a->h + C64_TH - 1) / C64_TH;
  if (c64_num_cus(err) < 0) return MTX_ERR_HIP;
  const long total = (long)p.tiles_x * p.tiles_y * p.n;
  const unsigned grid = (unsigned)(total < g_num_cus ? total : g_num_cus);
  if (a->chan_sum != nullptr &&
      hipMemsetAsync(a->chan_sum, 0, (size_t)a->n * grid * a->cout * sizeof(float), (hipStream_t)stream) != hipSuccess) {
    *err = "conv2d: chan_sum memset failed"; return MTX_ERR_HIP;
  }
  if (a->dtype == MTX_BF16) MTX_LAUNCH((conv3x3_c64_kernel<__bf16>), dim3(grid), dim3(512), 0, stream, p);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((conv3x3_c64_kernel<_Float16>), dim3(grid), dim3(512), 0, stream, p);
  else { *err = "conv2d: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

}  // namespace mtx
