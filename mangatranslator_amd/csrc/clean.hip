// clean.hip — pixel half of the OpenCV bubble-cleaning chain on gfx950.
//
// The reference runs ~10 full-page cv2 passes per bubble on the CPU (core/image/cleaning.py:296-337).  Here all
// bubbles of a page go through five byte kernels restricted to their own crops: HBM-bound integer work,
// one thread per crop pixel, coalesced along x; the page and masks are read in place (no copies).
#include "mtx_device.h"

namespace mtx {

constexpr int CL_INIT = 0x1fffffff;       // "far" (INT_MAX >> 2, like the reference's temporary)
constexpr int CL_HV = 65536, CL_DIAG = 91750, CL_LONG = 143976;

__device__ __forceinline__ int gray_of(const unsigned char* bgr) {
  return ((int)bgr[0] * 1868 + (int)bgr[1] * 9617 + (int)bgr[2] * 4899 + (1 << 13)) >> 14;   // cv2 BGR2GRAY (8-bit path)
}

// crop pixel -> (bubble, x, y); returns false past the end of the crop
__device__ __forceinline__ bool crop_pixel(const mtx_clean_args& a, int& bub, int& x, int& y, int& w, int& h, int& X, int& Y, long& o) {
  bub = blockIdx.y;
  const int* r = a.rois + bub * 4;
  w = r[2]; h = r[3];
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)w * h) return false;
  x = (int)(idx % w); y = (int)(idx / w);
  X = r[0] + x; Y = r[1] + y;
  o = a.offsets[bub] + idx;
  return true;
}

// base crop, dilation, erosion, grey statistics
__global__ __launch_bounds__(256) void clean_morph_kernel(mtx_clean_args a) {
  __shared__ int s_sum, s_cnt;
  if (threadIdx.x == 0) { s_sum = 0; s_cnt = 0; }
  __syncthreads();
  int bub, x, y, w, h, X, Y; long o;
  const bool live = crop_pixel(a, bub, x, y, w, h, X, Y, o);
  if (live) {
    const unsigned char* M = reinterpret_cast<const unsigned char*>(a.masks) + (size_t)bub * a.page_h * a.page_w;
    const bool base = M[(size_t)Y * a.page_w + X] != 0;
    bool dil = false, ero = true;
    for (int dy = -a.dil_r; dy <= a.dil_r && !dil; ++dy) {
      const int yy = Y + dy;
      if (yy < 0 || yy >= a.page_h) continue;
      const int dx = a.dil_dx[dy + a.dil_r];
      for (int xx = X - dx; xx <= X + dx; ++xx)
        if (xx >= 0 && xx < a.page_w && M[(size_t)yy * a.page_w + xx] != 0) { dil = true; break; }
    }
    for (int dy = -a.ero_r; dy <= a.ero_r && ero; ++dy) {
      const int yy = Y + dy;
      if (yy < 0 || yy >= a.page_h) continue;          // outside the page never lowers the minimum
      const int dx = a.ero_dx[dy + a.ero_r];
      for (int xx = X - dx; xx <= X + dx; ++xx)
        if (xx >= 0 && xx < a.page_w && M[(size_t)yy * a.page_w + xx] == 0) { ero = false; break; }
    }
    reinterpret_cast<unsigned char*>(a.base)[o] = base ? 255 : 0;
    reinterpret_cast<unsigned char*>(a.roi)[o] = dil ? 255 : 0;
    reinterpret_cast<unsigned char*>(a.eroded)[o] = ero ? 255 : 0;
    if (base) {
      const int g = gray_of(reinterpret_cast<const unsigned char*>(a.page_bgr) + ((size_t)Y * a.page_w + X) * 3);
      atomicAdd(&s_sum, g); atomicAdd(&s_cnt, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) { atomicAdd(a.stats + blockIdx.y * 260 + 256, s_sum); atomicAdd(a.stats + blockIdx.y * 260 + 257, s_cnt); }
}

// is_black from the mean grey under the base mask
__global__ void clean_polarity_kernel(mtx_clean_args a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int* st = a.stats + i * 260;
  st[258] = (st[257] > 0 && (long)st[256] < 128L * st[257]) ? 1 : 0;      // mean < 128
}

// histogram of the thresholding image (grey, inverted for dark bubbles) over the dilated ROI — Otsu's input
__global__ __launch_bounds__(256) void clean_hist_kernel(mtx_clean_args a) {
  __shared__ int hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  int bub, x, y, w, h, X, Y; long o;
  if (crop_pixel(a, bub, x, y, w, h, X, Y, o) && reinterpret_cast<const unsigned char*>(a.roi)[o]) {
    int g = gray_of(reinterpret_cast<const unsigned char*>(a.page_bgr) + ((size_t)Y * a.page_w + X) * 3);
    if (a.stats[bub * 260 + 258]) g = 255 - g;
    atomicAdd(&hist[g], 1);
  }
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(a.stats + blockIdx.y * 260 + threadIdx.x, hist[threadIdx.x]);
}

// threshold per bubble: fixed, or Otsu's between-class variance maximum over the ROI histogram
__global__ void clean_threshold_value_kernel(mtx_clean_args a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int* st = a.stats + i * 260;
  if (!a.use_otsu) { st[259] = a.threshold; return; }
  double n = 0.0, mu = 0.0;
  for (int k = 0; k < 256; ++k) { n += st[k]; mu += (double)k * st[k]; }
  if (n <= 0.0) { st[259] = 0; return; }
  const double scale = 1.0 / n;
  mu *= scale;
  double q1 = 0.0, mu1 = 0.0, best = 0.0; int best_k = 0;
  for (int k = 0; k < 256; ++k) {
    const double p = st[k] * scale;
    mu1 *= q1; q1 += p;
    const double q2 = 1.0 - q1;
    const double lo = q1 < q2 ? q1 : q2, hi = q1 < q2 ? q2 : q1;
    if (lo < 1.1920929e-07 || hi > 1.0 - 1.1920929e-07) continue;
    mu1 = (mu1 + k * p) / q1;
    const double mu2 = (mu - q1 * mu1) / q2;
    const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
    if (sigma > best) { best = sigma; best_k = k; }
  }
  st[259] = best_k;
}

// text candidates (ROI pixels brighter than the threshold) and the chamfer seed
__global__ __launch_bounds__(256) void clean_threshold_kernel(mtx_clean_args a) {
  int bub, x, y, w, h, X, Y; long o;
  if (!crop_pixel(a, bub, x, y, w, h, X, Y, o)) return;
  const bool roi = reinterpret_cast<const unsigned char*>(a.roi)[o] != 0;
  int g = gray_of(reinterpret_cast<const unsigned char*>(a.page_bgr) + ((size_t)Y * a.page_w + X) * 3);
  const int* st = a.stats + bub * 260;
  if (st[258]) g = 255 - g;
  reinterpret_cast<unsigned char*>(a.thresholded)[o] = (roi && g > st[259]) ? 255 : 0;
  a.dist_a[o] = roi ? CL_INIT : 0;
}

// one relaxation sweep of the 5x5 chamfer metric (Jacobi form of the reference's two raster passes: both
// converge to the same shortest-path distance; k sweeps make every distance < k exact)
__global__ __launch_bounds__(256) void clean_chamfer_kernel(mtx_clean_args a, const int* __restrict__ src, int* __restrict__ dst) {
  int bub, x, y, w, h, X, Y; long o;
  if (!crop_pixel(a, bub, x, y, w, h, X, Y, o)) return;
  int d = src[o];
  if (d > 0) {
    const long base = a.offsets[bub];
    auto at = [&](int dx, int dy) -> int {
      const int xx = x + dx, yy = y + dy;
      if (X + dx < 0 || X + dx >= a.page_w || Y + dy < 0 || Y + dy >= a.page_h) return CL_INIT;     // outside the page: far
      if (xx < 0 || xx >= w || yy < 0 || yy >= h) return 0;                                        // outside the crop: background
      return src[base + (long)yy * w + xx];
    };
    int m = d;
#define CL_RELAX(dx, dy, wt) { const int v = at(dx, dy) + wt; m = v < m ? v : m; }
    CL_RELAX(-1, 0, CL_HV) CL_RELAX(1, 0, CL_HV) CL_RELAX(0, -1, CL_HV) CL_RELAX(0, 1, CL_HV)
    CL_RELAX(-1, -1, CL_DIAG) CL_RELAX(1, -1, CL_DIAG) CL_RELAX(-1, 1, CL_DIAG) CL_RELAX(1, 1, CL_DIAG)
    CL_RELAX(-1, -2, CL_LONG) CL_RELAX(1, -2, CL_LONG) CL_RELAX(-2, -1, CL_LONG) CL_RELAX(2, -1, CL_LONG)
    CL_RELAX(-2, 1, CL_LONG) CL_RELAX(2, 1, CL_LONG) CL_RELAX(-1, 2, CL_LONG) CL_RELAX(1, 2, CL_LONG)
#undef CL_RELAX
    d = m;
  }
  dst[o] = d;
}

// shrunk ROI (distance >= shrink, relaxed to the junction radius inside junction zones) and the final AND
__global__ __launch_bounds__(256) void clean_finalize_kernel(mtx_clean_args a, const int* __restrict__ dist) {
  int bub, x, y, w, h, X, Y; long o;
  if (!crop_pixel(a, bub, x, y, w, h, X, Y, o)) return;
  const int d = dist[o];
  bool keep = d >= a.shrink_fixed;
  if (!keep && a.zones != nullptr && d >= a.junction_fixed) {
    for (int z = 0; z < a.max_zones; ++z) {
      const int* q = a.zones + ((size_t)bub * a.max_zones + z) * 4;
      if (q[2] > q[0] && X >= q[0] && X < q[2] && Y >= q[1] && Y < q[3]) { keep = true; break; }
    }
  }
  reinterpret_cast<unsigned char*>(a.shrunk)[o] = keep ? 255 : 0;
  if (!keep) reinterpret_cast<unsigned char*>(a.thresholded)[o] = 0;
}

int clean_launch(const mtx_clean_args* a, void* stream, const char** err) {
  if (!a->page_bgr || !a->masks || !a->rois || !a->offsets || !a->base || !a->roi || !a->eroded || !a->thresholded || !a->shrunk ||
      !a->dist_a || !a->dist_b || !a->stats) { *err = "bubble_clean: null operand"; return MTX_ERR_INVALID; }
  if (a->n < 1 || a->max_pixels < 1) return MTX_OK;
  if (a->dil_r < 0 || a->dil_r > 31 || a->ero_r < 0 || a->ero_r > 31) { *err = "bubble_clean: structuring elements up to 63 x 63"; return MTX_ERR_INVALID; }
  if (a->sweeps < 0 || a->sweeps > 4096) { *err = "bubble_clean: bad sweep count"; return MTX_ERR_INVALID; }
#ifdef MTX_EMU
  memset(a->stats, 0, (size_t)a->n * 260 * sizeof(int));
#else
  zero_words_async(a->stats, (size_t)a->n * 260 * sizeof(int), stream);        // a kernel, not a memset node (mtx_device.h)
#endif
  const dim3 grid((unsigned)((a->max_pixels + 255) / 256), (unsigned)a->n), small((unsigned)((a->n + 63) / 64));
  MTX_LAUNCH(clean_morph_kernel, grid, dim3(256), 0, stream, *a);
  MTX_LAUNCH(clean_polarity_kernel, small, dim3(64), 0, stream, *a);
  MTX_LAUNCH(clean_hist_kernel, grid, dim3(256), 0, stream, *a);
  MTX_LAUNCH(clean_threshold_value_kernel, small, dim3(64), 0, stream, *a);
  MTX_LAUNCH(clean_threshold_kernel, grid, dim3(256), 0, stream, *a);
  int* src = a->dist_a; int* dst = a->dist_b;
  for (int s = 0; s < a->sweeps; ++s) {
    MTX_LAUNCH(clean_chamfer_kernel, grid, dim3(256), 0, stream, *a, (const int*)src, dst);
    int* t = src; src = dst; dst = t;
  }
  MTX_LAUNCH(clean_finalize_kernel, grid, dim3(256), 0, stream, *a, (const int*)src);
  return MTX_OK;
}

}  // namespace mtx
