// detr.hip — the RT-DETR-v2 decoder pieces that are not GEMM / attention / norm (secondary bubble detector,
// reference core/ml/rtdetr_adapter.py:61-113 -> HF RTDetrV2ForObjectDetection).
//
//   multi-scale deformable attention: 300 queries x 8 heads x 12 sampling points, each point a bilinear tap
//   into one of three value maps ([H_l*W_l][heads*d] rows of the projected encoder memory).  Gather-bound:
//   one wave-quarter (16 lanes x 2 channels... ) — here one thread owns (query, head, 8-channel chunk) and walks
//   the points, so the 4 taps of a point are 16-byte loads; the softmax over the points is recomputed per thread
//   (12 values) instead of materialised.
//   reference-box refinement: sigmoid(delta + logit(ref)) in fp32 with the model's clamps.
#include "mtx_device.h"

namespace mtx {

template <typename T>
__global__ __launch_bounds__(256) void deform_attn_kernel(mtx_detr_args p) {
  const int cpd = p.d / 8;                                   // 16-byte chunks per head
  const long total = (long)p.rows * p.heads * cpd;
  const int LP = p.levels * p.points;
  const T* V = reinterpret_cast<const T*>(p.value);
  const T* OFF = reinterpret_cast<const T*>(p.off);
  const T* AW = reinterpret_cast<const T*>(p.aw);
  T* O = reinterpret_cast<T*>(p.out);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ch = (int)(idx % cpd);
    const int h = (int)((idx / cpd) % p.heads);
    const long r = idx / ((long)cpd * p.heads);
    const float cx = p.ref[r * 8 + 0], cy = p.ref[r * 8 + 1], bw = p.ref[r * 8 + 2], bh = p.ref[r * 8 + 3];
    // softmax over this head's L*P logits
    const T* aw = AW + r * p.ld_aw + (long)h * LP;
    float mx = -3.0e38f;
    for (int k = 0; k < LP; ++k) { const float v = to_f32(aw[k]); mx = v > mx ? v : mx; }
    float den = 0.f;
    for (int k = 0; k < LP; ++k) den += __expf(to_f32(aw[k]) - mx);
    const float inv_den = 1.0f / den;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const T* off = OFF + r * p.ld_off + (long)h * LP * 2;
    const float pscale = p.offset_scale / (float)p.points;
    for (int l = 0; l < p.levels; ++l) {
      const int H = p.lh[l], W = p.lw[l];
      const T* Vl = V + (long)p.lstart[l] * p.ld_value + h * p.d + ch * 8;
      for (int pt = 0; pt < p.points; ++pt) {
        const int k = l * p.points + pt;
        const float wgt = __expf(to_f32(aw[k]) - mx) * inv_den;
        // sampling location in [0,1] -> pixel coordinates of grid_sample(align_corners=False)
        const float lx = cx + to_f32(off[k * 2 + 0]) * pscale * bw;
        const float ly = cy + to_f32(off[k * 2 + 1]) * pscale * bh;
        const float px = lx * (float)W - 0.5f, py = ly * (float)H - 0.5f;
        const float fx = floorf(px), fy = floorf(py);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = px - fx, ay = py - fy;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
          if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
          const float wt = wgt * ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
          float g[8];
          unpack8<T>(*reinterpret_cast<const u32x4*>(Vl + ((long)yy * W + xx) * p.ld_value), g);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += wt * g[e];
        }
      }
    }
    *reinterpret_cast<u32x4*>(O + r * p.ld_out + h * p.d + ch * 8) = pack8<T>(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void box_refine_kernel(mtx_detr_args p) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= p.rows) return;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 4; ++c) {
    float x;
    if (p.kind == 2) {
      x = p.ref[r * 8 + c];
    } else {
      float v = p.ref[r * 8 + c];
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      const float x1 = v < 1e-5f ? 1e-5f : v, x2 = (1.f - v) < 1e-5f ? 1e-5f : (1.f - v);
      x = to_f32(reinterpret_cast<const T*>(p.delta)[(long)r * p.ld_delta + c]) + logf(x1 / x2);
    }
    o[c] = 1.0f / (1.0f + expf(-x));
  }
  for (int c = 0; c < 8; ++c) p.ref_out[r * 8 + c] = o[c];
  if (p.ref_t) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.ref_t) + (long)r * 8) = pack8<T>(o);
}

int detr_launch(const mtx_detr_args* a, void* stream, const char** err) {
  if (a->rows < 1) return MTX_OK;
  if (a->dtype != MTX_BF16 && a->dtype != MTX_F16) { *err = "detr: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  if (a->kind == 0) {
    if (!a->value || !a->off || !a->aw || !a->ref || !a->out) { *err = "detr: null operand"; return MTX_ERR_INVALID; }
    if (a->d % 8 || a->levels < 1 || a->levels > 4 || a->points < 1 || a->ld_value % 8 || a->ld_out % 8) { *err = "detr: bad deformable-attention layout"; return MTX_ERR_INVALID; }
    const long total = (long)a->rows * a->heads * (a->d / 8);
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (a->dtype == MTX_BF16) MTX_LAUNCH((deform_attn_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
    else MTX_LAUNCH((deform_attn_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
    return MTX_OK;
  }
  if (a->kind == 1 || a->kind == 2) {
    if (!a->ref || !a->ref_out || (a->kind == 1 && !a->delta)) { *err = "detr: null operand"; return MTX_ERR_INVALID; }
    const dim3 grid((unsigned)((a->rows + 255) / 256));
    if (a->dtype == MTX_BF16) MTX_LAUNCH((box_refine_kernel<__bf16>), grid, dim3(256), 0, stream, *a);
    else MTX_LAUNCH((box_refine_kernel<_Float16>), grid, dim3(256), 0, stream, *a);
    return MTX_OK;
  }
  *err = "detr: unknown kind";
  return MTX_ERR_INVALID;
}

}  // namespace mtx
