// quant.hip — MX block quantisation to OCP fp8 e4m3 (include/mtx_hip.h mtx_quant_args): the operand format of the
// fp8 GEMM path (gemm.hip gemm256_f8_kernel) behind FLUX.2-Klein's linears — the reference's low-precision route
// for this model is SDNQ packed weights + a quantised matmul (core/ml/model_manager.py:1296-1312).
//
// op MTX_QUANT_SWIGLU fuses the producer: q = quant(silu(x) * b) in one pass (the 16-bit SwiGLU result is written only when a 16-bit
// consumer exists) — the separate SwiGLU pass wrote 2 B per element that the quantiser read straight back.
// HBM-bound: 2 B in + 1 B + 1/32 B out per element.  A lane owns 8 consecutive k (one 16-byte load); 4 adjacent lanes
// form an MX block of 32 (amax by two xor-shuffles), 16 adjacent lanes one uint32 of four scale bytes.
#include "mtx_device.h"

namespace mtx {

template <typename T>
__global__ __launch_bounds__(256) void quant_mx_kernel(mtx_quant_args p) {
  const long K8 = p.k / 8;
  const long total = p.rows * K8;            // a multiple of 16: lanes 16 g .. 16 g + 15 are valid together
  const T* X = reinterpret_cast<const T*>(p.x);
  unsigned char* Q = reinterpret_cast<unsigned char*>(p.q);
  unsigned* S = reinterpret_cast<unsigned*>(p.scale);
  for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {      // wave-uniform trip count (shuffles inside)
    const long idx = base + threadIdx.x;
    const bool valid = idx < total;
    const long row = valid ? idx / K8 : 0, c8 = valid ? idx % K8 : 0;
    float f[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(X + row * p.ldx + c8 * 8), f);
    if (p.op == MTX_QUANT_SWIGLU) {          // x := silu(x) * b, rounded to T like the stand-alone SwiGLU pass (MTX_EW_SWIGLU) before it is quantised
      float g[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.b) + row * p.ldb + c8 * 8), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = to_f32(from_f32<T>(div_by_1p(f[e], 1.f + __expf(-f[e])) * g[e]));
      if (p.y != nullptr && valid) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.y) + row * p.ldy + c8 * 8) = pack8<T>(f);
    }
    unsigned w0, w1, word;
    mx_quantize_chunk(f, c8, w0, w1, word);
    if (valid) {
      *reinterpret_cast<u32x2*>(Q + row * p.ldq + c8 * 8) = u32x2{w0, w1};
      if ((c8 & 15) == 0) S[(c8 >> 4) * p.lds + row] = word;
    }
  }
}

int quant_launch(const mtx_quant_args* a, void* stream, const char** err) {
  if (!a->x || !a->q || !a->scale) { *err = "quantize_mx: null operand"; return MTX_ERR_INVALID; }
  if (a->op != MTX_QUANT_PLAIN && a->op != MTX_QUANT_SWIGLU) { *err = "quantize_mx: unknown op"; return MTX_ERR_INVALID; }
  if (a->op == MTX_QUANT_SWIGLU && (!a->b || a->ldb % 8 || (a->y && a->ldy % 8))) { *err = "quantize_mx (SwiGLU): needs b with ldb % 8 == 0 (and ldy % 8 == 0 with y)"; return MTX_ERR_INVALID; }
  if (a->rows < 1 || a->k < 128 || a->k % 128 || a->ldx % 8 || a->ldq % 8 || a->lds < a->rows) { *err = "quantize_mx: needs K % 128 == 0, ldx / ldq % 8 == 0, lds >= rows"; return MTX_ERR_INVALID; }
  const long total = a->rows * (a->k / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (a->dtype == MTX_BF16) MTX_LAUNCH((quant_mx_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else if (a->dtype == MTX_F16) MTX_LAUNCH((quant_mx_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  else { *err = "quantize_mx: dtype must be bf16 or f16"; return MTX_ERR_INVALID; }
  return MTX_OK;
}

}  // namespace mtx
