// quant.hip — MX block quantisation to OCP fp8 e4m3 (include/mtx_hip.h mtx_quant_args): the operand format of the
// fp8 GEMM path (gemm.hip gemm256_f8_kernel) behind FLUX.2-Klein's linears — the reference's low-precision route
// for this model is SDNQ packed weights + a quantised matmul (core/ml/model_manager.py:1296-1312).
//
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
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float a = fabsf(f[e]); amax = a > amax ? a : amax; }
    { float o = __shfl_xor(amax, 1, 64); amax = o > amax ? o : amax; }
    { float o = __shfl_xor(amax, 2, 64); amax = o > amax ? o : amax; }
    // smallest power of two 2^(eb - 127) >= amax / 448
    const float r = amax * (1.0f / 448.0f);
    const unsigned u = __builtin_bit_cast(unsigned, r);
    int eb = (int)((u >> 23) & 0xff) + ((u & 0x7fffffu) ? 1 : 0);
    eb = amax == 0.f ? 127 : (eb < 1 ? 1 : (eb > 253 ? 253 : eb));
    const float inv = __builtin_bit_cast(float, (unsigned)(254 - eb) << 23);      // 2^(127 - eb)
#pragma unroll
    for (int e = 0; e < 8; ++e) { float v = f[e] * inv; v = v > 448.f ? 448.f : (v < -448.f ? -448.f : v); f[e] = v; }
    unsigned w0 = 0, w1 = 0;
    w0 = cvt_pk_fp8<false>(f[0], f[1], w0); w0 = cvt_pk_fp8<true>(f[2], f[3], w0);
    w1 = cvt_pk_fp8<false>(f[4], f[5], w1); w1 = cvt_pk_fp8<true>(f[6], f[7], w1);
    unsigned word = (unsigned)eb << (8 * (int)((c8 & 15) >> 2));
    word |= __shfl_xor(word, 4, 64);
    word |= __shfl_xor(word, 8, 64);
    if (valid) {
      *reinterpret_cast<u32x2*>(Q + row * p.ldq + c8 * 8) = u32x2{w0, w1};
      if ((c8 & 15) == 0) S[(c8 >> 4) * p.lds + row] = word;
    }
  }
}

int quant_launch(const mtx_quant_args* a, void* stream, const char** err) {
  if (!a->x || !a->q || !a->scale) { *err = "quantize_mx: null operand"; return MTX_ERR_INVALID; }
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
