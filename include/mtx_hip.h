/*
 * mtx_hip.h — C ABI of libmtx_hip.so, the MI355X (gfx950) kernel library under the
 * MangaTranslator vision hot path (YOLO detect -> SAM-2.1 masks -> FLUX inpaint -> RCAN upscale).
 *
 * The reference (meangrinch/MangaTranslator) is pure Python and has no FFI of its own; the
 * arithmetic this library replaces lives in third-party wheels called from:
 *   - upscaler   model(tensor)                core/image/image_utils.py:369-374
 *   - YOLO       model(img, conf=, imgsz=..)  core/image/detection.py:1337-1345
 *   - SAM-2.1    sam_model(**inputs)          core/image/detection.py:494-509
 *   - FLUX       pipeline(**kw).images[0]     core/image/inpainting.py:877-887, 1577-1589
 * Each network is a static graph ("plan") of the ops below, built once per input shape by the
 * Python host (mangatranslator_amd/hip/plan.py) and executed here with no Python in the loop.
 *
 * Conventions:
 *   - every entry point returns 0 on success, a negative mtx_status otherwise; nothing throws
 *     across the ABI; mtx_last_error() gives a thread-local message.
 *   - all pointers are DEVICE pointers owned by the caller (torch tensor .data_ptr());
 *     the library never allocates activation memory and never synchronises a stream.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - activations are channels-last: [N, H, W, C] with an explicit per-pixel stride in
 *     elements ("ld"), so a channel slice of a wider buffer is a valid operand (concat is free).
 *   - 16-bit element types: MTX_BF16 / MTX_F16; accumulation is always fp32.
 */
#ifndef MTX_HIP_H
#define MTX_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MTX_API __attribute__((visibility("default")))
#else
#define MTX_API
#endif

#define MTX_ABI_VERSION 8

typedef enum mtx_status {
  MTX_OK = 0,
  MTX_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  MTX_ERR_HIP = -2,       /* a HIP runtime call failed */
  MTX_ERR_UNSUPPORTED = -3,
  MTX_ERR_STATE = -4
} mtx_status;

/* MTX_F32 as the `dtype` of mtx_gemm / mtx_attention / mtx_norm / mtx_elementwise (late round 4): every operand and the result are fp32 and
 * the arithmetic runs on the vector ALUs (csrc/f32ops.hip) — GEMM with bias / act / res / strided batches (no gate, glu or fp8 operands),
 * attention with head dim 8 / 16 / 32 / 64, LayerNorm (+ act), element-wise ADD / MUL / ACT / COPY / ROW_GATHER / SHUFFLE2_ADD / CVT_F32 /
 * CVT_16 (fp32 -> 16-bit, the bridge back into a 16-bit GEMM).
 * Small problems only: SAM-2.1's mask decoder under Sam2Hip(precision="high") (hardware-verified in round 5: profiles/r05_parity.json). */
typedef enum mtx_dtype { MTX_BF16 = 0, MTX_F16 = 1, MTX_F32 = 2, MTX_U8 = 3, MTX_I32 = 4,
                         MTX_F8 = 5 /* OCP e4m3fn bytes with MX block scales, see mtx_quant_args */ } mtx_dtype;

typedef enum mtx_act {
  MTX_ACT_NONE = 0, MTX_ACT_RELU = 1, MTX_ACT_SILU = 2, MTX_ACT_GELU = 3 /* erf */,
  MTX_ACT_GELU_TANH = 4, MTX_ACT_SIGMOID = 5, MTX_ACT_LEAKY = 6 /* slope = act_param */
} mtx_act;

/* ---- op argument blocks (plain data; mirrored 1:1 by ctypes in hip/abi.py) ------------- */

/* 2-D convolution, NHWC, implicit GEMM on MFMA, LDS halo tile.  ksize 1 or 3, stride 1 or 2,
 * pad = ksize/2.  w is packed [Cout][ksize*ksize][Cin] (tap-major, channel-contiguous).
 *   y = act(conv(x) + bias) [+ res_scale * res]
 * pixel_shuffle = r (0 or 2): output channel (dy*r+dx)*Cout/r^2 + c is stored at
 *   y[n, oy*r+dy, ox*r+dx, c]  (the caller permutes torch-order channels c*r^2+dy*r+dx at pack
 *   time) — ldy/ldres then describe the shuffled tensor.
 * chan_sum (optional, fp32 [N][tiles][Cout]): per-tile sums of act(conv(x) + bias) as rounded to the storage type — BEFORE out_scale
 *   and before the residual — for a fused global average pool; tiles = mtx_conv2d_tiles().  Both conv kernels (the 64 -> 64
 *   persistent one and the generic one) sum this same quantity.  With out_scale the 64 -> 64 kernel computes
 *   out_scale * act(..) + res_scale * res in fp32 and rounds once; the generic kernel rounds act(..) to the storage type first.   */
typedef struct mtx_conv2d_args {
  const void* x; const void* w; const float* bias; const void* res; void* y; float* chan_sum;
  int32_t n, h, w_in, cin, cout;
  int32_t ksize, stride;
  int32_t ldx, ldy, ldres;       /* per-pixel strides in elements */
  int32_t act; float act_param; float res_scale;
  int32_t pixel_shuffle;
  int32_t dtype;
  int32_t res_broadcast_n;       /* 1: res has batch 1 and is shared by all n images */
  int32_t pad_mode;              /* 0: pad k/2 on every side; 1: no pad top/left, 1 bottom/right
                                    (diffusers Downsample2D: F.pad(0,1,0,1) + conv k3 s2 p0)     */
  int32_t act_after_res;     /* 1: y = act(conv(x) + bias + res_scale * res)  (ResNet bottleneck), 0: act before the residual */
  /* optional DEVICE pointer to {valid_h, valid_w} (stride-1 convs): the image occupies only the top-left valid_h x valid_w pixels of
   * the h x w_in canvas; every output pixel at or beyond it is written as ZERO (and left out of chan_sum), so the next layer sees
   * exactly the zero padding an image of that size would have.  One plan built on a bucket size then serves every smaller image
   * (bubble crops of arbitrary size, reference core/services/translation.py:2097-2258) by rewriting two integers. */
  const int32_t* valid_hw;
  /* optional DEVICE [n][cout] per-channel factor on the conv result, applied after the activation and before the residual:
   * y = out_scale * act(conv(x) + bias) + res_scale * res.  RCAN's second RCAB conv writes x + s * conv2(t) with it (the channel
   * attention s is known before the conv runs, see mtx_ca_args.t), so no separate scale-and-add pass reads the activations again. */
  const float* out_scale;
} mtx_conv2d_args;

/* C[M,N] = epilogue(A[M,K] * W[N,K]^T)   A row stride lda, C row stride ldc (elements).
 *   v = acc + bias[n];  v = act(v);  v = v * (gate ? gate[gate_row(m), n] : 1) ;
 *   v += res ? res[m, n] : 0
 * gate rows: gate_row(m) = m / gate_rows_per (broadcast a per-sample modulation vector).
 * batch > 1: strided batched GEMM (a/w/c advance by *_bstride elements).                    */
typedef struct mtx_gemm_args {
  const void* a; const void* w; const float* bias; const void* res; const void* gate; void* c;
  int64_t m, n, k;
  int64_t lda, ldw, ldc, ldres, ldgate;
  int64_t batch, a_bstride, w_bstride, c_bstride;
  int64_t res_bstride;           /* batch stride of res (0 = the same res for every batch) */
  int32_t gate_rows_per;
  int32_t act; float act_param; float alpha;   /* acc *= alpha before bias */
  int32_t dtype; int32_t out_dtype;            /* out_dtype: MTX_BF16/F16 (=dtype) or MTX_F32 */
  /* optional scratch for the 256-tile kernel's K-slice tail (fp32 partial tiles + one ticket per tile in its last 4 KiB); NULL = never
   * split.  MTX_GEMM_WORKSPACE_BYTES exactly; ZEROED ONCE by the caller — the library leaves the tickets at zero after every launch. */
  void* workspace; int64_t workspace_bytes;
  /* in_dtype == MTX_F8 (the CDNA4 fp8 path, BASELINE.json config 5): a and w are e4m3 bytes ([M, K] / [N, K], lda / ldw in
   * elements = bytes) with MX scale planes as written by mtx_quantize_mx: a_scale[(k / 128) * lds_a + m], w_scale[(k / 128) *
   * lds_w + n].  `dtype` stays the 16-bit type of bias-free epilogue operands (gate, res) and of c.  K % 128 == 0, lda / ldw % 16 == 0.
   * in_dtype == 0 (or == dtype): the 16-bit path above. */
  const void* a_scale; const void* w_scale; int64_t lds_a, lds_w; int32_t in_dtype;
  int32_t flags;                 /* MTX_GEMM_* bits (tests / kernel benches; 0 in production graphs) */
  /* SwiGLU + MX-fp8 epilogue (in_dtype == MTX_F8 only; FLUX.2's gated MLP, reference diffusers Flux2FeedForward / Flux2SwiGLU behind
   * core/image/inpainting.py:1577-1589): the GEMM columns from glu_col0 on are 64-column spans [32 x a | 32 x b] — the caller has
   * permuted W's rows so — and leave the kernel as the MX e4m3 operand of the next linear instead of 16-bit values:
   *   h = round_T(silu(round_T(a)) * round_T(b))  (what MTX_QUANT_SWIGLU computes from the 16-bit projection),  span u -> the 32
   *   bytes glu_q[m * glu_ldq + 32 u ..] and byte (u & 3) of the scale word glu_scale[(u >> 2) * glu_lds + m].
   * Columns below glu_col0 take the usual epilogue into c.  glu_col0 % 256 == 0, (n - glu_col0) % 256 == 0, no bias / gate / res /
   * act on the launch, glu_q 16-byte aligned with glu_ldq % 16 == 0.  NULL glu_q = off.  (Hardware-verified in round 4 — tests/test_ops_gpu.py::test_gemm_f8_glu_epilogue —
   * and the default of the FLUX.2 graphs since.) */
  void* glu_q; void* glu_scale; int64_t glu_ldq, glu_lds, glu_col0;
  /* ABI 8.  w_lo: the weights are a PAIR W = w + w_lo of the storage type (same ldw / w_bstride) — both halves are multiplied with the same A
   * tile and add into one fp32 accumulator (SAM-2.1 precision "high": the weights' rounding to 16 bits goes away at twice the matrix
   * work, with no [x | x] operand and no copy pass; 16-bit operands, 128-tile kernel).  NULL = off.
   * res_dtype: 0 / dtype = res has the storage type; MTX_F32 (with out_dtype MTX_F32 only) = res is fp32 [.., ldres] — an fp32 residual
   * stream is added inside the GEMM that closes a branch. */
  const void* w_lo; int32_t res_dtype;
} mtx_gemm_args;
#define MTX_GEMM_FORCE_TILE256 1   /* use the 256-tile LDS-DMA kernel whatever the tile count (small-shape tests of that kernel) */
#define MTX_GEMM_NO_SPLIT 2        /* never hand left-over tiles to the K-slice tail */
#define MTX_GEMM_SLICES(n) ((n) << 8) /* tuning: cut the left-over tiles into exactly n K slices (2..255) instead of the launcher's choice */
#define MTX_GEMM_WORKSPACE_BYTES (2 * 320 * 256 * 256 * 4)

/* softmax(scale * Q K^T) V, non-causal, one launch for [batch, heads].
 * q: [batch, sq, heads, d] with strides; k,v: [batch, sk, heads, d]; o like q.
 * Strides in elements: *_bs (batch), *_ss (sequence), *_hs (head). d <= 128, d % 8 == 0. */
typedef struct mtx_attn_args {
  const void* q; const void* k; const void* v; void* o;
  int64_t batch, heads, sq, sk, d;
  int64_t q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
  float scale;
  int32_t dtype;
  /* optional scratch for the long-sequence kernel: when the workgroup count leaves a partial last wave on the
   * chip, the left-over query blocks are split over key ranges and merged from fp32 partials kept here
   * (MTX_ATTN_WORKSPACE_BYTES is always enough); NULL = never split */
  void* workspace; int64_t workspace_bytes;
  int32_t flags;                 /* MTX_ATTN_* bits */
  /* MX fp8 output (long-sequence kernel only: d = 128, sq >= 1024, sk >= 256, batch 1): the rows leave as the e4m3 operand of the next
   * linear — q8[row * ldq8 + head * 128 + ..] and one scale word per head and row, q8_scale[head * lds_q8 + row] — exactly what
   * mtx_quantize_mx makes of the 16-bit output (rounded to `dtype` first); `o` may then be NULL.  (Hardware-verified in round 4 —
   * tests/test_ops_gpu.py::test_attention_mx_fp8_output — and the default of the FLUX.2 graphs since.) */
  void* q8; void* q8_scale; int64_t ldq8, lds_q8;
  /* fp8 scores (ABI 8; long-sequence kernel with MTX_ATTN_Q_PRESCALED only, d = 128, batch 1): plain e4m3 copies of the query and key rows —
   * q_f8[row * qf8_ss + head * 128 + ..], k_f8[row * kf8_ss + head * 128 + ..], strides in BYTES, 16-byte aligned — as the rotary kernel
   * writes them (mtx_ew_args.y8).  The base-2 logits are 2^qk_f8_exp * sum_d q_f8 k_f8 on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the
   * 16-bit rate); q and k are then not read.  P V stays 16-bit.  NULL = 16-bit scores. */
  const void* q_f8; const void* k_f8; int64_t qf8_ss, kf8_ss; int32_t qk_f8_exp;
  /* fp8 P V (ABI 8; only together with q_f8 / k_f8 and the MX fp8 output q8): v_f8t = the values as MTX_EW_V_F8T writes them — e4m3 [head][128][vf8_ld]
   * — and the probabilities of a tile are rounded to e4m3 as well: O^T += V^T P on the fp8 matrix instruction, one MFMA of K = 64 per 32 x 32
   * block instead of four of K = 16; `v` is then not read.  NULL = 16-bit P V. */
  const void* v_f8t; int64_t vf8_ld;
} mtx_attn_args;
/* q already carries scale * log2(e) (folded into the producer, e.g. the pre-scaled rotary table of MTX_EW_QK_NORM_ROPE): `scale`
 * is ignored, scores are used as base-2 logits as they come out of the matrix pipe.  The long-sequence kernel then seeds its score
 * accumulators with minus the running maximum and needs no per-score multiply-add. */
#define MTX_ATTN_Q_PRESCALED 1
#define MTX_ATTN_WORKSPACE_BYTES (256 * (256 * 128 * 4 + 256 * 2 * 4))

/* row-wise normalisation over the last dim C of [rows, C] (row stride ld):
 *  kind 0 LayerNorm (gamma,beta optional), 1 RMSNorm (gamma optional).
 *  optional adaLN modulation: y = norm(x) * (1 + scale[r/rows_per, :]) + shift[r/rows_per, :] */
typedef struct mtx_norm_args {
  const void* x; void* y; const float* gamma; const float* beta;
  const void* mod_scale; const void* mod_shift;
  int64_t rows, c, ldx, ldy, rows_per, ldmod;
  float eps; int32_t kind; int32_t dtype;
  int32_t act;                   /* activation applied last (Sam2 upscaler: LayerNorm -> GELU) */
  /* optional MX fp8 twin of the result (the operand format of the fp8 GEMM, see mtx_quant_args): q [rows][ldq] e4m3 bytes and the
   * E8M0 scale plane q_scale[(c / 128) * lds_q + row], quantised from the result as rounded to `dtype` — bit-identical to running
   * mtx_quantize_mx on y.  With q given, y may be NULL (no 16-bit consumer: the 16-bit store is skipped).  C % 128 == 0. */
  void* q; void* q_scale; int64_t ldq, lds_q;
  /* ABI 8.  dtype == MTX_F32 only (ignored otherwise): the type y is written in — MTX_F32, or MTX_BF16 / MTX_F16 = the 16-bit operand of
   * the linear that follows (an fp32 residual stream is normalised in fp32 and rounded once, by the kernel that produced the value).
   * MTX_BF16 is 0: a caller of the fp32 norm must SET this field (MTX_F32 for the round-4 behaviour). */
  int32_t out_dtype;
} mtx_norm_args;

/* GroupNorm over NHWC [N, HW, C] with G groups, optional fused SiLU.  `workspace`: MTX_GROUPNORM_WS_FLOATS(n, hw, c, groups) floats
 * ([N][G] {mean, rstd} + per-block per-channel partial sums, added in a fixed order: identical calls give identical bytes); its content on
 * entry does not matter (ABI 6; up to ABI 5: N*C*2 + N*G*2 floats accumulated by atomics). */
#define MTX_GN_PIX_PER_BLOCK 1024
#define MTX_GROUPNORM_WS_FLOATS(n, hw, c, g) ((n) * (g) * 2 + (n) * (((hw) + MTX_GN_PIX_PER_BLOCK - 1) / MTX_GN_PIX_PER_BLOCK) * (c) * 2)
typedef struct mtx_groupnorm_args {
  const void* x; void* y; const float* gamma; const float* beta; float* workspace;
  int64_t n, hw, c, groups; float eps; int32_t act; int32_t dtype;
} mtx_groupnorm_args;

/* generic element-wise / data-movement ops on NHWC tensors (kind = mtx_ew_kind). */
typedef enum mtx_ew_kind {
  MTX_EW_SCALE_RES = 0,   /* y = a * s[n, c] + b        (RCAN channel attention + skip)      */
  MTX_EW_ADD = 1,         /* y = a + b                                                        */
  MTX_EW_MUL = 2,         /* y = a * b                                                        */
  MTX_EW_ACT = 3,         /* y = act(a)                                                       */
  MTX_EW_UPSAMPLE2X = 4,  /* nearest 2x: y[n, 2h+i, 2w+j, c] = a[n, h, w, c] (+ b if given)  */
  MTX_EW_MAXPOOL = 5,     /* k x k, stride s (i0 = k, i1 = s), pad k/2 for odd k, 0 for even k */
  MTX_EW_COPY = 6,        /* strided channel-slice copy                                       */
  MTX_EW_GATE_RES = 7,    /* y = b + a * g[row / rows_per, c]   (DiT gated residual)         */
  MTX_EW_ROW_GATHER = 8,  /* y[r, :] = a[idx[r], :], idx = (const int32_t*)s, r < n*h*w       */
  MTX_EW_IM2COL = 9,      /* y[r, tap*C + c] = a[n, oy*s-pad+ky, ox*s-pad+kx, c] (zero outside);
                             i0 = k, i1 = stride, pad = k/2; optional s = int32 row map
                             (output row r takes raster output pixel idx[r]); ldy >= k*k*C       */
  MTX_EW_SOFTMAX_ROWS = 10, /* y[r, :c] = softmax(act_param * a[r, :c]) over rows r < n*h*w (VAE attention); i0 > 0: only the first i0
                               columns carry weight, the others (padding up to the 16-byte chunk) come out 0 */
  MTX_EW_TRANSPOSE = 11,  /* y[c, r] = a[r, c] for r < h*w rows, c columns (per n; ldy = row stride of y) */
  MTX_EW_AVGPOOL2 = 13,   /* 2x2 stride-2 average pool, ceil mode, divisor = in-bounds taps (ResNet-vd shortcut) */
  MTX_EW_DWCONV = 15,     /* depthwise k x k convolution, stride 1, pad k/2 (i0 = k: 3 or 7): y[.., c] = act(sum_t a[.. + t, c] * w[t][c] + bias[c]);
                             s = weights T [k*k][C] (tap-major), b = fp32 bias [C] (may be null) — YOLO11 head / C2PSA and YOLO12 area-attention
                             positional convs (ultralytics DWConv, Conv(g = c)) */
  MTX_EW_SWIGLU = 14,     /* y = silu(a) * b   (FLUX.2 feed-forward: a, b = the two column halves of linear_in's output) */
  MTX_EW_SHUFFLE2_ADD = 16, /* fp32 only: a [n, h, w, 4c] holds per input pixel the four output pixels' c channels (column (dy * 2 + dx) * c + ch: a
                               ConvTranspose2d(k = 2, s = 2) done as a GEMM); y[n, 2h + dy, 2w + dx, ch] = a[...] + b[(n), 2h + dy, 2w + dx, ch] with b
                               optional and lds = its per-sample stride (0 = one image for every n) — SAM's mask upscaling                        */
  MTX_EW_CVT_F32 = 17,    /* fp32 only: y = (float) a, a of the 16-bit type i0 (MTX_F16 / MTX_BF16)                                              */
  MTX_EW_CVT_16 = 18,     /* fp32 only: y = a rounded to the 16-bit type i0, written i1 (1..4) times side by side: y[px][j * c + ch]             */
  MTX_EW_SUB = 19,        /* y = a - b                                                                                                             */
  MTX_EW_RESIDUAL_DIST = 20, /* first-block cache probe: r = a - b rounded to the 16-bit type, prev = s (same layout, row stride lds); y = fp32
                                [2 * MTX_RESDIST_PARTS]: y[2 g] = part g of sum |prev - r|, y[2 g + 1] = part g of sum |prev| (fixed summation order; the
                                caller adds the parts in index order)                                                                          */
  MTX_EW_V_F8T = 21,      /* the value operand of the attention with fp8 P V (mtx_attn_args.v_f8t): a = v rows [rows][c = heads * 128] (16-bit, row stride lda)
                             -> y8 = e4m3 bytes [head][128 d][ldy8 keys], ldy8 = rows padded to a multiple of 64, zero beyond `rows`.  Inside every 64-key
                             tile the keys of a d-row are stored in the order the score accumulators hold them (byte j of the tile: key
                             32 (j >> 5) + (j & 3) + 8 ((j & 15) >> 2) + 4 ((j >> 4) & 1)), so that the probabilities need no shuffle.  Saturated at +-448. */
  MTX_EW_QK_NORM_ROPE = 12 /* FLUX attention prep, in place friendly: for every token r and head hd (c = heads*d,
                             i0 = d): x <- RMSNorm_d(x) * gamma[d] (s = fp32 gamma, eps = act_param), then
                             rotary on interleaved pairs with b = fp32 [rows][d] cos|sin table laid out as
                             [rows][2][d/2]: (x0, x1) -> (x0*cos - x1*sin, x1*cos + x0*sin).
                             i1 > 0: heads >= i1 use the second gamma vector (fused q|k slices); with ldb > 0 the heads
                             < i1 (the q slice) read their table at b + ldb floats instead — a copy pre-multiplied by
                             the attention scale * log2(e), see MTX_ATTN_Q_PRESCALED                    */
} mtx_ew_kind;

#define MTX_RESDIST_PARTS 256

typedef struct mtx_ew_args {
  const void* a; const void* b; const void* s; void* y;
  int64_t n, h, w, c;          /* logical INPUT dims (h*w = rows per sample) */
  int64_t lda, ldb, ldy, lds;  /* per-pixel strides; lds = per-sample stride of s */
  int32_t kind; int32_t act; float act_param; int32_t i0, i1; int32_t dtype;
  /* ABI 8, MTX_EW_QK_NORM_ROPE only: an e4m3 twin of the result, y8[row * ldy8 + column] (bytes; ldy8 % 16 == 0), values of the heads < i1
   * (the q slice) multiplied by y8_mul first, saturated at +-448 — the operands of mtx_attn_args.q_f8 / k_f8.  NULL = none. */
  void* y8; int64_t ldy8; float y8_mul;
} mtx_ew_args;

/* RCAN channel attention squeeze/excite: s[n, c] = sigmoid(W2 relu(W1 mean + b1) + b2),
 * mean = sum over tiles of chan_sum / hw. w1: [Cr][C] fp32, w2: [C][Cr] fp32.            */
typedef struct mtx_ca_args {
  const float* chan_sum; const float* w1; const float* b1; const float* w2; const float* b2;
  float* s; int32_t n, tiles, c, cr; float inv_hw;
  const float* inv_hw_dev;       /* optional DEVICE scalar that replaces inv_hw (bucket plans: 1 / (valid_h * valid_w)) */
  /* "pool before the conv" (t != NULL): chan_sum holds the channel sums of t [n, h, w, ldt] (dtype `dtype`), and the pooled vector the
   * MLP sees is mean(conv3x3(t; conv_w, zero padding 1) + conv_b) — obtained WITHOUT running the conv, by linearity:
   *   sum_p conv(t)[p][co] = sum_tap sum_ci W[co][tap][ci] * S_tap[ci],  S_tap = total - excluded border row - excluded border column + corner
   * with the border sums read straight from t (2 rows + 2 columns).  conv_w is the packed conv weight [c][9][c] of type `dtype`.
   * valid_hw (optional DEVICE {h, w}) replaces h, w (bucket plans). */
  const void* t; const void* conv_w; const float* conv_b;
  int32_t h, w, ldt, dtype;
  const int32_t* valid_hw;
  /* optional, pool-before-conv form only: with a scratch the launch uses MTX_CA_SPLIT workgroups per image instead of one (one
   * workgroup reading 2 048 sum rows and 5 120 border pixels was 29 us of latency between an RCAB's two convs, 200 times per page):
   * each reduces a share of the sum rows and border strips, hands its partial record over write-through and draws a ticket; the
   * workgroup that draws the last ticket adds the records up in a fixed order, runs the MLP and puts the counter back to zero.
   * Layout: n * MTX_CA_SPLIT * MTX_CA_RECORD floats, then n uint32 counters that MUST BE ZERO before the first launch (MTX_CA_SCRATCH_BYTES(n)). */
  float* scratch;
} mtx_ca_args;
#define MTX_CA_SPLIT 32
#define MTX_CA_RECORD 320      /* channel totals [64] + border strips [4][64] */
#define MTX_CA_SCRATCH_BYTES(n) ((size_t)(n) * (MTX_CA_SPLIT * MTX_CA_RECORD * 4 + 4))

/* image <-> tensor conversions at the page boundary (core/image/image_utils.py:351-366). */
typedef enum mtx_img_kind {
  MTX_IMG_NCHW_F32_TO_NHWC = 0, /* [N,3,H,W] f32 -> [N,H/u,W/u,Cpad] T, x*mul + add[c], u = unshuffle */
  MTX_IMG_NHWC_TO_NCHW_F32 = 1, /* [N,H,W,ld] T (first 3 ch) -> [N,3,H,W] f32, x*mul + add[c]     */
  MTX_IMG_NHWC_TO_HWC_U8 = 2,   /* clamp(x*mul+add,0,1)*255 truncated -> uint8 [N,H,W,3]           */
  MTX_IMG_HWC_U8_TO_NHWC = 3    /* uint8 [N,H,W,3] -> T  (x/255*mul + add[c])                      */
} mtx_img_kind;

typedef struct mtx_img_args {
  const void* src; void* dst;
  int64_t n, h, w;        /* source spatial dims */
  int32_t c_pad;          /* channels of the NHWC side (ld) */
  int32_t unshuffle;      /* 1 or 2 (pixel-unshuffle factor folded into the layout change) */
  float mul; float add[4];
  int32_t kind; int32_t dtype;
  const int32_t* valid_hw;       /* optional DEVICE {valid_h, valid_w} in SOURCE pixels (kinds 0 and 3): destination pixels beyond are zero */
} mtx_img_args;

/* bilinear resize of fp32/T logits to page size fused with the >0 threshold
 * (HF Sam2ImageProcessor.post_process_masks, called at core/image/detection.py:507-510). */
typedef struct mtx_resize_thresh_args {
  const void* src; uint8_t* dst;    /* src [N, hs, ws, pix_stride] ; dst [N, hd, wd] 0/1 */
  int64_t n, hs, ws, hd, wd; float thresh; int32_t dtype;
  int32_t pix_stride;               /* elements between source pixels (0 -> 1) */
  const int32_t* sel;               /* optional [N]: channel picked per sample */
  int64_t batch_stride;             /* elements between samples (-1 -> hs*ws*pix_stride; 0 = shared map) */
  int32_t roi_y, roi_x, roi_h, roi_w;  /* source window that is resized (roi_h == 0 -> whole map) */
  const float* crop_xyxy;           /* optional [N][4] page-space boxes: mask = 0 outside [x1,x2) x [y1,y2)
                                       (ultralytics ops.crop_mask)                                   */
} mtx_resize_thresh_args;

/* SAM-2.1 single-mask selection (transformers Sam2MaskDecoder._dynamic_multimask_via_stability,
 * modeling_sam2.py:1265-1311): per box, stability = |logit0 > +delta| / |logit0 > -delta|;
 * sel = 0 if stability >= thresh else 1 + argmax(iou[1:4]).  logits fp32 [N, pix, 4].        */
typedef struct mtx_mask_select_args {
  const float* logits; const float* iou; int32_t* counts /* [N][2] scratch */; int32_t* sel /* [N] */;
  int64_t n, pix; float delta, thresh;
} mtx_mask_select_args;

/* antialiased bilinear resize of a uint8 HWC page to (oh, ow), rounded to uint8 levels, then
 * (x/255 - mean[c]) / std[c] into NHWC T with c_pad channels (Sam2ImageProcessor,
 * image_processing_sam2.py:370-380; called at core/image/detection.py:494-495).             */
typedef struct mtx_preproc_args {
  const uint8_t* src; void* dst;
  int64_t h, w, oh, ow; int32_t c_pad; float mean[3]; float std[3]; int32_t dtype;
  /* mode 1 = ultralytics LetterBox (called at core/image/detection.py:1337-1345): the page is resized
   * with plain bilinear (no antialias) to new_h x new_w, placed at (pad_top, pad_left) of the oh x ow
   * canvas filled with pad_value (114), channels flipped BGR -> RGB, scaled by 1/255.          */
  int32_t mode; int32_t new_h, new_w, pad_top, pad_left; float pad_value;
} mtx_preproc_args;

/* YOLOv8 Detect/Segment head decode (ultralytics nn/modules/head.py Detect._inference + DFL):
 * per anchor: box = dist2bbox(softmax-expectation over reg_max bins of the 4 sides) * stride (xyxy in
 * letterboxed pixels), class scores = sigmoid, mask coefficients copied.  Level l holds NHWC rows of
 * [4*reg_max | nc | nm] channels (ld elements per pixel).  out: fp32 [anchors][4 + nc + nm].      */
typedef struct mtx_yolo_decode_args {
  const void* level[4]; int32_t lh[4], lw[4], lld[4], lstride[4];
  int32_t n_levels, nc, nm, reg_max; float* out; int32_t dtype;
  int32_t cls_off, mc_off;       /* channel offsets of the class / mask-coefficient slices (0 -> packed) */
  /* ABI 8.  box_f32[l] (optional): the 4 * reg_max DFL logits of level l as fp32 rows [lh * lw][4 * reg_max] — the box branch's last 1x1
   * convolution written by an fp32-output GEMM; the box channels of level[l] are then not read.  The reference runs this branch in fp32 end
   * to end (ultralytics behind core/image/detection.py:1337-1345) and its boxes feed IoU > 0.7 / IoA > 0.9 index decisions: the logits'
   * rounding to 16 bits alone is 0.03 bin = 1 px at stride 32. */
  const float* box_f32[4];
} mtx_yolo_decode_args;

/* ---- bubble cleaning, pixel half (replaces the cv2 chain of reference core/image/cleaning.py:296-337 ------
 * `process_single_bubble`: dilate(ellipse) -> threshold [Otsu] -> distanceTransform(DIST_L2, 5) >= shrink ->
 * erode(ellipse), and :155-207 `_build_adaptive_shrink_mask`) for all N bubbles of a page at once, each on
 * its own crop rois[i] = (x0, y0, w, h) (the dilated mask's box + 2 px, clipped to the page).  Planes hold
 * the crops back to back (bubble i starts at byte offsets[i]); values are 0 / 255 like the reference's
 * masks.  stats[i] = { hist[256] of the thresholding image over the ROI, sum and count of grey under the
 * base mask, is_black, threshold used }.  The chamfer distance is 16.16 fixed point (weights 65536, 91750,
 * 143976) relaxed `sweeps` (>= ceil(shrink)) times: exact for every distance below the shrink radius.   */
typedef struct mtx_clean_args {
  const void* page_bgr;                 /* u8 [H][W][3] */
  const void* masks;                    /* u8 [N][H][W], nonzero = inside the bubble */
  const int32_t* rois;                  /* [N][4] */
  const int64_t* offsets;               /* [N] */
  void* base; void* roi; void* eroded; void* thresholded; void* shrunk;     /* u8 planes (out) */
  int32_t* dist_a; int32_t* dist_b;     /* i32 planes (scratch) */
  int32_t* stats;                       /* [N][260] (out; zeroed by the call) */
  const int32_t* zones;                 /* [N][max_zones][4] junction zones x1,y1,x2,y2 (page coords), or NULL */
  int32_t n, page_h, page_w, max_zones;
  int32_t dil_r, ero_r;                 /* half heights of the two elliptical structuring elements */
  int8_t dil_dx[64], ero_dx[64];        /* half widths of their rows dy = -r .. r (index dy + r) */
  int32_t threshold, use_otsu;
  int32_t shrink_fixed, junction_fixed; /* ceil(radius * 65536) */
  int32_t sweeps, max_pixels;           /* max_pixels = largest crop area */
} mtx_clean_args;

/* ---- RT-DETR decoder pieces (HF RTDetrV2, reference core/ml/rtdetr_adapter.py:61-113) -------------------------
 * kind 0  multi-scale deformable attention ("default" method): for query r, head h, the heads*d channels of
 *         out[r] = sum_p softmax_p(aw[r, h, :])[p] * bilinear(value_level(p)[h], ref[r].xy + off[r, h, p] * ref[r].wh * scale / n_points)
 *         (grid_sample align_corners = False, zero padding).  value: [sum(H_l*W_l)][heads*d] T; off: [rows][heads*L*P*2] T;
 *         aw: [rows][heads*L*P] T; ref: fp32 [rows][8] (cx, cy, w, h, ...).
 * kind 1  reference refinement: ref_out = sigmoid(delta[r, :4] + logit(clamp(ref[r]))) (fp32) and its T copy (8 columns, zero padded)
 * kind 2  ref_out = sigmoid(x[r, :4]) from fp32 logits x, plus the T copy                                              */
typedef struct mtx_detr_args {
  const void* value; const void* off; const void* aw; const float* ref; void* out;     /* kind 0 */
  const void* delta; float* ref_out; void* ref_t;                                       /* kinds 1, 2 (ref = input) */
  int32_t kind, rows, heads, d, levels, points;
  int32_t lh[4], lw[4], lstart[4];
  int32_t ld_value, ld_off, ld_aw, ld_out, ld_delta;
  float offset_scale; int32_t dtype;
} mtx_detr_args;

/* MX block quantisation of a 16-bit [rows, K] matrix (row stride ldx) to OCP fp8 e4m3 for the scaled matrix instructions
 * (v_mfma_scale_f32_32x32x64_f8f6f4): every 32 consecutive k of a row share one E8M0 scale 2^(e - 127), e chosen so that
 * the block's largest magnitude lands in (224, 448] (never saturates); q = RNE(x * 2^-(e - 127)).  q: [rows][ldq] bytes.
 * scale: one uint32 per (row, 128 k): byte b = e of block 4 * (k / 128) + b, laid out scale[(k / 128) * lds + row] so the
 * 32 rows a GEMM wave reads are contiguous.  K % 128 == 0. */
typedef struct mtx_quant_args {
  const void* x; void* q; void* scale;
  int64_t rows, k, ldx, ldq, lds;
  int32_t dtype;                 /* MTX_BF16 / MTX_F16: type of x */
  /* op MTX_QUANT_SWIGLU: the matrix that is quantised is silu(x) * b (b: [rows, K], row stride ldb), rounded to `dtype` first —
   * bit-identical to MTX_EW_SWIGLU followed by a plain quantise; y (optional, row stride ldy) also receives that 16-bit result. */
  int32_t op; const void* b; int64_t ldb; void* y; int64_t ldy;
} mtx_quant_args;
#define MTX_QUANT_PLAIN 0
#define MTX_QUANT_SWIGLU 1

/* The inpainting stage's image arithmetic around the diffusion pipeline, on the device (csrc/pagetail.hip; reference
 * core/image/inpainting.py:543-611, 877-968, 1187-1313, 1577-1665): all on uint8 HWC images, row strides ld_* in BYTES.
 *  MTX_TAIL_RESAMPLE   one axis of Pillow's 8-bit resize (Image.resize with BILINEAR / BICUBIC / LANCZOS), bit-exact: output
 *                      coordinate o reads src positions bounds[2o] .. + bounds[2o+1] with the fixed-point taps coeff[o * ksize ..]
 *                      (22 fractional bits, built like Pillow's precompute_coeffs + normalize_coeffs_8bpc by the caller).
 *                      axis 0: dst [out_h][out_w] from src rows src_row0 .. (horizontal pass); axis 1: vertical pass.
 *  MTX_TAIL_COMPOSITE  dst (the page, page_c channels, IN PLACE) window [y, y + out_h) x [x, x + out_w) = patch * alpha + page * (1 - alpha)
 *                      in fp32 with the reference's operation order, truncated to uint8 (bit-exact with the numpy expression).
 *  MTX_TAIL_LAB_STATS  OpenCV fixed-point RGB -> Lab of src (generated patch) and other (original crop); over the pixels with
 *                      mask == 0: sums[0..8] += {n, L, L^2, a, b of src, L, L^2, a, b of other} as exact uint64 (zero them first).
 *  MTX_TAIL_LAB_REMAP  dst = Lab -> RGB of (Lab(src) with L' = (L - params[0]) * params[1] + params[2], a' = a + params[3] if params[5],
 *                      b' = b + params[4] if params[6] on the pixels with mask != 0, clipped and truncated to uint8).
 * gamma_tab [256], cbrt_tab [cbrt_n], lab_coef [9]: OpenCV's tables (int32), uploaded once by the caller.
 *  MTX_TAIL_EDT_COLS   src = mask uint8 [out_h][out_w] (ld_src; nonzero = set), dst = uint8 g (ld_dst): distance to the nearest set pixel of
 *                      the same column, ksize = radius R (1..254); R + 1 = none within R.
 *  MTX_TAIL_EDT_ROWS   src = g, dst = float32 weight (ld_dst in floats): 1 on the mask, params[d^2] off it where d^2 = the exact squared
 *                      Euclidean distance to the mask if <= R^2, else 0 (params: float [R * R + 1], the host's float64 ramp
 *                      clip(1 - sqrt(d^2) / R, 0, 1) rounded to float32 — scipy.ndimage.distance_transform_edt + the numpy expression of
 *                      reference core/image/inpainting.py:1126-1163, bit for bit).  axis != 0: strict (0 off the mask).  Weights outside the
 *                      rectangle [x, page_c) x [y, cbrt_n) are 0 (composite_clip_bbox; pass 0, out_w, 0, out_h for none). */
typedef enum mtx_tail_kind { MTX_TAIL_RESAMPLE = 0, MTX_TAIL_COMPOSITE = 1, MTX_TAIL_LAB_STATS = 2, MTX_TAIL_LAB_REMAP = 3,
                             MTX_TAIL_EDT_COLS = 4, MTX_TAIL_EDT_ROWS = 5 } mtx_tail_kind;
typedef struct mtx_tail_args {
  int32_t kind;
  const void* src; void* dst;
  int32_t out_h, out_w, c;
  int64_t ld_src, ld_dst;
  const int32_t* bounds; const int32_t* coeff; int32_t ksize, axis, src_row0, coeff_bits;     /* RESAMPLE (coeff_bits: fractional bits of the taps, 0 = Pillow's 22) */
  const float* alpha; int64_t ld_alpha; int32_t x, y, page_c, src_c;                           /* COMPOSITE (ld_alpha in floats; c = channels blended, src_c = bytes per patch pixel, 0 = c) */
  const int32_t* gamma_tab; const int32_t* cbrt_tab; const int32_t* lab_coef; int32_t cbrt_n; /* LAB_* */
  const uint8_t* mask; int64_t ld_mask; const void* other; int64_t ld_other;
  unsigned long long* sums; const float* params;
} mtx_tail_args;

typedef enum mtx_op_kind {
  MTX_OP_CONV2D = 1, MTX_OP_GEMM = 2, MTX_OP_ATTN = 3, MTX_OP_NORM = 4, MTX_OP_GROUPNORM = 5,
  MTX_OP_EW = 6, MTX_OP_CA = 7, MTX_OP_IMG = 8, MTX_OP_RESIZE_THRESH = 9, MTX_OP_MEMSET = 10,
  MTX_OP_MASK_SELECT = 11, MTX_OP_PREPROC = 12, MTX_OP_YOLO_DECODE = 13, MTX_OP_DETR = 14, MTX_OP_QUANT = 15, MTX_OP_TAIL = 16
} mtx_op_kind;

typedef struct mtx_memset_args { void* ptr; int64_t bytes; int32_t value; } mtx_memset_args;

/* `lane`: 0 = the plan's main stream.  MTX_LANE_SIDE puts the op on the plan's side stream: a run of side ops starts after everything
   the main lane has issued before it and then runs BESIDE the main ops that follow; the first later main op that needs its results
   carries MTX_LANE_JOIN (the main lane waits for the side lane there; the end of a plan always joins).  Program order is a valid
   serial order, so an executor may ignore lanes (mtx_plan_run_range, the timing entry points and the simulator do). */
#define MTX_LANE_SIDE 1
#define MTX_LANE_JOIN 2
typedef struct mtx_op {
  int32_t kind; int32_t lane;
  union {
    mtx_conv2d_args conv; mtx_gemm_args gemm; mtx_attn_args attn; mtx_norm_args norm;
    mtx_groupnorm_args gn; mtx_ew_args ew; mtx_ca_args ca; mtx_img_args img;
    mtx_resize_thresh_args rt; mtx_memset_args ms; mtx_mask_select_args sel; mtx_preproc_args pre; mtx_yolo_decode_args yd; mtx_detr_args detr; mtx_quant_args quant; mtx_tail_args tail;
  } u;
} mtx_op;

/* ---- library ---------------------------------------------------------------------------- */
MTX_API int mtx_abi_version(void);
MTX_API size_t mtx_abi_sizeof(int op_kind);          /* sizeof the args struct; 0 = sizeof(mtx_op) */
MTX_API const char* mtx_last_error(void);
MTX_API int mtx_init(int device_ordinal);            /* hipSetDevice + arch check (gfx950) */
MTX_API int mtx_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ---- op-level entry points (tests, and the building blocks of plans) --------------------- */
MTX_API int mtx_conv2d(const mtx_conv2d_args* a, void* stream);
MTX_API int mtx_conv2d_tiles(const mtx_conv2d_args* a);   /* spatial tiles per image (chan_sum rows) */
MTX_API int mtx_gemm(const mtx_gemm_args* a, void* stream);
/* how THIS THREAD's last 256-tile GEMM launch was cut (test / tooling introspection): tiles computed whole, K slices of the left-over
 * tiles (1 = none) and the number of (slice, tile) pieces of the K-slice tail */
MTX_API int mtx_gemm_last_split(int* whole_tiles, int* k_slices, int* tail_pieces);
MTX_API int mtx_attention(const mtx_attn_args* a, void* stream);
MTX_API int mtx_norm(const mtx_norm_args* a, void* stream);
MTX_API int mtx_groupnorm(const mtx_groupnorm_args* a, void* stream);
MTX_API int mtx_elementwise(const mtx_ew_args* a, void* stream);
MTX_API int mtx_channel_attention(const mtx_ca_args* a, void* stream);
MTX_API int mtx_image_convert(const mtx_img_args* a, void* stream);
MTX_API int mtx_resize_threshold(const mtx_resize_thresh_args* a, void* stream);
MTX_API int mtx_mask_select(const mtx_mask_select_args* a, void* stream);
MTX_API int mtx_preprocess(const mtx_preproc_args* a, void* stream);
MTX_API int mtx_yolo_decode(const mtx_yolo_decode_args* a, void* stream);
MTX_API int mtx_bubble_clean(const mtx_clean_args* a, void* stream);
MTX_API int mtx_detr(const mtx_detr_args* a, void* stream);
MTX_API int mtx_quantize_mx(const mtx_quant_args* a, void* stream);
MTX_API int mtx_page_tail(const mtx_tail_args* a, void* stream);
/* contour half of the same chain, host side on one crop (cleaning.py:340-386): external contours of the
 * thresholded crop -> area / centroid filter -> filled union -> largest blob -> final mask + bounding box.
 * Returns the number of accepted text fragments (0 = nothing to clean).                                */
/* host side: cv2.distanceTransform(src, DIST_L2, 5) on a crop (two-pass 16.16 fixed-point chamfer), used by the
 * conjoined-bubble partition (reference core/image/detection.py:932-968) */
MTX_API int mtx_host_chamfer_l2_5x5(const uint8_t* src, int w, int h, float* dist);
MTX_API int mtx_host_text_mask(const uint8_t* thr, const uint8_t* eroded, int w, int h, int ox, int oy, int page_w, int page_h,
                               double min_area, uint8_t* final_mask, int* bbox);
/* outline polygon of the largest 8-connected blob of a [h][w] mask (nonzero = set), outer border walked pixel centre to pixel
 * centre — what ultralytics exposes as `results.masks[i].xy[0]` (reference core/image/detection.py:525-556 reads it when SAM gave no
 * mask).  Writes up to `cap` (x, y) pairs, returns the outline's point count (call again if it exceeds cap), 0 for an empty mask. */
MTX_API int mtx_host_mask_outline(const uint8_t* mask, int w, int h, int* xy, int cap);

/* host side: PNG writer of the batch harness (csrc/host_png.cpp; replaces Pillow + oxipng of reference core/image/image_utils.py:140-150):
 * uint8 [h][w][channels] (1 = L, 2 = LA, 3 = RGB, 4 = RGBA) -> a PNG file image in `out`.  reduce != 0: lossless colour-type reductions
 * (opaque alpha dropped, R == G == B -> grey), as oxipng does.  level 0..9 = zlib level, `threads` stripes deflated concurrently.
 * Returns the file's length (call again with a larger buffer when it exceeds out_cap; out may be NULL to ask), < 0 on error. */
MTX_API int64_t mtx_host_png_encode(const uint8_t* pixels, int w, int h, int channels, int level, int threads, int reduce,
                                    uint8_t* out, int64_t out_cap);

/* ---- plans: a network forward as one native call ----------------------------------------- */
MTX_API int mtx_plan_create(const mtx_op* ops, int n_ops, void** plan);
MTX_API int mtx_plan_run(void* plan, void* stream);            /* eager launch of every op, in order */
MTX_API int mtx_plan_run_graph(void* plan, void* stream);      /* hipGraph replay (captured on first use) */
MTX_API int mtx_plan_num_ops(void* plan);
MTX_API int mtx_plan_run_range(void* plan, int first, int last, void* stream);  /* debugging / parity */
MTX_API void mtx_plan_destroy(void* plan);

/* HIP-event timing helper for bench.py: records events on `stream` around plan replays
 * and returns the average milliseconds per replay (synchronises the stream).               */
MTX_API int mtx_plan_time(void* plan, void* stream, int iters, int use_graph, float* ms_per_iter);
/* in-context time of the ops listed in op_idx: the whole plan replayed as a hipGraph minus the same graph without those ops, HIP
 * events around each replay on a stream of the function's own; *ms_total = that difference summed over `iters` pairs of replays
 * (caches, clocks and launch mode as in the production path).  The skipped ops leave their outputs stale: timing only.
 * With MTX_TIME_OPS=stamp in the environment the figure comes instead from one replay graph that stores the device wall clock
 * before and after each listed op (no second graph, so the power mix of the replay is the production one). */
MTX_API int mtx_plan_time_ops(void* plan, void* stream, const int* op_idx, int n_idx, int iters, float* ms_total);
/* time only ops [first,last] of a plan: avg ms over iters (events on `stream`)              */
MTX_API int mtx_plan_time_range(void* plan, int first, int last, void* stream, int iters, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* MTX_HIP_H */
