"""GPU tier: op-level parity of libmtx_hip.so (through the C ABI) against torch fp32 on MI355X."""
import pytest

import op_checks as oc
from mangatranslator_amd.hip import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [abi.BF16, abi.F16])
@pytest.mark.parametrize("cfg", [
    dict(n=1, h=20, w=19, cin=64, cout=64, ksize=3, stride=1, act=abi.ACT_RELU),
    dict(n=2, h=9, w=33, cin=16, cout=24, ksize=3, stride=1, with_res=True),
    dict(n=1, h=17, w=18, cin=72, cout=136, ksize=3, stride=1, act=abi.ACT_SILU, ldx_extra=8),
    dict(n=1, h=21, w=35, cin=48, cout=96, ksize=3, stride=2, act=abi.ACT_SILU),
    dict(n=1, h=16, w=16, cin=128, cout=64, ksize=1, stride=1, act=abi.ACT_LEAKY),
    dict(n=1, h=10, w=18, cin=32, cout=128, ksize=3, stride=1, pixel_shuffle=2, with_res=True),
    dict(n=2, h=18, w=20, cin=64, cout=64, ksize=3, stride=1, with_sum=True, act=abi.ACT_RELU),
    dict(n=2, h=52, w=50, cin=64, cout=64, ksize=3, stride=1, with_sum=True, act=abi.ACT_RELU),      # interior tiles: the descriptor DMA / packed epilogue paths
    dict(n=1, h=50, w=67, cin=40, cout=64, ksize=3, stride=1, ldx_extra=8),
    # page-scale shapes
    dict(n=1, h=384, w=256, cin=64, cout=64, ksize=3, stride=1, with_res=True, with_sum=True),
    dict(n=1, h=200, w=136, cin=192, cout=384, ksize=3, stride=2, act=abi.ACT_SILU),
    dict(n=1, h=100, w=68, cin=576, cout=192, ksize=1, stride=1, act=abi.ACT_SILU),
])
def test_conv(hip_lib, dtype, cfg):
    oc.check_conv(hip_lib, dtype, **cfg)


@pytest.mark.parametrize("cfg", [
    dict(m=130, n=136, k=72, act=abi.ACT_GELU),
    dict(m=64, n=20, k=144, with_res=True, with_gate=True),
    dict(m=200, n=260, k=200, act=abi.ACT_GELU_TANH, out_f32=True),
    dict(m=40, n=48, k=64, batch=3, alpha=0.5, with_bias=False),
    dict(m=4096, n=1728, k=576, act=abi.ACT_GELU, with_res=True),
    dict(m=8704, n=3072, k=3072, with_gate=True, with_res=True),
    dict(m=65536, n=432, k=144),
])
def test_gemm(hip_lib, cfg):
    oc.check_gemm(hip_lib, abi.BF16, **cfg)


def test_gemm_f16(hip_lib):
    oc.check_gemm(hip_lib, abi.F16, m=96, n=72, k=96, with_res=True)
    oc.check_gemm(hip_lib, abi.F16, m=1000, n=512, k=1024)


@pytest.mark.parametrize("cfg", [
    dict(batch=1, heads=2, sq=70, sk=150, d=72),
    dict(batch=2, heads=1, sq=16, sk=16, d=32),
    dict(batch=1, heads=1, sq=130, sk=64, d=128),
    dict(batch=1, heads=3, sq=9, sk=100, d=16),
    dict(batch=1, heads=2, sq=64, sk=64, d=64),
    dict(batch=16, heads=8, sq=256, sk=256, d=72),
    dict(batch=1, heads=8, sq=4096, sk=4096, d=72),
    dict(batch=1, heads=4, sq=2500, sk=2500, d=128),
    dict(batch=64, heads=2, sq=64, sk=64, d=72),
])
def test_attention(hip_lib, cfg):
    oc.check_attention(hip_lib, abi.BF16, **cfg)


def test_attention_f16(hip_lib):
    oc.check_attention(hip_lib, abi.F16, batch=1, heads=1, sq=33, sk=75, d=72)


@pytest.mark.parametrize("cfg", [
    dict(rows=7, c=144, kind=0),
    dict(rows=5, c=3072, kind=0, affine=False, modulate=True),
    dict(rows=9, c=128, kind=1),
    dict(rows=3, c=1152, kind=0),
    dict(rows=8704, c=3072, kind=0, affine=False, modulate=True),
    dict(rows=611, c=1024, kind=0, affine=False, modulate=True),
    dict(rows=1030, c=2048, kind=1, affine=False),
    dict(rows=517, c=6144, kind=0, affine=False, modulate=True),
    dict(rows=4001, c=1152, kind=0),                                 # affine, ragged chunks, enough elements for a last-bit difference between the kernel forms to show
    dict(rows=3000, c=1024, kind=1),
])
def test_norm(hip_lib, cfg):
    oc.check_norm(hip_lib, abi.BF16, **cfg)


def test_groupnorm(hip_lib):
    oc.check_groupnorm(hip_lib, abi.BF16, n=2, h=9, w=11, c=128, groups=32)
    oc.check_groupnorm(hip_lib, abi.F16, n=1, h=40, w=30, c=256, groups=32, silu=False)
    oc.check_groupnorm(hip_lib, abi.BF16, n=1, h=128, w=96, c=512, groups=32)


def test_elementwise(hip_lib):
    oc.check_ew(hip_lib, abi.BF16)
    oc.check_ew(hip_lib, abi.F16)


def test_resize_threshold(hip_lib):
    oc.check_resize_threshold(hip_lib)


def test_image_convert(hip_lib):
    oc.check_image_convert(hip_lib, abi.F16)
    oc.check_image_convert(hip_lib, abi.BF16)


@pytest.mark.parametrize("cfg", [dict(batch=1, heads=2, sq=1030, sk=330, d=128), dict(batch=2, heads=3, sq=2048, sk=2048, d=128),
                                 dict(batch=1, heads=4, sq=4100, sk=4100, d=128, qmul=6.0), dict(batch=1, heads=2, sq=1500, sk=8652, d=128),
                                 dict(batch=1, heads=2, sq=1100, sk=320, d=128), dict(batch=1, heads=2, sq=1100, sk=449, d=128)])
def test_attention_long_sequence_kernel(hip_lib, cfg):
    oc.check_attention(hip_lib, abi.BF16, **cfg)
    oc.check_attention(hip_lib, abi.F16, **cfg)


@pytest.mark.parametrize("cfg", [dict(m=8652, n=3072, k=3072, with_res=True, with_gate=True), dict(m=4100, n=9216, k=1024, act=abi.ACT_GELU_TANH),
                                 dict(m=2048, n=5000 // 8 * 8, k=320, act=abi.ACT_SILU, with_bias=False, with_res=True)])
def test_gemm_256_tile_kernel(hip_lib, cfg):
    oc.check_gemm(hip_lib, abi.BF16, **cfg)
    oc.check_gemm(hip_lib, abi.F16, **cfg)


def test_quantize_mx(hip_lib):
    """MX fp8 quantiser (v_cvt_pk_fp8_f32, OCP e4m3) bit-exact against the torch restatement"""
    oc.check_quantize_mx(hip_lib, abi.BF16, rows=8652, k=3072)
    oc.check_quantize_mx(hip_lib, abi.F16, rows=333, k=1152, ld_extra=8, spread=8.0)


@pytest.mark.parametrize("cfg", [dict(m=8652, n=3072, k=3072, with_res=True, with_gate=True), dict(m=4100, n=9216, k=1024, act=abi.ACT_GELU_TANH),
                                 dict(m=2048, n=5000 // 8 * 8, k=384, act=abi.ACT_SILU, with_bias=False, with_res=True, spread=2.0),
                                 dict(m=8704, n=3072, k=12288, with_gate=True, with_res=True), dict(m=300, n=264, k=128, flags=abi.GEMM_FORCE_TILE256)])
def test_gemm_fp8(hip_lib, cfg):
    """v_mfma_scale_f32_32x32x64_f8f6f4 path against an fp32 product of the same quantised operands (FLUX.2-Klein shapes)"""
    err, qerr = oc.check_gemm_f8(hip_lib, abi.BF16, **cfg)
    print(f"fp8 gemm {cfg}: kernel rel err {err:.2e}, quantisation rel err vs the 16-bit product {qerr:.3f}")
    assert qerr < 0.06


def test_swiglu(hip_lib):
    oc.check_swiglu(hip_lib, abi.BF16, rows=8652, hid=9216)


def test_attention_prescaled_q(hip_lib):
    oc.check_attention(hip_lib, abi.BF16, batch=1, heads=3, sq=2100, sk=2100, d=128, qmul=4.0, prescaled=True)
    oc.check_attention(hip_lib, abi.BF16, batch=1, heads=24, sq=8652, sk=8652, d=128, prescaled=True)          # FLUX shape, key-split tail
    oc.check_attention(hip_lib, abi.F16, batch=1, heads=2, sq=1100, sk=449, d=128, qmul=40.0, prescaled=True)    # refresh path
    oc.check_attention(hip_lib, abi.BF16, batch=2, heads=2, sq=300, sk=200, d=64, prescaled=True)
    # a forced stale maximum (cdna_hip_programming.md §5.4 rule 26): keys from 200 on score far above everything before them, so the first tile's
    # maximum is stale by then and the partial-row-sum check must refresh it
    oc.check_attention(hip_lib, abi.BF16, batch=1, heads=2, sq=1024, sk=320, d=128, qmul=8.0, prescaled=True, late_keys=(200, 6.0))
    oc.check_attention(hip_lib, abi.BF16, batch=1, heads=2, sq=1100, sk=449, d=128, qmul=40.0, prescaled=True)


def test_f32_ops(hip_lib):
    """MTX_F32 instantiations of gemm / attention / norm / element-wise (csrc/f32ops.hip: SAM's fp32 mask decoder under precision "high")"""
    assert oc.check_f32_ops(hip_lib) < 2e-5


def test_hi_lo_weight_pairs(hip_lib):
    """W = W_hi + W_lo as ONE GEMM over K' = 2K ([x | x] against [W_hi | W_lo], fp32 accumulation): SAM's trunk under precision 'high'"""
    oc.check_hi_lo_weights(hip_lib)
    oc.check_hi_lo_weights(hip_lib, m=4096, n=2304, k=576)          # Hiera-L stage 3, fc1
    oc.check_hi_lo_weights(hip_lib, dtype=abi.BF16, m=1024, n=1152, k=4608, seed=1)
    assert oc.check_norm_f32_to_16(hip_lib, abi.F16) < 1e-3


def test_activation_epilogues_at_extreme_values(hip_lib):
    oc.check_gemm_act_extremes(hip_lib, abi.BF16)
    oc.check_gemm_act_extremes(hip_lib, abi.F16)


def test_first_block_cache_probe(hip_lib):
    """MTX_EW_RESIDUAL_DIST + MTX_EW_SUB at the Kontext image stream's size: the distance torch computes on the same rounded operands, the same
    parts from two launches (no atomics: the cache decision cannot depend on scheduling)"""
    oc.check_residual_dist(hip_lib, abi.BF16, rows=8300, c=3072)
    oc.check_residual_dist(hip_lib, abi.F16, rows=333, c=136, ld_extra=24, seed=1)


def test_flux_prep_kernels(hip_lib):
    oc.check_qk_norm_rope(hip_lib, abi.BF16, rows=1000, heads=24, d=128)
    oc.check_qk_norm_rope(hip_lib, abi.BF16, rows=1000, heads=24, d=128, q_fold=0.1275)
    oc.check_qk_norm_rope(hip_lib, abi.F16, rows=333, heads=2, d=64, fused=False)
    oc.check_softmax_transpose(hip_lib, abi.BF16, rows=1024, cols=1024)
    oc.check_softmax_transpose(hip_lib, abi.F16, rows=70, cols=136)


def test_gemm_matrix_beyond_4gb(hip_lib):
    """VAE im2col shape class: 1.05 M rows x 2304 columns of bf16 = 4.8 GB; tile-relative descriptors keep the 256-tile kernel"""
    oc.check_gemm_huge_rows(hip_lib, abi.BF16, m=1050000, n=256, k=2304)


def test_gemm_k_slice_tail(hip_lib):
    """the K-slice tail with its last-arriver fix-up at FLUX shapes on 256 CUs; three runs of a plan must agree bit for bit"""
    oc.check_gemm(hip_lib, abi.BF16, m=512, n=3072, k=12288, with_res=True, with_gate=True, runs=3, expect_split=(0, "sliced", None))     # whole problem in K slices
    oc.check_gemm(hip_lib, abi.BF16, m=8624, n=3072, k=15360, act=abi.ACT_NONE, with_res=True, with_gate=True, runs=3, expect_split=(256, "sliced", None))   # left-over tiles only
    oc.check_gemm(hip_lib, abi.BF16, m=8812, n=3072, k=15360, with_res=True, with_gate=True, runs=2, expect_split=(256, "sliced", None))
    oc.check_gemm(hip_lib, abi.F16, m=8112, n=3072, k=12288, runs=2, expect_split=(256, "sliced", None))


@pytest.mark.parametrize("dtype", [abi.BF16, abi.F16])
def test_rcab_tail_pool_before_conv(hip_lib, dtype):
    """RCAN's RCAB tail: channel attention from the sums of conv2's INPUT, conv2 writing x + s * conv2(t) (interior and border tiles)"""
    oc.check_rcab_tail(hip_lib, dtype, n=2, h=37, w=29)
    oc.check_rcab_tail(hip_lib, dtype, n=1, h=50, w=52)
    oc.check_rcab_tail(hip_lib, dtype, n=1, h=21, w=40, canvas=(64, 64))


def test_producers_write_their_fp8_twins(hip_lib):
    """adaLN norm and SwiGLU quantise their own output for the fp8 linears (Klein-4B width): bit-identical to producer + mtx_quantize_mx"""
    oc.check_fused_quantisers(hip_lib, abi.BF16, rows=8512, c=3072, hid=9216)
    oc.check_fused_quantisers(hip_lib, abi.BF16, rows=333, c=1152, hid=384, seed=1)


# --- the two epilogue fusions of the FLUX.2-Klein fp8 path at Klein's real shapes (they had only ever run on the simulator)
@pytest.mark.parametrize("cfg", [
    dict(m=8000, col0=0, hid=9216, k=3072, row_off=512),                       # double-block MLP-in
    dict(m=8512, col0=9216, hid=9216, k=3072, q_col_off=3072, seed=1),          # single-block fused projection, qkv columns in front
    dict(m=512, col0=0, hid=9216, k=3072, seed=2),                              # ragged text stream
    dict(m=300, col0=0, hid=256, k=256, seed=3),
])
def test_gemm_f8_glu_epilogue(hip_lib, cfg):
    """gemm256_f8_glu_kernel: e4m3 bytes + E8M0 scale words equal GEMM -> MTX_QUANT_SWIGLU byte for byte."""
    oc.check_gemm_f8_glu(hip_lib, abi.BF16, **cfg)


def test_attention_fp8_scores(hip_lib):
    """attn_mma32_k8_kernel / _k8q_kernel (mtx_attn_args.q_f8 / k_f8): scores from e4m3 q and k on the MX-scaled fp8 matrix instruction —
    against the exact softmax of the same e4m3 products; with the MX fp8 output form: the bytes of attention + quantiser.  The rotary
    kernel's e4m3 twin (mtx_ew_args.y8) byte for byte."""
    oc.check_attention_f8_scores(hip_lib, abi.BF16, heads=3, sq=2100, sk=2100)
    oc.check_attention_f8_scores(hip_lib, abi.BF16, heads=24, sq=8704, sk=8704, seed=3)          # Klein shape at 2048 x 3072 incl. the key-split tail
    oc.check_attention_f8_scores(hip_lib, abi.F16, heads=2, sq=1100, sk=449, exponent=-2, seed=1)
    oc.check_attention_f8_scores(hip_lib, abi.BF16, heads=2, sq=1024, sk=320, qmul=4.0, late_keys=(200, 6.0), seed=2)      # a forced stale maximum
    oc.check_rope_f8_twin(hip_lib, abi.BF16, rows=8704, heads=24)
    oc.check_rope_f8_twin(hip_lib, abi.F16, rows=333, heads=3, q_mul=1.0, seed=1)


def test_attention_fp8_scores_and_values(hip_lib):
    """attn_mma32_k8v8q_kernel (mtx_attn_args.v_f8t, an experiment of the FLUX.2 fp8 path) + MTX_EW_V_F8T: V^T bytes equal the restatement; output rows
    within 1.5 points (rms, relative) of the 16-bit-P-V kernel's distance to the exact softmax — Klein's shape incl. the key-split tail, a ragged key count,
    a forced stale maximum"""
    e1 = oc.check_attention_f8_pv(hip_lib, abi.BF16, heads=24, sq=8704, sk=8704, seed=3)
    e2 = oc.check_attention_f8_pv(hip_lib, abi.BF16, heads=3, sq=2100, sk=2100)
    e3 = oc.check_attention_f8_pv(hip_lib, abi.F16, heads=2, sq=1100, sk=449, exponent=-2, seed=1)
    e4 = oc.check_attention_f8_pv(hip_lib, abi.BF16, heads=2, sq=1024, sk=320, qmul=4.0, late_keys=(200, 6.0), seed=2)
    from parity_log import record
    record("attention.fp8_pv.rms_rel_err_16bit_pv_vs_fp8_pv", T8704=e1, T2100=e2, ragged_449=e3, stale_max=e4)


@pytest.mark.parametrize("cfg", [
    dict(heads=24, sq=8512, sk=8512),                                           # key-split tail blocks included
    dict(heads=24, sq=8652, sk=8652, extra_cols=9216, seed=1),                  # into the single blocks' concatenation buffer
    dict(heads=4, sq=1072, sk=1072, seed=2),
])
def test_attention_mx_fp8_output(hip_lib, cfg):
    """attn_mma32_q8_kernel / attn_merge_q8_kernel: bytes and scale words equal attention -> mtx_quantize_mx."""
    oc.check_attention_q8(hip_lib, abi.BF16, **cfg)


def test_memset_op(hip_lib):
    oc.check_memset(hip_lib)
