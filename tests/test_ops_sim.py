"""CPU tier: the gfx950 kernel sources, compiled for the host by tests/emu (SIMT simulator),
checked op by op against torch fp32.  Validates indexing (halo tiles, swizzles, MFMA fragment
maps, epilogues) where no GPU exists; the same checks run on hardware in test_ops_gpu.py."""
import pytest
import torch

import op_checks as oc
from mangatranslator_amd.hip import abi


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
    dict(n=1, h=40, w=24, cin=64, cout=64, ksize=3, stride=1, with_res=True, with_sum=True),       # sums AND residual: the generic kernel and its own row count
])
def test_conv(emu_lib, dtype, cfg):
    if dtype == abi.F16 and cfg.get("stride") == 2:
        pytest.skip("one dtype is enough for the slow stride-2 case")
    oc.check_conv(emu_lib, dtype, **cfg)


@pytest.mark.parametrize("cfg", [
    dict(m=130, n=136, k=72, act=abi.ACT_GELU),
    dict(m=64, n=20, k=144, with_res=True, with_gate=True),
    dict(m=200, n=260, k=200, act=abi.ACT_GELU_TANH, out_f32=True),
    dict(m=40, n=48, k=64, batch=3, alpha=0.5, with_bias=False),
])
def test_gemm(emu_lib, cfg):
    oc.check_gemm(emu_lib, abi.BF16, **cfg)


def test_gemm_f16(emu_lib):
    oc.check_gemm(emu_lib, abi.F16, m=96, n=72, k=96, with_res=True)


@pytest.mark.parametrize("cfg", [
    dict(batch=1, heads=2, sq=70, sk=150, d=72),
    dict(batch=2, heads=1, sq=16, sk=16, d=32),
    dict(batch=1, heads=1, sq=130, sk=64, d=128),
    dict(batch=1, heads=3, sq=9, sk=100, d=16),
    dict(batch=1, heads=2, sq=64, sk=64, d=64),
])
def test_attention(emu_lib, cfg):
    oc.check_attention(emu_lib, abi.BF16, **cfg)


def test_attention_f16(emu_lib):
    oc.check_attention(emu_lib, abi.F16, batch=1, heads=1, sq=33, sk=75, d=72)


@pytest.mark.parametrize("cfg", [
    dict(rows=7, c=144, kind=0),
    dict(rows=5, c=3072, kind=0, affine=False, modulate=True),
    dict(rows=9, c=128, kind=1),
    dict(rows=3, c=1152, kind=0),
    dict(rows=6, c=1024, kind=0, affine=False, modulate=True),      # whole 512-chunks, no affine: the straight-line kernel
    dict(rows=5, c=2048, kind=1, affine=False),
])
def test_norm(emu_lib, cfg):
    oc.check_norm(emu_lib, abi.BF16, **cfg)


def test_groupnorm(emu_lib):
    oc.check_groupnorm(emu_lib, abi.BF16, n=2, h=9, w=11, c=128, groups=32)
    oc.check_groupnorm(emu_lib, abi.F16, n=1, h=40, w=30, c=256, groups=32, silu=False)


def test_elementwise(emu_lib):
    oc.check_ew(emu_lib, abi.BF16)
    oc.check_ew(emu_lib, abi.F16)


def test_resize_threshold(emu_lib):
    oc.check_resize_threshold(emu_lib)


def test_image_convert(emu_lib):
    oc.check_image_convert(emu_lib, abi.F16)
    oc.check_image_convert(emu_lib, abi.BF16)


def test_attention_long_sequence_kernel(emu_lib):
    """d = 128, sq >= 1024: the 8-wave 32x32x16 kernel (ragged q block and ragged last key tile)"""
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=2, sq=1030, sk=330, d=128)
    oc.check_attention(emu_lib, abi.F16, batch=1, heads=1, sq=1024, sk=256, d=128, qmul=5.0)
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=1, sq=1024, sk=320, d=128)      # 5 tiles: two left after the 3-step loop
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=1, sq=1024, sk=448, d=128, qmul=3.0)      # 7 tiles: one left


def test_attention_prescaled_q(emu_lib):
    """MTX_ATTN_Q_PRESCALED (what the FLUX graph passes): the long-sequence kernel runs without a per-score multiply-add and takes
    the row maximum only on the first tile or when a partial row sum explodes; short sequences go through the generic kernel"""
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=1, sq=1030, sk=330, d=128, qmul=3.0, prescaled=True)
    oc.check_attention(emu_lib, abi.F16, batch=1, heads=1, sq=1024, sk=320, d=128, prescaled=True)
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=1, sq=1024, sk=448, d=128, qmul=40.0, prescaled=True)      # logits of +-500: refresh path
    oc.check_attention(emu_lib, abi.F16, batch=1, heads=1, sq=1024, sk=449, d=128, qmul=40.0, prescaled=True)       # f16 probabilities: limit 3e4
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=2, sq=100, sk=130, d=64, prescaled=True)
    oc.check_attention(emu_lib, abi.BF16, batch=1, heads=1, sq=1024, sk=320, d=128, qmul=8.0, prescaled=True, late_keys=(200, 6.0))      # a forced stale maximum


def test_activation_epilogues_at_extreme_values(emu_lib):
    oc.check_gemm_act_extremes(emu_lib, abi.BF16)
    oc.check_gemm_act_extremes(emu_lib, abi.F16)


def test_first_block_cache_probe(emu_lib):
    """MTX_EW_RESIDUAL_DIST + MTX_EW_SUB (the first-block cache of the Kontext path, reference core/ml/model_manager.py:1159-1162)"""
    oc.check_residual_dist(emu_lib, abi.BF16, rows=70, c=3072)
    oc.check_residual_dist(emu_lib, abi.F16, rows=33, c=136, ld_extra=24, seed=1)


def test_attention_fp8_scores(emu_lib):
    """attn_mma32_k8_kernel / _k8q_kernel (mtx_attn_args.q_f8 / k_f8): scores from e4m3 q and k on the MX-scaled fp8 matrix instruction;
    the rotary kernel's e4m3 twin (mtx_ew_args.y8)"""
    oc.check_attention_f8_scores(emu_lib, abi.BF16, heads=2, sq=1030, sk=330)
    oc.check_attention_f8_scores(emu_lib, abi.F16, heads=1, sq=1024, sk=256, exponent=-2, seed=1)
    oc.check_attention_f8_scores(emu_lib, abi.BF16, heads=1, sq=1024, sk=320, qmul=4.0, late_keys=(200, 6.0), seed=2)      # a forced stale maximum
    oc.check_rope_f8_twin(emu_lib, abi.BF16, rows=70, heads=2)
    oc.check_rope_f8_twin(emu_lib, abi.F16, rows=33, heads=1, q_mul=1.0, seed=1)


def test_attention_fp8_scores_and_values(emu_lib):
    """attn_mma32_k8v8q_kernel (mtx_attn_args.v_f8t) + MTX_EW_V_F8T: P V on the fp8 instruction as well — ragged key counts, the key-split tail, a forced
    stale maximum (the stale-maximum path must be left before a probability leaves e4m3's range)"""
    oc.check_attention_f8_pv(emu_lib, abi.BF16, heads=2, sq=1030, sk=330)
    oc.check_attention_f8_pv(emu_lib, abi.F16, heads=1, sq=1024, sk=256, exponent=-2, seed=1)
    oc.check_attention_f8_pv(emu_lib, abi.BF16, heads=1, sq=1024, sk=320, qmul=4.0, late_keys=(200, 6.0), seed=2)


def test_attention_fp8_output(emu_lib):
    """the long-sequence kernel writing the MX fp8 operand of the next linear (mtx_attn_args.q8) == the same kernel -> mtx_quantize_mx, byte
    for byte and scale word for scale word; 10 query blocks on 3 simulated CUs: one goes through the key-split tail + the quantising merge"""
    oc.check_attention_q8(emu_lib, abi.BF16, heads=2, sq=1030, sk=330)
    oc.check_attention_q8(emu_lib, abi.F16, heads=1, sq=1024, sk=256, prescaled=False, col_off=128, extra_cols=128, seed=1)


def test_gemm_256_tile_map_strips(emu_lib, monkeypatch):
    """round 6: the workgroup -> tile map cut into 1 / 2 / 4 / 8 column strips (gemm256_tile_origin) is a bijection on ragged tile planes —
    3 x 5 and 5 x 3 tiles of 256 x 256, whole tiles + K-slice tail (3 simulated CUs) — for every forced strip count and the launcher's choice"""
    f = abi.GEMM_FORCE_TILE256
    for st in (0, 1, 2, 4, 8):
        monkeypatch.setenv("MTX_GEMM_STRIPS", str(st))
        oc.check_gemm(emu_lib, abi.BF16, m=700, n=1200, k=128, with_res=True, flags=f)
        oc.check_gemm(emu_lib, abi.F16, m=1100, n=600, k=64, act=abi.ACT_SILU, flags=f, seed=st)


def test_gemm_256_tile_kernel(emu_lib):
    """the 256 x 256 LDS-DMA kernel (normally used from 24 tiles up) on ragged small problems: ping-pong loop with descriptor-based
    LDS-DMA (range-checked zero fill), pieces spread 3/3/2/0"""
    f = abi.GEMM_FORCE_TILE256
    oc.check_gemm(emu_lib, abi.BF16, m=300, n=264, k=128, act=abi.ACT_GELU_TANH, with_res=True, with_gate=True, flags=f)
    oc.check_gemm(emu_lib, abi.F16, m=256, n=512, k=64, batch=2, alpha=0.5, with_bias=False, flags=f)
    oc.check_gemm(emu_lib, abi.BF16, m=260, n=250 // 8 * 8, k=448, with_res=True, flags=f)
    oc.check_gemm(emu_lib, abi.BF16, m=300, n=264, k=4160, with_res=True, flags=f)


def test_quantize_mx(emu_lib):
    """MX fp8 quantiser: bytes and E8M0 scale words bit-exact against the torch restatement"""
    oc.check_quantize_mx(emu_lib, abi.BF16, rows=37, k=256)
    oc.check_quantize_mx(emu_lib, abi.F16, rows=70, k=128, ld_extra=8, spread=1.0)
    oc.check_quantize_mx(emu_lib, abi.BF16, rows=3, k=1152, spread=8.0)


def test_gemm_fp8_tile_kernel(emu_lib):
    """fp8 (MX e4m3) 256-tile kernel: ragged M / N, 1, 2 and several K tiles, fused epilogue, K-slice tail (3 simulated CUs)"""
    f = abi.GEMM_FORCE_TILE256
    oc.check_gemm_f8(emu_lib, abi.BF16, m=300, n=264, k=128, act=abi.ACT_GELU_TANH, with_res=True, with_gate=True, flags=f)
    oc.check_gemm_f8(emu_lib, abi.F16, m=256, n=512, k=256, with_bias=False, flags=f, spread=1.5)
    oc.check_gemm_f8(emu_lib, abi.BF16, m=260, n=248, k=640, with_res=True, flags=f)
    oc.check_gemm_f8(emu_lib, abi.BF16, m=1024, n=1024, k=512, with_gate=True, flags=f)      # 16 tiles on 3 CUs: one left over -> tail + merge


def test_gemm_fp8_gated_epilogue(emu_lib):
    """SwiGLU + MX quantisation inside the fp8 GEMM's epilogue == fp8 GEMM -> SwiGLU quantiser, byte for byte (ragged m, rows and
    bytes landing inside a wider operand buffer, ungated columns in front as in FLUX.2's fused single-block projection)"""
    oc.check_gemm_f8_glu(emu_lib, abi.BF16, m=300, col0=0, hid=128, k=256)
    oc.check_gemm_f8_glu(emu_lib, abi.BF16, m=260, col0=256, hid=256, k=128, row_off=5, q_col_off=128, seed=1, spread=1.5)
    oc.check_gemm_f8_glu(emu_lib, abi.F16, m=256, col0=0, hid=128, k=384, seed=2)


def test_swiglu(emu_lib):
    oc.check_swiglu(emu_lib, abi.BF16, rows=37, hid=72)
    oc.check_swiglu(emu_lib, abi.F16, rows=5, hid=384)


def test_flux_prep_kernels(emu_lib):
    oc.check_qk_norm_rope(emu_lib, abi.BF16, rows=70, heads=3, d=64)
    oc.check_qk_norm_rope(emu_lib, abi.BF16, rows=37, heads=2, d=128, q_fold=0.1275)
    oc.check_qk_norm_rope(emu_lib, abi.F16, rows=33, heads=2, d=128, fused=False)
    oc.check_softmax_transpose(emu_lib, abi.BF16, rows=24, cols=40)
    oc.check_softmax_transpose(emu_lib, abi.F16, rows=70, cols=136)


def test_gemm_k_slice_tail(emu_lib):
    """tiles % CUs != 0 (the simulator reports 3 CUs): the left-over tile is cut into K slices whose last arriver sums the partials in
    slice order and runs the epilogue (gemm256_slice_kernel); a second and third run of the same plan must reproduce the first bit for
    bit (tickets back at zero, no dependence on which piece came last)"""
    f = abi.GEMM_FORCE_TILE256
    # 1024 x 256 -> 4 tiles, 4 % 3 = 1 left over, K = 4096 -> 64 iterations in 3 slices of 22 / 22 / 20
    oc.check_gemm(emu_lib, abi.BF16, m=1024, n=256, k=4096, act=abi.ACT_GELU_TANH, with_res=True, with_gate=True, flags=f, runs=3, expect_split=(3, 3, 3))
    # ragged M and N on the left-over tile (7 tiles on 3 CUs), f16
    oc.check_gemm(emu_lib, abi.F16, m=1700, n=200, k=3072, with_bias=False, flags=f, runs=2, expect_split=(6, "sliced", None))
    # short K: slicing does not pay, the left-over tiles run whole
    oc.check_gemm(emu_lib, abi.BF16, m=2048, n=1024, k=512, flags=f, expect_split=(32, 1, 0))


def test_gemm_k_slices_whole_problem(emu_lib):
    """few tiles, long K: every tile's K range is cut into slices (no full-tile launch at all)"""
    oc.check_gemm(emu_lib, abi.BF16, m=256, n=200, k=8192, with_res=True, with_gate=True, runs=2, expect_split=(0, "sliced", None))


@pytest.mark.parametrize("dtype", [abi.BF16, abi.F16])
def test_rcab_tail_pool_before_conv(emu_lib, dtype):
    """RCAN's RCAB tail: channel attention from the sums of conv2's INPUT, conv2 writing x + s * conv2(t) (interior and border tiles)"""
    oc.check_rcab_tail(emu_lib, dtype, n=2, h=37, w=29)
    oc.check_rcab_tail(emu_lib, dtype, n=1, h=50, w=52)
    oc.check_rcab_tail(emu_lib, dtype, n=1, h=21, w=40, canvas=(64, 64))


@pytest.mark.parametrize("dtype", [abi.BF16, abi.F16])
def test_conv_out_scale_and_sums_on_both_kernels(emu_lib, dtype):
    """out_scale + chan_sum with the same arguments on the persistent 64 -> 64 kernel (no residual) and on the generic kernel (with a
    residual, and at 32 -> 48 channels): the sums are those of act(conv + bias) before the scale on both (include/mtx_hip.h)"""
    oc.check_conv(emu_lib, dtype, n=2, h=21, w=19, cin=64, cout=64, ksize=3, stride=1, act=abi.ACT_RELU, with_sum=True, with_scale=True)
    oc.check_conv(emu_lib, dtype, n=2, h=21, w=19, cin=64, cout=64, ksize=3, stride=1, act=abi.ACT_RELU, with_sum=True, with_scale=True, with_res=True)
    oc.check_conv(emu_lib, dtype, n=1, h=17, w=23, cin=32, cout=48, ksize=3, stride=1, act=abi.ACT_SILU, with_sum=True, with_scale=True)


def test_producers_write_their_fp8_twins(emu_lib):
    """adaLN norm and SwiGLU quantise their own output for the fp8 linears: bit-identical to producer + mtx_quantize_mx"""
    oc.check_fused_quantisers(emu_lib, abi.BF16, rows=37, c=256, hid=128)
    oc.check_fused_quantisers(emu_lib, abi.F16, rows=70, c=1152, hid=384, seed=1)


def test_memset_op(emu_lib):
    oc.check_memset(emu_lib)


def test_f32_ops(emu_lib):
    assert oc.check_f32_ops(emu_lib) < 2e-5


def test_norm_f32_row_to_16_bit_operand(emu_lib):
    assert oc.check_norm_f32_to_16(emu_lib, abi.F16) < 1e-3
    assert oc.check_norm_f32_to_16(emu_lib, abi.BF16, rows=9) < 8e-3


def test_hi_lo_weight_pairs(emu_lib):
    e_fast, e_high = oc.check_hi_lo_weights(emu_lib)
    assert e_fast > 1e-5            # the weights' rounding is what the fast form is left with
