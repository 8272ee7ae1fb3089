"""Op-level parity checks of libmtx_hip against plain torch fp32 references of the same op.

Written once, run twice: on the CPU kernel simulator (tests/test_ops_sim.py, `-m "not gpu"`)
to validate kernel index arithmetic, and on a real MI355X through the product library
(tests/test_ops_gpu.py, `-m gpu`).  All calls go through the C ABI.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn.functional as F

from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.plan import Act, PlanBuilder

TD = {abi.BF16: torch.bfloat16, abi.F16: torch.float16}
TOL = {abi.BF16: 1e-2, abi.F16: 4e-3}   # max |error| relative to the output's max |value|; inputs are rounded to T first (bf16's own rounding step is 3.9e-3)


def _dev(lib):
    return torch.device("cpu") if lib.is_simulator else torch.device("cuda:0")


def _sync(lib):
    if not lib.is_simulator:
        torch.cuda.synchronize()


def _relerr(y, ref):
    return ((y.float() - ref.float()).abs().max() / ref.float().abs().max().clamp_min(1e-6)).item()


def _run(pb):
    plan = pb.build()
    plan.run()
    _sync(pb.lib)
    return plan


def check_conv(lib, dtype, n, h, w, cin, cout, ksize, stride, act=abi.ACT_NONE, with_res=False,
               pixel_shuffle=0, with_sum=False, ldx_extra=0, seed=0, with_scale=False):
    """with_scale: y = out_scale * act(conv + bias) (+ res); the channel sums stay those of act(conv + bias), BEFORE the scale — by either
    conv kernel (64 -> 64 with sums and a residual runs on the generic kernel, without the residual on the persistent one)"""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, ksize, ksize, generator=g) / math.sqrt(cin * ksize * ksize)
    b = torch.randn(cout, generator=g)
    xq, wq = x.to(td).float(), wt.to(td).float()
    ref = F.conv2d(xq.permute(0, 3, 1, 2), wq, b, stride=stride, padding=ksize // 2)
    if act == abi.ACT_RELU:
        ref = F.relu(ref)
    elif act == abi.ACT_SILU:
        ref = F.silu(ref)
    elif act == abi.ACT_LEAKY:
        ref = F.leaky_relu(ref, 0.1)
    act_sum = ref.sum(dim=(2, 3)) if with_sum else None
    osc = None
    if with_scale:
        osc = torch.rand(n, cout, generator=g) * 1.5 + 0.25
        ref = ref * osc[:, :, None, None]
    if pixel_shuffle:
        ref = F.pixel_shuffle(ref, pixel_shuffle)
    res = None
    if with_res:
        res = torch.randn(ref.shape, generator=g).to(td).float()
        ref = ref + 0.5 * res
    ref = ref.permute(0, 2, 3, 1)

    pb = PlanBuilder(lib, dev, dtype)
    xb = pb.act(n, h, w, cin, ld=cin + ldx_extra)
    xb.t[..., :cin] = xq.to(td)
    if ldx_extra:
        xb.t[..., cin:] = 7.0   # garbage in the unused channels must not leak
    wp = wt
    bp = b
    if pixel_shuffle:
        c = cout // 4
        wp = wt.view(c, 4, cin, ksize, ksize).permute(1, 0, 2, 3, 4).reshape(cout, cin, ksize, ksize)
        bp = b.view(c, 4).t().reshape(-1)
    wpk = pb.const(wp.permute(0, 2, 3, 1).reshape(cout, ksize * ksize, cin), td)
    bias = pb.const(bp, torch.float32)
    rb = None
    if with_res:
        rb = pb.act(ref.shape[0], ref.shape[1], ref.shape[2], ref.shape[3])
        rb.t.copy_(res.permute(0, 2, 3, 1).to(td))
    cs = None
    if with_sum:
        tiles = pb.conv_tiles(xb, ksize, stride, cout=cout, with_res=with_res, with_scale=with_scale, act=act, pixel_shuffle=pixel_shuffle)
        cs = pb.buf((n, tiles, cout), torch.float32, zero=True)
        cs.fill_(777.0)          # the conv owns every row: stale values must not survive a launch (no memset in front of it)
    y = pb.conv2d(xb, wpk, bias, cout, ksize, stride, act=act, act_param=0.1, res=rb, res_scale=0.5,
                  pixel_shuffle=pixel_shuffle, chan_sum=cs, out_scale=pb.const(osc, torch.float32) if with_scale else None)
    _run(pb)
    err = _relerr(y.torch().cpu(), ref)
    assert err < TOL[dtype], f"conv mismatch rel err {err}"
    if with_sum:
        s = cs.sum(dim=1).cpu()
        e2 = ((s - act_sum).abs().max() / act_sum.abs().max()).item()
        assert e2 < TOL[dtype], f"chan_sum mismatch {e2}"
    return err


def check_gemm(lib, dtype, m, n, k, act=abi.ACT_NONE, with_bias=True, with_res=False, with_gate=False,
               out_f32=False, batch=1, alpha=1.0, seed=0, flags=0, runs=1, expect_split=None, keep=None):
    """runs > 1: the same plan is run again over a poisoned output — a K-slice launch must not depend on what an earlier launch left in its
    scratch (and must leave its tickets at zero).  expect_split = (whole tiles, K slices, tail pieces) the launch must report
    (mtx_gemm_last_split): the test shape really went through the path it is meant to cover."""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    a = torch.randn(batch, m, k, generator=g).to(td)
    w = (torch.randn(batch, n, k, generator=g) / math.sqrt(k)).to(td)
    b = torch.randn(n, generator=g) if with_bias else None
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float()) * alpha
    if b is not None:
        ref = ref + b
    if act == abi.ACT_GELU:
        ref = F.gelu(ref)
    elif act == abi.ACT_GELU_TANH:
        ref = F.gelu(ref, approximate="tanh")
    elif act == abi.ACT_SILU:
        ref = F.silu(ref)
    gate = res = None
    rows_per = max(m // 2, 1)
    if with_gate:
        gate = torch.randn((m + rows_per - 1) // rows_per, n, generator=g).to(td)
        ref = ref * gate.float().repeat_interleave(rows_per, dim=0)[:m]
    if with_res:
        res = torch.randn(batch, m, n, generator=g).to(td)
        ref = ref + res.float()
    pb = PlanBuilder(lib, dev, dtype)
    at, wt = pb.const(a), pb.const(w)
    out = pb.gemm(at, wt, m, n, k, bias=pb.const(b) if b is not None else None, act=act,
                  res=pb.const(res) if res is not None else None,
                  gate=pb.const(gate) if gate is not None else None, gate_rows_per=rows_per,
                  alpha=alpha, batch=batch, a_bs=m * k, w_bs=n * k, c_bs=m * n, out_f32=out_f32, flags=flags)
    plan = _run(pb)
    if expect_split is not None:
        got = lib.gemm_last_split()           # None in expect_split = any value; "sliced" = at least two K slices
        ok = all(e is None or (e == "sliced" and g >= 2) or e == g for e, g in zip(expect_split, got))
        assert ok, f"launch split {got} != {tuple(expect_split)}"
    err = _relerr(out.cpu().view(batch, m, n), ref)
    assert err < TOL[dtype], f"gemm mismatch rel err {err}"
    first = out.clone()
    if keep is not None:                    # the caller compares the bytes of several launch forms
        keep.append(first.cpu())
    for _ in range(runs - 1):
        out.fill_(float("nan"))
        plan.run()
        _sync(lib)
        assert torch.equal(out, first), "a second run of the same plan differs (stale scratch read, or an order-dependent sum)"
    return err


def check_gemm_huge_rows(lib, dtype, m, n, k, seed=0):
    """A matrix larger than 4 GB (a descriptor's reach): the rows are generated on the device and row blocks at the start, at the end
    and around the 4 GB byte offset are compared with a torch fp32 product of those rows"""
    dev, td = _dev(lib), TD[dtype]
    g = torch.Generator(device=dev).manual_seed(seed)
    pb = PlanBuilder(lib, dev, dtype)
    at = pb.buf((m, k), td)
    step = 1 << 16
    for r0 in range(0, m, step):                       # filled in slices: a 4.5 GB fp32 temporary per call otherwise
        at[r0:r0 + step] = torch.randn((min(step, m - r0), k), device=dev, generator=g).to(td)
    w = (torch.randn(n, k, generator=torch.Generator().manual_seed(seed + 1)) / math.sqrt(k)).to(td)
    wt = pb.const(w)
    out = pb.gemm(at, wt, m, n, k)
    _run(pb)
    edge = (1 << 32) // (k * at.element_size())        # first row whose bytes start beyond 4 GB
    for r0 in (0, edge - 300, m - 517):
        rows = slice(max(r0, 0), min(max(r0, 0) + 517, m))
        ref = at[rows].float() @ wt.float().t()
        err = _relerr(out[rows].float().cpu(), ref.cpu())
        assert err < TOL[dtype], f"gemm rows {rows} mismatch rel err {err}"


def check_attention(lib, dtype, batch, heads, sq, sk, d, seed=0, qmul=1.0, prescaled=False, late_keys=None):
    """prescaled: q carries scale * log2(e) before its rounding to the storage type (MTX_ATTN_Q_PRESCALED): the reference is the
    base-2 softmax of q k^T, i.e. SDPA with scale = ln 2 on the very same rounded q"""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    scale = 1.0 / math.sqrt(d)
    q = torch.randn(batch, sq, heads, d, generator=g) * qmul              # qmul > 1: peaked rows, exercises max refreshes
    q = (q * (scale * 1.4426950408889634)).to(td) if prescaled else q.to(td)
    k = torch.randn(batch, sk, heads, d, generator=g)
    if late_keys is not None:                                              # (first key, factor): keys from there on score far above everything before them
        k[:, late_keys[0]:] *= late_keys[1]
    k = k.to(td)
    v = torch.randn(batch, sk, heads, d, generator=g).to(td)
    ref = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2),
                                         v.float().transpose(1, 2), scale=math.log(2.0) if prescaled else scale).transpose(1, 2)
    pb = PlanBuilder(lib, dev, dtype)
    qt, kt, vt = pb.const(q), pb.const(k), pb.const(v)
    o = pb.buf((batch, sq, heads, d), td, zero=True)
    pb.attention(qt, kt, vt, o, batch, heads, sq, sk, d,
                 (sq * heads * d, heads * d, d), (sk * heads * d, heads * d, d),
                 (sk * heads * d, heads * d, d), (sq * heads * d, heads * d, d), scale, q_prescaled=prescaled)
    _run(pb)
    err = _relerr(o.cpu(), ref)
    assert err < TOL[dtype] * 1.5, f"attention mismatch rel err {err}"
    return err


def check_attention_f8_scores(lib, dtype, heads, sq, sk, seed=0, exponent=-3, qmul=1.0, late_keys=None):
    """the long-sequence kernel with fp8 scores (mtx_attn_args.q_f8 / k_f8): q and k as plain e4m3 rows, logits 2^exponent * q k^T.  The
    reference is the base-2 softmax of exactly those products (e4m3 products are exact in fp32) times the 16-bit v; and the same rows
    through the MX fp8 output form must give the bytes of the two launches it replaces."""
    d = 128
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    D = heads * d
    q8 = (torch.randn(sq, heads, d, generator=g) * qmul).clamp(-448, 448).to(torch.float8_e4m3fn)
    k = torch.randn(sk, heads, d, generator=g)
    if late_keys is not None:
        k[late_keys[0]:] *= late_keys[1]
    k8 = k.clamp(-448, 448).to(torch.float8_e4m3fn)
    v = torch.randn(1, sk, heads, d, generator=g).to(td)
    logits = torch.einsum("qhd,khd->hqk", q8.float(), k8.float()) * (2.0 ** exponent)
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(logits * math.log(2.0), dim=-1), v[0].float())
    rows = max(sq, sk)
    packed = torch.zeros(rows, 2 * D, dtype=torch.uint8)                      # [row][q bytes | k bytes]: the layout the rotary kernel leaves
    packed[:sq, :D] = q8.view(torch.uint8).reshape(sq, D)
    packed[:sk, D:] = k8.view(torch.uint8).reshape(sk, D)
    pb = PlanBuilder(lib, dev, dtype)
    unused = pb.buf((1, rows, heads, d), td, zero=True)                      # q / k are not read in this form
    vt, pk = pb.const(v), pb.const(packed)
    o = pb.buf((1, sq, heads, d), td, zero=True)
    strides = ((rows * D, D, d), (rows * D, D, d), (sk * D, D, d), (sq * D, D, d))
    pb.attention(unused, unused, vt, o, 1, heads, sq, sk, d, *strides, 1.0, q_prescaled=True, qk_f8=(pk, 0, D, 2 * D, exponent))
    lds = (sq + 63) // 64 * 64
    o8a, sca = pb.buf((sq, D), torch.uint8, zero=True), pb.buf((D // 128, lds), torch.int32, zero=True)
    o8b, scb = pb.buf((sq, D), torch.uint8, zero=True), pb.buf((D // 128, lds), torch.int32, zero=True)
    pb.attention(unused, unused, vt, None, 1, heads, sq, sk, d, *strides, 1.0, q_prescaled=True, qk_f8=(pk, 0, D, 2 * D, exponent), q8=(o8a, sca, D, lds, 0))
    pb.quantize(o, sq, D, q=o8b, scale=scb, lds=lds, ldq=D)
    _run(pb)
    err = _relerr(o.cpu()[0], ref)
    assert err < TOL[dtype] * 1.5, f"attention with fp8 scores: rel err {err}"
    a, b = o8a.cpu().numpy(), o8b.cpu().numpy()
    assert a.any() and np.array_equal(a, b), f"fp8 scores + fp8 output: {(a != b).sum()} of {a.size} e4m3 bytes differ from attention + quantiser"
    assert np.array_equal(sca.cpu().numpy(), scb.cpu().numpy())
    return err


def v_f8t_ref(v: torch.Tensor):
    """MTX_EW_V_F8T restated: v [rows, heads, 128] -> uint8 [heads * 128, ld], ld = rows padded to 64, byte j of a 64-key tile = key
    32 (j >> 5) + (j & 3) + 8 ((j & 15) >> 2) + 4 ((j >> 4) & 1), zeros past the rows"""
    rows, heads, d = v.shape
    ld = (rows + 63) // 64 * 64
    q = torch.zeros(ld, heads, d, dtype=torch.uint8)
    q[:rows] = v.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    j = torch.arange(64)
    key = 32 * (j >> 5) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * ((j >> 4) & 1)
    tiles = q.view(ld // 64, 64, heads, d)[:, key]                      # [tile, byte j, head, d]
    return tiles.permute(2, 3, 0, 1).reshape(heads * d, ld).contiguous(), ld


def check_attention_f8_pv(lib, dtype, heads, sq, sk, seed=0, exponent=-3, qmul=1.0, late_keys=None, tol=0.045):
    """fp8 scores + fp8 P V (mtx_attn_args.v_f8t): the V^T operand the producer kernel writes is byte-identical to its restatement; the output rows —
    MX fp8, dequantised here — are compared with the exact softmax over the e4m3 products times the values (which are e4m3-representable in this
    test, so both P V forms see the same numbers) by their rms error relative to the rms of the reference.  The rows' own e4m3 rounding is 2-3 % by that
    measure (the 16-bit-P-V kernel's figure, asserted beside it); the probabilities' rounding to e4m3 may add at most 1.5 points."""
    d = 128
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    D = heads * d
    q8 = (torch.randn(sq, heads, d, generator=g) * qmul).clamp(-448, 448).to(torch.float8_e4m3fn)
    k = torch.randn(sk, heads, d, generator=g)
    if late_keys is not None:
        k[late_keys[0]:] *= late_keys[1]
    k8 = k.clamp(-448, 448).to(torch.float8_e4m3fn)
    v = torch.randn(sk, heads, d, generator=g).to(torch.float8_e4m3fn).float().to(td)          # e4m3-representable values in the 16-bit type
    logits = torch.einsum("qhd,khd->hqk", q8.float(), k8.float()) * (2.0 ** exponent)
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(logits * math.log(2.0), dim=-1), v.float())
    rows = max(sq, sk)
    packed = torch.zeros(rows, 2 * D, dtype=torch.uint8)
    packed[:sq, :D] = q8.view(torch.uint8).reshape(sq, D)
    packed[:sk, D:] = k8.view(torch.uint8).reshape(sk, D)
    lds = (sq + 63) // 64 * 64
    errs = []
    for fp8_pv in (False, True):
        pb = PlanBuilder(lib, dev, dtype)
        unused = pb.buf((1, rows, heads, d), td, zero=True)
        vt, pk = pb.const(v.view(1, sk, heads, d)), pb.const(packed)
        vt8, ldv = pb.v_f8t(vt, sk, heads, D)
        o8, sc = pb.buf((sq, D), torch.uint8, zero=True), pb.buf((D // 128, lds), torch.int32, zero=True)
        strides = ((rows * D, D, d), (rows * D, D, d), (sk * D, D, d), (sq * D, D, d))
        pb.attention(unused, unused, vt, None, 1, heads, sq, sk, d, *strides, 1.0, q_prescaled=True, qk_f8=(pk, 0, D, 2 * D, exponent), q8=(o8, sc, D, lds, 0),
                     pv_f8=(vt8, ldv) if fp8_pv else None)
        _run(pb)
        if fp8_pv:
            want_vt, ld_ref = v_f8t_ref(v)
            assert ld_ref == ldv and np.array_equal(vt8.cpu().numpy(), want_vt.numpy()), "MTX_EW_V_F8T: bytes differ from the restatement"
        # dequantise the MX fp8 rows: byte * 2^(scale byte - 127), scale word [head][row], one byte per 32-column block of the head
        ob = o8.cpu().view(torch.float8_e4m3fn).float().view(sq, heads, 4, 32)
        sw = sc.cpu().numpy().astype(np.uint32)[:, :sq]                                           # [heads, sq]
        eb = np.stack([(sw >> (8 * i)) & 0xff for i in range(4)], -1).astype(np.int32)            # [heads, sq, 4]
        scale = torch.from_numpy(np.exp2((eb - 127).astype(np.float32))).permute(1, 0, 2)          # [sq, heads, 4]
        got = (ob * scale[..., None]).view(sq, heads, d)
        assert torch.isfinite(got).all()
        errs.append(float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
    assert errs[0] < 0.035, f"fp8 scores, 16-bit P V, MX fp8 rows: rms rel err {errs[0]}"
    assert errs[1] < tol and errs[1] < errs[0] + 0.015, f"fp8 P V: rms rel err {errs[1]} against {errs[0]} with 16-bit P V"
    return errs


def check_rope_f8_twin(lib, dtype, rows, heads, seed=0, q_mul=8.0):
    """MTX_EW_QK_NORM_ROPE with mtx_ew_args.y8: the e4m3 twin is RNE_e4m3(clamp(y * (q_mul on the q heads))) of the 16-bit values the
    same launch stores, byte for byte"""
    d = 128
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    c = 2 * heads * d
    x = (torch.randn(rows, c, generator=g) * 3.0).to(td)
    gamma = torch.randn(2 * d, generator=g).abs() + 0.5
    ang = torch.rand(rows, d // 2, generator=g) * 6.28
    cs = torch.cat([torch.cos(ang), torch.sin(ang)], 1).contiguous()          # [rows][2][d/2]
    pb = PlanBuilder(lib, dev, dtype)
    xt, gt, ct = pb.const(x), pb.const(gamma), pb.const(torch.cat([cs, cs * 0.1275], 0))
    y8 = pb.buf((rows, c), torch.uint8, zero=True)
    e = abi.EwArgs()
    e.a, e.b, e.s, e.y = xt.data_ptr(), ct.data_ptr(), gt.data_ptr(), xt.data_ptr()
    e.n, e.h, e.w, e.c = 1, 1, rows, c
    e.lda, e.ldb, e.ldy, e.lds = c, rows * d, c, 0
    e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_QK_NORM_ROPE, 0, 1e-6, d, heads, dtype
    e.y8, e.ldy8, e.y8_mul = y8.data_ptr(), c, q_mul
    pb._add(abi.OP_EW, e, "rope")
    _run(pb)
    y = xt.cpu().float()
    assert torch.isfinite(y).all() and y.abs().max() > 0.1
    y[:, :heads * d] *= q_mul
    want = y.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = y8.cpu().numpy()
    assert np.array_equal(got, want), f"rotary fp8 twin: {(got != want).sum()} of {got.size} bytes differ"


def check_attention_q8(lib, dtype, heads, sq, sk, seed=0, prescaled=True, col_off=0, extra_cols=0):
    """the long-sequence attention kernel with MX fp8 output (mtx_attn_args.q8) against the two launches it replaces — the same kernel
    into a 16-bit [sq, heads * 128] matrix, then mtx_quantize_mx — on the same operands: e4m3 bytes and scale words IDENTICAL, also for
    the query blocks that go through the key-split tail (their rows are quantised by the merge kernel).  col_off / extra_cols: the
    result lands inside a wider operand buffer (FLUX.2's single-block concatenation)."""
    d = 128
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    scale = 1.0 / math.sqrt(d)
    q = torch.randn(1, sq, heads, d, generator=g)
    q = (q * (scale * 1.4426950408889634)).to(td) if prescaled else q.to(td)
    k = torch.randn(1, sk, heads, d, generator=g).to(td)
    v = (torch.randn(1, sk, heads, d, generator=g) * torch.exp(torch.randn(1, sk, heads, 1, generator=g))).to(td)
    D = heads * d
    ldq = col_off + D + extra_cols
    lds = (sq + 63) // 64 * 64
    outs = []
    for fused in (False, True):
        pb = PlanBuilder(lib, dev, dtype)
        qt, kt, vt = pb.const(q), pb.const(k), pb.const(v)
        q8 = pb.buf((sq, ldq), torch.uint8, zero=True)
        sc = pb.buf((ldq // 128, lds), torch.int32, zero=True)
        strides = ((sq * D, D, d), (sk * D, D, d), (sk * D, D, d), (sq * D, D, d))
        if fused:
            pb.attention(qt, kt, vt, None, 1, heads, sq, sk, d, *strides, scale, q_prescaled=prescaled, q8=(q8, sc, ldq, lds, col_off))
        else:
            o = pb.buf((sq, D), td, zero=True)
            pb.attention(qt, kt, vt, o, 1, heads, sq, sk, d, *strides, scale, q_prescaled=prescaled)
            pb.quantize(o, sq, D, q=q8, scale=sc, lds=lds, ldq=ldq, q_col_off=col_off)
        _run(pb)
        outs.append((q8.cpu().numpy().copy(), sc.cpu().numpy().copy()))
    (q0, s0), (q1, s1) = outs
    assert q0.any() and s0.any()
    assert np.array_equal(q0, q1), f"attention with fp8 output: {(q0 != q1).sum()} of {q0.size} e4m3 bytes differ"
    assert np.array_equal(s0, s1), f"attention with fp8 output: {(s0 != s1).sum()} scale words differ"


def check_norm(lib, dtype, rows, c, kind=0, affine=True, modulate=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = (torch.randn(rows, c, generator=g) * 2 + 0.5).to(td)
    gamma = torch.randn(c, generator=g) if affine else None
    beta = torch.randn(c, generator=g) if (affine and kind == 0) else None
    xf = x.float()
    if kind == 0:
        ref = F.layer_norm(xf, (c,), gamma, beta, 1e-6)
    else:
        ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
        if gamma is not None:
            ref = ref * gamma
    ms = mh = None
    rows_per = max(rows // 2, 1)
    if modulate:
        nm = (rows + rows_per - 1) // rows_per
        ms = torch.randn(nm, c, generator=g).to(td)
        mh = torch.randn(nm, c, generator=g).to(td)
        ref = ref * (1 + ms.float().repeat_interleave(rows_per, 0)[:rows]) + mh.float().repeat_interleave(rows_per, 0)[:rows]
    pb = PlanBuilder(lib, dev, dtype)
    xt = pb.const(x)
    y = pb.buf((rows, c), td)
    pb.norm(xt, y, rows, c, gamma=pb.const(gamma) if gamma is not None else None,
            beta=pb.const(beta) if beta is not None else None, eps=1e-6, kind=kind,
            mod_scale=pb.const(ms) if ms is not None else None, mod_shift=pb.const(mh) if mh is not None else None,
            rows_per=rows_per if modulate else 0, ldmod=c if modulate else 0)
    plan = _run(pb)
    err = _relerr(y.cpu(), ref)
    assert err < TOL[dtype], f"norm mismatch rel err {err}"
    return err


def check_groupnorm(lib, dtype, n, h, w, c, groups, silu=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = (torch.randn(n, h, w, c, generator=g) * 1.5 + 0.3).to(td)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    ref = F.group_norm(x.float().permute(0, 3, 1, 2), groups, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    pb = PlanBuilder(lib, dev, dtype)
    xa = pb.act(n, h, w, c)
    xa.t.copy_(x)
    y = pb.groupnorm(xa, pb.const(gamma), pb.const(beta), groups, 1e-6, abi.ACT_SILU if silu else abi.ACT_NONE)
    _run(pb)
    err = _relerr(y.torch().cpu(), ref)
    assert err < TOL[dtype], f"groupnorm mismatch rel err {err}"
    return err


def check_residual_dist(lib, dtype, rows=700, c=3072, seed=0, ld_extra=0):
    """MTX_EW_RESIDUAL_DIST (first-block cache probe) and MTX_EW_SUB against torch on the same rounded operands; the probe twice: identical parts"""
    from mangatranslator_amd.hip.plan import residual_distance
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    ld = c + ld_extra
    before = torch.randn(rows, ld, generator=g).to(td)
    after = (before.float() + 0.3 * torch.randn(rows, ld, generator=g)).to(td)
    prev = ((after.float() - before.float())[:, :c] + 0.05 * torch.randn(rows, c, generator=g)).to(td)
    r = (after.float() - before.float()).to(td).float()[:, :c]
    want = ((prev.float() - r).abs().sum() / prev.float().abs().sum()).item()
    pb = PlanBuilder(lib, dev, dtype)
    a_t, b_t, p_t = pb.const(after), pb.const(before), pb.const(prev)
    parts = pb.residual_dist(a_t, b_t, p_t, rows, c, ld=ld)
    parts2 = pb.residual_dist(a_t, b_t, p_t, rows, c, ld=ld)
    diff = pb.ew(abi.EW_SUB, Act(a_t.view(1, 1, rows, ld), 1, 1, rows, c), b=Act(b_t.view(1, 1, rows, ld), 1, 1, rows, c), label="sub")
    _run(pb)
    got = residual_distance(parts)
    assert abs(got - want) < 1e-4 * max(want, 1e-6) + 1e-6, (got, want)
    assert torch.equal(parts.cpu(), parts2.cpu())
    assert torch.equal(diff.t.reshape(rows, c).cpu().float(), r)
    return got


def check_ew(lib, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    n, h, w, c = 2, 5, 7, 24
    a = torch.randn(n, h, w, c, generator=g).to(td)
    b = torch.randn(n, h, w, c, generator=g).to(td)
    s = torch.rand(n, c, generator=g)
    pb = PlanBuilder(lib, dev, dtype)
    A, B = pb.act(n, h, w, c), pb.act(n, h, w, c)
    A.t.copy_(a); B.t.copy_(b)
    y1 = pb.ew(abi.EW_SCALE_RES, A, b=B, s=pb.const(s), lds=c)
    y2 = pb.ew(abi.EW_UPSAMPLE2X, A)
    y3 = pb.ew(abi.EW_MAXPOOL, A, i0=5, i1=1)
    y4 = pb.ew(abi.EW_MAXPOOL, A, i0=2, i1=2)
    y5 = pb.ew(abi.EW_ADD, A, b=B, act=abi.ACT_SILU)
    gs = torch.randn(n, c, generator=g).to(td)
    y6 = pb.ew(abi.EW_GATE_RES, A, b=B, s=pb.const(gs), lds=c)
    _run(pb)
    af, bf = a.float(), b.float()
    assert _relerr(y1.torch().cpu(), af * s[:, None, None, :] + bf) < TOL[dtype]
    assert _relerr(y2.torch().cpu(), af.repeat_interleave(2, 1).repeat_interleave(2, 2)) < 1e-6
    mp = F.max_pool2d(af.permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1)
    assert _relerr(y3.torch().cpu(), mp) < 1e-6
    mp2 = F.max_pool2d(af.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert y4.torch().shape == mp2.shape and _relerr(y4.torch().cpu(), mp2) < 1e-6
    assert _relerr(y5.torch().cpu(), F.silu(af + bf)) < TOL[dtype]
    assert _relerr(y6.torch().cpu(), bf + af * gs.float()[:, None, None, :]) < TOL[dtype]


def check_resize_threshold(lib, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev = _dev(lib)
    src = torch.randn(3, 16, 16, generator=g)
    hd, wd = 37, 29
    ref = F.interpolate(src[None], (hd, wd), mode="bilinear", align_corners=False)[0] > 0.0
    pb = PlanBuilder(lib, dev, abi.BF16)
    st = pb.const(src)
    dst = pb.buf((3, hd, wd), torch.uint8, zero=True)
    pb.resize_threshold(st, dst, 3, 16, 16, hd, wd, 0.0, abi.F32)
    _run(pb)
    mism = (dst.cpu().bool() != ref).float().mean().item()
    # ties at exactly 0 are measure-zero; interpolation order differences may flip ~1e-7 logits
    assert mism < 2e-3, f"mask mismatch fraction {mism}"
    return mism


def check_image_convert(lib, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = torch.rand(1, 3, 6, 8, generator=g)
    pb = PlanBuilder(lib, dev, dtype)
    xt = pb.const(x)
    a = pb.act(1, 3, 4, 16)
    pb.image_convert(abi.IMG_NCHW_F32_TO_NHWC, xt, a.t, 1, 6, 8, 16, unshuffle=2, mul=2.0, add=(0.1, 0.2, 0.3))
    b = pb.act(1, 6, 8, 8)
    pb.image_convert(abi.IMG_NCHW_F32_TO_NHWC, xt, b.t, 1, 6, 8, 8, unshuffle=1, mul=1.0, add=(0, 0, 0))
    back = pb.buf((1, 3, 6, 8), torch.float32)
    pb.image_convert(abi.IMG_NHWC_TO_NCHW_F32, b.t, back, 1, 6, 8, 8, mul=1.0, add=(0, 0, 0))
    u8 = pb.buf((1, 6, 8, 3), torch.uint8)
    pb.image_convert(abi.IMG_NHWC_TO_HWC_U8, b.t, u8, 1, 6, 8, 8, mul=1.0, add=(0, 0, 0))
    _run(pb)
    add = torch.tensor([0.1, 0.2, 0.3]).view(1, 3, 1, 1)
    ref = F.pixel_unshuffle(x * 2.0 + add, 2).permute(0, 2, 3, 1)
    assert _relerr(a.t.cpu()[..., :12], ref) < TOL[dtype]
    assert a.t.cpu()[..., 12:].abs().max() == 0
    assert _relerr(back.cpu(), x) < TOL[dtype]
    xq = x.to(td).float()
    ref8 = (xq.permute(0, 2, 3, 1).clamp(0, 1) * 255).to(torch.uint8)
    assert (u8.cpu().int() - ref8.int()).abs().max() <= 0


def check_qk_norm_rope(lib, dtype, rows, heads, d, fused=True, seed=0, q_fold=None):
    """per-head RMSNorm * gamma then rotary on interleaved pairs, in place on the q|k column slices of a [rows, 3*H*d] buffer"""
    from mangatranslator_amd.hip.plan import Act
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    D = heads * d
    qkv = torch.randn(rows, 3 * D, generator=g).to(td)
    gam = 1.0 + 0.2 * torch.randn(2, d, generator=g)
    ang = torch.rand(rows, d // 2, generator=g) * 6.28
    cs = torch.stack([ang.cos(), ang.sin()], 1).contiguous()            # [rows, 2, d/2]
    ref = qkv.float().clone()
    for part in range(2):
        x = ref[:, part * D:(part + 1) * D].reshape(rows, heads, d)
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * gam[part]
        x = x.to(td).float()                                             # the model rounds before the rotary product
        x0, x1 = x[..., 0::2], x[..., 1::2]
        c, s_ = ang.cos()[:, None], ang.sin()[:, None]
        if q_fold is not None and part == 0:                                # q heads rotate with the pre-scaled table copy
            c, s_ = c * q_fold, s_ * q_fold
        ref[:, part * D:(part + 1) * D] = torch.stack([x0 * c - x1 * s_, x1 * c + x0 * s_], -1).reshape(rows, D)
    pb = PlanBuilder(lib, dev, dtype)
    if q_fold is not None:
        assert fused
        cs = torch.stack([cs, cs * q_fold]).contiguous()
    buf, cst, gm = pb.const(qkv), pb.const(cs), pb.const(gam.reshape(-1).contiguous())
    def add(col, ncols, gamma_off, split):
        e = abi.EwArgs()
        e.a = e.y = buf.data_ptr() + col * buf.element_size()
        e.b, e.s = cst.data_ptr(), gm.data_ptr() + gamma_off * 4
        e.n, e.h, e.w, e.c = 1, 1, rows, ncols
        e.lda = e.ldy = 3 * D
        e.ldb = rows * d if q_fold is not None else 0
        e.kind, e.act_param, e.i0, e.i1, e.dtype = abi.EW_QK_NORM_ROPE, 1e-6, d, split, dtype
        pb._add(abi.OP_EW, e, "rope")
    if fused:
        add(0, 2 * D, 0, heads)
    else:
        add(0, D, 0, 0); add(D, D, d, 0)
    _run(pb)
    err = _relerr(buf.cpu()[:, :2 * D], ref[:, :2 * D])
    assert err < TOL[dtype], f"qk_norm_rope mismatch rel err {err}"
    assert torch.equal(buf.cpu()[:, 2 * D:], qkv[:, 2 * D:]), "v slice must stay untouched"
    return err


def check_softmax_transpose(lib, dtype, rows, cols, seed=0):
    from mangatranslator_amd.hip.plan import Act
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = (torch.randn(rows, cols, generator=g) * 3).to(td)
    pb = PlanBuilder(lib, dev, dtype)
    xt = pb.const(x)
    a = Act(xt.view(1, 1, rows, cols), 1, 1, rows, cols)
    sm = pb.ew(abi.EW_SOFTMAX_ROWS, a, act_param=0.37)
    rp = (rows + 7) // 8 * 8
    tr = pb.buf((cols, rp), td, zero=True)
    pb.ew(abi.EW_TRANSPOSE, a, out=Act(tr.view(1, 1, cols, rp), 1, 1, cols, rp))
    _run(pb)
    ref = torch.softmax(x.float() * 0.37, -1)
    err = _relerr(sm.t.cpu().view(rows, cols), ref)
    assert err < TOL[dtype], f"softmax_rows mismatch rel err {err}"
    assert torch.equal(tr.cpu()[:, :rows], x.t()), "transpose must be exact"
    return err


# ---- fp8 path (BASELINE.json config 5) --------------------------------------------------------------------------------
def mx_quantize_ref(x: torch.Tensor):
    """torch restatement of mtx_quantize_mx (include/mtx_hip.h): per 32 k one E8M0 exponent eb = the smallest with
    2^(eb - 127) >= amax / 448 (fp32 arithmetic as in the kernel), q = RNE_e4m3(x * 2^(127 - eb)).
    -> (q uint8 [rows, k], eb int32 [rows, k / 32], dequantised fp32 [rows, k])"""
    rows, k = x.shape
    xb = x.float().view(rows, k // 32, 32)
    amax = xb.abs().amax(-1)
    r = (amax * torch.tensor(1.0 / 448.0, dtype=torch.float32)).contiguous()
    u = r.view(torch.int32)
    eb = ((u >> 23) & 0xff) + ((u & 0x7fffff) != 0).int()
    eb = torch.where(amax == 0, torch.full_like(eb, 127), eb.clamp(1, 253))
    inv = torch.ldexp(torch.ones_like(amax), 127 - eb)
    q8 = (xb * inv[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    deq = q8.float() * torch.ldexp(torch.ones_like(amax), eb - 127)[..., None]
    return q8.view(torch.uint8).view(rows, k), eb, deq.view(rows, k)


def check_quantize_mx(lib, dtype, rows, k, ld_extra=0, seed=0, spread=4.0):
    """bytes and scale words bit-exact against the torch restatement (inputs with a wide dynamic range across blocks)"""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = torch.randn(rows, k, generator=g) * torch.exp(spread * torch.randn(rows, k // 32, generator=g)).repeat_interleave(32, 1)
    x[0, :32] = 0.0                                   # an all-zero block
    x = x.to(td)
    q_ref, eb_ref, _ = mx_quantize_ref(x)
    pb = PlanBuilder(lib, dev, dtype)
    xt = pb.buf((rows, k + ld_extra), td)
    xt[:, :k] = x.to(dev)
    q, sc, lds = pb.quantize(xt, rows, k, ldx=k + ld_extra)
    _run(pb)
    assert torch.equal(q.cpu(), q_ref), f"fp8 bytes differ in {(q.cpu() != q_ref).sum().item()} places"
    words = sc.cpu()[:, :rows].t().contiguous()                    # [rows, k/128] int32
    got = torch.stack([(words >> (8 * b)) & 0xff for b in range(4)], -1).view(rows, k // 32)
    assert torch.equal(got.int(), eb_ref.int()), "scale bytes differ"


def check_fused_quantisers(lib, dtype, rows, c, hid, seed=0):
    """The fp8 twins written by the producers themselves — adaLN LayerNorm (mtx_norm_args.q) and SwiGLU (MTX_QUANT_SWIGLU) — must be
    bit-identical to the two-pass form (producer writes 16-bit, mtx_quantize_mx reads it back): bytes, scale words, and the optional
    16-bit copies; rows land at a row offset inside larger twin buffers, as in the FLUX.2 graph."""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = (torch.randn(rows, c, generator=g) * torch.exp(1.5 * torch.randn(rows, 1, generator=g)) + 0.3).to(td)
    ms, mh = (0.3 * torch.randn(2, c, generator=g)).to(td), (0.2 * torch.randn(2, c, generator=g)).to(td)
    ab = (2.0 * torch.randn(rows, 2 * hid, generator=g)).to(td)
    rows_per = (rows + 1) // 2
    r_off, R = 5, rows + 9
    lds = (R + 63) // 64 * 64
    pb = PlanBuilder(lib, dev, dtype)
    xt, mst, mht, abt = pb.const(x), pb.const(ms), pb.const(mh), pb.const(ab)
    # two-pass reference path
    y2 = pb.buf((rows, c), td)
    pb.norm(xt, y2, rows, c, eps=1e-6, kind=0, mod_scale=mst, mod_shift=mht, rows_per=rows_per, ldmod=c)
    q2, s2 = pb.buf((R, c), torch.uint8, zero=True), pb.buf((c // 128, lds), torch.int32, zero=True)
    pb.quantize(y2, rows, c, q=q2, scale=s2, row_off=r_off, lds=lds)
    va = Act(abt.view(1, 1, rows, 2 * hid), 1, 1, rows, hid, 0)
    vb = Act(abt.view(1, 1, rows, 2 * hid), 1, 1, rows, hid, hid)
    sw2 = pb.ew(abi.EW_SWIGLU, va, b=vb)
    qs2, ss2 = pb.buf((R, 2 * hid), torch.uint8, zero=True), pb.buf((2 * hid // 128, lds), torch.int32, zero=True)
    pb.quantize(sw2.t, rows, hid, q=qs2, scale=ss2, row_off=r_off, lds=lds, ldq=2 * hid, q_col_off=hid)      # lands in the right half of a wider twin
    # fused path
    y1 = pb.buf((rows, c), td)
    q1, s1 = pb.buf((R, c), torch.uint8, zero=True), pb.buf((c // 128, lds), torch.int32, zero=True)
    pb.norm(xt, y1, rows, c, eps=1e-6, kind=0, mod_scale=mst, mod_shift=mht, rows_per=rows_per, ldmod=c, q8=(q1, s1), q_row_off=r_off, lds_q=lds)
    q1b, s1b = pb.buf((R, c), torch.uint8, zero=True), pb.buf((c // 128, lds), torch.int32, zero=True)
    pb.norm(xt, None, rows, c, eps=1e-6, kind=0, mod_scale=mst, mod_shift=mht, rows_per=rows_per, ldmod=c, q8=(q1b, s1b), q_row_off=r_off, lds_q=lds)
    qs1, ss1 = pb.buf((R, 2 * hid), torch.uint8, zero=True), pb.buf((2 * hid // 128, lds), torch.int32, zero=True)
    sw1 = pb.buf((rows, hid), td)
    pb.quantize(abt, rows, hid, ldx=2 * hid, q=qs1, scale=ss1, row_off=r_off, lds=lds, ldq=2 * hid, q_col_off=hid,
                swiglu_b=abt, b_off=hid, ldb=2 * hid, y=sw1, ldy=hid)
    plan = _run(pb)
    assert torch.equal(y1, y2), "16-bit norm output changed"
    for a_, b_, what in ((q1, q2, "norm bytes"), (s1, s2, "norm scales"), (q1b, q2, "norm bytes (no 16-bit output)"), (s1b, s2, "norm scales (no 16-bit output)"),
                         (qs1, qs2, "SwiGLU bytes"), (ss1, ss2, "SwiGLU scales")):
        assert torch.equal(a_, b_), f"{what}: {(a_ != b_).sum().item()} differ"
    assert torch.equal(sw1, sw2.t.view(rows, hid)), "16-bit SwiGLU copy differs"
    assert q1.any() and qs1.any()


def check_gemm_act_extremes(lib, dtype):
    """tanh-GELU / SiLU epilogues far outside the usual range: pre-activations of -600 .. +600 (e^t overflows fp32 on one side): finite, the
    limits of the functions (0 and v), never NaN — the reciprocal form of the division (csrc/mtx_device.h div_by_1p) must survive d = inf"""
    dev, td = _dev(lib), TD[dtype]
    m, n, k = 256, 256, 64
    a = torch.linspace(-9.5, 9.5, m)[:, None].repeat(1, k).to(td)
    w = torch.ones(n, k).to(td)
    ref_v = a.float() @ w.float().t()                       # rows of constant value -608 .. 608
    for act, fn in ((abi.ACT_GELU_TANH, lambda v: F.gelu(v, approximate="tanh")), (abi.ACT_SILU, F.silu)):
        pb = PlanBuilder(lib, dev, dtype)
        out = pb.gemm(pb.const(a), pb.const(w), m, n, k, act=act, flags=abi.GEMM_FORCE_TILE256)
        _run(pb)
        got = out.cpu().float().view(m, n)
        assert torch.isfinite(got).all(), f"act {act}: non-finite values"
        want = fn(ref_v).to(td).float()
        assert (got - want).abs().max() <= 4.0 * torch.finfo(td).eps * want.abs().max(), (act, (got - want).abs().max().item())


def check_gemm_f8(lib, dtype, m, n, k, act=abi.ACT_NONE, with_bias=True, with_res=False, with_gate=False, seed=0, flags=0,
                  spread=1.0):
    """C = epilogue(dequant(Aq) dequant(Wq)^T): the kernel against an fp32 product of the SAME quantised operands (so the check is
    the kernel's arithmetic: e4m3 decode, block scales, k pairing, accumulation), plus the distance to the unquantised product."""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    a = (torch.randn(m, k, generator=g) * torch.exp(spread * torch.randn(m, k // 32, generator=g)).repeat_interleave(32, 1)).to(td)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(td)
    b = torch.randn(n, generator=g) if with_bias else None
    _, _, ad = mx_quantize_ref(a)
    _, _, wd = mx_quantize_ref(w)
    ref = ad @ wd.t()
    full = a.float() @ w.float().t()
    qerr = ((ref - full).norm() / full.norm()).item()
    if b is not None:
        ref = ref + b
    if act == abi.ACT_GELU_TANH:
        ref = F.gelu(ref, approximate="tanh")
    elif act == abi.ACT_SILU:
        ref = F.silu(ref)
    gate = res = None
    rows_per = max(m // 2, 1)
    if with_gate:
        gate = torch.randn((m + rows_per - 1) // rows_per, n, generator=g).to(td)
        ref = ref * gate.float().repeat_interleave(rows_per, dim=0)[:m]
    if with_res:
        res = torch.randn(m, n, generator=g).to(td)
        ref = ref + res.float()
    pb = PlanBuilder(lib, dev, dtype)
    aq, asc, lds_a = pb.quantize(pb.const(a), m, k)
    wq, wsc, lds_w = pb.quantize(pb.const(w), n, k)
    out = pb.gemm(aq, wq, m, n, k, bias=pb.const(b) if b is not None else None, act=act,
                  res=pb.const(res) if res is not None else None, gate=pb.const(gate) if gate is not None else None,
                  gate_rows_per=rows_per, f8=(asc, lds_a, wsc, lds_w, 0, 0), flags=flags)
    _run(pb)
    err = _relerr(out.cpu().view(m, n), ref)
    assert err < TOL[dtype], f"fp8 gemm mismatch rel err {err}"
    return err, qerr


def check_gemm_f8_glu(lib, dtype, m, col0, hid, k, seed=0, row_off=0, q_col_off=0, spread=1.0):
    """The gated epilogue of the fp8 GEMM (mtx_gemm_args.glu_*) against the two launches it replaces — fp8 GEMM into a 16-bit
    [m, col0 + 2 hid] projection, then the SwiGLU quantiser (MTX_QUANT_SWIGLU) — on the same operands: the e4m3 bytes, the E8M0 scale
    bytes and the ungated columns [0, col0) must be IDENTICAL (the fused form rounds a and b to the 16-bit type exactly where the
    projection would have been stored).  Rows land at row_off, bytes at q_col_off of a wider operand buffer, like in the FLUX.2 graphs."""
    from mangatranslator_amd.hip.plan import glu_interleave
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    n = col0 + 2 * hid
    a = (torch.randn(m, k, generator=g) * torch.exp(spread * torch.randn(m, k // 32, generator=g)).repeat_interleave(32, 1)).to(td)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k) * 1.5).to(td)
    R, QW = row_off + m + 3, q_col_off + hid
    lds = (R + 63) // 64 * 64
    outs = []
    for fused in (False, True):
        pb = PlanBuilder(lib, dev, dtype)
        aq, asc, lds_a = pb.quantize(pb.const(a), m, k)
        wq, wsc, lds_w = pb.quantize(pb.const(w[glu_interleave(col0, hid)].contiguous() if fused else w), n, k)
        q8 = pb.buf((R, QW), torch.uint8, zero=True)
        sc = pb.buf((QW // 128, lds), torch.int32, zero=True)
        if fused:
            c = pb.gemm(aq, wq, m, n, k, out=pb.buf((m, n), TD[dtype], zero=True) if col0 else None, f8=(asc, lds_a, wsc, lds_w, 0, 0),
                        flags=abi.GEMM_FORCE_TILE256, glu=(q8, sc, QW, lds, col0, row_off, q_col_off))
        else:
            # NO_SPLIT: the gated kernel computes whole tiles; the K-slice tail FORCE_TILE256 would otherwise give this reference on a 256-CU
            # chip (8512 x 27648: 88 left-over tiles) sums K in slabs, i.e. in another fp32 order — found on hardware, round 4's first visit
            c = pb.gemm(aq, wq, m, n, k, f8=(asc, lds_a, wsc, lds_w, 0, 0), flags=abi.GEMM_FORCE_TILE256 | abi.GEMM_NO_SPLIT)
            pb.quantize(c, m, hid, ldx=n, x_off=col0, q=q8, scale=sc, row_off=row_off, lds=lds, ldq=QW, q_col_off=q_col_off,
                        swiglu_b=c, b_off=col0 + hid, ldb=n)
        _run(pb)
        outs.append((q8.cpu().numpy().copy(), sc.cpu().numpy().copy(), c.cpu().float().numpy()[:, :col0].copy() if col0 else None))
    (q0, s0, c0), (q1, s1, c1) = outs
    assert q0.any() and s0.any()
    assert np.array_equal(q0, q1), f"gated epilogue: {(q0 != q1).sum()} of {q0.size} e4m3 bytes differ"
    assert np.array_equal(s0, s1), f"gated epilogue: {(s0 != s1).sum()} scale words differ"
    if col0:
        assert np.array_equal(c0, c1), "gated epilogue: the ungated columns differ"
    assert not q1[:row_off].any() and not q1[row_off + m:].any() and not q1[:, :q_col_off].any(), "bytes outside the target window were written"


def check_swiglu(lib, dtype, rows, hid, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = torch.randn(rows, 2 * hid, generator=g).to(td)
    ref = F.silu(x[:, :hid].float()) * x[:, hid:].float()
    pb = PlanBuilder(lib, dev, dtype)
    xt = pb.const(x)
    va = Act(xt.view(1, 1, rows, 2 * hid), 1, 1, rows, hid, 0)
    vb = Act(xt.view(1, 1, rows, 2 * hid), 1, 1, rows, hid, hid)
    y = pb.ew(abi.EW_SWIGLU, va, b=vb)
    _run(pb)
    err = _relerr(y.torch().cpu().view(rows, hid), ref)
    assert err < TOL[dtype], f"swiglu mismatch {err}"


def check_rcab_tail(lib, dtype, n=2, h=37, w=29, c=64, cr=4, canvas=None, seed=0):
    """The two ops that close an RCAB in the pool-before-conv form, against torch on the same rounded operands:
    (1) channel attention of mean(conv3x3(t) + b) from the channel sums of t (linearity; border rows / columns / corners read from t),
    (2) conv with a per-channel output factor and a residual: y = s * (conv(t) + b) + x.
    canvas=(H, W): t lives on a larger canvas and the image size comes from a device-side valid_hw (bucket plans)."""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    H, W = canvas if canvas else (h, w)
    t = torch.zeros(n, H, W, c)
    t[:, :h, :w] = torch.relu(torch.randn(n, h, w, c, generator=g))
    x = torch.randn(n, H, W, c, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) / math.sqrt(c * 9)
    b = torch.randn(c, generator=g) * 0.1
    w1, b1 = torch.randn(cr, c, generator=g) / math.sqrt(c), torch.randn(cr, generator=g) * 0.1
    w2, b2 = torch.randn(c, cr, generator=g) / math.sqrt(cr), torch.randn(c, generator=g) * 0.1
    tq, xq, wq = t.to(td).float(), x.to(td).float(), wt.to(td).float()
    u = F.conv2d(tq[:, :h, :w].permute(0, 3, 1, 2), wq, b, padding=1)                   # the image alone: its own zero padding
    m = u.mean((2, 3))
    s_ref = torch.sigmoid(F.linear(torch.relu(F.linear(m, w1, b1)), w2, b2))
    y_ref = (s_ref[:, :, None, None] * u + xq[:, :h, :w].permute(0, 3, 1, 2)).permute(0, 2, 3, 1)

    pb = PlanBuilder(lib, dev, dtype)
    tb = pb.act(n, H, W, c); tb.t.copy_(tq.to(td))
    xb = pb.act(n, H, W, c); xb.t.copy_(xq.to(td))
    wpk = pb.const(wt.permute(0, 2, 3, 1).reshape(c, 9, c), td)
    bias = pb.const(b, torch.float32)
    valid = None
    if canvas:
        valid = pb.buf((2,), torch.int32); valid.copy_(torch.tensor([h, w], dtype=torch.int32))
    # channel sums of t as a conv with SUM would leave them: any split into partial rows
    tiles = 5
    cs = pb.buf((n, tiles, c), torch.float32, zero=True)
    tot = tq.sum((1, 2))
    cs[:, 0] = (tot * 0.25).to(cs.device); cs[:, 3] = (tot * 0.75).to(cs.device)
    s_buf = pb.buf((n, c), torch.float32)
    pb.channel_attention(cs, pb.const(w1, torch.float32), pb.const(b1, torch.float32), pb.const(w2, torch.float32), pb.const(b2, torch.float32),
                         s_buf, n, tiles, c, cr, 1.0 / (h * w), before_conv=(tb, wpk, bias), valid_hw=valid)
    y = pb.conv2d(tb, wpk, bias, c, 3, 1, res=xb, out_scale=s_buf, valid_hw=valid)
    _run(pb)
    e_s = (s_buf.cpu() - s_ref).abs().max().item()
    assert e_s < 2e-3, f"channel attention (pool before the conv) off by {e_s}"
    out = y.torch().cpu().float()
    err = _relerr(out[:, :h, :w], y_ref)
    assert err < TOL[dtype], f"scaled conv + residual mismatch rel err {err}"
    if canvas:
        assert float(out[:, h:].abs().max()) == 0.0 and float(out[:, :, w:].abs().max()) == 0.0      # beyond the image: zeros
    return e_s, err


def check_memset(lib):
    """MTX_OP_MEMSET as a plan op (a fill kernel, not hipMemsetAsync): every byte of ragged ranges at ragged offsets takes the value, the
    bytes around them keep theirs, also on a second run"""
    dev = _dev(lib)
    pb = PlanBuilder(lib, dev, abi.BF16)
    cases = [(0, 184, 0), (3, 1, 7), (5, 2, 255), (1, 1027, 9), (2, 4, 1), (6, 4099, 0), (0, 0, 5)]
    bufs = []
    for off, n, v in cases:
        t = pb.buf((n + 16,), torch.uint8)
        t.fill_(0x5A)
        pb.memset(t[off:off + n], v)
        bufs.append(t)
    plan = _run(pb)
    for rep in range(2):
        for (off, n, v), t in zip(cases, bufs):
            h = t.cpu()
            assert (h[off:off + n] == v).all() and (h[:off] == 0x5A).all() and (h[off + n:] == 0x5A).all(), (off, n, v, rep)
            t[off:off + n] = 0x33
        plan.run()
        _sync(lib)


# ---- fp32 plans (csrc/f32ops.hip: SAM's high-precision mask decoder) ------------------------------------------------------------------------
def check_f32_ops(lib, seed=0):
    """every op of the fp32 path against torch fp32 on the same values: GEMM (ragged sizes, bias, activation, residual broadcast over a
    batch, operand offsets and strides, strided batches), attention (strided q / k / v slices of fused projections, few and many keys),
    LayerNorm (+ GELU), and the element-wise kinds incl. the ConvTranspose pixel shuffle and the 16-bit -> fp32 conversion"""
    g = torch.Generator().manual_seed(seed)
    dev = _dev(lib)
    rnd = lambda *s: torch.randn(*s, generator=g)
    pb = PlanBuilder(lib, dev, abi.F32)
    checks = []
    # GEMM 1: ragged, bias + GELU + residual
    m, n, k = 203, 77, 50
    a, w, b, r = rnd(m, k), rnd(n, k) / math.sqrt(k), rnd(n), rnd(m, n)
    out = pb.gemm(pb.const(a), pb.const(w), m, n, k, bias=pb.const(b), act=abi.ACT_GELU, res=pb.const(r))
    checks.append(("gemm gelu+res", out, F.gelu(a @ w.t() + b) + r))
    # GEMM 2: batch with a residual shared by every batch (res_bs = 0), alpha
    bt, m, n, k = 3, 40, 24, 32
    a, w, r = rnd(bt, m, k), rnd(n, k), rnd(m, n)
    out = pb.gemm(pb.const(a), pb.const(w), m, n, k, res=pb.const(r), batch=bt, a_bs=m * k, c_bs=m * n, res_bs=0, alpha=0.5)
    checks.append(("gemm batch shared res", out, 0.5 * torch.einsum("bmk,nk->bmn", a, w) + r))
    # GEMM 2b: from 256 rows up the fp32 GEMM runs on the matrix pipe (v_mfma_f32_32x32x2_f32): ragged in every dimension, an unaligned lda
    # (scalar loads) and an aligned one (16-byte loads), bias + activation + residual, batches, and the forced form on a small problem
    for m, n, k, ld_extra in ((300, 70, 50, 1), (513, 129, 256, 0)):
        a, w, b, r = rnd(m, k + ld_extra), rnd(n, k) / math.sqrt(k), rnd(n), rnd(m, n)
        out = pb.gemm(pb.const(a), pb.const(w), m, n, k, lda=k + ld_extra, bias=pb.const(b), act=abi.ACT_GELU, res=pb.const(r))
        checks.append((f"gemm (matrix pipe) {m}x{n}x{k}", out, F.gelu(a[:, :k] @ w.t() + b) + r))
    bt, m, n, k = 2, 260, 33, 18
    a, w = rnd(bt, m, k), rnd(bt, n, k)
    out = pb.gemm(pb.const(a), pb.const(w), m, n, k, batch=bt, a_bs=m * k, w_bs=n * k, c_bs=m * n, alpha=0.25)
    checks.append(("gemm (matrix pipe) batch", out, 0.25 * torch.einsum("bmk,bnk->bmn", a, w)))
    a, w = rnd(40, 24), rnd(9, 24)
    out = pb.gemm(pb.const(a), pb.const(w), 40, 9, 24, flags=abi.GEMM_FORCE_TILE256)
    checks.append(("gemm (matrix pipe, forced) 40x9x24", out, a @ w.t()))
    # GEMM 2c: few rows, long K (SAM's token-side MLP-out): a wave per output column and 8 rows, lanes striding K
    for m, n, k, bt in ((37, 50, 1024, 1), (72, 19, 2048, 2)):
        a, w, b, r = rnd(bt, m, k), rnd(n, k) / math.sqrt(k), rnd(n), rnd(bt, m, n)
        out = pb.gemm(pb.const(a), pb.const(w), m, n, k, bias=pb.const(b), act=abi.ACT_RELU, res=pb.const(r), batch=bt, a_bs=m * k, c_bs=m * n, res_bs=m * n)
        checks.append((f"gemm (few rows, long K) {bt}x{m}x{n}x{k}", out, F.relu(torch.einsum("bmk,nk->bmn", a, w) + b) + r))
    # GEMM 3: rows picked out of a wider matrix (lda, a_off), output into a column slice (ldc, c_off), relu; strided weight batches
    rows, NT, D, c2 = 5, 9, 16, 8
    q = rnd(rows * NT, D)
    w1, b1 = rnd(D, D) / 4, rnd(D)
    t = pb.gemm(pb.const(q), pb.const(w1), rows, D, D, lda=NT * D, a_off=2 * D, bias=pb.const(b1), act=abi.ACT_RELU)
    checks.append(("gemm token rows", t, F.relu(q.view(rows, NT, D)[:, 2] @ w1.t() + b1)))
    hyper = pb.buf((rows, 4, c2), torch.float32, zero=True)
    w3 = rnd(c2, D)
    pb.gemm(t, pb.const(w3), rows, c2, D, out=hyper, ldc=4 * c2, c_off=1 * c2)
    up = rnd(rows, 30, c2)
    hyp = rnd(rows, 4, c2)
    masks = pb.gemm(pb.const(up), pb.const(hyp), 30, 4, c2, batch=rows, a_bs=30 * c2, w_bs=4 * c2, c_bs=30 * 4, out_f32=True)
    checks.append(("gemm per-box weights", masks, torch.einsum("bpc,bkc->bpk", up, hyp)))
    # attention: q / k from one fused projection (k_off), 70 keys (two key rounds of a wave) and 3 keys
    # (130 x 3 and 300 x 9: many queries against few keys — a lane per query)
    for nb, heads, sq, sk, d in ((2, 4, 5, 70, 16), (1, 2, 130, 3, 32), (1, 1, 2, 64, 8), (3, 2, 300, 9, 16)):
        Dm = heads * d
        qk, v = rnd(nb * max(sq, sk), 2 * Dm), rnd(nb * sk, Dm)
        o = pb.buf((nb * sq, Dm), torch.float32, zero=True)
        L = max(sq, sk)
        qkT = pb.const(qk)
        pb.attention(qkT, qkT, pb.const(v), o, nb, heads, sq, sk, d, (L * 2 * Dm, 2 * Dm, d), (L * 2 * Dm, 2 * Dm, d),
                     (sk * Dm, Dm, d), (sq * Dm, Dm, d), 1.0 / math.sqrt(d), k_off=Dm)
        qq = qk.view(nb, L, 2, heads, d)[:, :sq, 0].transpose(1, 2)
        kk_ = qk.view(nb, L, 2, heads, d)[:, :sk, 1].transpose(1, 2)
        vv = v.view(nb, sk, heads, d).transpose(1, 2)
        checks.append((f"attention {sq}x{sk} d{d}", o, F.scaled_dot_product_attention(qq, kk_, vv).transpose(1, 2).reshape(nb * sq, Dm)))
    # LayerNorm (+ GELU), row stride wider than C
    rows, c = 37, 200
    x, gm, bt_ = rnd(rows, 256) * 3 + 1, rnd(c), rnd(c)
    y = pb.norm(pb.const(x), pb.buf((rows, c), torch.float32), rows, c, ldx=256, gamma=pb.const(gm), beta=pb.const(bt_), eps=1e-5, act=abi.ACT_GELU)
    checks.append(("layernorm gelu", y, F.gelu(F.layer_norm(x[:, :c], (c,), gm, bt_, 1e-5))))
    # element-wise kinds
    n_, h_, w_, c_ = 2, 3, 5, 8
    xa, xb = rnd(n_, h_, w_, c_), rnd(n_, h_, w_, c_)
    A, B = Act(pb.const(xa), n_, h_, w_, c_), Act(pb.const(xb), n_, h_, w_, c_)
    checks.append(("add", pb.ew(abi.EW_ADD, A, b=B).t, xa + xb))
    checks.append(("gelu", pb.ew(abi.EW_ACT, A, act=abi.ACT_GELU).t, F.gelu(xa)))
    idx = torch.tensor([4, 0, 3, 3, 1], dtype=torch.int32)
    src = rnd(6, 16)
    checks.append(("row gather", pb.row_gather(pb.const(src), pb.buf((5, 16), torch.float32), pb.hold(idx.to(dev)), 5, 16), src[idx.long()]))
    cols = rnd(n_, h_, w_, 4 * c_)
    skip = rnd(1, 2 * h_, 2 * w_, c_)
    want = cols.view(n_, h_, w_, 2, 2, c_).permute(0, 1, 3, 2, 4, 5).reshape(n_, 2 * h_, 2 * w_, c_) + skip
    outS = pb.act(n_, 2 * h_, 2 * w_, c_)
    e = abi.EwArgs()
    colsT = pb.const(cols)
    skipT = pb.const(skip)
    e.a, e.b, e.s, e.y = colsT.data_ptr(), skipT.data_ptr(), None, outS.ptr
    e.n, e.h, e.w, e.c = n_, h_, w_, c_
    e.lda, e.ldb, e.ldy, e.lds = 4 * c_, c_, c_, 0
    e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_SHUFFLE2_ADD, 0, 0.0, 0, 0, abi.F32
    pb._add(abi.OP_EW, e, "shuffle2_add")
    checks.append(("pixel shuffle + broadcast skip", outS.t, want))
    for src_dt, code in ((torch.float16, abi.F16), (torch.bfloat16, abi.BF16)):
        x16 = rnd(1, 2, 3, 16).to(src_dt)
        y32 = pb.act(1, 2, 3, 16)
        e = abi.EwArgs()
        x16d = pb.const(x16)
        e.a, e.b, e.s, e.y = x16d.data_ptr(), None, None, y32.ptr
        e.n, e.h, e.w, e.c = 1, 2, 3, 16
        e.lda, e.ldb, e.ldy, e.lds = 16, 0, 16, 0
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_CVT_F32, 0, 0.0, code, 0, abi.F32
        pb._add(abi.OP_EW, e, "cvt")
        checks.append((f"cvt {src_dt}", y32.t, x16.float()))
    # fp32 ops inside a 16-bit plan: LayerNorm on an fp32 stream, its result rounded into both halves of a wide 16-bit operand
    pb16 = PlanBuilder(lib, dev, abi.F16)
    x32 = rnd(21, 40)
    y32 = pb16.norm(pb16.const(x32), pb16.buf((21, 40), torch.float32), 21, 40, eps=1e-6, dtype=abi.F32)
    wide = pb16.cvt16(y32, pb16.buf((21, 80), torch.float16), 21, 40, copies=2)
    s32 = pb16.buf((21, 40), torch.float32)
    pb16.ew(abi.EW_ADD, Act(y32.view(1, 1, 21, 40), 1, 1, 21, 40), b=Act(pb16.const(x32).view(1, 1, 21, 40), 1, 1, 21, 40), out=Act(s32.view(1, 1, 21, 40), 1, 1, 21, 40), dtype=abi.F32)
    _run(pb16)
    want_ln = F.layer_norm(x32, (40,), eps=1e-6)
    assert torch.equal(wide.cpu()[:, :40], wide.cpu()[:, 40:]) and torch.equal(wide.cpu()[:, :40], y32.cpu().to(torch.float16))
    checks.append(("layernorm fp32 in a 16-bit plan", y32, want_ln))
    checks.append(("fp32 add in a 16-bit plan", s32, want_ln + x32))
    _run(pb)
    worst = 0.0
    for name, got, want in checks:
        got, want = got.float().cpu().reshape(want.shape), want.float()
        err = ((got - want).abs().max() / want.abs().max().clamp_min(1e-6)).item()
        assert err < 2e-5, f"fp32 op '{name}': rel err {err}"
        worst = max(worst, err)
    return worst


def check_hi_lo_weights(lib, dtype=abi.F16, m=300, n=96, k=144, seed=0):
    """`Sam2Hip(precision="high")`'s weight trick at op level: W = W_hi + W_lo in the storage type, both halves multiplied with the same
    operand tile into one fp32 accumulator (mtx_gemm_args.w_lo, round 6; round 5 ran it as a GEMM over K' = 2K against a copied [x | x]) —
    against x @ W^T in float64 the error falls from the weights' rounding (2^-12 relative per weight) to fp32 accumulation noise.  Also the
    epilogue forms the trunk uses: fp32 output with an fp32 residual (the branch-closing linears of the fp32 residual stream), 16-bit output
    with bias + GELU (fc1), and the pair form must agree with the 2K form up to fp32 summation order."""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    x = torch.randn(m, k, generator=g).to(td)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    r32 = torch.randn(m, n, generator=g)
    ref = (x.double() @ w.double().t()).float()
    w_hi = w.to(td)
    w_lo = (w - w_hi.float()).to(td)
    pb = PlanBuilder(lib, dev, dtype)
    xc, whc, wlc = pb.const(x), pb.const(w_hi), pb.const(w_lo)
    fast = pb.gemm(xc, whc, m, n, k, out_f32=True)
    high = pb.gemm(xc, whc, m, n, k, out_f32=True, w_lo=wlc)
    xx, ww = pb.const(torch.cat([x, x], 1)), pb.const(torch.cat([w_hi, w_lo], 1))
    high2k = pb.gemm(xx, ww, m, n, 2 * k, out_f32=True)
    high16 = pb.gemm(xc, whc, m, n, k, w_lo=wlc, bias=pb.const(b), act=abi.ACT_GELU)                 # 16-bit output: the form the qkv / fc1 linears use
    closed = pb.gemm(xc, whc, m, n, k, w_lo=wlc, bias=pb.const(b), res=pb.const(r32), res_f32=True, out_f32=True)      # stream <- res + x W^T + b, all fp32
    _run(pb)
    e_fast, e_high = _relerr(fast.cpu(), ref), _relerr(high.cpu(), ref)
    assert e_high < e_fast / 20 and e_high < 1e-5, (e_fast, e_high)          # what is left is fp32 accumulation over K (2.3e-6 at K = 1152 on the simulator)
    assert _relerr(high.cpu(), high2k.cpu()) < 1e-5          # (two fp32 summation orders of the same products: noise grows with K)
    assert _relerr(high16.cpu(), F.gelu(ref + b)) < TOL[dtype]
    assert _relerr(closed.cpu(), ref + b + r32) < 1e-5
    return e_fast, e_high


def check_norm_f32_to_16(lib, dtype=abi.F16, rows=37, c=576, seed=0):
    """LayerNorm of an fp32 row written in the 16-bit operand type of the next linear (mtx_norm_args.out_dtype, round 6): equal to the fp32
    result rounded once; widths on the register path (c % 4 == 0, c <= 1280) and off it"""
    g = torch.Generator().manual_seed(seed)
    dev, td = _dev(lib), TD[dtype]
    worst = 0.0
    for cc in (c, 144, 1152, 1284, 75):
        x = torch.randn(rows, cc, generator=g) * 3 + 0.5
        gam, bet = torch.randn(cc, generator=g), torch.randn(cc, generator=g)
        pb = PlanBuilder(lib, dev, dtype)
        xc = pb.const(x, torch.float32)
        y32 = pb.norm(xc, pb.buf((rows, cc), torch.float32), rows, cc, gamma=pb.const(gam, torch.float32), beta=pb.const(bet, torch.float32), eps=1e-6, dtype=abi.F32)
        y16 = pb.norm(xc, pb.buf((rows, cc), td), rows, cc, gamma=pb.const(gam, torch.float32), beta=pb.const(bet, torch.float32), eps=1e-6, dtype=abi.F32, out_dtype=dtype)
        _run(pb)
        ref = F.layer_norm(x, (cc,), gam, bet, 1e-6)
        assert _relerr(y32.cpu(), ref) < 2e-6
        assert torch.equal(y16.cpu(), y32.cpu().to(td)), "16-bit norm output is not the fp32 result rounded once"
        worst = max(worst, _relerr(y16.float().cpu(), ref))
    return worst
