"""GPU micro-benchmarks of the MFMA-bound kernels at FLUX shapes (HIP-event timing via mtx_plan_time).
usage: python tools/bench_kernels.py [attn T [heads]] [gemm M N K] [gemm8 M N K] [quant ROWS K] [conv H W] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.hip.plan import PlanBuilder

lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")


def _time(plan, iters):
    plan.run(); torch.cuda.synchronize()
    plan.time(5)
    return min(plan.time(iters) for _ in range(3))


def attn(T, heads=24, d=128, iters=10):
    pb = PlanBuilder(lib, dev, abi.BF16)
    D = heads * d
    qkv = pb.buf((T, 3 * D), torch.bfloat16); qkv.normal_()
    o = pb.buf((T, D), torch.bfloat16)
    qkv[:, :D] *= d ** -0.5 * 1.4426950408889634         # the FLUX graphs' form: q pre-multiplied by scale * log2(e)
    pb.attention(qkv, qkv, qkv, o, 1, heads, T, T, d, (0, 3 * D, d), (0, 3 * D, d), (0, 3 * D, d), (0, D, d), d ** -0.5, k_off=D, v_off=2 * D,
                 q_prescaled=True)
    ms = _time(pb.build(), iters)
    print(f"attn T={T} heads={heads}: {ms:.3f} ms  {4 * T * T * D / ms / 1e9:.0f} TFLOP/s", flush=True)
    return ms


def attn_q8(T, heads=24, d=128, iters=10, fused=True):
    """FLUX.2 form: the rows leave as the MX fp8 operand of the next linear (fused: mtx_attn_args.q8; else attention + quantiser launch)"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    D = heads * d
    qkv = pb.buf((T, 3 * D), torch.bfloat16); qkv.normal_()
    qkv[:, :D] *= d ** -0.5 * 1.4426950408889634
    lds = (T + 63) // 64 * 64
    q8 = pb.buf((T, D), torch.uint8, zero=True)
    sc = pb.buf((D // 128, lds), torch.int32, zero=True)
    strides = ((0, 3 * D, d), (0, 3 * D, d), (0, 3 * D, d), (0, D, d))
    if fused:
        pb.attention(qkv, qkv, qkv, None, 1, heads, T, T, d, *strides, d ** -0.5, k_off=D, v_off=2 * D, q_prescaled=True, q8=(q8, sc, D, lds, 0))
    else:
        o = pb.buf((T, D), torch.bfloat16)
        pb.attention(qkv, qkv, qkv, o, 1, heads, T, T, d, *strides, d ** -0.5, k_off=D, v_off=2 * D, q_prescaled=True)
        pb.quantize(o, T, D, q=q8, scale=sc, lds=lds, ldq=D)
    ms = _time(pb.build(), iters)
    print(f"attn -> MX fp8 T={T} heads={heads} [{'fused epilogue' if fused else 'attention + quantiser launch'}]: {ms:.3f} ms  {4 * T * T * D / ms / 1e9:.0f} TFLOP/s", flush=True)


def attn_f8_pv(T, heads=24, d=128, iters=10):
    """fp8 scores + fp8 P V (mtx_attn_args.v_f8t), MX fp8 rows out; the V^T producer launch (MTX_EW_V_F8T) timed with it and alone"""
    D = heads * d
    lds = (T + 63) // 64 * 64
    def build(with_attn, with_prod):
        pb = PlanBuilder(lib, dev, abi.BF16)
        qkv = pb.buf((T, 3 * D), torch.bfloat16); qkv.normal_()
        qk8 = pb.buf((T, 2 * D), torch.uint8)
        qk8.copy_(qkv[:, :2 * D].float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
        q8 = pb.buf((T, D), torch.uint8, zero=True)
        sc = pb.buf((D // 128, lds), torch.int32, zero=True)
        vt8 = pb.buf((D, lds), torch.uint8, zero=True)
        if with_prod:
            pb.v_f8t(qkv, T, heads, 3 * D, v_off=2 * D, out=vt8)
        if with_attn:
            pb.attention(qkv, qkv, qkv, None, 1, heads, T, T, d, (0, 3 * D, d), (0, 3 * D, d), (0, 3 * D, d), (0, D, d), d ** -0.5, k_off=D, v_off=2 * D,
                         q_prescaled=True, q8=(q8, sc, D, lds, 0), qk_f8=(qk8, 0, D, 2 * D, -3), pv_f8=(vt8, lds))
        return pb.build()
    both, prod = _time(build(True, True), iters), _time(build(False, True), iters)
    print(f"attn, fp8 scores + fp8 P V T={T} heads={heads} [MX fp8 rows out]: {both - prod:.3f} ms  {4 * T * T * D / (both - prod) / 1e9:.0f} TFLOP/s; "
          f"with the V^T producer launch {both:.3f} ms (producer alone {prod * 1e3:.1f} us)", flush=True)


def attn_f8_scores(T, heads=24, d=128, iters=10, out8=True):
    """the fp8-score form (mtx_attn_args.q_f8 / k_f8): q and k as plain e4m3 rows; out8: rows leave as MX fp8 (the Klein graph's form), else 16-bit"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    D = heads * d
    qkv = pb.buf((T, 3 * D), torch.bfloat16); qkv.normal_()
    qk8 = pb.buf((T, 2 * D), torch.uint8)
    qk8.copy_(qkv[:, :2 * D].float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
    lds = (T + 63) // 64 * 64
    q8 = pb.buf((T, D), torch.uint8, zero=True)
    sc = pb.buf((D // 128, lds), torch.int32, zero=True)
    o = None if out8 else pb.buf((T, D), torch.bfloat16)
    pb.attention(qkv, qkv, qkv, o, 1, heads, T, T, d, (0, 3 * D, d), (0, 3 * D, d), (0, 3 * D, d), (0, D, d), d ** -0.5, k_off=D, v_off=2 * D,
                 q_prescaled=True, q8=(q8, sc, D, lds, 0) if out8 else None, qk_f8=(qk8, 0, D, 2 * D, -3))
    ms = _time(pb.build(), iters)
    print(f"attn, fp8 scores T={T} heads={heads} [{'MX fp8 rows out' if out8 else '16-bit rows out'}]: {ms:.3f} ms  {4 * T * T * D / ms / 1e9:.0f} TFLOP/s", flush=True)


def gemm8_glu(M, hid, K, col0=0, iters=20, fused=True):
    """FLUX.2 MLP-in: fp8 GEMM [M, col0 + 2 hid] whose gated half leaves as silu(a) * b in MX fp8 (fused: mtx_gemm_args.glu_*; else GEMM + SwiGLU quantiser)"""
    from mangatranslator_amd.hip.plan import glu_interleave
    pb = PlanBuilder(lib, dev, abi.BF16)
    N = col0 + 2 * hid
    a = pb.buf((M, K), torch.bfloat16); a.normal_()
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5)
    if fused:
        w.copy_(w[glu_interleave(col0, hid).to(dev)].clone())
    q = PlanBuilder(lib, dev, abi.BF16)
    aq, asc, la = q.quantize(a, M, K)
    wq, wsc, lw = q.quantize(w, N, K)
    q.build().run(); torch.cuda.synchronize()
    pb.keep += [aq, asc, wq, wsc]
    lds = (M + 63) // 64 * 64
    q8 = pb.buf((M, hid), torch.uint8, zero=True)
    sc = pb.buf((hid // 128, lds), torch.int32, zero=True)
    if fused:
        pb.gemm(aq, wq, M, N, K, out=pb.buf((M, N), torch.bfloat16) if col0 else None, f8=(asc, la, wsc, lw, 0, 0), glu=(q8, sc, hid, lds, col0, 0, 0))
    else:
        c = pb.gemm(aq, wq, M, N, K, f8=(asc, la, wsc, lw, 0, 0))
        pb.quantize(c, M, hid, ldx=N, x_off=col0, q=q8, scale=sc, lds=lds, ldq=hid, swiglu_b=c, b_off=col0 + hid, ldb=N)
    ms = _time(pb.build(), iters)
    print(f"gemm8 + SwiGLU -> MX fp8 M={M} N={N} K={K} [{'gated epilogue' if fused else 'GEMM + SwiGLU quantiser launch'}]: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)


def gemm(M, N, K, iters=20, f8=False, pad=0, flags=0, act=0):
    """pad > 0: rows of A and W are `pad` elements apart more than K (leading-dimension padding, an address-interleave probe)"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((M, K + pad), torch.bfloat16); a.normal_()
    w = pb.buf((N, K + pad), torch.bfloat16); w.normal_(0, K ** -0.5)
    if f8:
        q = PlanBuilder(lib, dev, abi.BF16)
        aq, asc, la = q.quantize(a, M, K)
        wq, wsc, lw = q.quantize(w, N, K)
        q.build().run(); torch.cuda.synchronize()
        pb.keep += [aq, asc, wq, wsc]
        pb.gemm(aq, wq, M, N, K, f8=(asc, la, wsc, lw, 0, 0), flags=flags)
    else:
        pb.gemm(a, w, M, N, K, lda=K + pad, ldw=K + pad, flags=flags, act=act, bias=(pb.buf((N,), torch.float32, zero=True) if act else None))
    ms = _time(pb.build(), iters)
    split = lib.gemm_last_split()
    tag = " [whole tiles only]" if flags & abi.GEMM_NO_SPLIT else f" [whole tiles, K slices, pieces = {split}]"
    if act:
        tag += " [bias + tanh-GELU epilogue]"
    print(f"gemm{'8' if f8 else ''} M={M} N={N} K={K}{f' ld+{pad}' if pad else ''}{tag}: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)


def gemm_epi(M, N, K, kind, f8=False, reps=3, iters=20):
    """The 256-tile kernels with the epilogue a FLUX linear really has — kind "b": bias; "g": bias + tanh-GELU; "r": bias + gate + residual
    (attention / MLP output projections).  (Round 5 compared the batched epilogue with the one-at-a-time form here: profiles/r05_visit_q_*.log.)"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    g = torch.Generator(device=dev).manual_seed(5)
    a = pb.buf((M, K), torch.bfloat16); a.normal_(generator=g)
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5, generator=g)
    bias = pb.buf((N,), torch.float32); bias.normal_(generator=g)
    gate = res = None
    if kind == "r":
        gate = pb.buf((2, N), torch.bfloat16); gate.normal_(generator=g)
        res = pb.buf((M, N), torch.bfloat16); res.normal_(generator=g)
    kw = dict(bias=bias, act=abi.ACT_GELU_TANH if kind == "g" else 0, res=res, gate=gate, gate_rows_per=(M + 1) // 2)
    if f8:
        q = PlanBuilder(lib, dev, abi.BF16)
        aq, asc, la = q.quantize(a, M, K)
        wq, wsc, lw = q.quantize(w, N, K)
        q.build().run(); torch.cuda.synchronize()
        pb.keep += [aq, asc, wq, wsc]
        pb.gemm(aq, wq, M, N, K, f8=(asc, la, wsc, lw, 0, 0), **kw)
    else:
        pb.gemm(a, w, M, N, K, **kw)
    plan = pb.build()
    best = min(_time(plan, iters) for _ in range(reps))
    what = {"b": "bias", "g": "bias + tanh-GELU", "r": "bias + gate + residual"}[kind]
    print(f"gemm{'8' if f8 else ''} M={M} N={N} K={K} [{what}]: {best:.3f} ms ({2 * M * N * K / best / 1e9:.0f} TFLOP/s)", flush=True)


def gemm_strips(M, N, K, f8=False, reps=3, iters=20, kind="b"):
    """Round 6: the workgroup -> tile map of the 256-tile kernels with 1 (rounds 1-5), 2, 4, 8 column strips and the launcher's own choice (0),
    alternating in one process (the launcher reads MTX_GEMM_STRIPS per launch); the outputs must be the same bytes."""
    pb = PlanBuilder(lib, dev, abi.BF16)
    g = torch.Generator(device=dev).manual_seed(5)
    a = pb.buf((M, K), torch.bfloat16); a.normal_(generator=g)
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5, generator=g)
    bias = pb.buf((N,), torch.float32); bias.normal_(generator=g)
    kw = dict(bias=bias, act=abi.ACT_GELU_TANH if kind == "g" else 0)
    if f8:
        q = PlanBuilder(lib, dev, abi.BF16)
        aq, asc, la = q.quantize(a, M, K)
        wq, wsc, lw = q.quantize(w, N, K)
        q.build().run(); torch.cuda.synchronize()
        pb.keep += [aq, asc, wq, wsc]
        out = pb.gemm(aq, wq, M, N, K, f8=(asc, la, wsc, lw, 0, 0), **kw)
    else:
        out = pb.gemm(a, w, M, N, K, **kw)
    plan = pb.build()
    variants = (1, 2, 4, 0)                    # strips; 0 = the launcher's choice
    best, ref, same = {}, None, True
    for _ in range(reps):
        for st in variants:
            os.environ["MTX_GEMM_STRIPS"] = str(st)
            out.zero_()
            ms = _time(plan, iters)
            best[st] = min(best.get(st, 1e9), ms)
            if ref is None:
                ref = out.clone()
            same = same and torch.equal(out, ref)
    os.environ.pop("MTX_GEMM_STRIPS", None)
    print(f"gemm{'8' if f8 else ''} M={M} N={N} K={K} strips -> ms: " + "  ".join(f"{'auto' if st == 0 else st}: {best[st]:.4f} ({2 * M * N * K / best[st] / 1e9:.0f} TF)" for st in variants)
          + f"  same bytes: {same}", flush=True)


def quant(rows, K, iters=20):
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((rows, K), torch.bfloat16); a.normal_()
    pb.quantize(a, rows, K)
    ms = _time(pb.build(), iters)
    print(f"quantize_mx rows={rows} K={K}: {ms * 1e3:.1f} us  {rows * K * (3 + 1 / 32) / ms / 1e6:.0f} GB/s", flush=True)


def conv(H, W, iters=20):
    """RCAN body conv 64 -> 64 (f16) at page resolution"""
    pb = PlanBuilder(lib, dev, abi.F16)
    x = pb.act(1, H, W, 64); x.t.normal_()
    w = pb.buf((64, 9, 64), torch.float16); w.normal_(0, 0.04)
    b = pb.buf((64,), torch.float32, zero=True)
    pb.conv2d(x, w, b, 64, act=abi.ACT_RELU)
    ms = _time(pb.build(), iters)
    byts = 2 * 64 * H * W * 2 + 9 * 64 * 64 * 2
    print(f"conv3x3 64->64 {W}x{H}: {ms * 1e3:.1f} us  {byts / ms / 1e6:.0f} GB/s  {2 * 9 * 64 * 64 * H * W / ms / 1e9:.0f} TFLOP/s", flush=True)


def norm(rows, c, reps=3, iters=50, q8=False):
    """adaLN LayerNorm rows of a FLUX block (no affine, modulation rows per stream); q8: the FLUX.2 form (MX fp8 twin from the registers, no
    16-bit store).  (Round 5 compared four kernel forms here: profiles/r05_visit_o / _p logs.)"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    x = pb.buf((rows, c), torch.bfloat16); x.normal_()
    ms_, mh_ = pb.buf((2, c), torch.bfloat16), pb.buf((2, c), torch.bfloat16)
    ms_.normal_(); mh_.normal_()
    rows_per = (rows + 1) // 2
    lds = (rows + 63) // 64 * 64
    if q8:
        q, sc = pb.buf((rows, c), torch.uint8, zero=True), pb.buf((c // 128, lds), torch.int32, zero=True)
        pb.norm(x, None, rows, c, eps=1e-6, kind=0, mod_scale=ms_, mod_shift=mh_, rows_per=rows_per, ldmod=c, q8=(q, sc), lds_q=lds)
    else:
        y = pb.buf((rows, c), torch.bfloat16)
        pb.norm(x, y, rows, c, eps=1e-6, kind=0, mod_scale=ms_, mod_shift=mh_, rows_per=rows_per, ldmod=c)
    plan = pb.build()
    t_ms = min(_time(plan, iters) for _ in range(reps))
    byts = rows * c * (2 + (1 if q8 else 2))
    print(f"norm {rows}x{c} {'-> MX fp8' if q8 else 'bf16'}: {t_ms * 1e3:.1f} us  {byts / t_ms / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    while args:
        if args[0] == "attn":
            attn(int(args[1])); args = args[2:]
        elif args[0] == "attn88":                       # fp8 scores + fp8 P V
            attn_f8_pv(int(args[1])); args = args[2:]
        elif args[0] in ("attn8", "attn8w"):          # fp8 scores: MX fp8 rows out / 16-bit rows out
            attn_f8_scores(int(args[1]), out8=args[0] == "attn8"); args = args[2:]
        elif args[0] in ("attnq", "attnqs"):          # attention with MX fp8 output: fused epilogue / separate quantiser
            attn_q8(int(args[1]), fused=args[0] != "attnqs"); args = args[2:]
        elif args[0] in ("glu", "glus"):              # glu M hid K col0
            gemm8_glu(int(args[1]), int(args[2]), int(args[3]), int(args[4]), fused=args[0] == "glu"); args = args[5:]
        elif args[0] in ("norm", "normq"):              # norm ROWS C: every norm kernel form in turn
            norm(int(args[1]), int(args[2]), q8=args[0] == "normq"); args = args[3:]
        elif args[0] == "quant":
            quant(int(args[1]), int(args[2])); args = args[3:]
        elif args[0] == "conv":
            conv(int(args[1]), int(args[2])); args = args[3:]
        elif args[0] in ("gemmeb", "gemmeg", "gemmer", "gemm8eb", "gemm8er"):      # gemmeK M N K: with a real epilogue (K = b / g / r), fp8 with gemm8eK
            gemm_epi(int(args[1]), int(args[2]), int(args[3]), args[0][-1], f8=args[0].startswith("gemm8")); args = args[4:]
        elif args[0] in ("gemmst", "gemm8st", "gemmgst"):     # strip-count A/B of the 256-tile kernels' tile map
            gemm_strips(int(args[1]), int(args[2]), int(args[3]), f8=args[0] == "gemm8st", kind="g" if args[0] == "gemmgst" else "b"); args = args[4:]
        elif args[0] == "gemmg":                       # bf16 GEMM with the bias + tanh-GELU epilogue (FLUX ff1 / proj_mlp)
            gemm(int(args[1]), int(args[2]), int(args[3]), act=abi.ACT_GELU_TANH); args = args[4:]
        elif args[0] == "gemmp":
            gemm(int(args[1]), int(args[2]), int(args[3]), pad=int(args[4])); args = args[5:]
        elif args[0].rstrip("0123456789") in ("gemm", "gemm8", "gemmn", "gemm8n", "gemms", "gemm8s", "gemmfs", "gemm8fs"):
            # ...n: no split at all, ...sN: exactly N K slices (e.g. gemms4, gemm8s2)
            name = args[0].rstrip("0123456789")
            fl = abi.GEMM_NO_SPLIT if name.endswith("n") else 0
            if name.endswith("s"):
                fl = int(args[0][len(name):]) << 8
            if name.endswith("fs"):                    # gemmfsN: N slices also where the launcher would not slice (K shorter than 64 iterations)
                fl |= abi.GEMM_FORCE_TILE256
            gemm(int(args[1]), int(args[2]), int(args[3]), f8=args[0].startswith("gemm8"), flags=fl); args = args[4:]
        else:
            raise SystemExit(f"unknown benchmark {args[0]}")
