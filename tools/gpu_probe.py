"""Micro-benchmarks of the individual kernels on the GPU box (HIP-event timing through
mtx_plan_time_range).  Writes gpurun_out/probe.json; guides kernel optimisation."""
import json
import math
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mangatranslator_amd.hip import abi  # noqa: E402
from mangatranslator_amd.hip.lib import get_library  # noqa: E402
from mangatranslator_amd.hip.plan import PlanBuilder  # noqa: E402

lib = get_library()
lib.init(0)
dev = torch.device("cuda:0")
out = []


def timeit(pb, iters=20):
    plan = pb.build()
    plan.run()
    torch.cuda.synchronize()
    plan.time(3)
    return plan.time(iters)


def conv(h, w, cin, cout, k=3, s=1, dtype=abi.F16):
    pb = PlanBuilder(lib, dev, dtype)
    x = pb.act(1, h, w, cin)
    x.t.normal_()
    wt = pb.const(torch.randn(cout, k * k, cin) / math.sqrt(cin * k * k), pb.tdtype)
    b = pb.const(torch.zeros(cout))
    y = pb.conv2d(x, wt, b, cout, k, s, act=abi.ACT_RELU)
    ms = timeit(pb)
    fl = 2.0 * k * k * cin * cout * y.h * y.w
    by = (cin * h * w + cout * y.h * y.w) * 2 + k * k * cin * cout * 2
    out.append(dict(op="conv", h=h, w=w, cin=cin, cout=cout, k=k, s=s, ms=ms, tflops=fl / ms / 1e9, gbs=by / ms / 1e6))
    print(out[-1], flush=True)


def gemm(m, n, k, dtype=abi.BF16):
    pb = PlanBuilder(lib, dev, dtype)
    a = pb.const(torch.randn(m, k), pb.tdtype)
    w = pb.const(torch.randn(n, k) / math.sqrt(k), pb.tdtype)
    pb.gemm(a, w, m, n, k, bias=pb.const(torch.zeros(n)))
    ms = timeit(pb)
    out.append(dict(op="gemm", m=m, n=n, k=k, ms=ms, tflops=2.0 * m * n * k / ms / 1e9,
                    gbs=(m * k + n * k + m * n) * 2 / ms / 1e6))
    print(out[-1], flush=True)


def attn(batch, heads, s, d, dtype=abi.BF16):
    pb = PlanBuilder(lib, dev, dtype)
    q = pb.const(torch.randn(batch, s, heads, d), pb.tdtype)
    k = pb.const(torch.randn(batch, s, heads, d), pb.tdtype)
    v = pb.const(torch.randn(batch, s, heads, d), pb.tdtype)
    o = pb.buf((batch, s, heads, d), pb.tdtype)
    st = (s * heads * d, heads * d, d)
    pb.attention(q, k, v, o, batch, heads, s, s, d, st, st, st, st, 1.0 / math.sqrt(d))
    ms = timeit(pb, 10)
    out.append(dict(op="attn", batch=batch, heads=heads, s=s, d=d, ms=ms, tflops=4.0 * batch * heads * s * s * d / ms / 1e9))
    print(out[-1], flush=True)


def ew(h, w, c):
    pb = PlanBuilder(lib, dev, abi.F16)
    a, b = pb.act(1, h, w, c), pb.act(1, h, w, c)
    s = pb.const(torch.rand(1, c))
    pb.ew(abi.EW_SCALE_RES, a, b=b, s=s, lds=c)
    ms = timeit(pb)
    out.append(dict(op="scale_res", h=h, w=w, c=c, ms=ms, gbs=3 * h * w * c * 2 / ms / 1e6))
    print(out[-1], flush=True)


conv(1536, 1024, 64, 64)
conv(1536, 1024, 64, 64, dtype=abi.BF16)
conv(1536, 1024, 64, 256)
conv(3072, 2048, 64, 8)
conv(1536, 1024, 8, 64)
conv(800, 544, 48, 96, 3, 2)
conv(400, 272, 96, 96)
conv(200, 136, 192, 192)
conv(100, 68, 384, 384)
conv(1024, 1024, 128, 128)
conv(512, 512, 512, 512)
conv(200, 136, 384, 192, 1, 1)
ew(1536, 1024, 64)
gemm(8704, 3072, 3072)
gemm(8704, 12288, 3072)
gemm(8704, 3072, 15360)
gemm(65536, 432, 144)
gemm(4096, 1728, 576)
gemm(4096, 2304, 576)
gemm(4096, 576, 2304)
attn(1, 24, 8704, 128)
attn(1, 8, 4096, 72)
attn(16, 8, 256, 72)
attn(1024, 2, 64, 72)
os.makedirs(ROOT / "gpurun_out", exist_ok=True)
json.dump(out, open(ROOT / "gpurun_out" / "probe.json", "w"), indent=1)
