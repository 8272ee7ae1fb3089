"""Plan lanes (include/mtx_hip.h, MTX_LANE_*): a chain of GEMMs on the side lane beside a chain on the main lane, joined by an op that
reads both; the result must equal the single-lane plan's bit for bit — eagerly and as a hipGraph replay."""
import torch

from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.plan import PlanBuilder


def build(lib, device, lanes, m_main=320, m_side=64, d=128, depth=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    pb = PlanBuilder(lib, device, abi.BF16, lanes=lanes)
    x = pb.buf((m_main + m_side, d), torch.bfloat16)
    x.copy_(torch.randn(m_main + m_side, d, generator=g).to(torch.bfloat16))
    ws = [pb.const(torch.randn(d, d, generator=g) / d ** 0.5, torch.bfloat16) for _ in range(2 * depth + 1)]
    a = pb.buf((m_main + m_side, d), torch.bfloat16)
    b = pb.buf((m_main + m_side, d), torch.bfloat16)
    src, dst = x, a
    for k in range(depth):
        with pb.side():           # rows [0, m_side): the short stream
            pb.gemm(src, ws[2 * k], m_side, d, d, out=dst, act=abi.ACT_GELU_TANH, label=f"side{k}")
        pb.gemm(src, ws[2 * k + 1], m_main, d, d, out=dst, a_off=m_side * d, c_off=m_side * d, act=abi.ACT_GELU_TANH, label=f"main{k}")
        src, dst = dst, (b if dst is a else a)
    pb.join()
    y = pb.gemm(src, ws[-1], m_main + m_side, d, d, label="joint")          # reads every row: both lanes must have landed
    plan = pb.build()
    return plan, y, pb


def check(lib, device, graph):
    p1, y1, pb1 = build(lib, device, lanes=True)
    p0, y0, _ = build(lib, device, lanes=False)
    lanes = [op.lane for op in pb1.ops]
    assert lanes == [abi.LANE_SIDE, 0] * 3 + [abi.LANE_JOIN], lanes
    for _ in range(3):            # replays too
        p1.run(graph=graph); p0.run(graph=graph)
    if device != "cpu":
        torch.cuda.synchronize()
    assert torch.equal(y1.cpu(), y0.cpu())
    assert float(y1.float().abs().mean()) > 0
