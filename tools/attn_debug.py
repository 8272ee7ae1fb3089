"""one-wave-per-SIMD attention against the 8-wave kernel on the same inputs, same process: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.hip.plan import PlanBuilder
lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")
def run(T, Tk, heads, w64, seed=0, qmul=1.0, dt=torch.bfloat16):
    os.environ["MTX_ATTN_W64"] = "1" if w64 else "0"
    g = torch.Generator(device=dev).manual_seed(seed)
    d = 128; D = heads * d
    q = (torch.randn(T, D, device=dev, generator=g) * qmul * d ** -0.5 * 1.4426950408889634).to(dt)
    k = torch.randn(Tk, D, device=dev, generator=g).to(dt)
    v = torch.randn(Tk, D, device=dev, generator=g).to(dt)
    pb = PlanBuilder(lib, dev, abi.BF16 if dt == torch.bfloat16 else abi.F16)
    o = pb.buf((T, D), dt, zero=True)
    pb.attention(q, k, v, o, 1, heads, T, Tk, d, (0, D, d), (0, D, d), (0, D, d), (0, D, d), d ** -0.5, q_prescaled=True)
    pb.build().run(); torch.cuda.synchronize()
    return o.float().view(T, heads, d)
for dbg in ("0", "1"):
  os.environ["MTX_ATTN_W64_DBG"] = dbg
  print("==== MTX_ATTN_W64_DBG =", dbg, "(1: every block redone on the slow path)")
  for (T, Tk, heads, qmul, dt) in ((1024, 256, 1, 1.0, torch.bfloat16), (1100, 449, 2, 40.0, torch.bfloat16), (1100, 449, 2, 40.0, torch.float16), (1100, 449, 2, 1.0, torch.float16)):
    a, b = run(T, Tk, heads, True, qmul=qmul, dt=dt), run(T, Tk, heads, False, qmul=qmul, dt=dt)
    print(dt, "qmul", qmul)
    err = (a - b).abs()
    ratio = (a.flatten(1).abs().sum(1) / b.flatten(1).abs().sum(1).clamp_min(1e-9))
    print("   ratio |w64| / |ref| of rows 0..39:", [round(v, 2) for v in ratio[:40].tolist()])
    rows = err.amax(dim=(1, 2)); bad = (rows > 0.05).nonzero().flatten()
    print(f"T={T} Tk={Tk} heads={heads}: max abs diff {err.max().item():.4f}, nan {torch.isnan(a).sum().item()}, bad rows {len(bad)} of {T}: first {bad[:12].tolist()} ... per head bad {[(err[:, h].amax(1) > 0.05).sum().item() for h in range(heads)]}")
    if len(bad):
        r = bad[0].item(); h = err[r].amax(1).argmax().item()
        print("   row", r, "head", h, "w64", a[r, h, :8].tolist(), "\n   ref", b[r, h, :8].tolist(), "\n   bad cols", (err[r, h] > 0.05).nonzero().flatten()[:40].tolist())
