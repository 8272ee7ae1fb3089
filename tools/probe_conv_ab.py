"""interleaved A/B of the RCAN conv (64->64 @1024x1536, f16) variants selected by MTX_C64_ABL in one process"""
import os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.hip.plan import PlanBuilder
lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")
pb = PlanBuilder(lib, dev, abi.F16)
x = pb.act(1, 1536, 1024, 64); x.t.normal_()
wt = pb.const(torch.randn(64, 9, 64) / 24.0, pb.tdtype)
y = pb.conv2d(x, wt, pb.const(torch.zeros(64)), 64, 3, 1, act=abi.ACT_RELU)
plan = pb.build(); plan.run(); torch.cuda.synchronize()
ref = y.t.clone()
modes = sys.argv[1:] or ["0", "5"]
res = {m: [] for m in modes}
for r in range(5):
    for m in modes:
        os.environ["MTX_C64_ABL"] = m
        plan.run(); torch.cuda.synchronize()
        if r == 0 and m not in ("1", "3", "4"):
            print("mode", m, "max abs diff vs default", float((y.t.float() - ref.float()).abs().max()))
        plan.time(5)
        res[m].append(plan.time(30))
for m, v in res.items():
    ms = min(v)
    print(f"ABL={m} conv64 {ms*1000:.1f} us  {2*9*64*64*1536*1024/ms/1e9:.0f} TF/s  {402.7/ms:.0f} GB/s")
