"""conv 64->64 @1024x1536 only (for rocprofv3 --pmc and the ablation variants MTX_C64_ABL=1..4)."""
import math, os, sys
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
plan.time(5)
iters = int(os.environ.get("ITERS", "30"))
ms = plan.time(iters)
print(f"ABL={os.environ.get('MTX_C64_ABL','0')} conv64 {ms*1000:.1f} us  {2*9*64*64*1536*1024/ms/1e9:.0f} TF/s  {402.7/ms:.0f} GB/s", flush=True)
