import os, sys
os.environ["MTX_C64_ABL"] = "9"
from pathlib import Path
import torch, numpy as np
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
dbg = pb.buf((256 * 16 * 8,), torch.int64, zero=True)
y = pb.conv2d(x, wt, pb.const(torch.zeros(64)), 64, 3, 1, act=abi.ACT_RELU, chan_sum=dbg)
plan = pb.build()
for _ in range(3): plan.run()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(256, 16, 8)[:, :12, :7].astype(np.float64)
names = ["issue loads", "mfma", "barrier3", "write_halo(wait loads)", "stores issue", "barrier6"]
ph = d[:, :, 1:] - d[:, :, :-1]
print("per-tile phase cycles (mean over 256 blocks x 12 tiles; counter = s_memtime/readcyclecounter):")
for i, n in enumerate(names): print(f"  {n:26s} mean {ph[:, :, i].mean():9.0f}  p90 {np.percentile(ph[:, :, i], 90):9.0f}")
tot = d[:, :, 6] - d[:, :, 0]
print("  tile total mean", tot.mean(), " block total", (d[:, 11, 6] - d[:, 0, 0]).mean())
print("  tile-to-tile gap", (d[:, 1:, 0] - d[:, :-1, 6]).mean())
