"""SAM-2.1 Hiera-L at 1024x1536 with trained-model logit spread: logit error and mask mismatch of the HIP path against the fp32 oracle for
bf16 and f16 storage, each also with precision "high" (round 4: which precision the `> 0` masks need; tests/sam2_checks.py does the comparison;
the timing of the segment stage with precision "high" is `bench.py --config 2 --sam-precision high`).
    python tools/sam_dtype_probe.py [out.json]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import torch  # noqa: E402

from mangatranslator_amd.hip import abi  # noqa: E402
from mangatranslator_amd.hip.lib import get_library  # noqa: E402
import sam2_checks as sc  # noqa: E402


def main():
    lib = get_library()
    lib.init(0)
    out = {}
    variants = [("bf16", dict(dtype=abi.BF16)), ("f16", dict(dtype=abi.F16)),
                ("f16 high", dict(dtype=abi.F16, precision="high")), ("bf16 high", dict(dtype=abi.BF16, precision="high"))]      # hi + lo trunk weights, fp32 mask decoder
    for name, kw in variants:
        t = time.perf_counter()
        try:
            sc.check_sam2(lib, "cuda:0", "hiera_large", h=1536, w=1024, n_boxes=8, seed=2, logit_tol=0.06, mask_tol=0.01, calibrated=True, **kw)
            status = "ok"
        except AssertionError as e:
            status = f"assert: {str(e)[:200]}"
        torch.cuda.synchronize()
        out[name] = dict(status=status, seconds=round(time.perf_counter() - t, 1), **sc.stats)
        print(name, json.dumps(out[name]))
    if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
