"""SAM-2.1 Hiera-L at 1024x1536, trained-model logit spread: what each part of precision "high" buys and costs (VERDICT r05 #1a).
Per arithmetic — "fast", each of the three parts alone (hi + lo weight pairs / fp32 residual stream / fp32 mask decoder), pairs, "high" — the
mask mismatch and logit error against the fp32 oracle (tests/sam2_checks.py) and the GPU time of the encoder / decoder graphs of the very
model that was checked.      python tools/sam_frontier.py out.json [variant ...]"""
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

VARIANTS = ["fast", "hilo", "stream32", "dec32", "hilo+stream32", "hilo+dec32", "high"]


def main():
    lib = get_library()
    lib.init(0)
    out = {}
    names = sys.argv[2:] or VARIANTS
    H, W, NB = 1536, 1024, 8
    for name in names:
        t = time.perf_counter()
        try:
            sc.check_sam2(lib, "cuda:0", "hiera_large", h=H, w=W, n_boxes=NB, seed=2, logit_tol=0.06, mask_tol=0.01, calibrated=True, dtype=abi.F16, precision=name)
            status = "ok"
        except AssertionError as e:
            status = f"assert: {str(e)[:200]}"
        torch.cuda.synchronize()
        row = dict(status=status, seconds=round(time.perf_counter() - t, 1), **sc.stats)
        hipm = sc.last.get("hip")
        if hipm is not None:
            pre, enc, dec, post = hipm.plans(NB, H, W)
            enc.time(3); dec.time(3)
            row["segment_ms"] = {"preprocess": pre.time(5), "encoder": min(enc.time(5) for _ in range(3)), "decoder": min(dec.time(5) for _ in range(3)), "upsample_threshold": post.time(5)}
            row["encoder_launches"] = len(enc.labels)
        out[name] = row
        print(name, json.dumps({k: v for k, v in row.items() if k != "stability_scores"}), flush=True)
        sc.last.clear()
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
