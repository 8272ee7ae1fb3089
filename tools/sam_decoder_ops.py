"""Per-op GPU time of SAM-2.1's mask decoder plan (Hiera-L geometry, seeded weights, 8 boxes) for precision "fast" and "high": where the
decoder's milliseconds go (round 5: the fp32 decoder is 4.4 ms against 1.6 ms in 16-bit storage).
    python tools/sam_decoder_ops.py [boxes]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from mangatranslator_amd.core.ml.sam2 import Sam2Hip  # noqa: E402
from mangatranslator_amd.hip import abi  # noqa: E402
from mangatranslator_amd.hip.lib import get_library  # noqa: E402
from mangatranslator_amd.utils import synthetic_checkpoints as synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lib = get_library(); lib.init(0)
    cfg = synth.sam2_hiera_large_config()
    sd = synth.sam2_state_dict(cfg, seed=11)
    for precision in ("fast", "high"):
        sam = Sam2Hip(sd, cfg, device="cuda:0", lib=lib, dtype=abi.F16, precision=precision)
        pre, enc, dec, post = sam.plans(n, 1536, 1024)
        dec.run(); torch.cuda.synchronize()
        total = dec.time(10)
        rows = []
        for i in range(dec.n_ops):
            dec.time_range(i, i, 2)
            rows.append((dec.time_range(i, i, 10), dec.labels[i]))
        print(f"== precision {precision}: decoder {total:.3f} ms as a plan, {sum(r[0] for r in rows):.3f} ms op by op, {dec.n_ops} ops")
        groups = {}
        for ms, lab in rows:
            key = lab.split(".")[-1] if lab.startswith("dec") and lab[3:4].isdigit() else lab
            groups.setdefault(key, [0.0, 0])
            groups[key][0] += ms; groups[key][1] += 1
        for key, (ms, k) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:18]:
            print(f"   {ms:7.3f} ms  x{k:<3d} {key}")


if __name__ == "__main__":
    main()
