"""Summarise a rocprofv3 counter pass over tools/bench_kernels.py (tools/pmc_kernels.sh with
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE")
into profiles/r01_pmc_mfma_util.json:   python tools/summarize_pmc.py <counter_collection.csv> <kernel_trace.csv> "<command line>" """
import collections
import csv
import json
import sys
from pathlib import Path

import numpy as np

counters, trace, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
out_path = Path(sys.argv[4]) if len(sys.argv) > 4 else Path(__file__).resolve().parent.parent / "profiles" / "r01_pmc_mfma_util.json"
# one entry per kernel AND grid size: the same kernel runs several problem sizes in one pass
agg = collections.defaultdict(lambda: collections.defaultdict(list))
names = set()
for r in csv.DictReader(open(counters)):
    agg[f'{r["Kernel_Name"]} grid={r.get("Grid_Size", "")}'][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names.add(r["Counter_Name"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    g = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
    dur[f'{r["Kernel_Name"]} grid={g}'].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"_doc": f"rocprofv3 --pmc {' '.join(sorted(names))} "
               f"--kernel-trace -- python tools/bench_kernels.py {cmd} (counters only; the attention launches carry "
               "MTX_ATTN_Q_PRESCALED like the FLUX graph). Per-launch averages. mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
               "GRBM_GUI_ACTIVE / 8 XCDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs (cross-check: it gives a 1.8-2.2 GHz clock against "
               "the traced duration). lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.", "kernels": {}}
for k, d in agg.items():
    if "mtx" not in k:
        continue
    m = {c: float(np.mean(v)) for c, v in d.items()}
    du = float(np.mean(dur[k])) if k in dur else None
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    out["kernels"][k] = {"launches": len(d["GRBM_GUI_ACTIVE"]), "avg_duration_us": du / 1e3 if du else None,
                         "counters_per_launch": {c: round(v) for c, v in m.items()}, "gpu_cycles": round(cyc),
                         "effective_clock_ghz": round(cyc / du, 3) if du else None,
                         "mfma_util": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc), 4),
                         "lds_conflict_frac": round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 5) if m.get("SQ_LDS_IDX_ACTIVE") else None}
    e = out["kernels"][k]
    print(k[:70], e["avg_duration_us"], e["effective_clock_ghz"], e["mfma_util"], e["lds_conflict_frac"])
json.dump(out, open(out_path, "w"), indent=1)
