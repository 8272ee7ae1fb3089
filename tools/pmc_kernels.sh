#!/bin/bash
# PMC pass over the two MFMA-bound kernels (counters only; never combined with sys/hip trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ARGS="${@:-gemm 8192 8192 8192 attn 8652}"
(cd /tmp && rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc" -o k -- python "$GRAFT_REPO_ROOT/tools/bench_kernels.py" $ARGS > "$GRAFT_REPO_ROOT/gpurun_out/pmc/log.txt" 2>&1)
tail -3 gpurun_out/pmc/log.txt
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    if "mtx" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / n:16.0f}")
PY
