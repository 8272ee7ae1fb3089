#!/bin/bash
# round 5, visit d: the tuned long-sequence kernel with 16-byte row stores (17) and with matrix-pipe row sums + 16-byte stores (18)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity (tests/test_ops_gpu.py): default kernel, schedules 17 / 18"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "attention_prescaled or (alternative_schedules and (17 or 18))" 2>&1 | tail -4
  echo "== A/B, T = 8812, 24 heads, 4 rounds in one process: 0 = default, 17 = wide stores, 18 = matrix-pipe row sums + wide stores, 15 = attn_x staged + sums + wide"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,17,18,15 4 2>&1 | grep -v "^$" | tail -20
  echo "== other sequence lengths (Klein at 2048x3072: T = 12800 + 512; SAM-free sanity at 4096)"
  timeout 600 python tools/bench_kernels.py attnx 13312 0,18 2 attnx 4096 0,18 2 2>&1 | grep "best of"
  echo "== counters: default kernel and schedule 18"
  mkdir -p gpurun_out/pmc_attn2
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_attn2" -o k -- python "$GRAFT_REPO_ROOT/tools/bench_kernels.py" attnx 8812 0,18 1 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_attn2/log.txt" 2>&1)
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob("gpurun_out/pmc_attn2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    if "attn_mma32" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / n:16.0f}")
    if d.get("GRBM_GUI_ACTIVE"): print(f"   mfma busy = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['GRBM_GUI_ACTIVE'] / 8):.3f}")
PY
  rm -rf gpurun_out/pmc_attn2
} > gpurun_out/r05_visit_d.log 2>&1
cat gpurun_out/r05_visit_d.log
