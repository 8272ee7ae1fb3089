#!/bin/bash
# round 5, visit c: attn_x with the default kernel's register-staged transport (+8): parity, same-process A/B, LDS counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity of the register-staged schedules (tests/test_ops_gpu.py)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "alternative_schedules and (9 or 10 or 11 or 12 or 13 or 15 or 16)" 2>&1 | tail -4
  echo "== A/B, T = 8812, 24 heads: 0 = default kernel; 9 = attn_x staged; 10 = + stagger; 11 = + matrix-pipe sums; 13 = + wide stores; 15 = sums + wide; 12 / 16 = stagger + sums (+ wide); 4 = DMA + sums"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,9,10,11,13,15,12,16,4 3 2>&1 | grep -v "^$" | tail -32
  echo "== LDS counters, default kernel and schedule 11"
  mkdir -p gpurun_out/pmc_attn
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_attn" -o k -- python "$GRAFT_REPO_ROOT/tools/bench_kernels.py" attnx 8812 0,11 1 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_attn/log.txt" 2>&1)
  tail -3 gpurun_out/pmc_attn/log.txt
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob("gpurun_out/pmc_attn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    if "attn" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / n:16.0f}")
PY
  rm -rf gpurun_out/pmc_attn/*/*.db 2>/dev/null
} > gpurun_out/r05_visit_c.log 2>&1
cat gpurun_out/r05_visit_c.log
