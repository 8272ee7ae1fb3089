#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of a tools/bench_kernels.py invocation
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/kt
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/kt" -o k -- python "$GRAFT_REPO_ROOT/tools/bench_kernels.py" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/kt/log.txt" 2>&1)
grep -v rocprofv3 gpurun_out/kt/log.txt | tail -4
python - <<'PY'
import csv, glob
for fn in glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:8]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
find gpurun_out/kt -name "*kernel_trace.csv" -delete
