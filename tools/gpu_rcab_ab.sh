#!/bin/bash
# round 2, GPU visit W: RCAB in three launches (pool before the conv) — parity and timing against the four-launch form
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== parity"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_bubble_crops_gpu.py -q -m gpu -s -k "conv or rcan or rcab or bubble" 2>&1 | grep -v "^$" | tail -14
echo "== whole RCAN, new form"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_upscale_only.json; python -c "import json; d=json.load(open('gpurun_out/r02_bench_upscale_only.json')); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
echo "== whole RCAN, four-launch form"; timeout 300 python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.core.ml.rcan import RCANUpscaler
from oracle.rcan_ref import make_state_dict
lib = get_library(); lib.init(0)
sd = make_state_dict(n_feats=64, n_resgroups=10, n_resblocks=20, seed=0)
for flag in (True, False, True, False):
    m = RCANUpscaler(sd, device="cuda:0", lib=lib, pool_before_conv=flag)
    p = m.plan_for(1, 1536, 1024)
    p.time(3, graph=True)
    print("pool_before_conv", flag, "ms/page", round(p.time(10, graph=True), 2), flush=True)
    del m, p
PY
echo "== rocprof upscale"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_up -o up -- python $GRAFT_REPO_ROOT/bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1); for f in $(find /tmp/prof_up -name "*kernel_stats.csv"); do cp $f gpurun_out/r02_bench_upscale_only_kernel_stats.csv; head -9 $f | cut -c1-150; done
} > gpurun_out/r02_w.log 2>&1
cat gpurun_out/r02_w.log
