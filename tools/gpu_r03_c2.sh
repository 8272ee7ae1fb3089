#!/bin/bash
# round 3, GPU visit C2: whole RCAN per page, channel attention on 32 workgroups per image vs one (same library, same box, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
{
timeout 600 python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.core.ml.rcan import RCANUpscaler
from oracle.rcan_ref import make_state_dict
lib = get_library(); lib.init(0)
sd = make_state_dict(n_feats=64, n_resgroups=10, n_resblocks=20, seed=0)
for hw in ((1536, 1024), (3072, 2048)):
    for flag in (True, False, True, False):
        m = RCANUpscaler(sd, device="cuda:0", lib=lib, ca_split=flag)
        p = m.plan_for(1, *hw)
        p.time(3, graph=True)
        print(hw, "ca_split", flag, "ms/page", round(min(p.time(10, graph=True) for _ in range(2)), 2), flush=True)
        del m, p
PY
echo "== rocprof upscale"; export TMPDIR=/tmp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_up -o up -- python $GRAFT_REPO_ROOT/bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1); for f in $(find /tmp/prof_up -name "*kernel_stats.csv"); do cp $f gpurun_out/r03_bench_upscale_only_kernel_stats.csv; head -9 $f | cut -c1-150; done
} > gpurun_out/r03_c2.log 2>&1
cat gpurun_out/r03_c2.log
