#!/bin/bash
# round 2, GPU visit B: fp8 path after the layout fix, the whole GPU suite, config-5 bench with its stdout kept
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== ops"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -8
echo "== flux2"; timeout 1500 python -m pytest tests/test_flux2_gpu.py -x -q -s 2>&1 | grep -i "flux\|klein\|passed\|failed\|error" | tail -30
echo "== rest of the gpu suite"; timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_ops_gpu.py --deselect tests/test_flux2_gpu.py 2>&1 | tail -15
echo "== bench config 5 (short)"; timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench5_short.out 2>gpurun_out/bench5_short.err; grep '^{' gpurun_out/bench5_short.out > gpurun_out/bench5_short.json; grep -v '^{' gpurun_out/bench5_short.out | grep -v "OSB text model unavailable" | sort | uniq -c | sort -rn | head -12; tail -3 gpurun_out/bench5_short.err
} > gpurun_out/r02_b.log 2>&1
tail -100 gpurun_out/r02_b.log
