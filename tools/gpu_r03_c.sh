#!/bin/bash
# round 3, GPU visit C: channel attention on 32 workgroups per image (ticket + last arriver) vs round 2's single workgroup: op + RCAN tests,
# whole-RCAN time per page with both libraries on the same box; the op suite under the tightened bf16 tolerance
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
echo "== op + rcan tests on the GPU"
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_bubble_crops_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -8
for rep in 1 2; do
  echo "== upscale only, round 3 library (rep $rep)"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
  echo "== upscale only, round 2 library (rep $rep)"; MTX_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/probes/build/libmtx_r02.so timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
done
} > gpurun_out/r03_c.log 2>&1
tail -40 gpurun_out/r03_c.log
