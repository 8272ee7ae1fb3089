#!/bin/bash
# round 2, GPU visit E: the four tests that failed in visit D with their full output; RCAN conv A/B (r01 loop / buffer stores only / new loop)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
echo "== conv A/B"
for lib in tools/probes/build/libmtx_conv_HEAD1.so tools/probes/build/libmtx_conv_HEAD.so mangatranslator_amd/csrc/libmtx_hip.so; do
  echo "-- $lib"
  MTX_HIP_LIBRARY=$R/$lib timeout 300 python tools/bench_kernels.py conv 1536 1024 conv 3072 2048 2>&1 | tail -3
  MTX_HIP_LIBRARY=$R/$lib timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
done
echo "== conv / RCAN parity on the new kernel"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_bubble_crops_gpu.py -q -m gpu -s -k "conv or rcan or bubble" 2>&1 | grep -v "^$" | tail -30
echo "== failing tests, full output"; timeout 1500 python -m pytest tests/test_yolo11_gpu.py tests/test_bench_launch.py -q -s -m gpu --tb=short 2>&1 | grep -v "^$" | cut -c1-400 | head -300
} > gpurun_out/r02_e.log 2>&1
tail -150 gpurun_out/r02_e.log
