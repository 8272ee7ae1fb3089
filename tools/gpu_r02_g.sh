#!/bin/bash
# round 2, GPU visit G: RCAN conv with the interior-tile fast paths — probe, parity, whole RCAN
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
echo "== conv probe 1024x1536"; timeout 120 tools/probes/conv_probe 1536 1024 2>&1 | head -62
echo "== conv probe 2048x3072"; timeout 120 tools/probes/conv_probe 3072 2048 2>&1 | head -8
echo "== parity"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_bubble_crops_gpu.py tests/test_yolo_gpu.py -q -m gpu -s -k "conv or rcan or bubble or yolo" 2>&1 | grep -v "^$" | tail -25
echo "== bench kernels"; timeout 300 python tools/bench_kernels.py conv 1536 1024 conv 3072 2048 2>&1 | tail -2
echo "== whole RCAN"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_upscale_only.json; python -c "import json; d=json.load(open('gpurun_out/r02_bench_upscale_only.json')); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
} > gpurun_out/r02_g.log 2>&1
tail -120 gpurun_out/r02_g.log
