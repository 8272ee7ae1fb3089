#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== conv parity (hardware permlane16_swap semantics)"; timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py -q -m gpu -k "conv or rcan" 2>&1 | tail -4
echo "== conv probe"; timeout 120 tools/probes/conv_probe 1536 1024 2>&1 | head -18; timeout 120 tools/probes/conv_probe 3072 2048 2>&1 | head -18
echo "== bench kernels"; timeout 300 python tools/bench_kernels.py conv 1536 1024 conv 3072 2048 2>&1 | tail -2
echo "== whole RCAN"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
} > gpurun_out/r02_n.log 2>&1
cat gpurun_out/r02_n.log
