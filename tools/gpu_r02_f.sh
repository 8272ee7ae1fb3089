#!/bin/bash
# round 2, GPU visit F: slot decomposition of the RCAN conv (tools/probes/conv_probe), YOLO11 / launch tests after their fixes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
echo "== conv probe 1024x1536"; timeout 120 tools/probes/conv_probe 1536 1024 2>&1 | head -150
echo "== conv probe 2048x3072 (timings only)"; timeout 120 tools/probes/conv_probe 3072 2048 2>&1 | head -8
echo "== tests"; timeout 1500 python -m pytest tests/test_yolo11_gpu.py tests/test_bench_launch.py -q -s -m gpu --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | head -120
} > gpurun_out/r02_f.log 2>&1
tail -200 gpurun_out/r02_f.log
