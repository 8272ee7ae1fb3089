#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== conv probe 1024x1536"; timeout 120 tools/probes/conv_probe 1536 1024 2>&1 | head -62
echo "== conv probe 2048x3072"; timeout 120 tools/probes/conv_probe 3072 2048 2>&1 | head -9
echo "== bench kernels"; timeout 300 python tools/bench_kernels.py conv 1536 1024 conv 3072 2048 2>&1 | tail -2
} > gpurun_out/r02_h.log 2>&1
tail -80 gpurun_out/r02_h.log
