#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{ timeout 600 python tools/detector_concurrency_probe.py 2>&1 | grep -v amdgpu.ids; echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 600 python tools/detector_concurrency_probe.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r03_f4.log 2>&1
cat gpurun_out/r03_f4.log
