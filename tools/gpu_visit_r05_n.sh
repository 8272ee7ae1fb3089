#!/bin/bash
# round 5, visit n: Newton-refined reciprocal in the GELU / SwiGLU epilogues only — extreme values, detectors back at their figures, FLUX parity, timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_yolo11_gpu.py tests/test_flux_gpu.py tests/test_flux2_gpu.py -q -x -s -p no:cacheprovider -k "extreme or yolo11l or yolo12x or gemm or glu or dit_step or kontext_loop or klein or full_width" 2>&1 | grep -E "^YOLO|^\.+YOLO|passed|failed|Error" | tail -8
  timeout 300 python tools/bench_kernels.py gemmg 8812 12288 3072 gemmg 8300 12288 3072 glu 8512 9216 3072 9216 gemmn 8812 12288 3072 2>&1 | grep "^gemm"
} > gpurun_out/r05_visit_n.log 2>&1
cat gpurun_out/r05_visit_n.log
