set -x
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "gemm" 2>&1 | tail -3
timeout 900 python tools/bench_kernels.py clamp 8704 9216 3072 clamp 8704 12288 3072 clamp 8704 3072 12288 clamp 8704 21504 3072 clamp 8704 3072 15360 clamp 8652 3072 3072 2>&1 | tail -13
