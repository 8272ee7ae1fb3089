set -x
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "gemm" 2>&1 | tail -3
timeout 900 python tools/bench_kernels.py ab 8704 9216 3072 ab 8704 12288 3072 ab 8704 3072 12288 ab 8704 21504 3072 ab 8192 8192 8192 2>&1 | tail -16
