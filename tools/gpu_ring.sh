set -x
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "gemm" 2>&1 | tail -5
