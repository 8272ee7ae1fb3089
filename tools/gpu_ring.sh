set -x
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -3
timeout 900 python tools/bench_kernels.py attn_ab 8704 attn_ab 4096 2>&1 | tail -12
