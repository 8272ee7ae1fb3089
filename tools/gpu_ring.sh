set -x
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "variants" 2>&1 | tail -5
timeout 900 python tools/bench_kernels.py attn_ab 8192 attn_ab 8704 attn_ab 4096 2>&1 | tail -8
MTX_ATTN_NOSPLIT=1 timeout 900 python tools/bench_kernels.py attn_ab 8704 2>&1 | tail -2
