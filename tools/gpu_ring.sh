set -x
timeout 900 python tools/bench_kernels.py attn_ab 8192 attn_ab 8704 2>&1 | tail -7
