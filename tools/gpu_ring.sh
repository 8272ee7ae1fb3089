set -x
timeout 900 python tools/bench_kernels.py mintiles 512 9216 3072 mintiles 512 3072 3072 mintiles 512 12288 3072 mintiles 4096 3072 3072 mintiles 1024 4608 1152 mintiles 4096 1152 4608 2>&1 | tail -13
