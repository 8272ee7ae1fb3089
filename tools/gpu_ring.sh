set -x
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py -x -q --tb=short -k "conv or rcan" 2>&1 | tail -3
timeout 600 python tools/probe_conv_ab.py 0 7 5 2>&1 | tail -7
