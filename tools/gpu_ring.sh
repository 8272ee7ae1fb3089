set -x
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "prescaled or variants or long_sequence" 2>&1 | tail -5
