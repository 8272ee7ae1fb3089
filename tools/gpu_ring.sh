set -x
timeout 600 python tools/probe_rcan_ab.py 2>&1 | tail -5
