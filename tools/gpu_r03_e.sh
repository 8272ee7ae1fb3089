#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{ timeout 600 python tools/profile_detect_host.py 2>&1 | grep -v "^{" | cut -c1-170; } > gpurun_out/r03_e.log 2>&1
tail -90 gpurun_out/r03_e.log
