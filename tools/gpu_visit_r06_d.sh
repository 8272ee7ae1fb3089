#!/bin/bash
# round 6, visit D: where an iteration of the lock-step conv goes (stamps), same-process timing against the two-group kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{ timeout 200 tools/probes/conv_probe 1536 1024 | grep -v "slot\|^workgroup [0-9]* group\|---- stamps"; } > gpurun_out/r06_visit_d.log 2>&1
cat gpurun_out/r06_visit_d.log
