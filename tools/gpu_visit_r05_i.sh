#!/bin/bash
# round 5, visit i: attn_x with fragment reads four steps ahead of the MFMAs (+16), alone and under the half-tile stagger
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity (deep-prefetch schedules 17, 18, 25, 26, 30 and the renumbered 65 / 66 / 67)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "alternative_schedules and (17 or 18 or 25 or 26 or 30 or 65 or 66 or 67)" 2>&1 | tail -4
  echo "== A/B, T = 8812: 0 default | 1 DMA | 17 DMA+deep | 2 DMA+stagger | 18 DMA+stagger+deep | 9 staged | 25 staged+deep | 10 staged+stagger | 26 staged+stagger+deep | 30 = 26 + wide"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,1,17,2,18,9,25,10,26,30 3 2>&1 | grep -v "^$" | tail -34
} > gpurun_out/r05_visit_i.log 2>&1
cat gpurun_out/r05_visit_i.log
