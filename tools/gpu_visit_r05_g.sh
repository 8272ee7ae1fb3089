#!/bin/bash
# round 5, visit g: per-op time of SAM's decoder (fast / high), forced K slices on short-K GEMM remainders, weight-broadcast timing (two ranks, one device)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== SAM-2.1 mask decoder, op by op"
  timeout 300 python tools/sam_decoder_ops.py 8 2>&1 | grep -v "^$" | tail -42
  echo "== K slices on the last partial wave of short-K GEMMs (48 iterations): unsplit / forced 2, 3, 4, 6 slices"
  for shape in "8300 12288 3072" "8812 12288 3072" "8300 9216 3072" "8300 3072 3072"; do
    timeout 300 python tools/bench_kernels.py gemmn $shape gemmfs2 $shape gemmfs3 $shape gemmfs4 $shape gemmfs6 $shape 2>&1 | grep "^gemm"
  done
  echo "== weight broadcast, two ranks on one device (gloo)"
  timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/time_weight_broadcast.py 1.0 2>&1 | grep -v "^\[W\|^W0\|Gloo\|^$" | tail -6
} > gpurun_out/r05_visit_g.log 2>&1
cat gpurun_out/r05_visit_g.log
