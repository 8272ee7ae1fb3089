#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MTX_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 1 --warmup 1 --inpaint-steps 2 --backend gloo --no-cpu-baseline > gpurun_out/bench2.log 2> gpurun_out/bench2.err; echo "2-rank exit $?"; tail -c 200 gpurun_out/bench2.log; echo
bash tools/gpu_final.sh
