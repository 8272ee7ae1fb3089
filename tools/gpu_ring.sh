set -x
export MTX_BENCH_ONE_DEVICE=1
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --inpaint-steps 2 --backend gloo > gpurun_out/bench2.log 2> gpurun_out/bench2.err
echo "exit $?"; tail -c 700 gpurun_out/bench2.log; echo; grep -v "amdgpu.ids\|RuntimeWarning\|alive &=\|iou = " gpurun_out/bench2.err | tail -15
