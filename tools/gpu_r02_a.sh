#!/bin/bash
# round 2, GPU visit A: the new kernels on hardware (fp8 path, FLUX.2 graphs), kernel micro-benchmarks, a first config-5 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== ops (fp8 / quantise / swiglu / 256-tile gemm)"; timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -s -k "fp8 or quantize or swiglu or gemm_256 or stream_k" 2>&1 | tail -15
echo "== kernels"; timeout 300 python tools/bench_kernels.py gemm 8704 9216 3072 gemm8 8704 9216 3072 gemm 8704 3072 3072 gemm8 8704 3072 3072 gemm 8704 27648 3072 gemm8 8704 27648 3072 gemm 8704 3072 12288 gemm8 8704 3072 12288 gemm8 8704 18432 3072 gemm8 8704 3072 9216 quant 8704 3072 quant 8704 12288 attn 8704 conv 1536 1024 conv 3072 2048 2>&1 | tail -20
echo "== flux2"; timeout 1500 python -m pytest tests/test_flux2_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -30
echo "== bench config 5 (short)"; timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench5_short.err | grep '^{' > gpurun_out/bench5_short.json; tail -c 3000 gpurun_out/bench5_short.json; tail -5 gpurun_out/bench5_short.err
} > gpurun_out/r02_a.log 2>&1
tail -120 gpurun_out/r02_a.log
