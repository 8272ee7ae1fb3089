#!/bin/bash
# round 3, GPU visit B: the one-launch stream-K GEMM (whole tiles + tail units + last-arriver fix-up) against round 2's three-launch
# form (tools/probes/build/libmtx_r02.so = the round-2 library, same box, alternating) on every block-linear shape of a Kontext / Klein step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
echo "== gemm tests on the GPU"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" -p no:cacheprovider 2>&1 | tail -5
SH="gemm 8812 3072 15360 gemm 8300 3072 12288 gemm 8300 3072 3072 gemm 8300 12288 3072 gemm 8812 12288 3072 gemm 8812 9216 3072 gemm 8300 9216 3072 gemm 512 3072 12288 gemm 512 9216 3072 gemm8 8512 27648 3072 gemm8 8512 3072 12288 gemm8 8000 18432 3072 gemm8 8000 3072 9216"
for rep in 1 2; do
  echo "== round 3 library (rep $rep)"; timeout 300 python tools/bench_kernels.py $SH 2>&1 | grep -v amdgpu.ids
  echo "== round 2 library (rep $rep)"; MTX_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/probes/build/libmtx_r02.so timeout 300 python tools/bench_kernels.py $SH 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r03_b.log 2>&1
tail -80 gpurun_out/r03_b.log
