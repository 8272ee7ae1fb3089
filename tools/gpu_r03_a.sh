#!/bin/bash
# round 3, GPU visit A: (1) SQ counter pass (MFMA busy, VALU instructions, LDS issue stalls, wait buckets, clock) over the hot kernels at
# the bench's shapes -> gpurun_out/r03_pmc_mfma_util_visit_a.json; (2) isolated timings of every block-linear shape of a Kontext step,
# with and without leading-dimension padding (address-interleave probe for the long-K, N = 3072 shapes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="gemm 8812 9216 3072 gemm 8812 3072 15360 attn 8812 conv 1536 1024 gemm8 8512 27648 3072"
rm -rf /tmp/pmc_a; mkdir -p /tmp/pmc_a
{
echo "== SQ counter pass"
(cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_a -o k -- python $R/tools/bench_kernels.py $ARGS 2>&1 | tail -8)
CC=$(find /tmp/pmc_a -name "*counter_collection.csv" | head -1); KT=$(find /tmp/pmc_a -name "*kernel_trace.csv" | head -1)
python tools/summarize_pmc.py "$CC" "$KT" "$ARGS" gpurun_out/r03_pmc_mfma_util_visit_a.json
echo "== isolated block-linear shapes (no profiler)"
timeout 400 python tools/bench_kernels.py gemm 8812 9216 3072 gemm 8812 12288 3072 gemm 8812 3072 15360 gemmp 8812 3072 15360 64 gemmp 8812 3072 15360 32 \
  gemm 8300 3072 12288 gemmp 8300 3072 12288 64 gemm 8300 3072 3072 gemmp 8300 3072 3072 64 gemm 8300 12288 3072 gemmp 8812 9216 3072 64 \
  gemm 8192 8192 8192 gemm 8812 3072 15360 attn 8812 conv 1536 1024
} > gpurun_out/r03_a.log 2>&1
tail -60 gpurun_out/r03_a.log
