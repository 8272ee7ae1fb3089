#!/bin/bash
# round 6, visit N: (VERDICT r05 #4c) the bf16 256-tile GEMM with its tile plane in 1 / 2 / 4 / 8 column strips (4 strips = an A-panel raster two XCDs wide),
# each in its own counter pass: duration, effective clock and MFMA busy side by side; the same pass over the attention kernels with 16-bit and fp8 scores
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
pass() {   # pass <tag> <bench_kernels args>
  tag=$1; shift
  rm -rf /tmp/pmc_$tag; mkdir -p /tmp/pmc_$tag
  (cd /tmp && timeout 500 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc_$tag -o k -- python $R/tools/bench_kernels.py "$@" 2>&1 | grep -E "^gemm|^attn")
  python tools/summarize_pmc.py "$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_$tag -name '*kernel_trace.csv' | head -1)" "$*" gpurun_out/r06_pmc_$tag.json
}
{
  for st in 1 2 4 8; do
    echo "== MTX_GEMM_STRIPS=$st"
    MTX_GEMM_STRIPS=$st pass gemm_bf16_strips_$st gemm 8812 9216 3072 gemm 8812 12288 3072
  done
  echo "== attention: 16-bit scores (attn, attnq) and fp8 scores (attn8w, attn8), T = 8704; fp8 GEMM"
  pass attention_scores attn 8704 attnq 8704 attn8w 8704 attn8 8704 gemm8 8512 27648 3072
} > gpurun_out/r06_visit_n.log 2>&1
cat gpurun_out/r06_visit_n.log
