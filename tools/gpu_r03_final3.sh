#!/bin/bash
# round 3: LDS counter pass (bank conflicts, LDS-active cycles) on the hot kernels — the figure r03_pmc_mfma_util.json left null
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="gemm 8812 9216 3072 attn 8812 conv 1536 1024 gemm8 8512 27648 3072"
rm -rf /tmp/pmc_l; mkdir -p /tmp/pmc_l
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_l -o k -- python $R/tools/bench_kernels.py $ARGS 2>&1 | grep -v "^[WE]2026" | tail -6)
python tools/summarize_pmc.py "$(find /tmp/pmc_l -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_l -name '*kernel_trace.csv' | head -1)" "$ARGS" gpurun_out/r03_pmc_lds.json
