set -x
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" timeout 600 bash tools/pmc_kernels.sh attn 8704 gemm 8704 9216 3072 gemm 8704 3072 12288 > gpurun_out/pmc_mfma.log 2>&1; tail -4 gpurun_out/pmc_mfma.log
cp gpurun_out/pmc/k_counter_collection.csv gpurun_out/pmc_mfma_counters.csv; cp gpurun_out/pmc/k_kernel_trace.csv gpurun_out/pmc_mfma_trace.csv
