set -x
PMC="FETCH_SIZE" timeout 600 bash tools/pmc_kernels.sh attn 8652 gemm 8624 9216 3072 > gpurun_out/pmc_f.log 2>&1; cp gpurun_out/pmc/k_counter_collection.csv gpurun_out/pmc_fetch.csv
PMC="WRITE_SIZE" timeout 600 bash tools/pmc_kernels.sh attn 8652 gemm 8624 9216 3072 > gpurun_out/pmc_w.log 2>&1; cp gpurun_out/pmc/k_counter_collection.csv gpurun_out/pmc_write.csv
tail -3 gpurun_out/pmc_w.log
