#!/bin/bash
# GPU-box visit: the attention-variant test in full, a short bench sanity run, the whole GPU suite, the default bench,
# a PMC pass (MFMA busy / LDS conflicts; counters only) over the two MFMA kernels and the RCAN conv.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -k "variants" > gpurun_out/variants.log 2>&1; tail -25 gpurun_out/variants.log
timeout 600 python bench.py --steps 1 --warmup 1 --inpaint-steps 2 --no-cpu-baseline > gpurun_out/bench_short.log 2> gpurun_out/bench_short.err; tail -c 600 gpurun_out/bench_short.log; tail -5 gpurun_out/bench_short.err
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -c 3000 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" timeout 600 bash tools/pmc_kernels.sh attn 8704 gemm 8704 9216 3072 gemm 8704 3072 12288 > gpurun_out/pmc_mfma.log 2>&1; tail -60 gpurun_out/pmc_mfma.log
cp gpurun_out/pmc/k_counter_collection.csv gpurun_out/pmc_mfma_counters.csv 2>/dev/null
cp gpurun_out/pmc/k_kernel_trace.csv gpurun_out/pmc_mfma_trace.csv 2>/dev/null
