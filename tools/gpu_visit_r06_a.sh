#!/bin/bash
# round 6, visit A: counters before code.
#   (1) where a slot of the persistent RCAN conv goes, for the two forms the RCAB pair really runs (conv1 = ReLU + sums, conv2 = scale + residual)
#   (2) the conv's wait / LDS counters (the pass r05 had only for GEMM and attention)
#   (3) the 256-tile GEMMs' tile map in 1 / 2 / 4 / 8 column strips, same process, fp8 and bf16 FLUX shapes
#   (4) fabric-side read traffic of the fp8 GEMM with one strip (rounds 1-5) against the launcher's choice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
  echo "== (1) conv_probe 1024x1536"
  timeout 120 tools/probes/conv_probe 1536 1024
  echo "== (3) tile-map strips A/B"
  timeout 900 python tools/bench_kernels.py gemm8st 8512 27648 3072 gemm8st 8512 9216 3072 gemm8st 8512 3072 12288 gemm8st 8000 18432 3072 \
      gemmst 8812 9216 3072 gemmgst 8812 12288 3072 gemmst 8812 3072 15360 gemmst 8300 3072 12288 gemmst 8300 3072 3072 2>&1 | grep -v "^[WE]2026"
  echo "== (2) conv counters"
  rm -rf /tmp/pmc_c; mkdir -p /tmp/pmc_c
  ARGS="conv 1536 1024 conv 3072 2048"
  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_c -o k -- python $R/tools/bench_kernels.py $ARGS 2>&1 | grep -v "^[WE]2026" | tail -3)
  python tools/summarize_pmc.py "$(find /tmp/pmc_c -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_c -name '*kernel_trace.csv' | head -1)" "$ARGS" gpurun_out/r06_pmc_conv_waits.json
  rm -rf /tmp/pmc_d; mkdir -p /tmp/pmc_d
  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_d -o k -- python $R/tools/bench_kernels.py $ARGS 2>&1 | grep -v "^[WE]2026" | tail -3)
  python tools/summarize_pmc.py "$(find /tmp/pmc_d -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_d -name '*kernel_trace.csv' | head -1)" "$ARGS" gpurun_out/r06_pmc_conv_insts.json
  python - <<'PY'
import json
for f in ("gpurun_out/r06_pmc_conv_waits.json", "gpurun_out/r06_pmc_conv_insts.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "missing", e); continue
    for k, v in d["kernels"].items():
        print(k[:70], round(v["avg_duration_us"], 1), "us", v["effective_clock_ghz"], "GHz", {c: n for c, n in v["counters_per_launch"].items()})
PY
  echo "== (4) fp8 / bf16 GEMM fabric traffic: one strip, then the launcher's choice"
  for st in 1 0; do
    MTX_GEMM_STRIPS=$st bash tools/pmc_traffic.sh gemm8 8512 27648 3072 gemm 8812 9216 3072 2>&1 | tail -4
    cp gpurun_out/r02_pmc_traffic.json gpurun_out/r06_pmc_traffic_gemm_strips_$st.json
  done
} > gpurun_out/r06_visit_a.log 2>&1
cat gpurun_out/r06_visit_a.log
