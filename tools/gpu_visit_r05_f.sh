#!/bin/bash
# round 5, visit f: fp32 GEMM on the matrix pipe (SAM precision "high"), K slices on short-K GEMM remainders, two ranks on one device with the Kontext weight set
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== fp32 ops incl. the v_mfma_f32_32x32x2_f32 GEMM, hi + lo weights, SAM high precision (tests)"
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -q -x -p no:cacheprovider -k "f32_ops or hi_lo or high_precision" 2>&1 | tail -4
  echo "== config 2 with SAM precision high: segment_ms"
  timeout 300 python bench.py --config 2 --steps 30 --warmup 3 --no-cpu-baseline --no-traffic --sam-precision high > gpurun_out/r05_c2_high_mfma32.out 2> gpurun_out/r05_c2_high_mfma32.err
  python - <<'PY'
import json
line = [l for l in open("gpurun_out/r05_c2_high_mfma32.out") if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print(round(d.get("value", 0), 2), d.get("unit"), d.get("config", {}).get("segment_ms"))
PY
  echo "== K slices on the last partial wave of short-K GEMMs (48 iterations): unsplit / launcher's choice / forced 2, 3, 4, 6 slices"
  for shape in "8300 12288 3072" "8812 12288 3072" "8300 9216 3072" "8300 3072 3072"; do
    timeout 300 python tools/bench_kernels.py gemmn $shape gemm $shape gemmfs2 $shape gemmfs3 $shape gemmfs4 $shape gemmfs6 $shape 2>&1 | grep "^gemm"
  done
  echo "== two ranks on ONE device (gloo, host-staged): FLUX.1-Kontext weight set seeded on rank 0 and broadcast in 1 GiB buckets; inpaint stage, one step"
  MTX_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --stages inpaint --inpaint-steps 1 --steps 1 --warmup 0 --no-cpu-baseline --no-traffic > gpurun_out/r05_two_ranks_kontext.out 2> gpurun_out/r05_two_ranks_kontext.err
  echo "rc $?"
  python - <<'PY'
import json
line = [l for l in open("gpurun_out/r05_two_ranks_kontext.out") if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print("n_gpus", d.get("n_gpus"), "value", d.get("value"), d.get("config", {}).get("launch"), d.get("config", {}).get("host_placement"))
PY
  tail -3 gpurun_out/r05_two_ranks_kontext.err
} > gpurun_out/r05_visit_f.log 2>&1
cat gpurun_out/r05_visit_f.log
