#!/bin/bash
# round 5, visit q: the 256-tile GEMM epilogue with its bias / gate / residual requests in batches against one at a time (identical bytes?
# how much per FLUX shape?), then the Kontext page and the Klein page on it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "gemm" 2>&1 | tail -4
  timeout 400 python tools/bench_kernels.py gemmer 8812 3072 15360 gemmer 8300 3072 3072 gemmer 8300 3072 12288 gemmer 512 3072 3072 gemmeg 8300 12288 3072 gemmeg 8812 12288 3072 gemmeb 8812 9216 3072 gemmeb 8300 9216 3072 gemm8eb 8512 27648 3072 gemm8er 8512 3072 12288 gemm8eb 8512 9216 3072 2>&1 | grep "^gemm"
  timeout 500 python bench.py --steps 6 --warmup 2 --no-extra --no-traffic --no-cpu-baseline > gpurun_out/r05_bench_default_batched_epilogue.json 2> gpurun_out/r05_bench_q.err
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_default_batched_epilogue.json"))
print("config 4:", round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page |", {k: round(v["frac"], 3) for k, v in d.items() if k.startswith("roofline")})
PY
  timeout 400 python bench.py --config 5 --steps 6 --warmup 2 --no-extra --no-traffic --no-cpu-baseline > gpurun_out/r05_bench_config5_batched_epilogue.json 2>> gpurun_out/r05_bench_q.err
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_config5_batched_epilogue.json"))
print("config 5:", round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page |", {k: round(v["frac"], 3) for k, v in d.items() if k.startswith("roofline")})
PY
  tail -3 gpurun_out/r05_bench_q.err
} > gpurun_out/r05_visit_q.log 2>&1
cat gpurun_out/r05_visit_q.log
