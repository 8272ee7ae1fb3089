#!/bin/bash
# round 5, visit r: what the late kernel changes are worth on a page, same box, same process settings: config 4 and config 5 with the round-5a
# kernels (serial GEMM epilogue, fp32-register norm kernel: MTX_GEMM_SERIAL_EPILOGUE=1 MTX_NORM_FORM=0) against the defaults
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[2], round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page |", {k: round(v["frac"], 3) for k, v in d.items() if k.startswith("roofline")}, flush=True)
PY
}
{
  for rep in 1 2; do
    MTX_GEMM_SERIAL_EPILOGUE=1 MTX_NORM_FORM=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-traffic --no-cpu-baseline > gpurun_out/r05_ab_c4_old_$rep.json 2>/dev/null
    line gpurun_out/r05_ab_c4_old_$rep.json "config 4, round-5a kernels  :"
    timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-traffic --no-cpu-baseline > gpurun_out/r05_ab_c4_new_$rep.json 2>/dev/null
    line gpurun_out/r05_ab_c4_new_$rep.json "config 4, batched epilogue + packed norm:"
  done
  MTX_GEMM_SERIAL_EPILOGUE=1 MTX_NORM_FORM=0 timeout 300 python bench.py --config 5 --steps 6 --warmup 2 --no-extra --no-traffic --no-cpu-baseline > gpurun_out/r05_ab_c5_old.json 2>/dev/null
  line gpurun_out/r05_ab_c5_old.json "config 5, round-5a kernels  :"
  timeout 300 python bench.py --config 5 --steps 6 --warmup 2 --no-extra --no-traffic --no-cpu-baseline > gpurun_out/r05_ab_c5_new.json 2>/dev/null
  line gpurun_out/r05_ab_c5_new.json "config 5, batched epilogue + packed norm:"
} > gpurun_out/r05_visit_r.log 2>&1
cat gpurun_out/r05_visit_r.log
