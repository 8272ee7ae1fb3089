#!/bin/bash
# round 4, visit B: hardware tests of the fused epilogues (fixed reference) and of the K-slice tail; SAM bf16 / f16 probe; same-process A/B of
# the K-slice tail against round 3's stream-K tail + merge; a short default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ echo "== hardware tests"; timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_flux2_gpu.py -q -m gpu -p no:cacheprovider -k "glu or mx_fp8 or quantiser or k_slice or gemm" 2>&1 | tail -15
  echo "== GEMM tail A/B (same process per shape pair)"
  for rep in 1 2; do
  timeout 300 python tools/bench_kernels.py gemm 8812 3072 15360 gemmo 8812 3072 15360 gemmn 8812 3072 15360 gemm 8300 3072 12288 gemmo 8300 3072 12288 gemmn 8300 3072 12288 \
      gemm 512 3072 12288 gemmo 512 3072 12288 gemmn 512 3072 12288 gemm 8812 3072 3072 gemmn 8812 3072 3072 gemm8 8512 3072 12288 gemm8o 8512 3072 12288 gemm8n 8512 3072 12288 2>&1 | grep -v Warning
  done
  echo "== SAM dtype probe"; timeout 600 python tools/sam_dtype_probe.py gpurun_out/r04_sam_dtype_probe.json 2>&1 | tail -4
  echo "== default bench, short"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r04_b_bench_default_short.json
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_b_bench_default_short.json')); c=d['config']
print(round(d['value'],4), round(d['ms_per_step'],1), 'dit step', c['inpaint'].get('dit_step_ms'), d.get('roofline'))
PY
} > gpurun_out/r04_b.log 2>&1
cat gpurun_out/r04_b.log
