#!/bin/bash
# round 3, GPU visit G: norms and SwiGLU write the fp8 linears' operands themselves (config 5): parity, then same-library A/B
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== parity"; timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_flux2_gpu.py -q -x -p no:cacheprovider -k "fp8 or twins or klein or quantize or flux2 or norm" 2>&1 | tail -4
for flag in "" "--no-fused-quant" "" "--no-fused-quant"; do
  echo "== config 5 $flag"; timeout 900 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline $flag 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],4),'pages/s', round(d['ms_per_step'],1),'ms/page; DiT step', round(c['inpaint']['dit_step_ms'],2),'ms; roofline', round(d['roofline']['frac'],3))"
done
} > gpurun_out/r03_g.log 2>&1
cat gpurun_out/r03_g.log
