#!/bin/bash
# round 4, visit F: the two epilogue fusions in isolation, same process, fused vs separate launches (visit A's box: DiT step 52.3 -> 48.9 ms with
# them; visit E's box: 51.0 -> 52.3) — which kernel moves, and by how much on THIS box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ B="python tools/bench_kernels.py"
  for rep in 1 2; do
  timeout 300 $B attn 8512 attnq 8512 attnqs 8512 attn 8512 attnq 8512 attnqs 8512 2>&1 | grep -E "attn"
  timeout 300 $B glu 8512 9216 3072 9216 glus 8512 9216 3072 9216 glu 8000 9216 3072 0 glus 8000 9216 3072 0 glu 8512 9216 3072 9216 glus 8512 9216 3072 9216 2>&1 | grep gemm8
  done
  echo "== config 5 DiT step, on / off / on / off"
  for v in "" "--no-glu-epilogue" "" "--no-glu-epilogue"; do
    timeout 300 python bench.py --config 5 --stages inpaint --steps 3 --warmup 1 --no-cpu-baseline --no-traffic $v 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['inpaint']; print('[$v]', 'dit step', round(c['dit_step_ms'],2), [(g['kernel'], g.get('m'), g.get('n'), g.get('k'), round(g.get('ms', 0) or 0, 4)) for g in c.get('mfma_launch_groups', [])][:8])"
  done
} > gpurun_out/r04_f.log 2>&1
cat gpurun_out/r04_f.log
