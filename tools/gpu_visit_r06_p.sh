#!/bin/bash
# round 6, visit P: cross-page batches through the 640-px detector graphs (core/ml/detector_batch.py): GPU parity (bytes of the one-page call), then
# config 2 and config 1 with the batching wrapper off / on at 2, 3, 4 front halves, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity"
  timeout 900 python -m pytest tests/test_yolo11_gpu.py -q -x -p no:cacheprovider -k "batch" 2>&1 | tail -4
  for c in 2 1; do
    echo "== config $c"
    for v in "2 1" "2 2" "3 1" "3 3" "4 1" "4 4" "4 2" "2 1" "4 4"; do
      set -- $v
      timeout 600 python bench.py --config $c --steps 64 --warmup 8 --no-cpu-baseline --no-traffic --no-extra --front-replicas $1 --detector-batch $2 > gpurun_out/cb.out 2> gpurun_out/cb.err
      python - "$1" "$2" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/cb.out") if x.startswith("{")]
if not l:
    print("front", sys.argv[1], "batch", sys.argv[2], "no line", open("gpurun_out/cb.err").read()[-800:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
print(f"front halves {sys.argv[1]}, detector batch {sys.argv[2]}: {d['value']:.2f} pages/s {d['ms_per_step']:.2f} ms/page | aux ms", {k: round(v, 2) for k, v in c.get("detect_aux_ms", {}).items()},
      "|", c.get("detector_batch"), "| stages", c.get("stage_wall_ms_one_page"))
PY
    done
  done
} > gpurun_out/r06_visit_p.log 2>&1
cat gpurun_out/r06_visit_p.log
