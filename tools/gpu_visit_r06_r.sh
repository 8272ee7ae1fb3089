#!/bin/bash
# round 6, visit R: parity of the batch runs (SAM "high" instances staged), then config 2 with 2 / 3 / 4 front halves sharing detector batches against the
# round-5 arrangement, three rounds alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity"
  timeout 1200 python -m pytest tests/test_yolo11_gpu.py tests/test_page_vision_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error" | grep -v "SAM 2.1" | tail -6
  echo "== config 2"
  for r in 1 2 3; do
    for v in "2 1" "2 2" "3 3" "4 4"; do
      set -- $v
      timeout 600 python bench.py --config 2 --steps 64 --warmup 8 --no-cpu-baseline --no-traffic --no-extra --front-replicas $1 --detector-batch $2 > gpurun_out/cb.out 2> gpurun_out/cb.err
      python - "$1" "$2" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/cb.out") if x.startswith("{")]
if not l:
    print("front", sys.argv[1], "batch", sys.argv[2], "no line", open("gpurun_out/cb.err").read()[-800:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
print(f"front halves {sys.argv[1]}, detector batch {sys.argv[2]}: {d['value']:.2f} pages/s {d['ms_per_step']:.2f} ms/page |", c.get("detector_batch"))
PY
    done
  done
} > gpurun_out/r06_visit_r.log 2>&1
cat gpurun_out/r06_visit_r.log
