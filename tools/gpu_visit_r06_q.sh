#!/bin/bash
# round 6, visit Q: detector batches after the capture fix: parity tests (op level + the product's batch run), config 2 / 1 at their new defaults against the
# round-5 arrangement (2 front halves, a detector instance each), alternating; config 2 with four front halves
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity"
  timeout 900 python -m pytest tests/test_yolo11_gpu.py tests/test_page_vision_gpu.py -q -x -s -p no:cacheprovider -k "batch" 2>&1 | grep -E "passed|failed|Error|error|graph replays" | grep -v "SAM 2.1" | tail -8
  for c in 2 1; do
    echo "== config $c"
    for v in "2 1" "0 0" "4 4" "2 1" "0 0" "3 3"; do
      set -- $v
      fr=""; [ "$1" != "0" ] && fr="--front-replicas $1"
      timeout 600 python bench.py --config $c --steps 64 --warmup 8 --no-cpu-baseline --no-traffic --no-extra $fr --detector-batch $2 > gpurun_out/cb.out 2> gpurun_out/cb.err
      python - "$1" "$2" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/cb.out") if x.startswith("{")]
if not l:
    print("front", sys.argv[1], "batch", sys.argv[2], "no line", open("gpurun_out/cb.err").read()[-800:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
print(f"front halves {sys.argv[1] if sys.argv[1] != '0' else 'default'} -> {c.get('front_replicas')}, detector batch {sys.argv[2] if sys.argv[2] != '0' else 'default'}: {d['value']:.2f} pages/s {d['ms_per_step']:.2f} ms/page | aux ms",
      {k: round(v, 2) for k, v in c.get("detect_aux_ms", {}).items()}, "|", c.get("detector_batch"), "| stages", c.get("stage_wall_ms_one_page"))
PY
    done
  done
} > gpurun_out/r06_visit_q.log 2>&1
cat gpurun_out/r06_visit_q.log
