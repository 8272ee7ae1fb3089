#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for a in 0 1 2 3 4; do MTX_C64_ABL=$a python tools/probe_conv.py 2>/dev/null | tail -1; done | tee gpurun_out/conv_abl.log
cd /tmp
ITERS=5 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o c -- python $GRAFT_REPO_ROOT/tools/probe_conv.py > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1
ITERS=5 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o c -- python $GRAFT_REPO_ROOT/tools/probe_conv.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1
ITERS=5 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o c -- python $GRAFT_REPO_ROOT/tools/probe_conv.py > $GRAFT_REPO_ROOT/gpurun_out/pmc3.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc1 gpurun_out/pmc2; 
python - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3"):
    for f in glob.glob(f"gpurun_out/{d}/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "c64" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(d, k, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)))
PY
