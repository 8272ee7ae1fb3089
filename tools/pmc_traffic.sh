#!/bin/bash
# HBM-side traffic of the hot kernels: two counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; counters only,
# with --kernel-trace, never combined with sys/hip trace domains) over tools/bench_kernels.py, folded into gpurun_out/r02_pmc_traffic.json.
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte requests at 64 B -> doubled; WRITE_SIZE as reported.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${@:-conv 1536 1024 conv 3072 2048 attn 8812 gemm 8812 9216 3072 gemm 8812 3072 15360 gemm8 8512 27648 3072}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; mkdir -p /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o k -- python $R/tools/bench_kernels.py $ARGS > $R/gpurun_out/pmc_$c.log 2>&1)
  tail -2 gpurun_out/pmc_$c.log
done
python - <<'PY'
import csv, glob, collections, json
def per_kernel(counter):
    tot, n = collections.Counter(), collections.Counter()
    for fn in glob.glob(f"/tmp/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter:
                # one row per dispatch: key by kernel AND grid size (the same kernel runs several problem sizes here)
                k = (r["Kernel_Name"], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
                tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
f, nf = per_kernel("FETCH_SIZE"); w, nw = per_kernel("WRITE_SIZE")
out = []
for k in sorted(f, key=lambda k: -f[k]):
    if "mtx" not in k[0] or k not in w: continue
    rd = 2 * f[k] / nf[k] * 1024; wr = w[k] / nw[k] * 1024
    out.append({"kernel": k[0][:110], "grid": k[1], "launches": nf[k], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "bytes_per_launch": round(rd + wr)})
json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_kernels.py; FETCH_SIZE x 2 (gfx950), KB -> bytes",
           "kernels": out}, open("gpurun_out/r02_pmc_traffic.json", "w"), indent=1)
for o in out[:14]: print(o["kernel"][:60], o["grid"], o["launches"], f'{o["read_bytes_per_launch"]/1e6:.1f} MB read', f'{o["write_bytes_per_launch"]/1e6:.1f} MB written')
PY
