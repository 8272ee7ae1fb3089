"""Fold two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_kernels.sh, counters only) into profiles/r01_pmc_traffic.json:
    python tools/summarize_pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>
gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE (KB) counts 128-B requests at 64 B -> doubled; WRITE_SIZE (KB) as reported."""
import collections
import csv
import json
import sys
from pathlib import Path


def per_kernel(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out_path = Path(__file__).resolve().parent.parent / "profiles" / "r01_pmc_traffic.json"
data = json.loads(out_path.read_text())


def find(part):
    return [k for k in fetch if part in k]


def entry(main_part, merge_part=None):
    k = find(main_part)[0]
    e = {"kernel": k, "launches": nf[k], "fetch_size_kb_sum": round(fetch[k]), "write_size_kb_sum": round(write[k])}
    b = (2 * fetch[k] / nf[k] + write[k] / nw[k]) * 1024
    if merge_part:
        m = find(merge_part)[0]
        e["merge_fetch_kb_sum"], e["merge_write_kb_sum"] = round(fetch[m]), round(write[m])
        b += (2 * fetch[m] / nf[m] + write[m] / nw[m]) * 1024
    e["bytes_per_launch"] = round(b)
    return e


for key in list(data):
    if key.startswith("attn_mma32_kernel"):
        new = entry("attn_mma32_kernel", "attn_merge_kernel")
        new["algorithmic_bytes_per_launch"] = data[key].get("algorithmic_bytes_per_launch")
        data[key] = new
    elif key.startswith("gemm256_kernel"):
        new = entry("gemm256_kernel")
        new["algorithmic_bytes_per_launch"] = data[key].get("algorithmic_bytes_per_launch")
        data[key] = new
out_path.write_text(json.dumps(data, indent=1))
for k, v in data.items():
    if isinstance(v, dict):
        print(k, v.get("bytes_per_launch"), v.get("algorithmic_bytes_per_launch"))
