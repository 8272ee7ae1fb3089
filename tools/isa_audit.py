"""Static look at the gfx950 ISA of every kernel in csrc/ for memory requests that cannot overlap: a load followed by `s_waitcnt vmcnt(0)` before
the next load is issued is one dependent memory round trip of the wave (≈ 0.5 µs from the L2, 1–2 µs from HBM).  A kernel whose epilogue or
row loop is a chain of those is latency-bound however few bytes it moves — round 5 found the row-normalisation kernel (12 per row) and the
256-tile GEMM epilogue (≈ 40 per tile) that way (docs/experiments.md, round 5).  No GPU needed: hipcc cross-compiles.

    python tools/isa_audit.py [--min 6] [file.hip ...]        # default: every .hip under mangatranslator_amd/csrc

Columns: full-wait round trips in the kernel's code (static count: both sides of every branch are counted), loads, ISA lines, VGPRs, scratch
bytes per lane, kernel.  A static count is a lead, not a measurement — read the ISA around the waits before changing a kernel."""
import argparse
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "mangatranslator_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"


def compile_to_isa(src: Path, out_dir: Path):
    asm, res = out_dir / (src.stem + ".s"), out_dir / (src.stem + ".res")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT / 'include'}", "-S", "--cuda-device-only", str(src), "-o", str(asm),
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, cwd=src.parent, capture_output=True, text=True)
    res.write_text(r.stderr)
    if not asm.exists():
        raise SystemExit(f"{src.name}: hipcc failed\n{r.stderr[-2000:]}")
    return asm, res


def resources(res: Path):
    """kernel -> (VGPRs, scratch bytes per lane) from -Rpass-analysis=kernel-resource-usage"""
    out, cur = {}, None
    for line in res.read_text().splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); out[cur] = [None, None]; continue
        if cur:
            m = re.search(r" VGPRs: (\d+)", line)
            if m:
                out[cur][0] = int(m.group(1))
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m:
                out[cur][1] = int(m.group(1))
    return out


def audit(asm: Path):
    rows, cur, chains, loads, lines, pending = [], None, 0, 0, 0, False
    for line in asm.read_text().splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, chains, loads, lines, pending = m.group(1), 0, 0, 0, False
            continue
        if cur is None:
            continue
        lines += 1
        t = line.strip()
        if t.startswith(("global_load", "buffer_load")) and not t.endswith(" lds"):
            loads += 1; pending = True
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            chains += 1 if pending else 0
            pending = False
        elif t.startswith("s_endpgm"):
            rows.append((chains, loads, lines, cur)); cur = None
    return rows


def demangle(name: str) -> str:
    import shutil
    tool = shutil.which("c++filt")
    if not tool or "DF16" in name:          # binutils' demangler misreads the _Float16 / __bf16 template arguments: keep those mangled
        return name
    return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("sources", nargs="*", help="default: every .hip under mangatranslator_amd/csrc")
    ap.add_argument("--min", type=int, default=6, help="only kernels with at least this many full-wait round trips")
    a = ap.parse_args(argv)
    srcs = [Path(s).resolve() for s in a.sources] or sorted(CSRC.glob("*.hip"))
    with tempfile.TemporaryDirectory() as td:
        with ThreadPoolExecutor(max_workers=6) as pool:
            built = list(pool.map(lambda s: compile_to_isa(s, Path(td)), srcs))
        table = []
        for asm, res in built:
            rs = resources(res)
            for chains, loads, lines, k in audit(asm):
                v, sc = rs.get(k, (None, None))
                table.append((chains, loads, lines, v, sc, asm.stem, k))
    table.sort(key=lambda r: (-r[0], r[6]))
    print(f"{'waits':>5} {'loads':>5} {'lines':>6} {'VGPR':>4} {'scr':>4}  kernel")
    for chains, loads, lines, v, sc, stem, k in table:
        if chains >= a.min:
            demangled = demangle(k)
            print(f"{chains:5d} {loads:5d} {lines:6d} {v if v is not None else '-':>4} {sc if sc is not None else '-':>4}  {stem}: {demangled[:150]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
