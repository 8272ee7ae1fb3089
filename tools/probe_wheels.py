"""Import / reachability probe for the wheels whose arithmetic the unpinned oracles restate (VERDICT r03 missing #1).
Run anywhere (`python tools/probe_wheels.py [out.json]`); prints one JSON object: which wheels import (with version), whether pip can
see an index or a local wheelhouse, and where site-packages live.  Nothing here touches /root/reference."""
import importlib
import json
import os
import socket
import subprocess
import sys

WHEELS = ["cv2", "ultralytics", "spandrel", "diffusers", "torchvision", "oxipng", "sdnq", "PIL", "scipy", "transformers", "safetensors", "numpy", "torch"]


def main():
    out = {"python": sys.version.split()[0], "host": socket.gethostname(), "wheels": {}}
    for name in WHEELS:
        try:
            m = importlib.import_module(name)
            out["wheels"][name] = {"ok": True, "version": str(getattr(m, "__version__", "?"))}
        except Exception as e:  # noqa: BLE001
            out["wheels"][name] = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:80]}"}
    # is any package index or wheelhouse reachable?
    probes = {}
    for host in ("pypi.org", "files.pythonhosted.org", "download.pytorch.org"):
        try:
            socket.setdefaulttimeout(3)
            socket.create_connection((host, 443), timeout=3).close()
            probes[host] = "reachable"
        except Exception as e:  # noqa: BLE001
            probes[host] = f"{type(e).__name__}: {str(e)[:60]}"
    out["network"] = probes
    try:
        r = subprocess.run([sys.executable, "-m", "pip", "download", "--no-deps", "-d", "/tmp/_probe_dl", "opencv-python-headless"],
                           capture_output=True, text=True, timeout=40)
        out["pip_download_opencv"] = {"rc": r.returncode, "tail": (r.stdout + r.stderr)[-300:]}
    except Exception as e:  # noqa: BLE001
        out["pip_download_opencv"] = {"rc": None, "tail": f"{type(e).__name__}: {e}"}
    try:
        r = subprocess.run([sys.executable, "-m", "pip", "config", "list"], capture_output=True, text=True, timeout=20)
        out["pip_config"] = r.stdout.strip()[:400]
    except Exception as e:  # noqa: BLE001
        out["pip_config"] = str(e)
    houses = []
    for root in ("/opt", "/root", "/usr/share", "/tmp", "/wheels", "/wheelhouse", "/mnt"):
        if not os.path.isdir(root):
            continue
        try:
            r = subprocess.run(["find", root, "-maxdepth", "4", "-name", "*.whl", "-not", "-path", "*/site-packages/*"],
                               capture_output=True, text=True, timeout=30)
            houses += [ln for ln in r.stdout.splitlines() if ln][:200]
        except Exception:  # noqa: BLE001
            pass
    out["wheel_files_found"] = len(houses)
    out["wheel_files_matching"] = [h for h in houses if any(k in h.lower() for k in ("opencv", "ultralytics", "spandrel", "diffusers", "torchvision", "oxipng", "sdnq"))]
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
