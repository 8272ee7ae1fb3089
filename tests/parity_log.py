"""Every GPU parity check drops the figures it measured into gpurun_out/r06_parity.json (merged back from the GPU box by gpurun; the
round's copy is committed as profiles/r06_parity.json), so each number DESIGN.md quotes can be traced to a file."""
import json
import os
from pathlib import Path

_PATH = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out" / "r06_parity.json"


def record(name: str, **values):
    try:
        _PATH.parent.mkdir(parents=True, exist_ok=True)
        data = json.loads(_PATH.read_text()) if _PATH.exists() else {}
        data[name] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in values.items()}
        _PATH.write_text(json.dumps(data, indent=1, sort_keys=True))
    except Exception:
        pass
