import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """CPU SIMT simulator build of the kernel sources (tests/emu) — test infrastructure only."""
    from mangatranslator_amd.hip.lib import _open_simulator_for_tests
    so = ROOT / "tests" / "emu" / "libmtx_emu.so"
    csrc = ROOT / "mangatranslator_amd" / "csrc"
    r = subprocess.run(["make", "-s", "-j8", "emu"], cwd=csrc, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("building the kernel simulator failed:\n" + r.stdout + r.stderr)
    return _open_simulator_for_tests(so)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library on a real GPU."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mangatranslator_amd.hip.lib import get_library
    lib = get_library()
    lib.init(0)
    return lib


@pytest.fixture(autouse=True)
def fresh_stage_memo():
    """every test starts with an empty stage memo (core/caching.py is process-global, like the reference's)"""
    from mangatranslator_amd.core import caching
    caching.get_cache().reset()
    yield
