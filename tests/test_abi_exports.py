"""CPU tier: the product library `mangatranslator_amd/csrc/libmtx_hip.so` (built by `__graft_entry__.build()`, hipcc cross-compile)
loads without a GPU and exports every entry point `include/mtx_hip.h` declares; the ctypes mirror (`hip/abi.py`) lists the same
symbols and its structs have the sizes the library reports.  No compute call is made."""
import ctypes as C
import re
from pathlib import Path

import pytest

from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.lib import MtxLibrary

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "mtx_hip.h"
PRODUCT = ROOT / "mangatranslator_amd" / "csrc" / "libmtx_hip.so"
SIMULATOR = ROOT / "tests" / "emu" / "libmtx_emu.so"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"MTX_API\s+[\w\s\*]+?\b(mtx_\w+)\s*\(", text)))


def test_header_and_ctypes_mirror_list_the_same_entry_points():
    assert declared_symbols() == sorted(abi.EXPORTS)
    assert len(abi.EXPORTS) >= 30


@pytest.mark.parametrize("path,sim", [(PRODUCT, False), (SIMULATOR, True)])
def test_library_loads_and_exports_every_declared_symbol(path, sim):
    if not path.exists():
        pytest.skip(f"{path.name} not built")
    dll = C.CDLL(str(path))
    missing = [s for s in declared_symbols() if not hasattr(dll, s)]
    assert not missing, f"{path.name} does not export {missing}"
    assert dll.mtx_abi_version() == abi.ABI_VERSION
    lib = MtxLibrary(path, is_simulator=sim)                  # typed prototypes + struct-size handshake (mtx_abi_sizeof), no device call
    lib.mtx_abi_sizeof.restype = C.c_size_t
    assert lib.mtx_abi_sizeof(0) == C.sizeof(abi.Op)
    for kind, t in abi.ARG_TYPES.items():
        assert lib.mtx_abi_sizeof(kind) == C.sizeof(t), kind
    assert lib.mtx_abi_sizeof(12345) == 0


def test_missing_library_fails_loudly(tmp_path):
    from mangatranslator_amd.utils.exceptions import ModelError
    with pytest.raises(ModelError):
        MtxLibrary(tmp_path / "libmtx_hip.so")
