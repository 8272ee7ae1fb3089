"""Loader of libmtx_hip.so — the only gateway from Python to the HIP kernels.

There is NO fallback: if the library is missing or the device is not gfx950 this raises
ModelError (the reference's model-failure exception, utils/exceptions.py), exactly like a
failed model load in core/ml/model_manager.py — it never silently computes on the CPU.
"""
import ctypes as C
import os
import threading
from pathlib import Path

from ..utils.exceptions import ModelError
from . import abi

_CSRC = Path(__file__).resolve().parent.parent / "csrc"
_DEFAULT_SO = _CSRC / "libmtx_hip.so"


class MtxLibrary:
    """ctypes handle with typed prototypes; `check()` turns status codes into ModelError."""

    def __init__(self, path: Path, is_simulator: bool = False):
        if not Path(path).exists():
            raise ModelError(
                f"HIP kernel library not found: {path}. Build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)."
            )
        try:
            self._dll = C.CDLL(str(path))
        except OSError as e:  # missing ROCm runtime etc.
            raise ModelError(f"cannot load {path}: {e}") from e
        self.path = Path(path)
        self.is_simulator = is_simulator
        d = self._dll
        missing = [s for s in abi.EXPORTS if not hasattr(d, s)]
        if missing:
            raise ModelError(f"{path} does not export: {missing}")
        d.mtx_last_error.restype = C.c_char_p
        d.mtx_abi_sizeof.restype = C.c_size_t
        d.mtx_abi_sizeof.argtypes = [C.c_int]
        d.mtx_plan_destroy.restype = None
        d.mtx_plan_destroy.argtypes = [C.c_void_p]
        for name, t in (("mtx_conv2d", abi.ConvArgs), ("mtx_gemm", abi.GemmArgs),
                        ("mtx_attention", abi.AttnArgs), ("mtx_norm", abi.NormArgs),
                        ("mtx_groupnorm", abi.GroupNormArgs), ("mtx_elementwise", abi.EwArgs),
                        ("mtx_channel_attention", abi.CaArgs), ("mtx_image_convert", abi.ImgArgs),
                        ("mtx_resize_threshold", abi.ResizeThreshArgs),
                        ("mtx_mask_select", abi.MaskSelectArgs), ("mtx_preprocess", abi.PreprocArgs),
                        ("mtx_yolo_decode", abi.YoloDecodeArgs), ("mtx_bubble_clean", abi.CleanArgs), ("mtx_detr", abi.DetrArgs),
                        ("mtx_quantize_mx", abi.QuantArgs), ("mtx_page_tail", abi.TailArgs)):
            getattr(d, name).argtypes = [C.POINTER(t), C.c_void_p]
        d.mtx_host_text_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_double, C.c_void_p, C.POINTER(C.c_int)]
        d.mtx_host_chamfer_l2_5x5.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        d.mtx_host_mask_outline.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        d.mtx_host_png_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        d.mtx_host_png_encode.restype = C.c_int64
        d.mtx_conv2d_tiles.argtypes = [C.POINTER(abi.ConvArgs)]
        d.mtx_gemm_last_split.argtypes = [C.POINTER(C.c_int)] * 3
        d.mtx_plan_create.argtypes = [C.POINTER(abi.Op), C.c_int, C.POINTER(C.c_void_p)]
        d.mtx_plan_run.argtypes = [C.c_void_p, C.c_void_p]
        d.mtx_plan_run_graph.argtypes = [C.c_void_p, C.c_void_p]
        d.mtx_plan_num_ops.argtypes = [C.c_void_p]
        d.mtx_plan_run_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        d.mtx_plan_time.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        d.mtx_plan_time_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        d.mtx_plan_time_ops.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float)]
        d.mtx_device_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
        if d.mtx_abi_version() != abi.ABI_VERSION:
            raise ModelError(f"{path}: ABI version {d.mtx_abi_version()} != {abi.ABI_VERSION}")
        if d.mtx_abi_sizeof(0) != C.sizeof(abi.Op):
            raise ModelError("mtx_op layout mismatch between include/mtx_hip.h and hip/abi.py")
        if d.mtx_abi_sizeof(abi.CLEAN_ARGS_KIND) != C.sizeof(abi.CleanArgs):
            raise ModelError("mtx_clean_args layout mismatch between include/mtx_hip.h and hip/abi.py")
        for kind, t in abi.ARG_TYPES.items():
            if d.mtx_abi_sizeof(kind) != C.sizeof(t):
                raise ModelError(f"ABI struct mismatch for op kind {kind}: "
                                 f"{d.mtx_abi_sizeof(kind)} != {C.sizeof(t)}")

    def __getattr__(self, name):
        return getattr(self._dll, name)

    def last_error(self) -> str:
        msg = self._dll.mtx_last_error()
        return msg.decode("utf-8", "replace") if msg else ""

    def check(self, rc: int, what: str = "") -> None:
        if rc != 0:
            raise ModelError(f"{what or 'libmtx_hip'} failed ({rc}): {self.last_error()}")

    def gemm_last_split(self):
        """(whole tiles, K slices, tail pieces) of this thread's last 256-tile GEMM launch (mtx_gemm_last_split)"""
        v = [C.c_int(0) for _ in range(3)]
        self._dll.mtx_gemm_last_split(*[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def init(self, device_ordinal: int = 0) -> None:
        self.check(self._dll.mtx_init(int(device_ordinal)), "mtx_init")


_lock = threading.Lock()
_lib = None


def get_library() -> MtxLibrary:
    """Process-wide product library (gfx950).  Raises ModelError when it cannot be used."""
    global _lib
    with _lock:
        if _lib is None:
            _lib = MtxLibrary(Path(os.environ.get("MTX_HIP_LIBRARY", _DEFAULT_SO)))
        return _lib


def _open_simulator_for_tests(path) -> MtxLibrary:
    """TEST-ONLY: open tests/emu/libmtx_emu.so (CPU SIMT simulator of the same kernel sources).
    Never called from product code; tests pass the returned handle explicitly."""
    return MtxLibrary(Path(path), is_simulator=True)
