"""Device / dtype selection (reference core/device.py:7-103).  One process drives one MI355X:
`cuda:<LOCAL_RANK>` and bf16; there is no CPU compute path behind it."""
import gc
import os
from typing import Optional

import torch


def get_best_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cpu")       # lets host-only code (geometry, tests) import; kernels refuse it


def get_best_dtype(device: Optional[torch.device] = None) -> torch.dtype:
    device = device if device is not None else get_best_device()
    return torch.bfloat16 if device.type == "cuda" else torch.float32


def empty_cache(device: Optional[torch.device] = None) -> None:
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def synchronize(device: Optional[torch.device] = None) -> None:
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def get_device_info(device: Optional[torch.device] = None) -> dict:
    """reference core/device.py:116-166: device name + allocated / reserved GB as two-decimal strings, or `memory: "N/A"` on the host"""
    device = device if device is not None else get_best_device()
    if device.type == "cuda" and torch.cuda.is_available():
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return {"device": torch.cuda.get_device_name(idx), "allocated_gb": f"{torch.cuda.memory_allocated(idx) / 1024 ** 3:.2f}",
                "reserved_gb": f"{torch.cuda.memory_reserved(idx) / 1024 ** 3:.2f}"}
    return {"device": "cpu", "memory": "N/A"}


def is_gpu_available() -> bool:
    """reference core/device.py:169-191 (one backend here: ROCm behind torch.cuda)"""
    return torch.cuda.is_available()
