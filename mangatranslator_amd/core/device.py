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


# ---- host placement for N ranks on one node (no counterpart in the reference, which runs one process) ------------------------------------
def _parse_cpulist(text: str) -> list:
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_pci_addresses(sys_root: str = "/sys") -> list:
    """PCI addresses of the AMD GPUs / accelerators of this host in bus order — the order HIP numbers them in when HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES do not remap (display controller 0x03xx or processing accelerator 0x12xx, vendor 0x1002)"""
    from pathlib import Path
    out = []
    base = Path(sys_root) / "bus" / "pci" / "devices"
    if not base.is_dir():
        return out
    for d in sorted(base.iterdir()):
        try:
            if (d / "vendor").read_text().strip().lower() != "0x1002":
                continue
            cls = (d / "class").read_text().strip().lower()
            if cls.startswith("0x03") or cls.startswith("0x12"):
                out.append(d.name)
        except OSError:
            continue
    return out


def gpu_local_cpus(device_index: int, n_devices: Optional[int] = None, sys_root: str = "/sys") -> Optional[list]:
    """The CPUs this rank should run its host threads on: those of the NUMA node its GPU hangs off (`local_cpulist` of the GPU's PCI
    function), intersected with the CPUs the process may use, and — when several of the node's GPUs share that NUMA node — this GPU's
    equal share of them (GPUs in bus order).  None when the topology cannot be read (containers without /sys PCI entries) or says nothing
    (one NUMA node, numa_node = -1): the caller then leaves the affinity alone."""
    from pathlib import Path
    addrs = gpu_pci_addresses(sys_root)
    visible = None
    if torch.cuda.is_available() and sys_root == "/sys":
        # the runtime's own answer: the PCI address of EVERY visible device, in the runtime's order — HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES may
        # select any subset (GPUs 4-7 of 8): the share among the GPUs of one NUMA node is computed over the devices this process set can see
        try:
            vis = []
            for i in range(torch.cuda.device_count()):
                pr = torch.cuda.get_device_properties(i)
                if not (hasattr(pr, "pci_bus_id") and hasattr(pr, "pci_device_id")):
                    vis = None
                    break
                vis.append(f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
            if vis and all(a in addrs for a in vis):
                visible = vis
        except Exception:      # noqa: BLE001
            visible = None
    if visible is not None:
        addrs = visible                                  # index = the runtime's device index; nothing to truncate
    elif n_devices is not None and len(addrs) > n_devices:
        addrs = addrs[:n_devices]                        # sysfs-order fallback (no PCI ids from the runtime): the first n in bus order
    if not (0 <= device_index < len(addrs)):
        return None
    base = Path(sys_root) / "bus" / "pci" / "devices"
    try:
        lists = [tuple(_parse_cpulist((base / a / "local_cpulist").read_text())) for a in addrs]
    except (OSError, ValueError):
        return None
    mine = lists[device_index]
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cpus = [c for c in mine if c in allowed]
    if not cpus or (len(addrs) <= 1 and len(cpus) == len(allowed)):      # nothing usable, or one GPU whose node is the whole mask
        return None
    sharers = [i for i, l in enumerate(lists) if l == mine]
    k, n = sharers.index(device_index), len(sharers)
    share = cpus[k * len(cpus) // n:(k + 1) * len(cpus) // n]
    return share or None


def pin_host_threads_to_gpu(device_index: int, n_devices: Optional[int] = None, sys_root: str = "/sys") -> dict:
    """`os.sched_setaffinity` of the calling thread to `gpu_local_cpus` — threads started AFTERWARDS inherit it, thread pools that already exist
    (torch / OpenMP workers) stay where they were: call this before the first torch operation of the rank (bench.py and
    integration.pin_rank_to_gpu_cpus() do).  What it is for: a page's host work — NMS, mask
    logic, PNG codecs, the launches themselves — then runs on the socket its GPU is attached to instead of wherever the scheduler put the
    rank.  At configs 1 / 2 speeds (tens of pages per second and GPU) eight ranks are host-bound.  -> what was done."""
    cpus = gpu_local_cpus(device_index, n_devices, sys_root)
    if not cpus:
        return {"pinned": False, "reason": "no usable PCI / NUMA topology for this GPU"}
    try:
        os.sched_setaffinity(0, cpus)
    except (AttributeError, OSError) as e:
        return {"pinned": False, "reason": str(e)}
    return {"pinned": True, "cpus": len(cpus), "first": cpus[0], "last": cpus[-1]}
