"""De-quantising reader for SDNQ-packed checkpoints — the storage format of the FLUX repositories the reference downloads
(`Disty0/FLUX.1-Kontext-dev-SDNQ-uint4-svd-r32`, `Disty0/FLUX.2-klein-4B-SDNQ-4bit-dynamic`, `...-9B-SDNQ-4bit-dynamic-svd-r32`;
reference core/ml/model_manager.py:231-240, loaded through diffusers + the `sdnq` package at :1176-1337).

This package serves bf16 / MX-fp8 weights resident in HBM, so the packed weights are expanded ONCE at load:
    W[n, k] = zero_point[n, g] + q[n, k] * scale[n, g]   (asymmetric, unsigned storage)        g = k // group_size
    W[n, k] = q[n, k] * scale[n, g]                       (symmetric: stored value minus 2^(bits-1))
    W      += svd_up[n, r] @ svd_down[r, k]               (when the low-rank correction is stored: "-svd-r32")
and handed on as bf16 (the fp8 path then quantises them to MX e4m3 like any other bf16 checkpoint).

FORMAT PROVENANCE — read this before trusting it: neither the `sdnq` wheel nor any SDNQ checkpoint is reachable from the build
image.  The layout below is restated from sdnq 0.1.x's published source (`sdnq/packed_int.py`, `sdnq/dequantizer.py`,
`sdnq/quantizer.py`): per quantised linear the parameters `weight` (uint8 storage; sub-byte types packed little-nibble-first along
the flattened tensor), `scale`, optional `zero_point`, optional `svd_up` / `svd_down`; the weights' dtype per module from
`quantization_config` in config.json (`weights_dtype`, overridden per module by `modules_dtype_dict`).  Everything here is driven by
the SHAPES found in the file (packed element count vs the logical shape gives the bit width, the scale's shape gives the grouping),
so a mismatch with the real format fails loudly (`ModelError`) instead of producing wrong weights silently; the arithmetic is
covered by a round-trip test against a writer that follows the same published layout (tests/test_checkpoint_readiness.py) — that test cannot
vouch for the provenance, only for self-consistency.  tools/pin_oracles.py --models is where a real checkpoint gets compared.
"""
import json
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch

from ...utils.exceptions import ModelError

SUFFIXES = (".scale", ".zero_point", ".svd_up", ".svd_down")


def unpack_bits(packed: torch.Tensor, bits: int, count: int) -> torch.Tensor:
    """uint8 storage -> `count` unsigned values of `bits` bits each (8 / bits per byte, lowest bits first)"""
    p = packed.reshape(-1).to(torch.uint8)
    if bits == 8:
        out = p
    elif bits in (4, 2, 1):
        per = 8 // bits
        shifts = torch.arange(per, dtype=torch.uint8) * bits
        out = ((p[:, None] >> shifts[None, :]) & ((1 << bits) - 1)).reshape(-1)
    else:
        raise ModelError(f"SDNQ: {bits}-bit packing is not supported (uint8 / 4 / 2 / 1)")
    need = (count * bits + 7) // 8
    if p.numel() != need:
        # over-supply is as wrong as under-supply: a 4-bit reading of an 8-bit tensor would otherwise take its first nibbles silently
        raise ModelError(f"SDNQ: {p.numel()} packed bytes for {count} values of {bits} bits ({need} bytes expected) — the configured bit "
                         f"width does not match the stored tensor")
    return out[:count]


def pack_bits(values: torch.Tensor, bits: int) -> torch.Tensor:
    """inverse of `unpack_bits` (used by the tests' writer)"""
    v = values.reshape(-1).to(torch.uint8)
    if bits == 8:
        return v.clone()
    per = 8 // bits
    pad = (-v.numel()) % per
    if pad:
        v = torch.cat([v, torch.zeros(pad, dtype=torch.uint8)])
    v = v.reshape(-1, per)
    out = torch.zeros(v.shape[0], dtype=torch.uint8)
    for i in range(per):
        out |= v[:, i] << (i * bits)
    return out


def dequantize(weight: torch.Tensor, scale: torch.Tensor, shape: Tuple[int, ...], zero_point: Optional[torch.Tensor] = None,
               svd_up: Optional[torch.Tensor] = None, svd_down: Optional[torch.Tensor] = None, bits: Optional[int] = None) -> torch.Tensor:
    """packed `weight` -> fp32 tensor of the logical `shape` ([N, K] or conv [N, C, kh, kw], grouped along the flattened K axis)"""
    n = int(shape[0])
    k = 1
    for d in shape[1:]:
        k *= int(d)
    count = n * k
    if bits is None:
        if weight.dtype in (torch.int8, torch.uint8) and weight.numel() * 8 % count == 0 and weight.numel() * 8 // count in (1, 2, 4, 8):
            bits = weight.numel() * 8 // count
        else:
            raise ModelError(f"SDNQ: cannot tell the bit width of a {tuple(weight.shape)} {weight.dtype} tensor for logical shape {tuple(shape)}")
    elif weight.dtype in (torch.int8, torch.uint8) and weight.numel() != (count * bits + 7) // 8:
        raise ModelError(f"SDNQ: config says {bits}-bit weights but a {tuple(weight.shape)} {weight.dtype} tensor for logical shape "
                         f"{tuple(shape)} holds {weight.numel() * 8 / count:g} bits per value")
    q = unpack_bits(weight.view(torch.uint8) if weight.dtype == torch.int8 else weight, bits, count).to(torch.float32)
    if zero_point is None:
        q = q - float(1 << (bits - 1))                               # symmetric types are stored offset by half their range
    s = scale.to(torch.float32).reshape(n, -1)
    groups = s.shape[1]
    if k % groups:
        raise ModelError(f"SDNQ: {groups} scale groups do not divide K = {k}")
    q = q.reshape(n, groups, k // groups)
    w = q * s[:, :, None]
    if zero_point is not None:
        w = w + zero_point.to(torch.float32).reshape(n, groups)[:, :, None]
    w = w.reshape(n, k)
    if (svd_up is None) != (svd_down is None):
        raise ModelError("SDNQ: svd_up and svd_down come as a pair")
    if svd_up is not None:
        up, down = svd_up.to(torch.float32), svd_down.to(torch.float32)
        if up.shape[0] != n or down.shape[-1] != k or up.shape[1] != down.shape[0]:
            raise ModelError(f"SDNQ: low-rank factors {tuple(up.shape)} x {tuple(down.shape)} do not fit a {n} x {k} weight")
        w = w + up @ down
    return w.reshape(tuple(shape))


class SdnqReader:
    """name -> fp32 / bf16 tensor over the safetensors shards of one diffusers sub-folder, expanding SDNQ-packed parameters.
    A parameter counts as packed when the file holds `<base>.scale` beside `<base>.weight`; everything else is passed through."""

    def __init__(self, folder: Path):
        from safetensors import safe_open
        self.folder = Path(folder)
        files = sorted(self.folder.glob("*.safetensors"))
        if not files:
            raise ModelError(f"no safetensors shards under {folder}")
        self.handles = [safe_open(str(f), framework="pt", device="cpu") for f in files]
        self.where = {k: h for h in self.handles for k in h.keys()}
        self.qcfg = {}
        cfg = self.folder / "config.json"
        if cfg.exists():
            self.qcfg = json.loads(cfg.read_text()).get("quantization_config") or {}

    def is_packed(self, name: str) -> bool:
        return name.endswith(".weight") and name[:-len(".weight")] + ".scale" in self.where

    def keys(self):
        return [k for k in self.where if not k.endswith(SUFFIXES)]

    def _bits_from_config(self, base: str) -> Optional[int]:
        dt = (self.qcfg.get("modules_dtype_dict") or {})
        name = None
        for dtype_name, modules in dt.items():
            if any(base == m or base.endswith("." + m) for m in modules):
                name = dtype_name
        name = name or self.qcfg.get("weights_dtype")
        if not name:
            return None
        digits = "".join(c for c in str(name) if c.isdigit())
        return int(digits) if digits else None

    def get(self, name: str, shape: Tuple[int, ...]) -> torch.Tensor:
        if name not in self.where:
            raise ModelError(f"{self.folder}: parameter {name} is missing")
        t = self.where[name].get_tensor(name)
        if not self.is_packed(name):
            if tuple(t.shape) != tuple(shape):
                raise ModelError(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
            return t
        base = name[:-len(".weight")]
        part = {s: (self.where[base + s].get_tensor(base + s) if base + s in self.where else None) for s in SUFFIXES}
        return dequantize(t, part[".scale"], shape, part[".zero_point"], part[".svd_up"], part[".svd_down"], bits=self._bits_from_config(base))


def dequantized_state_dict(folder: Path, shapes: Dict[str, Tuple[int, ...]], dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """every parameter named in `shapes`, expanded, as `dtype` (tools/export_prompt_embeds.py: SDNQ-packed text encoders)"""
    r = SdnqReader(folder)
    return {k: r.get(k, s).to(dtype) for k, s in shapes.items() if k in r.where}
