"""Pickle-free reader of ultralytics `.pt` detector checkpoints.

The reference loads its four YOLO detectors with `YOLO(str(path))` (core/ml/model_manager.py:711-743, :780-838), i.e. `torch.load` of a
pickle whose objects are instances of `ultralytics.nn.*` classes — unreadable without the ultralytics package, and arbitrary code for
anyone who can replace the file.  This build needs the TENSORS and the class names only, so the file is read with a restricted
unpickler that never imports anything:

  * the container is torch's zip format (`<name>/data.pkl`, `<name>/data/<key>` per storage);
  * tensors are rebuilt from the storages by hand (`torch._utils._rebuild_tensor_v2` / `_rebuild_parameter` semantics: dtype from the
    storage class, offset / size / stride as pickled);
  * `collections.OrderedDict`, `torch.Size`, dtypes and a handful of builtins are themselves;
  * EVERY other global (ultralytics modules, torch.nn modules, pathlib, numpy scalars of the training log, ...) becomes an inert shell
    object that only records the state pickle hands it — no constructor, `__setstate__` or `__reduce__` of the original class runs.

The module tree is then walked the way `nn.Module.state_dict()` names things (`_parameters`, persistent `_buffers`, `_modules`), which
gives exactly the keys `YOLO(path).model.state_dict()` has (`model.0.conv.weight`, ...), plus the checkpoint's `names`.  As ultralytics'
own loader does, the EMA weights are preferred when the checkpoint holds them.  Everything is returned as fp32 (checkpoints store
half); BatchNorm stays un-fused and is folded by the graph builders.
"""
import io
import pickle
import zipfile
from collections import OrderedDict
from pathlib import Path
from typing import Dict, Tuple

import numpy as np
import torch

from ...utils.exceptions import ModelError

_STORAGE_DTYPES = {
    "FloatStorage": (torch.float32, np.float32), "HalfStorage": (torch.float16, np.float16), "DoubleStorage": (torch.float64, np.float64),
    "BFloat16Storage": (torch.bfloat16, np.uint16), "LongStorage": (torch.int64, np.int64), "IntStorage": (torch.int32, np.int32),
    "ShortStorage": (torch.int16, np.int16), "CharStorage": (torch.int8, np.int8), "ByteStorage": (torch.uint8, np.uint8),
    "BoolStorage": (torch.bool, np.bool_),
}
_SAFE_BUILTINS = {"set": set, "frozenset": frozenset, "dict": dict, "list": list, "tuple": tuple, "int": int, "float": float, "bool": bool,
                  "str": str, "bytes": bytes, "bytearray": bytearray, "complex": complex, "slice": slice, "range": range, "object": object}


class _StorageType:
    def __init__(self, name):
        self.torch_dtype, self.np_dtype = _STORAGE_DTYPES[name]


class _Storage:
    def __init__(self, stype: _StorageType, key: str, numel: int):
        self.stype, self.key, self.numel = stype, key, int(numel)


class Shell:
    """stands in for an instance of a class this reader does not know: keeps what pickle gives it, runs nothing"""
    _origin = ("?", "?")

    def __init__(self, *args, **kwargs):
        self.__dict__["_shell_args"] = args

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], (dict, type(None))):
            for part in state:                                   # (dict state, slots state)
                if isinstance(part, dict):
                    self.__dict__.update(part)
        else:
            self.__dict__["_shell_state"] = state

    # containers some pickled classes extend (list / dict subclasses use APPEND / SETITEM opcodes)
    def append(self, x):
        self.__dict__.setdefault("_shell_items", []).append(x)

    def extend(self, xs):
        self.__dict__.setdefault("_shell_items", []).extend(xs)

    def __setitem__(self, k, v):
        self.__dict__.setdefault("_shell_map", {})[k] = v

    def __repr__(self):
        return f"<Shell {'.'.join(self._origin)}>"


def _shell_class(module: str, name: str, _cache={}):
    key = (module, name)
    if key not in _cache:
        _cache[key] = type(name, (Shell,), {"_origin": key})
    return _cache[key]


class _Reader(pickle.Unpickler):
    def __init__(self, f, zf: zipfile.ZipFile, root: str):
        super().__init__(f)
        self.zf, self.root = zf, root
        self._bytes: Dict[str, bytes] = {}

    # ---- tensors ------------------------------------------------------------------------------------------------------------
    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise ModelError(f"ultralytics .pt: unexpected persistent id {pid!r:.80}")
        return _Storage(pid[1], str(pid[2]), pid[4])

    def _tensor(self, storage, offset, size, stride, *_ignored):
        if not isinstance(storage, _Storage):
            raise ModelError("ultralytics .pt: tensor without a storage record")
        raw = self._bytes.get(storage.key)
        if raw is None:
            try:
                raw = self._bytes[storage.key] = self.zf.read(f"{self.root}data/{storage.key}")
            except KeyError as e:
                raise ModelError(f"ultralytics .pt: storage {storage.key} missing from the archive") from e
        st = storage.stype
        flat = np.frombuffer(raw, dtype=st.np_dtype)
        if flat.size < storage.numel:
            raise ModelError(f"ultralytics .pt: storage {storage.key} holds {flat.size} elements, {storage.numel} recorded")
        t = torch.from_numpy(flat.copy())
        if st.torch_dtype == torch.bfloat16:
            t = t.view(torch.bfloat16)
        size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
        need = 1 + sum((n - 1) * s for n, s in zip(size, stride)) if all(n > 0 for n in size) else 0
        if int(offset) + need > t.numel():
            raise ModelError("ultralytics .pt: tensor view reaches past its storage")
        return torch.as_strided(t, size, stride, int(offset)).clone()

    # ---- globals --------------------------------------------------------------------------------------------------------------
    def find_class(self, module, name):
        if module == "torch._utils":
            if name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return self._tensor
            if name == "_rebuild_parameter":
                return lambda data, requires_grad=False, hooks=None: data
            if name == "_rebuild_parameter_with_state":
                return lambda data, requires_grad=False, hooks=None, state=None: data
        if module == "torch._tensor" and name == "_rebuild_from_type_v2":
            return lambda func, new_type, args, state: func(*args)
        if module in ("torch", "torch.storage") and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "torch" and name == "Size":
            return lambda *a: tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list)) else tuple(a)
        if module == "torch" and isinstance(getattr(torch, name, None), torch.dtype):
            return getattr(torch, name)
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module in ("builtins", "__builtin__") and name in _SAFE_BUILTINS:
            return _SAFE_BUILTINS[name]
        return _shell_class(module, name)                 # inert: ultralytics.*, torch.nn.*, pathlib.*, numpy.*, ...


def _module_tensors(mod, prefix: str, out: Dict[str, torch.Tensor], depth=0):
    """`nn.Module.state_dict()` naming over a tree of shells"""
    if depth > 64:
        raise ModelError("ultralytics .pt: module tree deeper than 64 levels")
    d = getattr(mod, "__dict__", {})
    for name, p in (d.get("_parameters") or {}).items():
        if isinstance(p, torch.Tensor):
            out[prefix + name] = p
    skip = d.get("_non_persistent_buffers_set") or ()
    for name, b in (d.get("_buffers") or {}).items():
        if isinstance(b, torch.Tensor) and name not in skip:
            out[prefix + name] = b
    for name, child in (d.get("_modules") or {}).items():
        if child is not None:
            _module_tensors(child, f"{prefix}{name}.", out, depth + 1)


def read_ultralytics_pt(path) -> Tuple[Dict[str, torch.Tensor], Dict[str, str]]:
    """(state dict as `YOLO(path).model.float().state_dict()` names and shapes it, {"names": repr(class names), "task": ..., "source": ...})"""
    path = Path(path)
    try:
        zf = zipfile.ZipFile(path)
    except (zipfile.BadZipFile, OSError) as e:
        raise ModelError(f"{path}: not a torch zip checkpoint ({e})") from e
    with zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            raise ModelError(f"{path}: expected one data.pkl in the archive, found {len(pkl)}")
        root = pkl[0][: -len("data.pkl")]
        try:
            ckpt = _Reader(io.BytesIO(zf.read(pkl[0])), zf, root).load()
        except ModelError:
            raise
        except Exception as e:                                  # truncated / foreign pickle
            raise ModelError(f"{path}: cannot read the checkpoint pickle ({type(e).__name__}: {e})") from e
    if isinstance(ckpt, dict):
        model, source = (ckpt.get("ema"), "ema") if isinstance(ckpt.get("ema"), Shell) else (ckpt.get("model"), "model")
    else:
        model, source = ckpt, "object"
    if not isinstance(model, Shell):
        raise ModelError(f"{path}: no pickled model object under 'ema' / 'model'")
    sd: Dict[str, torch.Tensor] = OrderedDict()
    _module_tensors(model, "", sd)
    sd = OrderedDict((k, v.float().contiguous()) for k, v in sd.items() if v.dtype.is_floating_point)
    if not sd:
        raise ModelError(f"{path}: the model object holds no floating-point tensors")
    meta = {"source": source}
    names = getattr(model, "__dict__", {}).get("names")
    if isinstance(names, (list, tuple)):
        names = dict(enumerate(names))
    if isinstance(names, dict) and all(isinstance(v, str) for v in names.values()):
        meta["names"] = repr({int(k): v for k, v in names.items()})
    task = getattr(model, "__dict__", {}).get("task")
    if isinstance(task, str):
        meta["task"] = task
    return sd, meta
