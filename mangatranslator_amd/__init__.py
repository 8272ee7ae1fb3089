"""mangatranslator_amd — MI355X (gfx950) implementation of MangaTranslator's vision hot path.

Layout:
  csrc/   hand-written HIP kernels + the C ABI (include/mtx_hip.h) -> csrc/libmtx_hip.so
  hip/    ctypes binding of that ABI and the static-graph ("plan") builder
  core/   host-side mirror of the reference operator surface (core.ml / core.image), so the
          reference's callers (core/pipeline.py, main.py, app.py) find the same names
  utils/  the reference's exception types and logging shim (error contract)
"""
from ._version import __version__, __version_info__  # noqa: F401
