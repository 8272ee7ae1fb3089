"""Resolution-scaled kernel sizes (reference core/scaling.py:64-96 `scale_kernel`; the page's
processing scale is sqrt(W*H / 1e6), reference core/pipeline.py:765-767)."""
import math
from typing import Optional, Tuple


def processing_scale(width: int, height: int) -> float:
    return math.sqrt(width * height / 1_000_000.0)


def _effective(scale: Optional[float]) -> float:
    if scale is None:
        return 1.0
    try:
        s = float(scale)
    except (TypeError, ValueError):
        return 1.0
    return s if (math.isfinite(s) and s > 0) else 1.0


def scale_kernel(kernel: Tuple[int, int], scale: Optional[float], *, minimum: int = 1, maximum: int = 63) -> Tuple[int, int]:
    """Linear scaling of a 2-D morphology kernel, clamped to [minimum, maximum] and forced odd
    (even results round up when that stays in range, else down)."""
    s = _effective(scale)

    def one(base: int) -> int:
        v = min(max(base * s, float(minimum)), float(maximum))
        d = min(maximum, max(minimum, int(round(v))))
        if d % 2 == 0:
            d = d + 1 if d + 1 <= maximum else d - 1
            d = max(minimum, d)
            if d % 2 == 0:
                d = max(minimum, d + 1)
        return max(minimum, d)

    return (one(kernel[0]), one(kernel[1]))


def scale_scalar(value: float, scale: Optional[float], *, minimum: Optional[float] = None, maximum: Optional[float] = None) -> float:
    """value * scale, clamped (reference core/scaling.py:18-31; a non-positive / missing scale counts as 1)."""
    s = 1.0 if (scale is None or scale <= 0) else float(scale)
    v = value * s
    if minimum is not None:
        v = max(minimum, v)
    if maximum is not None:
        v = min(maximum, v)
    return v


def scale_length(value: float, scale: Optional[float], *, minimum: Optional[float] = 1.0, maximum: Optional[float] = None) -> int:
    return max(1, int(round(scale_scalar(value, scale, minimum=minimum, maximum=maximum))))


def scale_area(value: float, scale: Optional[float], *, minimum: Optional[float] = 1.0, maximum: Optional[float] = None) -> int:
    """area-like quantities scale with scale**2 (reference core/scaling.py:49-62)"""
    s = 1.0 if (scale is None or scale <= 0) else float(scale)
    v = value * (s * s)
    if minimum is not None:
        v = max(minimum, v)
    if maximum is not None:
        v = min(maximum, v)
    return max(1, int(round(v)))
