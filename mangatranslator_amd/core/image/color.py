"""8-bit sRGB <-> CIE Lab (D65) with OpenCV's 8-bit conventions (L * 255 / 100, a + 128, b + 128) — what
`FluxKleinInpainter._match_luminance` needs (reference core/image/inpainting.py:1187-1256 calls cv2.cvtColor with
COLOR_RGB2LAB / COLOR_LAB2RGB; OpenCV is not a dependency of this build).

Forward: OpenCV's published fixed-point scheme (gamma table with 3 fractional bits, 12-bit XYZ matrix over the white point, cube-root
table in 2^-15 units), vectorised through three lookup tables.  Inverse: the float formulation rounded to uint8.  The primitives are
"parity unpinned" against OpenCV itself (oracle/cv2_color_ref.py says why); the luminance-matching flow around them is pinned.
"""
import numpy as np

_M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
_WHITE = np.array([0.950456, 1.0, 1.088754], np.float32)
_INV = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float32)
_GAMMA_SHIFT, _LAB_SHIFT, _LAB_SHIFT2 = 3, 12, 15

_x = np.arange(256, dtype=np.float32) / np.float32(255.0)
_lin = np.where(_x <= np.float32(0.04045), _x.astype(np.float64) / 12.92, ((_x.astype(np.float64) + 0.055) / 1.055) ** 2.4)
_GAMMA_TAB = np.clip(np.rint(np.float32(255.0 * (1 << _GAMMA_SHIFT)) * _lin.astype(np.float32)), 0, 65535).astype(np.int64)
_t = np.arange(256 * 3 // 2 * (1 << _GAMMA_SHIFT), dtype=np.float32) * np.float32(1.0 / (255.0 * (1 << _GAMMA_SHIFT)))
_c = np.where(_t < np.float32(0.008856), _t.astype(np.float64) * 7.787 + 0.13793103448275862, np.cbrt(_t).astype(np.float64))
_CBRT_TAB = np.clip(np.rint(np.float32(1 << _LAB_SHIFT2) * _c.astype(np.float32)), 0, 65535).astype(np.int64)
_COEF = np.rint(_M * (np.float32(1 << _LAB_SHIFT) / _WHITE)[:, None].astype(np.float32)).astype(np.int64)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


_GAMMA32, _CBRT32, _COEF32 = _GAMMA_TAB.astype(np.int32), _CBRT_TAB.astype(np.int32), _COEF.astype(np.int32)


def rgb_to_lab_u8(rgb: np.ndarray) -> np.ndarray:
    """uint8 [..., 3] RGB -> uint8 [..., 3] (L, a, b).  int32 throughout (every intermediate stays below 2^26): three table lookups and
    nine multiply-adds per pixel; the luminance match of a 1 MP crop calls this twice per region on the host"""
    src = np.asarray(rgb, np.uint8)
    r, g, b_ = _GAMMA32[src[..., 0]], _GAMMA32[src[..., 1]], _GAMMA32[src[..., 2]]
    half = np.int32(1 << (_LAB_SHIFT - 1))
    f = []
    for c in range(3):
        xyz = (r * _COEF32[c, 0] + g * _COEF32[c, 1] + b_ * _COEF32[c, 2] + half) >> _LAB_SHIFT
        np.clip(xyz, 0, _CBRT32.size - 1, out=xyz)
        f.append(_CBRT32[xyz])
    fx, fy, fz = f
    lshift = -((16 * 255 * (1 << _LAB_SHIFT2) + 50) // 100)
    h2 = np.int32(1 << (_LAB_SHIFT2 - 1))
    out = np.empty(src.shape[:-1] + (3,), np.uint8)
    L = (np.int32((116 * 255 + 50) // 100) * fy + np.int32(lshift) + h2) >> _LAB_SHIFT2
    a = (np.int32(500) * (fx - fy) + np.int32(128 * (1 << _LAB_SHIFT2)) + h2) >> _LAB_SHIFT2
    b = (np.int32(200) * (fy - fz) + np.int32(128 * (1 << _LAB_SHIFT2)) + h2) >> _LAB_SHIFT2
    out[..., 0], out[..., 1], out[..., 2] = np.clip(L, 0, 255), np.clip(a, 0, 255), np.clip(b, 0, 255)
    return out


def lab_to_rgb_u8(lab: np.ndarray) -> np.ndarray:
    """uint8 [..., 3] (L, a, b) -> uint8 [..., 3] RGB"""
    f32 = np.float32
    v = np.asarray(lab, np.uint8).astype(f32)
    L = v[..., 0] * f32(100.0 / 255.0)
    A, B = v[..., 1] - f32(128.0), v[..., 2] - f32(128.0)
    low = L <= f32(0.008856 * 903.3)
    fy_hi = (L + f32(16.0)) / f32(116.0)
    y = np.where(low, L / f32(903.3), fy_hi * fy_hi * fy_hi).astype(f32)
    fy = np.where(low, f32(7.787) * (L / f32(903.3)) + f32(16.0 / 116.0), fy_hi).astype(f32)

    def inv_f(f):
        return np.where(f <= f32(7.787 * 0.008856 + 16.0 / 116.0), (f - f32(16.0 / 116.0)) / f32(7.787), f * f * f).astype(f32)
    x = inv_f(fy + A / f32(500.0)) * _WHITE[0]
    z = inv_f(fy - B / f32(200.0)) * _WHITE[2]
    out = []
    for r in range(3):
        lin = np.clip(_INV[r, 0] * x + _INV[r, 1] * y + _INV[r, 2] * z, f32(0.0), f32(1.0)).astype(f32)
        g = np.where(lin <= f32(0.0031308), lin * f32(12.92), f32(1.055) * np.power(lin, f32(1.0 / 2.4), dtype=f32) - f32(0.055)).astype(f32)
        out.append(np.clip(np.rint(g * f32(255.0)), 0, 255))
    return np.stack(out, -1).astype(np.uint8)
