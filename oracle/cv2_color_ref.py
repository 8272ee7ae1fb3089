"""OpenCV 8-bit sRGB <-> CIE Lab conversions (cv2.cvtColor COLOR_RGB2LAB / COLOR_LAB2RGB) restated.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: OpenCV is absent from this image (reference requirement `opencv-python`); the only call sites on the hot path are
FluxKleinInpainter._compute_luminance_stats / _match_luminance (reference core/image/inpainting.py:1187-1256).  Restated from
OpenCV's published algorithm (modules/imgproc/src/color_lab.cpp):
  forward (RGB2Lab_b): sRGB gamma table with 3 fractional bits, 12-bit fixed-point XYZ matrix scaled by the D65 white point,
      cube-root table in 1/2^15 units, L = (296 fY - 16*255/100) etc. with CV_DESCALE rounding; output L*255/100, a+128, b+128.
  inverse: the float formulation (Lab2RGBfloat) on L*100/255, a-128, b-128, sRGB gamma, x255 rounded and saturated.  OpenCV's
      8-bit inverse is an integer approximation of this formula and can differ from it by one level.
Plain per-pixel Python/numpy, written for clarity — the product's vectorised version lives in mangatranslator_amd/core/image/color.py.
"""
import numpy as np

_M = [0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227]
_WHITE = [0.950456, 1.0, 1.088754]
_INV = [3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311]


def _round_half_even(v):
    return int(np.rint(v))


def _tables():
    gamma = np.zeros(256, np.int64)
    for i in range(256):
        x = np.float32(i) / np.float32(255.0)
        v = float(x) / 12.92 if x <= np.float32(0.04045) else ((float(x) + 0.055) / 1.055) ** 2.4
        gamma[i] = min(65535, max(0, _round_half_even(np.float32(255.0 * 8) * np.float32(v))))
    cbrt = np.zeros(256 * 3 // 2 * 8, np.int64)
    for i in range(cbrt.size):
        x = np.float32(i) * np.float32(1.0 / (255.0 * 8))
        v = float(x) * 7.787 + 0.13793103448275862 if x < np.float32(0.008856) else float(np.cbrt(np.float32(x)))
        cbrt[i] = min(65535, max(0, _round_half_even(np.float32(1 << 15) * np.float32(v))))
    coef = [_round_half_even(np.float32(_M[r * 3 + c]) * np.float32((1 << 12) / _WHITE[r])) for r in range(3) for c in range(3)]
    return gamma, cbrt, coef


_GAMMA, _CBRT, _COEF = _tables()


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def rgb_to_lab_u8(img: np.ndarray) -> np.ndarray:
    a = np.asarray(img, np.uint8)                     # 3 or 4 channels in (alpha ignored), 3 out — as OpenCV
    out = np.zeros(a.shape[:-1] + (3,), np.uint8)
    lshift = -((16 * 255 * (1 << 15) + 50) // 100)
    for idx in np.ndindex(a.shape[:-1]):
        R, G, B = (int(_GAMMA[int(v)]) for v in a[idx][:3])
        fx = int(_CBRT[_descale(R * _COEF[0] + G * _COEF[1] + B * _COEF[2], 12)])
        fy = int(_CBRT[_descale(R * _COEF[3] + G * _COEF[4] + B * _COEF[5], 12)])
        fz = int(_CBRT[_descale(R * _COEF[6] + G * _COEF[7] + B * _COEF[8], 12)])
        L = _descale(296 * fy + lshift, 15)
        A = _descale(500 * (fx - fy) + 128 * (1 << 15), 15)
        Bb = _descale(200 * (fy - fz) + 128 * (1 << 15), 15)
        out[idx] = [min(255, max(0, L)), min(255, max(0, A)), min(255, max(0, Bb))]
    return out


def lab_to_rgb_u8(img: np.ndarray) -> np.ndarray:
    a = np.asarray(img, np.uint8)
    out = np.zeros(a.shape, np.uint8)
    f32 = np.float32
    for idx in np.ndindex(a.shape[:-1]):
        L = f32(a[idx][0]) * f32(100.0 / 255.0)
        A, B = f32(a[idx][1]) - f32(128.0), f32(a[idx][2]) - f32(128.0)
        if L <= f32(0.008856 * 903.3):
            y = L / f32(903.3)
            fy = f32(7.787) * y + f32(16.0 / 116.0)
        else:
            fy = (L + f32(16.0)) / f32(116.0)
            y = fy * fy * fy
        xz = []
        for f in (fy + A / f32(500.0), fy - B / f32(200.0)):
            xz.append((f - f32(16.0 / 116.0)) / f32(7.787) if f <= f32(7.787 * 0.008856 + 16.0 / 116.0) else f * f * f)
        x, z = xz[0] * f32(_WHITE[0]), xz[1] * f32(_WHITE[2])
        rgb = []
        for r in range(3):
            lin = f32(_INV[r * 3]) * x + f32(_INV[r * 3 + 1]) * y + f32(_INV[r * 3 + 2]) * z
            lin = min(f32(1.0), max(f32(0.0), lin))
            g = lin * f32(12.92) if lin <= f32(0.0031308) else f32(1.055) * f32(np.power(lin, f32(1.0 / 2.4))) - f32(0.055)
            rgb.append(min(255, max(0, _round_half_even(g * f32(255.0)))))
        out[idx] = rgb
    return out
