"""Synthetic manga pages for benchmarks and tests: seed = 1234 + page_index,
RGB uint8, near-white paper, 6 screentone rectangles, B elliptical speech bubbles with dark
strokes inside, R outside-text blocks on a non-solid gradient background (placed clear of the bubbles).  Returns the page and
the generator's ground-truth bubble boxes (used in place of detector output when checkpoints are
absent, so every stage processes a fixed unit count)."""
import numpy as np


def make_page(index: int, width: int = 1024, height: int = 1536, bubbles: int = 8, osb_regions: int = 0):
    rng = np.random.default_rng(1234 + index)
    page = np.clip(rng.normal(250, 3, (height, width, 1)), 0, 255).repeat(3, axis=2)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(6):
        rw, rh = int(rng.integers(width // 6, width // 2)), int(rng.integers(height // 8, height // 3))
        x0, y0 = int(rng.integers(0, width - rw)), int(rng.integers(0, height - rh))
        grey = float(rng.integers(120, 201))
        tone = (((xx[y0:y0 + rh, x0:x0 + rw] // 2) + (yy[y0:y0 + rh, x0:x0 + rw] // 2)) % 2).astype(np.float64)
        page[y0:y0 + rh, x0:x0 + rw, :] = (grey + (255 - grey) * tone)[..., None]
    boxes = []
    for _ in range(bubbles):
        a, b = float(rng.uniform(80, 220)) * width / 1024 / 2, float(rng.uniform(80, 220)) * width / 1024 / 2
        cx, cy = float(rng.uniform(a + 4, width - a - 4)), float(rng.uniform(b + 4, height - b - 4))
        d = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2
        page[d <= 1.0] = 255
        ring = (d <= 1.0) & (((xx - cx) / (a - 3)) ** 2 + ((yy - cy) / (b - 3)) ** 2 > 1.0)
        page[ring] = 0
        for _ in range(int(rng.integers(4, 11))):
            sx = int(cx + rng.uniform(-0.5, 0.5) * a)
            sy = int(cy + rng.uniform(-0.5, 0.5) * b)
            ln = int(rng.integers(8, max(9, int(0.4 * b))))
            page[max(sy - ln // 2, 0):sy + ln // 2, max(sx - 1, 0):sx + 2] = 20
        boxes.append([cx - a, cy - b, cx + a, cy + b])
    regions = []
    for _ in range(osb_regions):
        rw, rh = 200 * width // 1024, 120 * width // 1024
        for _try in range(64):     # clear of the bubbles and earlier blocks (a text box inside a bubble is dialogue, not outside text)
            x0, y0 = int(rng.integers(0, width - rw)), int(rng.integers(0, height - rh))
            if not any(x0 < b[2] + 16 and b[0] - 16 < x0 + rw and y0 < b[3] + 16 and b[1] - 16 < y0 + rh for b in boxes + regions):
                break
        grad = np.linspace(90, 230, rw)[None, :, None] + np.linspace(-25, 25, rh)[:, None, None]
        page[y0:y0 + rh, x0:x0 + rw] = grad
        for _ in range(14):
            sx, sy = int(rng.integers(x0 + 6, max(x0 + 7, x0 + rw - 6))), int(rng.integers(y0 + 6, max(y0 + 7, y0 + rh - 26)))   # (max: pages narrower than ~280 px)
            page[sy:sy + 20, sx:sx + 3] = 15
        regions.append([x0, y0, x0 + rw, y0 + rh])
    return np.clip(page, 0, 255).astype(np.uint8), np.asarray(boxes, dtype=np.float32), regions
