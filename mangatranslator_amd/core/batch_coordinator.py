"""Shared request budget and FLUX wave scheduling.

Mirror of the reference's core/batch_coordinator.py:18-164 operator surface — same names, arguments
and results (`BatchRequestCoordinator.slot/run/map_ordered`, `bboxes_overlap`, `expanded_mask_bbox`,
`partition_non_overlapping_waves`, `paste_image_region`); pinned by tests/golden/batch_coordinator.json.
"""
import threading
from concurrent.futures import ThreadPoolExecutor
from contextlib import contextmanager
from typing import Callable, Iterable, List, Optional, Sequence, Tuple, TypeVar

import numpy as np

from ..utils.exceptions import CancellationError

T = TypeVar("T")
R = TypeVar("R")
BBox = Tuple[int, int, int, int]


class BatchRequestCoordinator:
    """At most `max_requests` jobs hold a slot at once; a thread already inside a slot re-enters freely."""

    def __init__(self, max_requests: int, cancellation_manager=None):
        self.max_requests = max(1, int(max_requests or 1))
        self._slots = threading.BoundedSemaphore(self.max_requests)
        self._cancel = cancellation_manager
        self._tls = threading.local()

    def _raise_if_cancelled(self):
        if self._cancel is not None and self._cancel.is_cancelled():
            raise CancellationError("Batch process cancelled by user.")

    def in_slot(self) -> bool:
        return getattr(self._tls, "depth", 0) > 0

    @contextmanager
    def slot(self):
        if self.in_slot():
            yield
            return
        self._raise_if_cancelled()
        self._slots.acquire()
        self._tls.depth = 1
        try:
            self._raise_if_cancelled()
            yield
        finally:
            self._tls.depth = 0
            self._slots.release()

    def run(self, fn: Callable[..., R], *args, **kwargs) -> R:
        with self.slot():
            return fn(*args, **kwargs)

    def map_ordered(self, jobs: Sequence[Callable[[], R]]) -> List[R]:
        if not jobs:
            return []
        if len(jobs) == 1:
            return [self.run(jobs[0])]
        with ThreadPoolExecutor(max_workers=min(len(jobs), self.max_requests)) as pool:
            futures = [pool.submit(self.run, job) for job in jobs]
            return [f.result() for f in futures]


def bboxes_overlap(first: BBox, second: BBox) -> bool:
    """Open-interval overlap test on (x1, y1, x2, y2): touching edges do not overlap."""
    return first[0] < second[2] and second[0] < first[2] and first[1] < second[3] and second[1] < first[3]


def expanded_mask_bbox(mask: np.ndarray, image_size: Tuple[int, int], padding_ratio: float = 0.5,
                       max_padding: int = 160, min_padding: int = 64, extra_padding: int = 16) -> Optional[BBox]:
    """Mask bbox grown by max(min_padding, min(ratio * longer side, max_padding)) + extra, clipped to the page."""
    m = np.asarray(mask)
    if m.ndim == 3:
        m = m[..., 0]
    m = m.astype(bool)
    rows, cols = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
    if rows.size == 0 or cols.size == 0:
        return None
    img_w, img_h = image_size
    x1, x2, y1, y2 = int(cols[0]), int(cols[-1]) + 1, int(rows[0]), int(rows[-1]) + 1
    pad = max(min_padding, int(min(max(x2 - x1, y2 - y1) * padding_ratio, max_padding))) + extra_padding
    return (max(0, x1 - pad), max(0, y1 - pad), min(img_w, x2 + pad), min(img_h, y2 + pad))


def partition_non_overlapping_waves(items: Iterable[T], get_bbox: Callable[[T], Optional[BBox]]) -> List[List[T]]:
    """Greedy in-order split: a wave closes as soon as the next bbox overlaps one already in it;
    an item without a bbox always runs alone."""
    waves: List[List[T]] = []
    wave: List[T] = []
    boxes: List[BBox] = []
    for item in items:
        bbox = get_bbox(item)
        if bbox is None:
            if wave:
                waves.append(wave)
            waves.append([item])
            wave, boxes = [], []
            continue
        if any(bboxes_overlap(bbox, other) for other in boxes):
            waves.append(wave)
            wave, boxes = [], []
        wave.append(item)
        boxes.append(bbox)
    if wave:
        waves.append(wave)
    return waves


def paste_image_region(target, source, bbox: BBox):
    out = target.copy()
    out.paste(source.crop(bbox), (bbox[0], bbox[1]))
    return out
