"""Shared request budget and FLUX wave scheduling.

Mirror of the reference's core/batch_coordinator.py:18-164 operator surface — same names, arguments
and results (`BatchRequestCoordinator.slot/run/map_ordered`, `bboxes_overlap`, `expanded_mask_bbox`,
`partition_non_overlapping_waves`, `paste_image_region`); pinned by tests/golden/batch_coordinator.json.
"""
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, List, Optional, Sequence, Tuple, TypeVar

import numpy as np

from ..utils.exceptions import CancellationError

T = TypeVar("T")
R = TypeVar("R")
BBox = Tuple[int, int, int, int]


class _SlotBudget:
    """A counted budget of request slots with an owner set: a thread that already owns a slot is let through
    again without taking a second one.  One condition variable guards the free count and the owner set, so
    ownership and the count can never disagree (the reference keeps a semaphore plus a thread-local flag)."""

    def __init__(self, capacity: int):
        self.capacity = capacity
        self._free = capacity
        self._owners = set()
        self._cv = threading.Condition()

    def owned_by_caller(self) -> bool:
        with self._cv:
            return threading.get_ident() in self._owners

    def take(self) -> bool:
        """True when this call took a slot (and must give it back), False when the thread already owned one."""
        me = threading.get_ident()
        with self._cv:
            if me in self._owners:
                return False
            while self._free == 0:
                self._cv.wait()
            self._free -= 1
            self._owners.add(me)
            return True

    def give_back(self) -> None:
        with self._cv:
            self._owners.discard(threading.get_ident())
            self._free += 1
            self._cv.notify()


class _HeldSlot:
    """Context manager of one `slot()` entry: cancellation is looked at before waiting and again once the slot is held."""

    def __init__(self, owner: "BatchRequestCoordinator"):
        self._owner = owner
        self._took = False

    def __enter__(self):
        budget = self._owner._budget
        if not budget.owned_by_caller():
            self._owner._stop_if_cancelled()
            self._took = budget.take()
            if self._took:
                try:
                    self._owner._stop_if_cancelled()
                except BaseException:
                    budget.give_back()
                    self._took = False
                    raise
        return None

    def __exit__(self, *exc):
        if self._took:
            self._took = False
            self._owner._budget.give_back()
        return False


class BatchRequestCoordinator:
    """Operator surface of the reference's class (core/batch_coordinator.py:18-75): at most `max_requests` jobs hold a
    slot at once, a thread inside a slot re-enters freely, `map_ordered` returns results in input order."""

    def __init__(self, max_requests: int, cancellation_manager=None):
        self.max_requests = max(1, int(max_requests or 1))
        self._budget = _SlotBudget(self.max_requests)
        self._cancel = cancellation_manager

    def _stop_if_cancelled(self) -> None:
        manager = self._cancel
        if manager is not None and manager.is_cancelled():
            raise CancellationError("Batch process cancelled by user.")

    def in_slot(self) -> bool:
        return self._budget.owned_by_caller()

    def slot(self) -> _HeldSlot:
        return _HeldSlot(self)

    def run(self, fn: Callable[..., R], *args, **kwargs) -> R:
        with _HeldSlot(self):
            return fn(*args, **kwargs)

    def map_ordered(self, jobs: Sequence[Callable[[], R]]) -> List[R]:
        jobs = list(jobs)
        width = min(len(jobs), self.max_requests)
        if width <= 1:                                  # nothing to overlap: stay on the caller's thread
            return [self.run(job) for job in jobs]
        with ThreadPoolExecutor(max_workers=width, thread_name_prefix="batch-request") as pool:
            return list(pool.map(self.run, jobs))


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
