"""Exception types of the reference's error contract (reference: utils/exceptions.py:1-50).

Callers of the hot path catch these and degrade (SAM -> YOLO masks, FLUX -> flat fill,
SURVEY.md §5), so the drop-in must raise the same classes with the same bases.
"""


class ValidationError(ValueError):
    pass


class ModelError(RuntimeError):
    pass


class ImageProcessingError(Exception):
    pass


class DetectionError(RuntimeError):
    pass


class CleaningError(Exception):
    pass


class CancellationError(Exception):
    pass
