"""`failed_paths.txt` of the batch harness (behaviour of the reference's utils/path_list.py:36-78): the absolute paths of the
pages that failed, each once, in first-failure order, one per line."""
from pathlib import Path
from typing import Iterable, Optional, Union

from .logging import log_message

IMAGE_EXTENSIONS = frozenset((".jpg", ".jpeg", ".png", ".webp"))
FAILED_PATHS_FILENAME = "failed_paths.txt"


def _absolute(text: str) -> str:
    try:
        return str(Path(text).resolve())
    except OSError:
        return text


def write_failed_paths(output_dir: Union[str, Path], paths: Iterable[str]) -> Optional[Path]:
    """-> the file written, or None when there was nothing to record (or the directory is not writable)"""
    cleaned = (str(p).strip() for p in paths if p is not None)
    ordered = list(dict.fromkeys(_absolute(t) for t in cleaned if t))
    if not ordered:
        return None
    target = Path(output_dir) / FAILED_PATHS_FILENAME
    try:
        target.parent.mkdir(parents=True, exist_ok=True)
        target.write_text("".join(line + "\n" for line in ordered), encoding="utf-8")
    except OSError as e:
        log_message(f"Warning: failed to write {FAILED_PATHS_FILENAME}: {e}", always_print=True)
        return None
    return target
