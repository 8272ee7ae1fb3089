"""`failed_paths.txt` writer of the batch harness (reference utils/path_list.py:8-9, 36-78)."""
from pathlib import Path
from typing import Iterable, List, Optional, Union

from .logging import log_message

IMAGE_EXTENSIONS = {".jpg", ".jpeg", ".png", ".webp"}
FAILED_PATHS_FILENAME = "failed_paths.txt"


def write_failed_paths(output_dir: Union[str, Path], paths: Iterable[str]) -> Optional[Path]:
    """unique absolute paths, first occurrence order, one per line; None when there is nothing to write"""
    unique: List[str] = []
    seen = set()
    for raw in paths:
        if raw is None:
            continue
        text = str(raw).strip()
        if not text:
            continue
        try:
            abs_path = str(Path(text).resolve())
        except OSError:
            abs_path = text
        if abs_path not in seen:
            seen.add(abs_path)
            unique.append(abs_path)
    if not unique:
        return None
    out_dir = Path(output_dir)
    try:
        out_dir.mkdir(parents=True, exist_ok=True)
        out_file = out_dir / FAILED_PATHS_FILENAME
        out_file.write_text("\n".join(unique) + "\n", encoding="utf-8")
        return out_file
    except OSError as e:
        log_message(f"Warning: failed to write {FAILED_PATHS_FILENAME}: {e}", always_print=True)
        return None
