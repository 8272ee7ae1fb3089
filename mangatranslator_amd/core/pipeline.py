"""Batch harness helpers of the hot path and the page sharding the reference does not have
(SURVEY.md §8 rows a11 / e).  `_natural_path_sort_key` / `_resolve_output_path` give the reference's
page order and output naming (core/pipeline.py:133-142, 2027-2064; pinned by tests/golden/harness.json);
`shard_pages` partitions the sorted page list one page per GPU, rank r taking pages r, r+G, r+2G, ...
with no data-path collective; `merge_batch_results` is the host-side gather of the per-rank result
dicts (`success_count`, `error_count`, `errors`, `failed_image_paths`, core/pipeline.py:2233-2239).
"""
import os
import re
from pathlib import Path
from typing import Callable, Dict, List, Sequence, Tuple

from ..utils.logging import log_message

NATURAL_SORT_TOKEN_RE = re.compile(r"(\d+)")


def _natural_text_sort_key(text: str):
    return tuple((0, int(tok), tok) if tok.isdigit() else (1, tok.lower(), tok)
                 for tok in NATURAL_SORT_TOKEN_RE.split(text) if tok)


def _natural_path_sort_key(path: Path):
    return tuple(_natural_text_sort_key(part) for part in Path(path).parts)


def _resolve_output_path(img_path: Path, input_dir: Path, output_dir: Path, config, preserve_structure: bool) -> Tuple[Path, str, str]:
    if preserve_structure:
        rel = img_path.relative_to(input_dir)
        out_dir = output_dir / rel.parent
        os.makedirs(out_dir, exist_ok=True)
        stem, display = rel.stem, str(rel)
    else:
        out_dir, stem, display = output_dir, img_path.stem, img_path.name
    fmt = config.output.output_format
    ext = {"jpeg": ".jpg", "png": ".png"}.get(fmt, img_path.suffix.lower())
    if fmt not in ("jpeg", "png", "auto"):
        log_message(f"Warning: Invalid output_format '{fmt}' in config. Using original extension '{ext}'.", always_print=True)
    return out_dir / f"{stem}_translated{ext}", display, display


def shard_pages(pages: Sequence, rank: int, world_size: int) -> List:
    """Static round-robin partition of the naturally sorted page list (pages are independent units)."""
    ordered = sorted(pages, key=lambda p: _natural_path_sort_key(Path(p)))
    return ordered[rank::world_size]


def merge_batch_results(per_rank: Sequence[Dict]) -> Dict:
    merged = {"success_count": 0, "error_count": 0, "errors": {}, "failed_image_paths": []}
    for r in per_rank:
        merged["success_count"] += int(r.get("success_count", 0))
        merged["error_count"] += int(r.get("error_count", 0))
        merged["errors"].update(r.get("errors", {}))
        merged["failed_image_paths"].extend(r.get("failed_image_paths", []))
    merged["failed_image_paths"].sort(key=lambda p: _natural_path_sort_key(Path(p)))
    return merged


def process_pages_sharded(pages: Sequence, process_page: Callable[[Path], None]) -> Dict:
    """The page-sharded batch loop: every rank of the initialised process group (one per
    GPU) walks its own slice of the sorted page list with `process_page`, failures are collected per page
    as the reference's batch loop does (core/pipeline.py:2233-2239), and the per-rank result dicts are
    gathered host-side.  Every rank returns the merged dict.  Without a process group: plain loop."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    local = {"success_count": 0, "error_count": 0, "errors": {}, "failed_image_paths": []}
    for page in shard_pages(pages, rank, world):
        try:
            process_page(Path(page))
            local["success_count"] += 1
        except Exception as e:      # noqa: BLE001 — a bad page must not stop the batch
            local["error_count"] += 1
            local["errors"][Path(page).name] = str(e)
            local["failed_image_paths"].append(str(page))
    if not multi:
        return merge_batch_results([local])
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    return merge_batch_results(gathered)
