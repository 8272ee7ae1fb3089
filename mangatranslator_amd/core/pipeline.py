"""Batch harness helpers of the hot path and the page sharding the reference does not have
(SURVEY.md §8 rows a11 / e).  `_natural_path_sort_key` / `_resolve_output_path` give the reference's
page order and output naming (core/pipeline.py:133-142, 2027-2064; pinned by tests/golden/harness.json);
`shard_pages` partitions the sorted page list one page per GPU, rank r taking pages r, r+G, r+2G, ...
with no data-path collective; `merge_batch_results` is the host-side gather of the per-rank result
dicts (`success_count`, `error_count`, `errors`, `failed_image_paths`, core/pipeline.py:2233-2239).
"""
import os
import re
import threading
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from ..utils.logging import log_message
from ..utils.path_list import IMAGE_EXTENSIONS, write_failed_paths

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
    ios = [r["io"] for r in per_rank if isinstance(r.get("io"), dict)]
    if ios:      # host I/O accounting of `batch_process_images` (not a key of the reference's dict): sums over the ranks, maxima where it is one
        merged["io"] = {k: (max(i[k] for i in ios) if k.startswith("max_") else sum(i[k] for i in ios)) for k in ios[0]}
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


# ---- the batch harness's image I/O around the GPU work ------------------------------------------------
def collect_image_files(input_dir: Path, preserve_structure: bool = False) -> List[Path]:
    """the page list of a batch in the reference's order (core/pipeline.py:2532-2560): .jpg/.jpeg/.png/.webp files of the directory
    (recursively with preserve_structure), naturally sorted on the path relative to the input directory"""
    input_dir = Path(input_dir)
    if preserve_structure:
        files = [Path(root) / f for root, _dirs, names in os.walk(input_dir) for f in names if Path(f).suffix.lower() in IMAGE_EXTENSIONS]
    else:
        files = [f for f in input_dir.iterdir() if f.is_file() and f.suffix.lower() in IMAGE_EXTENSIONS]

    def key(path: Path):
        try:
            rel = path.relative_to(input_dir) if preserve_structure else Path(path.name)
        except ValueError:
            rel = path
        return _natural_path_sort_key(rel)
    files.sort(key=key)
    return files


def load_page(img_path: Path, output_format: str):
    """decode + the reference's mode rule (core/pipeline.py:707-717): RGBA unless the output is JPEG; transparency is flattened onto
    white on the way to RGB (`convert_image_to_target_mode`, :715-717)"""
    from PIL import Image
    from .image.image_utils import convert_image_to_target_mode
    with Image.open(img_path) as im:
        im.load()
        page = im.copy()
    jpeg_out = output_format == "jpeg" or (output_format == "auto" and Path(img_path).suffix.lower() in (".jpg", ".jpeg"))
    target = "RGB" if jpeg_out else "RGBA"
    return convert_image_to_target_mode(page, target)


def batch_process_images(input_dir, config, output_dir=None, preserve_structure: bool = False,
                         process_image: Optional[Callable] = None, io_threads: int = 2,
                         process_front: Optional[Callable] = None, process_back: Optional[Callable] = None,
                         front_workers: int = 1, front_context: Optional[Callable] = None,
                         preload: Optional[Callable] = None) -> Dict:
    """The vision half of `batch_translate_images` (core/pipeline.py:2481-2733) on one GPU or page-sharded over the ranks of the
    initialised process group: same page list and order, same output naming (`_resolve_output_path`), same results dict
    (`success_count`, `error_count`, `errors` keyed by the display path, `failed_image_paths` absolute, `failed_paths_file`), a bad
    page never stops the batch.  `process_image(page: PIL.Image, path: Path) -> PIL.Image` is the per-page vision stack.
    Host codec work is kept off the GPU's critical path: `io_threads` workers decode the next pages while the current one is on the
    GPU and `io_threads` more encode / write finished pages behind it (PIL releases the GIL in its codecs); at most 2 x io_threads
    finished pages wait for an encoder (back-pressure on the page loop).  `results["io"]` accounts for the host side: decode / encode /
    process seconds, the waits, the deepest save queue.
    **Two pages in flight** (round 4): with `process_front(page, path) -> state` and `process_back(state) -> PIL.Image` instead of
    `process_image`, page i + 1's front half (detect / segment / OSB prepare: host-heavy, small GPU graphs on the models' own streams) runs
    on a worker thread beside page i's back half (diffusion, upscaling, cleaning: GPU-bound) — `process_page_vision_front` /
    `process_page_vision_back` below are that pair.  Pages still complete, are saved and are reported in batch order; a page whose front
    or back half raises is recorded as failed exactly like a failing `process_image`, and the pages around it are not affected.
    **`front_workers` = N > 1**: N front halves run at once (pages i + 1 .. i + N beside page i's back half), each holding one of N
    slots for its duration; `front_context(slot)` is entered around the front half — by default `ModelManager.front_replica(slot)`, which
    serves the detectors and SAM from instance set `slot` (a model instance holds one page at a time).  Pays off for stage sets whose back
    half is short (detect / clean only: the 640-pixel detector graphs do not fill the chip) and needs `GPU_MAX_HW_QUEUES=16` in the
    environment — two pages' model streams on ROCm's default four hardware queues run slower than one page.
    **Worker threads and the process group** (ADVICE r04): `preload()` is called once on the calling thread before the first front half
    starts — the place to load every model through the rank-collective path (`ModelManager.preload_for_config`) — and every front half runs
    with the calling thread's current device (the HIP device is per thread and defaults to 0; ranks > 0 work on cuda:LOCAL_RANK)."""
    from .image.image_utils import save_image_with_compression
    if (process_front is None) != (process_back is None):
        raise ValueError("process_front and process_back come as a pair")
    pipelined = process_front is not None
    front_workers = max(1, int(front_workers)) if pipelined else 1
    if front_workers > 1 and front_context is None:
        from .ml.model_manager import get_model_manager
        front_context = get_model_manager().front_replica
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
            log_message("front_workers > 1 without GPU_MAX_HW_QUEUES=16 in the environment: two pages' model streams will share the "
                        "runtime's default hardware queues (expect no gain)", always_print=True)
    import torch.distributed as dist
    empty = {"success_count": 0, "error_count": 0, "errors": {}, "failed_image_paths": []}
    input_dir = Path(input_dir)
    if not input_dir.is_dir():
        log_message(f"Input path '{input_dir}' is not a directory", always_print=True)
        return empty
    output_dir = Path(output_dir) if output_dir else Path("./output") / time.strftime("%Y%m%d_%H%M%S")
    os.makedirs(output_dir, exist_ok=True)
    files = collect_image_files(input_dir, preserve_structure)
    if not files:
        log_message(f"No image files found in '{input_dir}'", always_print=True)
        return empty
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    mine = files[rank::world]                      # already in batch order: rank r takes pages r, r + G, ...
    fmt = config.output.output_format
    local = {"success_count": 0, "error_count": 0, "errors": {}, "failed_image_paths": []}
    t0 = time.time()

    def fail(img_path, error_key, e):
        log_message(f"Error processing {error_key}: {e}", always_print=True)
        local["error_count"] += 1
        local["errors"][error_key] = str(e)
        try:
            local["failed_image_paths"].append(str(Path(img_path).resolve()))
        except OSError:
            local["failed_image_paths"].append(str(img_path))

    io = {"decode_s": 0.0, "encode_s": 0.0, "gpu_wait_for_decode_s": 0.0, "wait_for_save_slot_s": 0.0, "process_s": 0.0, "pages": 0,
          "max_pending_saves": 0, "front_s": 0.0, "back_s": 0.0, "pages_in_flight": 1 + front_workers if pipelined else 1}
    io_lock = threading.Lock()

    def timed(key, fn, *a):
        t = time.perf_counter()
        try:
            return fn(*a)
        finally:
            with io_lock:
                io[key] += time.perf_counter() - t

    def settle(entry):
        fut, img_path, error_key = entry
        try:
            fut.result()
            local["success_count"] += 1
        except Exception as e:      # noqa: BLE001
            fail(img_path, error_key, e)

    import contextlib
    import queue
    if preload is not None and pipelined:
        try:
            preload()
        except Exception as e:      # noqa: BLE001 — pages meet the same error on their own failure paths
            log_message(f"Model preload failed: {e}", always_print=True)
    main_device = None
    try:
        import torch
        if torch.cuda.is_available():
            main_device = torch.cuda.current_device()
    except Exception:      # noqa: BLE001
        main_device = None
    free_slots = queue.SimpleQueue()               # a front half holds one slot (= one instance set of the front-half models) while it runs
    for slot in range(front_workers):
        free_slots.put(slot)

    def front_task(i):
        """decoded page i through the front half (worker thread); what it returns or raises belongs to page i"""
        t = time.perf_counter()
        if main_device is not None:
            import torch
            torch.cuda.set_device(main_device)     # per thread: without it `torch.cuda.current_stream()` and hipSetDevice-scoped calls of this thread mean device 0
        page = decodes.pop(i).result()
        t1 = time.perf_counter()
        slot = free_slots.get()
        try:
            with (front_context(slot) if front_context is not None else contextlib.nullcontext()):
                state = process_front(page, mine[i])
        finally:
            free_slots.put(slot)
        with io_lock:
            io["gpu_wait_for_decode_s"] += t1 - t
            io["front_s"] += time.perf_counter() - t1
        return state

    queued_fronts = 0

    def queue_fronts(upto):
        """front halves of the pages below `upto` are queued (each exactly once, in page order)"""
        nonlocal queued_fronts
        while queued_fronts < min(upto, len(mine)):
            fronts[queued_fronts] = front_pool.submit(front_task, queued_fronts)
            queued_fronts += 1

    # decoders and encoders do not share a queue: the next page's decode must never wait behind the finished pages' (much slower) PNG encodes
    with ThreadPoolExecutor(max_workers=max(1, io_threads)) as pool, ThreadPoolExecutor(max_workers=max(1, io_threads)) as enc_pool, \
            ThreadPoolExecutor(max_workers=front_workers) as front_pool:
        ahead = max(1, io_threads, front_workers)
        max_pending = 2 * max(1, io_threads)       # finished pages waiting for a codec thread: a 4096x6144 RGBA page is 100 MB, and PNG
        decodes = {i: pool.submit(timed, "decode_s", load_page, mine[i], fmt) for i in range(min(ahead, len(mine)))}      # encoding is slower than the GPU
        saves = deque()
        fronts = {}
        if pipelined:
            queue_fronts(front_workers)
        for i, img_path in enumerate(mine):
            if i + ahead < len(mine):
                decodes[i + ahead] = pool.submit(timed, "decode_s", load_page, mine[i + ahead], fmt)
            error_key = img_path.name
            try:
                out_path, display, error_key = _resolve_output_path(img_path, input_dir, output_dir, config, preserve_structure)
                log_message(f"Processing {rank + i * world + 1}/{len(files)}: {display}", always_print=True)
                t = time.perf_counter()
                if pipelined:
                    queue_fronts(i + 1 + front_workers)      # the next pages' front halves start before this page's back half
                    state = fronts.pop(i).result()
                    t1 = time.perf_counter()
                    result = process_back(state)
                    t2 = time.perf_counter()
                    io["back_s"] += t2 - t1
                    t1 = t                          # process_s = this page's share of the page loop (waiting for its front half + its back half)
                else:
                    page = decodes.pop(i).result()
                    t1 = time.perf_counter()
                    result = process_image(page, img_path) if process_image is not None else page
                    t2 = time.perf_counter()
                    io["gpu_wait_for_decode_s"] += t1 - t
                while len(saves) >= max_pending:      # back-pressure: the page loop waits for the oldest save instead of queueing pages without bound
                    settle(saves.popleft())
                io["process_s"] += t2 - t1
                io["wait_for_save_slot_s"] += time.perf_counter() - t2
                io["pages"] += 1
                saves.append((enc_pool.submit(timed, "encode_s", save_image_with_compression, result, out_path, config.output.jpeg_quality,
                                          config.output.png_compression), img_path, error_key))
                io["max_pending_saves"] = max(io["max_pending_saves"], len(saves))
            except Exception as e:      # noqa: BLE001 — a bad page must not stop the batch
                decodes.pop(i, None)
                if pipelined:                       # this page may have failed before the next front halves were queued
                    fronts.pop(i, None)
                    queue_fronts(i + 1 + front_workers)
                fail(img_path, error_key, e)
        while saves:
            settle(saves.popleft())
    local["io"] = io

    if multi:
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        results = merge_batch_results(gathered)
    else:
        results = merge_batch_results([local])
    dt = time.time() - t0
    if "io" in results and results["io"]["pages"]:
        n_ = results["io"]["pages"]
        results["io"].update(wall_s=dt, decode_ms_per_page=1e3 * results["io"]["decode_s"] / n_, encode_ms_per_page=1e3 * results["io"]["encode_s"] / n_,
                             process_ms_per_page=1e3 * results["io"]["process_s"] / n_, io_threads=io_threads)
    log_message(f"Batch complete: {results['success_count']}/{len(files)} images in {dt:.2f}s ({dt / len(files):.2f}s/image)", always_print=True)
    if results["failed_image_paths"] and rank == 0:
        failed_file = write_failed_paths(output_dir, results["failed_image_paths"])
        if failed_file:
            results["failed_paths_file"] = str(failed_file)
    return results


# ---- the vision half of `translate_and_render` (reference core/pipeline.py:638-1000, cleaning-only flow) ----------------------------
def resolve_pre_upscale_factor(pre_cfg, verbose: bool = False) -> float:
    """initial upscaling factor of the page (reference `_resolve_pre_upscale_factor`, :602-614): off unless `preprocessing.enabled`,
    clamped to 1..8, anything up to 1.01 counts as off"""
    if pre_cfg is None or not getattr(pre_cfg, "enabled", False):
        return 1.0
    factor = max(1.0, min(float(getattr(pre_cfg, "factor", None) or 1.0), 8.0))
    if factor <= 1.01:
        return 1.0
    log_message(f"Initial upscaling enabled: {factor:.2f}x", verbose=verbose)
    return factor


def apply_pre_upscale_if_needed(image, config, verbose: bool = False):
    """(page, factor): the page through the RCAN upscaler (the output stage's model choice) before detection when initial upscaling is
    on (reference `_apply_pre_upscale_if_needed`, :617-635)"""
    factor = resolve_pre_upscale_factor(getattr(config, "preprocessing", None), verbose)
    if factor == 1.0:
        return image, 1.0
    from .image.image_utils import upscale_image
    model_type = getattr(config.output, "image_upscale_model", "model_lite") if hasattr(config, "output") else "model_lite"
    return upscale_image(image, factor, model_type=model_type, verbose=verbose), factor


def process_page_vision(page, config, image_path="page.png", image_format: Optional[str] = None, verbose: bool = False):
    """One page through the hot path in the reference's stage order: detect speech bubbles (+ SAM masks) -> OSB text stage (regions
    to FLUX or flat fill) -> bubble cleaning -> optional final upscale -> target mode.  `page` is the decoded PIL page already in its
    target mode (`load_page`); the result is what the reference hands to `save_image_with_compression` in `cleaning_only` mode.
    Stage failures degrade exactly as there: detection errors -> no bubbles (:804-807), cleaning errors -> the uncleaned page
    (:94-123), OSB errors -> the page as it was.  Panel detection (`use_panel_sorting`): `detect_panels` on the YOLO11-L graph (core/ml/yolo11.py); a failing loader or
    model leaves panels = None, the reference's own failure path.
    Returns `(page_out, info)` with the detections, the per-bubble cleaning records and the processing scale.
    = `process_page_vision_back(process_page_vision_front(...))`: the two halves exist so that a batch can keep two pages in flight
    (`batch_process_images(process_front=, process_back=)`)."""
    return process_page_vision_back(process_page_vision_front(page, config, image_path, image_format, verbose))


def process_page_vision_front(page, config, image_path="page.png", image_format: Optional[str] = None, verbose: bool = False) -> Dict:
    """Front half of a page: target mode, optional initial upscale, stage-memo page switch, bubble detection (+ SAM masks), panels, and the
    OSB stage's PREPARE part (outside-text detection, masks, region grouping, the FLUX / flat-fill decision) — the stages whose host share is
    large and whose GPU work is small graphs on the models' own streams.  Returns the state `process_page_vision_back` finishes."""
    import math
    from .caching import get_cache
    from .image.detection import detect_panels, detect_speech_bubbles
    from .image.image_utils import upscale_image
    from .outside_text_processor import prepare_outside_text_work
    target_mode = page.mode if page.mode in ("RGB", "RGBA") else "RGBA"
    if page.mode != target_mode:
        page = page.convert(target_mode)
    info = {"bubbles": [], "text_free_boxes": [], "cleaned": [], "processing_scale": 1.0}
    state = {"config": config, "image_path": image_path, "verbose": verbose, "target_mode": target_mode, "info": info, "done": None, "work": None, "osb_error": None}
    page, info["pre_upscale_factor"] = apply_pre_upscale_if_needed(page, config, verbose)      # :718-720, before anything looks at the page
    if getattr(config, "upscaling_only", False):
        out = page
        if config.output.upscale_final_image:
            out = upscale_image(out, config.output.image_upscale_factor, model_type=config.output.image_upscale_model, verbose=verbose)
        state["done"] = out if out.mode == target_mode else out.convert(target_mode)
        return state
    scale = math.sqrt(page.width * page.height / 1_000_000) if config.preprocessing.auto_scale else 1.0          # :765-771
    info["processing_scale"] = scale
    get_cache().set_current_image(page, verbose)       # :774 — a new page drops what the stage memo holds for the previous one
    det = config.detection
    # the panel network does not depend on the bubbles: queued now, it runs beside the page's other detectors and is collected where the
    # reference calls `detect_panels` (and meets the other front halves' panel calls in one graph replay: core/ml/detector_batch.py)
    panel_ticket = None
    if getattr(det, "use_panel_sorting", False):
        from .image.detection import submit_panels
        panel_ticket = submit_panels(page, det.panel_confidence)
    try:
        bubbles, text_free = detect_speech_bubbles(image_path, getattr(config, "yolo_model_path", None), det.confidence, verbose=verbose,
                                                   device=config.device, seg_model=det.seg_model, conjoined_detection=det.conjoined_detection,
                                                   conjoined_confidence=det.conjoined_confidence, image_override=page,
                                                   osb_enabled=config.outside_text.enabled,
                                                   osb_text_verification=det.use_osb_text_verification,
                                                   osb_text_hf_token=config.outside_text.huggingface_token,
                                                   bubble_detector_model=det.bubble_detector_model)
    except Exception as e:      # noqa: BLE001
        log_message(f"Error during detection: {e}", always_print=True)
        bubbles, text_free = [], []
    info["bubbles"], info["text_free_boxes"] = bubbles, text_free
    panels = None
    if getattr(det, "use_panel_sorting", False):                  # :804-831 — the OSB stage keeps its render boxes inside the panel
        try:
            panels = detect_panels(image_path, confidence=det.panel_confidence, device=config.device, verbose=verbose, image_override=page,
                                   **({"ticket": panel_ticket} if panel_ticket is not None else {}))       # (the reference's signature when nothing was queued early)
            log_message(f"Detected {len(panels)} panels" if panels else "No panels detected", always_print=bool(panels), verbose=verbose)
        except Exception as e:      # noqa: BLE001
            log_message(f"Panel detection failed: {e}. Using global sorting.", always_print=True)
            panels = None
    info["panels"] = panels
    try:
        state["work"] = prepare_outside_text_work(page, config, image_path, image_format, verbose=verbose, bubble_data=bubbles,
                                                  text_free_boxes=text_free, panels=panels)
    except Exception as e:      # noqa: BLE001 — raised again where `process_outside_text` would have raised it: in the back half
        state["osb_error"] = e
    state.update(page=page, scale=scale)
    return state


def process_page_vision_back(state: Dict):
    """Back half of a page: the OSB stage's FINISH part (FLUX waves / flat fills), bubble cleaning, optional final upscale, target mode —
    the GPU-bound stages.  Returns `(page_out, info)` like `process_page_vision`."""
    import numpy as np
    from PIL import Image
    from .image.cleaning import clean_speech_bubbles
    from .image.image_utils import upscale_image
    from .outside_text_processor import finish_outside_text_work
    from ..utils.exceptions import CleaningError
    info, config, verbose, target_mode = state["info"], state["config"], state["verbose"], state["target_mode"]
    if state["done"] is not None:
        return state["done"], info
    if state["osb_error"] is not None:
        raise state["osb_error"]
    page, scale, det, bubbles = state["page"], state["scale"], config.detection, info["bubbles"]
    if state["work"] is not None:
        page, _osb = finish_outside_text_work(state["work"])
    if bubbles:
        cl = config.cleaning
        try:
            cleaned_cv, info["cleaned"] = clean_speech_bubbles(
                page, getattr(config, "yolo_model_path", None), det.confidence, pre_computed_detections=bubbles, device=config.device,
                thresholding_value=cl.thresholding_value, use_otsu_threshold=cl.use_otsu_threshold, roi_shrink_px=cl.roi_shrink_px,
                verbose=verbose, processing_scale=scale, conjoined_confidence=det.conjoined_confidence,
                inpaint_colored_bubbles=cl.inpaint_colored_bubbles, bubble_detector_model=det.bubble_detector_model,
                request_coordinator=getattr(config, "request_coordinator", None))
            arr = np.asarray(cleaned_cv)
            page = Image.fromarray(np.ascontiguousarray(arr[..., [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])]))     # cv2_to_pil
        except (CleaningError, Exception) as e:      # noqa: BLE001
            log_message(f"Error during cleaning: {e}", always_print=True)
    if page.mode != target_mode:
        page = page.convert(target_mode)
    if config.output.upscale_final_image:
        page = upscale_image(page, config.output.image_upscale_factor, model_type=config.output.image_upscale_model, verbose=verbose)
        if page.mode != target_mode:
            page = page.convert(target_mode)
    return page, info


_PIL_FORMAT_BY_SUFFIX = {".png": "PNG", ".jpg": "JPEG", ".jpeg": "JPEG", ".webp": "WEBP", ".bmp": "BMP", ".tif": "TIFF", ".tiff": "TIFF", ".gif": "GIF"}


def resolve_sam_precision(config) -> str:
    """SAM-2.1 arithmetic of a batch: what `config.detection.sam_precision` pins ("fast" / "high"), otherwise "high" — for every batch since
    round 6, segment-only ones included (there the mask IS the product): weight pairs in one GEMM, the fp32 residual added in the GEMM epilogue
    and an fp32 -> 16-bit LayerNorm brought its encoder from 20.3 to 13.8 ms (11.6 for "fast"), 4.9e-5 of the mask pixels differ from the fp32
    reference instead of 1.8e-4, BASELINE config 2 runs 34.9 against 38.3 pages/s (profiles/r06_sam_frontier.json, r06_visit_f_...log)"""
    pinned = getattr(getattr(config, "detection", None), "sam_precision", None)
    return pinned if pinned in ("fast", "high") else "high"


def default_front_workers(config) -> int:
    """front halves a batch keeps in flight beside the running back half: 2 for configurations whose back half is short — no FLUX inpainting
    of outside text, no final upscale: then the page's time is its detect stage, whose graphs do not fill the chip (two pages' detect stages
    at once: config 2 33 -> 37 pages/s) — and only when the process was started with `GPU_MAX_HW_QUEUES` >= 16 (on the
    runtime's default four queues two pages' ten model streams run SLOWER than one page's five); 1 otherwise"""
    osb = getattr(config, "outside_text", None)
    heavy_back = bool(getattr(osb, "enabled", False)) or bool(getattr(getattr(config, "output", None), "upscale_final_image", False))
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    except ValueError:
        queues = 4
    return 1 if heavy_back or queues < 16 else 2


def batch_vision_images(input_dir, config, output_dir=None, preserve_structure: bool = False, io_threads: int = 2,
                        front_workers: Optional[int] = None) -> Dict:
    """The page stack of this package behind the batch harness — what the reference's `batch_translate_images` does with
    `cleaning_only` pages (core/pipeline.py:2481-2733 around `translate_and_render`, :638-1000): every image of `input_dir` through
    `process_page_vision`, page i + 1's front half (and, with `front_workers` > 1, the following pages' on further instance sets of the
    detectors and SAM) beside page i's back half; same files, order and results dict as the sequential loop."""
    verbose = bool(getattr(config, "verbose", False))
    from .ml.model_manager import get_model_manager
    manager = get_model_manager()
    previous_precision = getattr(manager, "sam_precision", "high")
    manager.sam_precision = resolve_sam_precision(config)      # for this batch only: restored below (the manager is a process-wide singleton)

    def front(page, path):
        # worker thread: whatever it still has to load lazily is read locally — collectives belong to the main thread (preload below)
        with manager.thread_local_reads():
            return process_page_vision_front(page, config, path, _PIL_FORMAT_BY_SUFFIX.get(Path(path).suffix.lower()), verbose)

    def back(state):
        return process_page_vision_back(state)[0]

    n = default_front_workers(config) if front_workers is None else max(1, int(front_workers))
    # front halves that run side by side share the panel / outside-text detectors' graph replays (core/ml/detector_batch.py; same bytes per page)
    previous_batch, manager.detector_batch = manager.detector_batch, (n if n > 1 else 1)
    try:
        return batch_process_images(input_dir, config, output_dir, preserve_structure, io_threads=io_threads, process_front=front, process_back=back,
                                    front_workers=n, preload=lambda: manager.preload_for_config(config, verbose))
    finally:
        manager.sam_precision = previous_precision
        manager.detector_batch = previous_batch
