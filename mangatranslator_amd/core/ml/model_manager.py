"""Model lifecycle for the MI355X hot path (SURVEY.md §8 rows a10 / b).

Same surface as the reference's `ModelManager` (core/ml/model_manager.py:57-1525) for the models on
the vision path: a process-wide singleton with `.device`, `.dtype`, `.models`, `.model_paths`,
`.flux_inference_lock`, `load_upscale()`, `load_upscale_lite()`, `load_sam2()`,
`load_flux_kontext_sdnq()`, `unload_*()`, `clear_cache()`, `set_hf_token()` — but what the loaders
return are libmtx_hip graph objects with the call shapes the operators use:

    upscale model   model(tensor[1,3,H,W] f32) -> tensor            (image_utils.py:369-374)
    SAM 2.1         (processor, model): processor(image, input_boxes=...), model(**inputs).pred_masks,
                    processor.post_process_masks(...)               (detection.py:494-509)

Checkpoints are read from `./models/...` exactly where the reference stores them; there is no
network code here (downloads are the reference's job).  A missing file raises ModelError, which the
reference's callers already catch and degrade on.  When several ranks are up, rank 0 reads the file
and the tensors travel to the other GPUs in ONE flat RCCL broadcast over xGMI (`broadcast_state_dict`).
"""
import threading
from enum import Enum
from pathlib import Path
from typing import Optional

import torch

from ...utils.exceptions import ModelError
from ...utils.logging import log_message
from ..device import empty_cache, get_best_device, get_best_dtype, get_device_info


class ModelType(Enum):
    UPSCALE = "upscale"
    UPSCALE_LITE = "upscale_lite"
    YOLO_SPEECH_BUBBLE = "yolo_speech_bubble"
    YOLO_SPEECH_BUBBLE_2 = "yolo_speech_bubble_2"
    SAM2 = "sam2"
    RTDETR_CONJOINED_BUBBLE = "rtdetr_conjoined_bubble"
    YOLO_OSBTEXT = "yolo_osbtext"
    YOLO_PANEL = "yolo_panel"
    FLUX_KONTEXT_SDNQ_PIPELINE = "flux_kontext_sdnq_pipeline"
    FLUX_KLEIN_9B_PIPELINE = "flux_klein_9b_pipeline"
    FLUX_KLEIN_4B_PIPELINE = "flux_klein_4b_pipeline"
    MANGA_OCR = "manga_ocr"
    PADDLE_OCR_VL = "paddle_ocr_vl"
    # members the reference's callers name (:31-54) whose backends are not this build's: the slots exist so `is_loaded` / `unload_model` on
    # them behave (never loaded), the nunchaku trio aliases the one native Kontext pipeline (`load_flux_models` below)
    SAM3 = "sam3"
    FLUX_TRANSFORMER = "flux_transformer"
    FLUX_TEXT_ENCODER = "flux_text_encoder"
    FLUX_PIPELINE = "flux_pipeline"
    SDCPP_SERVER = "sdcpp_server"


def _dist_on() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_status(error: Optional[str], src: int = 0) -> None:
    """Every rank learns whether rank `src` succeeded BEFORE anyone enters a data collective: rank `src` passes its error string
    (None = fine), the others pass None; a failure raises the same ModelError on every rank (so the reference's degrade-on-ModelError
    paths run everywhere instead of the peers hanging in a broadcast rank 0 never joins)."""
    if not _dist_on():
        if error:
            raise ModelError(error)
        return
    import torch.distributed as dist
    box = [error if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    if box[0]:
        raise ModelError(box[0])


def broadcast_state_dict(sd: Optional[dict], template: Optional[dict] = None, src: int = 0) -> dict:
    """One flat collective per dtype for a whole checkpoint.  Rank `src` passes `sd`; the others pass a `template`
    (name -> (shape, dtype) or same-shaped tensors) or the same-shaped dict.  Integer buffers travel as integers, zero-element tensors
    take no room.  No-op without an initialised process group."""
    import torch.distributed as dist
    if not _dist_on():
        return sd
    rank = dist.get_rank()
    ref = sd if rank == src else (template or sd)
    meta = {}
    for k, v in ref.items():
        if isinstance(v, torch.Tensor):
            meta[k] = (tuple(v.shape), v.dtype)
        else:
            meta[k] = (tuple(v[0]), v[1])
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    out = {}
    for dt in sorted({d for _, d in meta.values()}, key=str):
        keys = sorted(k for k, (_, d) in meta.items() if d == dt)
        sizes = [int(torch.Size(meta[k][0]).numel()) for k in keys]
        wire = torch.float32 if dt in (torch.float64,) and dev.type == "cuda" else dt
        flat = torch.empty(sum(sizes), dtype=wire, device=dev)
        if rank == src and flat.numel():
            flat.copy_(torch.cat([sd[k].detach().reshape(-1).to(wire) for k in keys]))
        if flat.numel():
            dist.broadcast(flat, src=src)
        flat = flat.cpu()
        off = 0
        for k, n in zip(keys, sizes):
            out[k] = flat[off:off + n].view(meta[k][0]).to(dt).clone()
            off += n
    return out


class _Sam2ProcessorShim:
    """The slice of `Sam2Processor` the operator uses; the heavy lifting is fused into the model."""

    def __call__(self, image, input_boxes=None, return_tensors="pt"):
        import numpy as np
        arr = np.asarray(image.convert("RGB")) if hasattr(image, "convert") else np.asarray(image)
        boxes = torch.as_tensor(input_boxes, dtype=torch.float32).reshape(1, -1, 4)
        return _Sam2Inputs(page=arr, input_boxes=boxes, original_sizes=torch.tensor([[arr.shape[0], arr.shape[1]]]))

    @staticmethod
    def post_process_masks(pred_masks, original_sizes, **kw):
        return pred_masks.resolved_masks()


class _Sam2Inputs(dict):
    def __init__(self, **kw):
        super().__init__(**kw)

    def to(self, device):
        return self


class _Sam2Outputs:
    def __init__(self, masks_u8, low_res, iou):
        self.pred_masks = _PredMasks(masks_u8, low_res)
        self.iou_scores = iou


class _PredMasks:
    """Carries the page-resolution bitmasks produced on the device; `post_process_masks` returns them in
    the reference's shape ([N,1,H,W] bool per image) without a second resize pass."""

    def __init__(self, masks_u8, low_res):
        self._m, self.low_res = masks_u8, low_res

    def resolved_masks(self):
        return [self._m.bool()[:, None]]


SAM_F16_PROBE_TOLERANCE = 0.08       # bf16's own error at Hiera-L is ~0.024 of the logit range (profiles/r04_sam_dtype_probe.json)


class _Sam2ModelShim:
    def __init__(self, hip_model, dtype):
        self.hip, self.dtype = hip_model, dtype

    def __call__(self, multimask_output=False, **inputs):
        if multimask_output:
            raise ModelError("SAM-2.1 on libmtx_hip implements multimask_output=False (the reference's call)")
        masks, low, iou, _ = self.hip.segment(inputs["page"], inputs["input_boxes"][0].numpy(), return_logits=True)
        return _Sam2Outputs(masks, low, iou)


class ModelManager:
    _instance = None
    _lock = threading.RLock()

    def __new__(cls):
        with cls._lock:
            if cls._instance is None:
                cls._instance = super().__new__(cls)
                cls._instance._initialized = False
        return cls._instance

    def __init__(self):
        with self._lock:
            if self._initialized:
                return
            self.device = get_best_device()
            self.dtype = get_best_dtype(self.device)
            self.models = {}
            model_dir = Path("./models").resolve()
            self.model_paths = {
                ModelType.UPSCALE: model_dir / "upscale" / "2x-AnimeSharpV4_RCAN.safetensors",
                ModelType.UPSCALE_LITE: model_dir / "upscale" / "2x-AnimeSharpV4_Fast_RCAN_PU.safetensors",
                ModelType.YOLO_SPEECH_BUBBLE: model_dir / "yolo" / "yolov8m_seg-speech-bubble.pt",            # the reference's own file names (:119-132);
                ModelType.YOLO_SPEECH_BUBBLE_2: model_dir / "yolo" / "manga109-segmentation-bubble.pt",       # a `.safetensors` sibling is read as well
                ModelType.SAM2: model_dir / "sam" / "sam2.1-hiera-large",
                ModelType.RTDETR_CONJOINED_BUBBLE: model_dir / "rtdetr" / "comic-text-and-bubble-detector",
                ModelType.YOLO_OSBTEXT: model_dir / "yolo" / "animetext_yolov12x.pt",
                ModelType.YOLO_PANEL: model_dir / "yolo" / "manga109_v2023.12.07_l_yolov11.pt",
                ModelType.FLUX_KONTEXT_SDNQ_PIPELINE: model_dir / "flux" / "kontext",
                ModelType.FLUX_KLEIN_4B_PIPELINE: model_dir / "flux" / "klein-4b",
                ModelType.FLUX_KLEIN_9B_PIPELINE: model_dir / "flux" / "klein-9b",
            }
            self.hf_token = None
            self.flux_hf_token = None
            self.flux_inference_lock = threading.Lock()
            self._tls = threading.local()           # .replica: which instance set of the front-half models this thread is served from
            # SAM-2.1 arithmetic.  "high" (the default): hi + lo weight pairs in the trunk / neck, fp32 residual stream and an fp32 mask decoder
            # (core/ml/sam2.py) — 5.1e-5 of a page's mask pixels differ from the fp32 reference after `> 0` (the reference keeps exactly
            # those masks, detection.py:494-510), for + 11.6 ms of GPU time per page: under 1 % of any page that is inpainted or upscaled.
            # "fast": 16-bit storage throughout (1.8e-4 of the pixels; encoder + decoder 13.0 instead of 24.6 ms) — what
            # `core.pipeline.batch_vision_images` selects for batches with neither inpainting nor upscaling behind the masks (detect +
            # segment at 37.9 instead of 27.6 pages/s, profiles/r05_bench_config2_sam_*.json), and what a caller sets here before `load_sam2`.
            self.sam_precision = "high"
            # storage type of SAM's 16-bit tensors: "auto" = f16 when the f16 and bf16 models agree on the load-time probe page (f16's logit
            # error is 8x smaller; this library's f16 conversions SATURATE at 65504, so an activation overflow on a page unlike the probe
            # would degrade masks without a NaN — ADVICE r04), "bf16" = the reference's own GPU dtype, no probe, "f16" = no probe either
            self.sam_storage = "auto"
            self._failed_reads = {}                 # checkpoint path -> the error all ranks were told (multi-rank only; see _read_safetensors_with_metadata)
            self.detector_batch = 1                 # pages per graph replay of the panel / outside-text detectors (see _maybe_batched)
            self._batchers = {}
            self._initialized = True
            log_message(f"Model Manager initialized on device: {self.device}", always_print=True)

    # ---- front-half replicas (round 4) ----------------------------------------------------------------
    # A model's plan owns ONE set of activation buffers and one stream, so two pages cannot be inside the same detector or SAM instance at
    # once.  `with manager.front_replica(i):` makes the loaders of the front-half models (FRONT_MODEL_TYPES: the four detectors and SAM)
    # hand this thread instance set i — built on first use from the same checkpoint files, < 2 GB of HBM per set — so that
    # `batch_process_images(front_workers=N)` can run N pages' detect stages at once (core/pipeline.py; what `bench.py --front-replicas`
    # measures: config 2 33 -> 37 pages/s with sixteen hardware queues).  Set 0 is the plain slot the reference's callers know; every other
    # model type ignores the setting.
    FRONT_MODEL_TYPES = frozenset({ModelType.YOLO_SPEECH_BUBBLE, ModelType.YOLO_SPEECH_BUBBLE_2, ModelType.RTDETR_CONJOINED_BUBBLE,
                                   ModelType.YOLO_OSBTEXT, ModelType.YOLO_PANEL, ModelType.SAM2})

    # checkpoint file names earlier builds of this package staged for a slot whose default name has since become the reference's own
    # (a staging directory that still holds the old panel export must keep working, not fall back to "global sorting")
    LEGACY_CHECKPOINT_NAMES = {"manga109_v2023.12.07_l_yolov11.pt": ("manga109_panel_yolo11l.safetensors", "manga109_panel_yolo11l.pt")}

    def thread_local_reads(self):
        """`with manager.thread_local_reads():` — loader calls of THIS thread read their checkpoints themselves and issue no collective.
        For worker threads of a page-sharded batch (core/pipeline.py front halves): the ranks' threads cannot be kept in one order, and a
        status / tensor broadcast entered in different orders on different ranks hangs or cross-wires.  The models a batch
        needs are loaded through the collective path on the main thread first (`preload_for_config`); what a worker still loads lazily
        afterwards (a replica set, a model the preload could not know about) comes from the filesystem the ranks of a node share."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = getattr(self._tls, "local_reads", False)
            self._tls.local_reads = True
            try:
                yield self
            finally:
                self._tls.local_reads = before
        return scope()

    def _local_reads(self) -> bool:
        return bool(getattr(self._tls, "local_reads", False))

    def preload_for_config(self, config, verbose: bool = False) -> dict:
        """Every model the page flow of `config` will ask for, loaded NOW, on the calling (main) thread, in one fixed order — so that in a
        multi-rank batch all collective loads happen before any worker thread exists and in the same order on every rank.  A model that
        cannot be loaded is reported and skipped (every rank sees the same verdict: the loaders share their status first); the page flow
        meets the same error later on its own failure path.  Returns {model name: "loaded" | error text}."""
        det = getattr(config, "detection", None)
        osb = getattr(config, "outside_text", None)
        out_cfg = getattr(config, "output", None)
        pre = getattr(config, "preprocessing", None)
        g = lambda o, n, d=None: getattr(o, n, d) if o is not None else d
        jobs = [("bubble detector", lambda: self.load_yolo_speech_bubble(getattr(config, "yolo_model_path", None) or g(det, "bubble_detector_model"), verbose=verbose))]
        if g(det, "conjoined_detection", True) or g(osb, "enabled", False):
            jobs.append(("RT-DETR secondary detector", lambda: self.load_rtdetr_conjoined_bubble(verbose=verbose)))
        if g(det, "seg_model", "sam2") == "sam2":
            jobs.append(("SAM 2.1", lambda: self.load_sam2(verbose=verbose)))
        if g(osb, "enabled", False) or g(det, "use_osb_text_verification", False):
            jobs.append(("outside-text detector", lambda: self.load_yolo_osbtext(token=g(osb, "huggingface_token", None), verbose=verbose)))
        if g(det, "use_panel_sorting", False):
            jobs.append(("panel detector", lambda: self.load_yolo_panel(verbose=verbose)))
        if g(out_cfg, "upscale_final_image", False) or g(pre, "enabled", False):
            lite = g(out_cfg, "image_upscale_model", "model_lite") == "model_lite"
            jobs.append(("upscaler", (lambda: self.load_upscale_lite(verbose=verbose)) if lite else (lambda: self.load_upscale(verbose=verbose))))
        if g(osb, "enabled", False):
            method = g(osb, "inpainting_method", "flux_klein_4b")
            if method == "flux_kontext":
                jobs.append(("FLUX.1 Kontext", lambda: self.load_flux_kontext_sdnq(verbose=verbose)))
            elif method in ("flux_klein_4b", "flux_klein_9b"):
                jobs.append((f"FLUX.2 Klein {method[-2:].upper()}", (lambda: self.load_flux_klein_9b(verbose=verbose)) if method.endswith("9b") else (lambda: self.load_flux_klein_4b(verbose=verbose))))
        report = {}
        for name, load in jobs:
            try:
                report[name] = "loaded" if load() is not None else "not staged"
            except Exception as e:      # noqa: BLE001 — the page flow has its own failure path for every one of these
                report[name] = f"{type(e).__name__}: {e}"
                log_message(f"preload: {name}: {e}", always_print=True)
        return report

    def front_replica(self, index: int):
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = getattr(self._tls, "replica", 0)
            self._tls.replica = max(0, int(index))
            try:
                yield self
            finally:
                self._tls.replica = before
        return scope()

    def current_front_replica(self) -> int:
        return getattr(self._tls, "replica", 0)

    # the 640-pixel detect-only networks that can carry several pages per graph replay (core/ml/detector_batch.py)
    BATCHED_DETECTOR_TYPES = frozenset({ModelType.YOLO_OSBTEXT, ModelType.YOLO_PANEL, ModelType.RTDETR_CONJOINED_BUBBLE})

    def _slot(self, model_type: ModelType):
        """key of `self.models` this thread's loader call uses"""
        r = self.current_front_replica()
        if self.detector_batch > 1 and model_type in self.BATCHED_DETECTOR_TYPES:
            return model_type            # every front half shares the one instance behind the batching wrapper
        return (model_type, r) if r and model_type in self.FRONT_MODEL_TYPES else model_type

    def _maybe_batched(self, slot, model):
        """`detector_batch` > 1 (set by `batch_vision_images(front_workers=N)`): the panel / outside-text detector is handed out behind ONE
        DetectorBatcher per slot — the pages whose front halves run side by side share a graph replay, each page's results are the one-page
        call's bytes.  Models that are not this package's detect-only YOLO11 / YOLO12 graphs (a test double, a segmentation head) pass through."""
        if self.detector_batch <= 1 or not hasattr(model, "_build"):
            return model
        from .detector_batch import DetectorBatcher, RTDetrBatcher
        if hasattr(model, "_build_decoder"):            # RT-DETR: backbone + encoder per batch, decoder per image
            cls = RTDetrBatcher
        elif getattr(model, "a", {}).get("nm", 1) == 0:  # detect-only YOLO11 / YOLO12 head
            cls = DetectorBatcher
        else:
            return model
        w = self._batchers.get(slot)
        if w is None or w.model is not model or w.batch != self.detector_batch:
            w = self._batchers[slot] = cls(model, self.detector_batch, peers=self.detector_batch)
        return w

    # ---- bookkeeping -------------------------------------------------------------------------------
    def is_loaded(self, model_type) -> bool:
        with self._lock:
            return self.models.get(model_type) is not None

    def set_hf_token(self, token):
        self.hf_token = token or None

    def set_flux_hf_token(self, token):
        self.flux_hf_token = token or None

    def clear_cache(self):
        empty_cache(self.device)

    def forget_failed_loads(self):
        """checkpoints reported missing / unreadable under a process group are remembered (no collective per retry); call this on every rank
        after staging one of them"""
        self._failed_reads.clear()

    def unload_model(self, model_type: ModelType, force_gc: bool = True, verbose: bool = False):
        with self._lock:
            replicas = [k for k in self.models if isinstance(k, tuple) and k[0] == model_type]      # its front-half replicas go with it
            if not self.is_loaded(model_type) and not replicas:
                return
            log_message(f"Unloading {model_type.value}...", verbose=verbose)
            self.models[model_type] = None
            for key in replicas:
                del self.models[key]
            if force_gc:
                empty_cache(self.device)

    def unload_upscale_models(self, verbose: bool = False):
        self.unload_model(ModelType.UPSCALE, force_gc=False, verbose=verbose)
        self.unload_model(ModelType.UPSCALE_LITE, force_gc=False, verbose=verbose)
        empty_cache(self.device)

    def unload_ocr_models(self, verbose: bool = False):
        """reference :1391-1396 (the OCR recognisers themselves are outside the hot path; only the OSB text detector slot lives here)"""
        self.unload_model(ModelType.YOLO_OSBTEXT, verbose=verbose)

    def unload_flux_kontext_sdnq_models(self, verbose: bool = False):
        self.unload_model(ModelType.FLUX_KONTEXT_SDNQ_PIPELINE, verbose=verbose)

    # the nunchaku / sd.cpp backend names of the reference's Kontext inpainter (core/image/inpainting.py:172-222, model_manager.py:1076-1174,
    # :1436-1452): there is ONE Kontext implementation here, so they resolve to it instead of raising AttributeError under a caller that
    # was configured for another backend
    def set_flux_residual_diff_threshold(self, threshold: float):
        """reference :1076-1082 — stored clamped to [0, 1] like there.  `FluxKontextInpainter.load_models` hands it to the pipeline
        (`FluxKontextHip.residual_diff_threshold`) for backend "nunchaku" — the reference applies its first-block cache in that loader only
        (:1159-1162) — and 0 (every step runs every block) for "sdnq" / "sdcpp"."""
        self.flux_residual_diff_threshold = max(0.0, min(1.0, float(threshold)))

    def load_flux_models(self, verbose: bool = False):
        """-> (transformer, text_encoder, pipeline) like reference :1084-1174; the native pipeline owns its transformer and works from exported
        prompt embeddings, so the first two are None — what the reference's own SDNQ branch leaves them at (inpainting.py:186-187)"""
        return None, None, self.load_flux_kontext_sdnq(verbose=verbose)

    def unload_flux_kontext_models(self, verbose: bool = False):
        self.unload_flux_kontext_sdnq_models(verbose=verbose)

    def shutdown_sdcpp_server(self, name: Optional[str] = None, verbose: bool = False):
        """no stable-diffusion.cpp server exists in this build; unload paths of the reference call this unconditionally (:1449, :1480)"""

    def shutdown_sdcpp_servers(self, verbose: bool = False):
        pass

    def load_sam3(self, token: Optional[str] = None, verbose: bool = False):
        """SAM 3 (`facebook/sam3`, gated; reference :1012-1046) is outside SURVEY §8 — fail the way a missing checkpoint does, so
        `detect_speech_bubbles(seg_model="sam3")` degrades through its ModelError path instead of an AttributeError"""
        raise ModelError("SAM 3 is not built in this package (SURVEY §8 a3 covers SAM 2.1); use seg_model='sam2'")

    def get_memory_stats(self):
        """reference :1495-1497"""
        return get_device_info(self.device)

    def print_memory_stats(self):
        """reference :1499-1510 (which indexes `stats["memory"]` and so raises KeyError on a GPU; this one prints)"""
        stats = self.get_memory_stats()
        if stats.get("memory") == "N/A":
            log_message(f"Device: {stats['device']}", always_print=True)
        else:
            log_message(f"GPU Memory - Allocated: {stats['allocated_gb']} GB, Reserved: {stats['reserved_gb']} GB", always_print=True)

    def unload_flux_klein_models(self, verbose: bool = False):
        """reference :1462-1480"""
        self.unload_model(ModelType.FLUX_KLEIN_9B_PIPELINE, force_gc=False, verbose=verbose)
        self.unload_model(ModelType.FLUX_KLEIN_4B_PIPELINE, force_gc=True, verbose=verbose)

    # ---- loaders -----------------------------------------------------------------------------------
    def _read_safetensors(self, path: Path, local: bool = False) -> dict:
        """the tensors of a checkpoint (see `_read_safetensors_with_metadata`)"""
        return self._read_safetensors_with_metadata(path, local)[0]

    def _read_safetensors_with_metadata(self, path: Path, local: bool = False):
        """(tensors, header metadata): rank 0 reads, then (with several ranks) a status broadcast, a shape / dtype / metadata broadcast
        and one flat broadcast per dtype; a missing or unreadable file raises ModelError on EVERY rank.  The metadata travels WITH the
        state dict it describes — nothing is remembered on the manager between reads.
        `local`: this rank reads for itself and no collective is issued — front-half replicas (`front_replica`) are built lazily from
        worker threads, where the ranks' calls cannot be kept in one order; their checkpoint has been read once through the collective
        path already (set 0), the ranks of a node share the filesystem."""
        import torch.distributed as dist
        local = local or self._local_reads()
        if not local and _dist_on():
            # a checkpoint every rank was already told is missing / unreadable: say so again without another collective (callers such as the
            # OSB stage ask for an unstaged optional model on EVERY page; a status broadcast per page and rank is a sync point the page loop
            # does not need).  Staging the file later takes `forget_failed_loads()`.
            known = self._failed_reads.get(str(path))
            if known is not None:
                raise ModelError(known)
        rank0 = local or not _dist_on() or dist.get_rank() == 0
        sd, error, metadata = None, None, {}
        if rank0:
            # detector checkpoints: the ultralytics `.pt` the reference downloads (read without ultralytics and without executing the
            # pickle: core/ml/ultralytics_pt.py) or its safetensors export (tools/export_ultralytics_state_dict.py), whichever is staged;
            # then the names earlier builds staged for the slot
            candidates = [path, path.with_suffix(".safetensors"), path.with_suffix(".pt")] + [path.with_name(n) for n in self.LEGACY_CHECKPOINT_NAMES.get(path.name, ())]
            found = next((c for c in candidates if c.exists()), None)
            if found is not None and found.name in self.LEGACY_CHECKPOINT_NAMES.get(path.name, ()):
                log_message(f"{path.name} is not staged; reading the earlier export {found.name} from the same directory", always_print=True)
            if found is None:
                error = f"checkpoint not found: {path} (stage it under ./models; this build never downloads)"
            else:
                try:
                    with open(found, "rb") as fh:
                        magic = fh.read(2)
                    if magic == b"PK":                      # torch's zip container
                        from .ultralytics_pt import read_ultralytics_pt
                        sd, metadata = read_ultralytics_pt(found)
                    else:
                        from safetensors import safe_open
                        with safe_open(str(found), framework="pt", device="cpu") as f:
                            sd = {k: f.get_tensor(k) for k in f.keys()}
                            metadata = dict(f.metadata() or {})
                except Exception as e:                      # truncated / foreign file
                    error = f"cannot read {found}: {e}"
        if local:
            if error:
                raise ModelError(error)
            return sd, metadata
        try:
            broadcast_status(error)
        except ModelError as e:
            if _dist_on():
                self._failed_reads[str(path)] = str(e)
            raise
        if _dist_on():
            meta = [{k: (tuple(v.shape), v.dtype) for k, v in sd.items()}, metadata] if rank0 else [None, None]
            dist.broadcast_object_list(meta, src=0)
            metadata = meta[1] or {}
            sd = broadcast_state_dict(sd, template=meta[0])
        return sd, metadata

    def _prompt_embeds_ready(self, root: Path, pipeline: str) -> bool:
        """rank 0's verdict, shared: `root / prompt_embeds.safetensors` exists — after encoding the fixed prompt ONCE from the snapshot's
        staged text encoder(s) if the file was absent (core/ml/prompt_embeds.py: the reference's first-use encode, inpainting.py:846-873 /
        :1110-1124, moved to load time; the encoders are dropped again)"""
        import torch.distributed as dist
        rank0 = not _dist_on() or dist.get_rank() == 0 or self._local_reads()
        have = [False]
        if rank0:
            from .prompt_embeds import ensure_prompt_embeds
            have = [bool(ensure_prompt_embeds(root, pipeline, device=self.device, log=lambda msg: log_message(msg, always_print=True)))]
        if _dist_on() and not self._local_reads():
            dist.broadcast_object_list(have, src=0)
        return bool(have[0])

    def _detector_from_state_dict(self, sd: dict, default_names: Optional[dict] = None, metadata: Optional[dict] = None):
        """ultralytics detector state dict (from the `.pt` itself or its safetensors export) -> the graph of its family: YOLOv8-seg
        (`YoloSegHip`) or YOLO11 / YOLO11-seg / YOLO12 (`Yolo11Hip`), told apart by the blocks the state dict holds.  Class names come
        from the `names` entry of the checkpoint's own header `metadata` (what ultralytics keeps in the .pt)."""
        import ast
        names = None
        raw = (metadata or {}).get("names")
        if raw:
            try:
                names = {int(k): str(v) for k, v in dict(ast.literal_eval(raw)).items()}
            except (ValueError, SyntaxError):
                names = None
        names = names or default_names
        if "model.10.m.0.attn.qkv.conv.weight" in sd or "model.6.m.0.0.attn.qkv.conv.weight" in sd:
            from .yolo11 import Yolo11Hip
            return Yolo11Hip(sd, device=self.device, names=names)
        from .yolo import YoloSegHip
        return YoloSegHip(sd, device=self.device, names=names)

    def _staged(self, path: Path, what: str) -> None:
        """filesystem check done by rank 0 only, verdict shared: every rank raises or none does"""
        import torch.distributed as dist
        missing = f"{what} not found: {path} (stage it under ./models; this build never downloads)"
        if self._local_reads():              # a worker thread of a page-sharded batch: no collective (see `thread_local_reads`)
            if not path.exists():
                raise ModelError(missing)
            return
        rank0 = not _dist_on() or dist.get_rank() == 0
        broadcast_status(missing if rank0 and not path.exists() else None)

    def _load_rcan(self, model_type: ModelType, verbose: bool):
        with self._lock:
            if self.is_loaded(model_type):
                return self.models[model_type]
            from .rcan import RCANUpscaler
            sd = self._read_safetensors(self.model_paths[model_type])
            model = RCANUpscaler(sd, device=self.device)
            self.models[model_type] = model
            log_message(f"Upscale model loaded ({model_type.value}).", verbose=verbose)
            return model

    def load_upscale(self, verbose: bool = False):
        return self._load_rcan(ModelType.UPSCALE, verbose)

    def load_upscale_lite(self, verbose: bool = False):
        return self._load_rcan(ModelType.UPSCALE_LITE, verbose)

    def _resolve_speech_bubble_model(self, model_path):
        """(slot, checkpoint path) for what the caller names.  The reference names the detector by checkpoint path (None = the first
        model, the second model's own path = the second, any other path = a custom checkpoint in the first slot; :702-709); this
        build's operators may also pass the detector's short name, "yolo_1" / "yolo_2"."""
        if model_path in ("yolo_1", "yolo_2"):
            mt = ModelType.YOLO_SPEECH_BUBBLE_2 if model_path == "yolo_2" else ModelType.YOLO_SPEECH_BUBBLE
            return mt, self.model_paths[mt]
        if model_path is None:
            return ModelType.YOLO_SPEECH_BUBBLE, self.model_paths[ModelType.YOLO_SPEECH_BUBBLE]
        if Path(model_path) == self.model_paths[ModelType.YOLO_SPEECH_BUBBLE_2]:
            return ModelType.YOLO_SPEECH_BUBBLE_2, self.model_paths[ModelType.YOLO_SPEECH_BUBBLE_2]
        return ModelType.YOLO_SPEECH_BUBBLE, Path(model_path)

    def load_yolo_speech_bubble(self, model_path: Optional[str] = None, verbose: bool = False):
        """YOLO-seg bubble detector as a libmtx_hip graph with the ultralytics call shape
        (reference :711-743; same `model_path` meaning, see `_resolve_speech_bubble_model`).  The checkpoint is the reference's own
        ultralytics `.pt` (read pickle-free, core/ml/ultralytics_pt.py) or its safetensors export (tools/export_ultralytics_state_dict.py)."""
        with self._lock:
            mt, path = self._resolve_speech_bubble_model(model_path)
            slot = self._slot(mt)
            if self.is_loaded(slot):
                return self.models[slot]
            sd, md = self._read_safetensors_with_metadata(path, local=slot is not mt)
            model = self._detector_from_state_dict(sd, {0: "speech_bubble"}, md)        # yolo_1 is a YOLOv8m-seg, the default yolo_2 a YOLO11-seg
            self.models[slot] = model
            log_message(f"YOLO bubble detector loaded ({mt.value}).", verbose=verbose)
            return model

    def unload_all(self, verbose: bool = False):
        """every slot emptied, device cache released (reference :1482-1493)"""
        log_message("Unloading all models...", verbose=verbose)
        with self._lock:
            for model_type in list(self.models):
                if isinstance(model_type, tuple):
                    del self.models[model_type]
                else:
                    self.models[model_type] = None
        self.clear_cache()
        log_message("All models unloaded.", verbose=verbose)

    def load_yolo_osbtext(self, token: Optional[str] = None, verbose: bool = False):
        """OSB text detector (AnimeText YOLO12x, reference :780-808) as a libmtx_hip graph (core/ml/yolo11.py).  A checkpoint that is not
        staged raises ModelError, and the callers degrade exactly as the reference does when the gated checkpoint cannot be fetched —
        `detect_outside_text` falls back to the secondary detector's text_free boxes (ocr_detection.py:447-468) and
        `detect_speech_bubbles` skips OSB text verification (detection.py:196-198)."""
        with self._lock:
            slot = self._slot(ModelType.YOLO_OSBTEXT)
            if self.is_loaded(slot):
                return self._maybe_batched(slot, self.models[slot])
            try:
                sd, md = self._read_safetensors_with_metadata(self.model_paths[ModelType.YOLO_OSBTEXT], local=slot is not ModelType.YOLO_OSBTEXT)
                model = self._detector_from_state_dict(sd, {0: "text"}, md)
            except ModelError:
                raise
            except Exception as e:
                raise ModelError(f"Failed to load OSB Text model: {e}") from e
            self.models[slot] = model
            log_message("OSB text detector loaded.", verbose=verbose)
            return self._maybe_batched(slot, model)

    def load_yolo_panel(self, verbose: bool = False):
        """Panel detector (YOLO11-L, reference :810-838) as a libmtx_hip graph (core/ml/yolo11.py); ModelError when the checkpoint is not
        staged, which `detect_panels` passes on and the page flow turns into "Panel detection failed ... Using global sorting"."""
        with self._lock:
            slot = self._slot(ModelType.YOLO_PANEL)
            if self.is_loaded(slot):
                return self._maybe_batched(slot, self.models[slot])
            try:
                sd, md = self._read_safetensors_with_metadata(self.model_paths[ModelType.YOLO_PANEL], local=slot is not ModelType.YOLO_PANEL)
                model = self._detector_from_state_dict(sd, {0: "frame"}, md)
            except ModelError:
                raise
            except Exception as e:
                raise ModelError(f"Failed to load panel detection model: {e}") from e
            self.models[slot] = model
            log_message("Panel detector loaded.", verbose=verbose)
            return self._maybe_batched(slot, model)

    def get_manga_ocr(self, verbose: bool = False):
        """manga-ocr recogniser slot (reference :856-904).  The OCR side is outside the MI355X hot path: whatever recogniser object a
        deployment put in the slot is handed out, otherwise ModelError — which `extract_text_with_manga_ocr` turns into the reference's
        "[OCR FAILED]" markers."""
        with self._lock:
            if self.is_loaded(ModelType.MANGA_OCR):
                return self.models[ModelType.MANGA_OCR]
            raise ModelError("manga-ocr is not part of this build (OCR side); load the reference's recogniser into ModelType.MANGA_OCR")

    def get_paddle_ocr_vl(self, verbose: bool = False):
        """(processor, model) slot of PaddleOCR-VL (reference :927-980); see `get_manga_ocr`"""
        with self._lock:
            if self.is_loaded(ModelType.PADDLE_OCR_VL):
                return self.models[ModelType.PADDLE_OCR_VL]
            raise ModelError("PaddleOCR-VL is not part of this build (OCR side); load the reference's pair into ModelType.PADDLE_OCR_VL")

    def load_rtdetr_conjoined_bubble(self, verbose: bool = False):
        """RT-DETR-v2 secondary detector as libmtx_hip graphs with the YOLO-shaped call of the reference's adapter
        (reference :745-778; core/ml/rtdetr_adapter.py).  Expects the HF repo layout (config.json + model.safetensors)."""
        mt = ModelType.RTDETR_CONJOINED_BUBBLE
        with self._lock:
            slot = self._slot(mt)
            if self.is_loaded(slot):
                return self._maybe_batched(slot, self.models[slot])
            from .rtdetr import RTDetrHip
            root = self.model_paths[mt]
            if slot is mt:
                self._staged(root / "config.json", "RT-DETR config")
            try:
                from transformers import RTDetrV2Config
                config = RTDetrV2Config.from_pretrained(str(root))
                sd = self._read_safetensors(root / "model.safetensors", local=slot is not mt)
                model = RTDetrHip(sd, config, device=self.device, names=getattr(config, "id2label", None))
            except ModelError:
                raise
            except Exception as e:
                raise ModelError(f"Failed to load RT-DETR conjoined model: {e}") from e
            self.models[slot] = model
            log_message("RT-DETR conjoined bubble model loaded.", verbose=verbose)
            return self._maybe_batched(slot, model)

    def load_sam2(self, verbose: bool = False):
        """-> (processor, model) like the reference (:982-1010), backed by the HIP graph."""
        with self._lock:
            slot = self._slot(ModelType.SAM2)
            if self.is_loaded(slot):
                loaded = self.models[slot][1].hip
                if slot is not ModelType.SAM2 or loaded.precision == getattr(self, "sam_precision", "high"):
                    return self.models[slot]
                self.unload_model(ModelType.SAM2, force_gc=False, verbose=verbose)      # `sam_precision` changed since the load: build the other arithmetic
            from .sam2 import Sam2Hip
            root = self.model_paths[ModelType.SAM2]
            weights, cfg = root / "model.safetensors", root / "config.json"
            if slot is ModelType.SAM2:
                self._staged(cfg, "SAM-2.1 config")
            from transformers import Sam2Config
            config = Sam2Config.from_pretrained(str(root))
            sd = self._read_safetensors(weights, local=slot is not ModelType.SAM2)
            want = getattr(self, "sam_precision", "high")
            if want not in ("fast", "high"):
                raise ModelError(f"ModelManager.sam_precision must be 'fast' or 'high', not {want!r}")
            if slot is not ModelType.SAM2 and self.is_loaded(ModelType.SAM2):
                # a replica takes the storage type set 0 settled on (its probe compared the two): one more model, no second probe
                first = self.models[ModelType.SAM2][1].hip
                hip = Sam2Hip(sd, config, device=self.device, dtype=first.dtype, precision=first.precision)
                self.models[slot] = (_Sam2ProcessorShim(), _Sam2ModelShim(hip, self.dtype))
                return self.models[slot]
            # f16 storage (8x smaller logit error than bf16 against the fp32 reference: the `> 0` masks are what the page flow keeps)
            # unless the checkpoint leaves the f16 range: then the two models disagree grossly on the load-time probe and bf16 — the
            # reference's own GPU dtype — is kept
            from ...hip import abi
            storage = getattr(self, "sam_storage", "auto")
            if storage not in ("auto", "bf16", "f16"):
                raise ModelError(f"ModelManager.sam_storage must be 'auto', 'bf16' or 'f16', not {storage!r}")
            if storage == "auto":
                hip = Sam2Hip(sd, config, device=self.device, dtype=abi.BF16)
                hip16 = Sam2Hip(sd, config, device=self.device, dtype=abi.F16)
                ref, got = hip.probe_logits(), hip16.probe_logits()
                gap = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item() if torch.isfinite(got).all() else float("inf")
                if gap < SAM_F16_PROBE_TOLERANCE:
                    hip, why = hip16, f"f16 and bf16 agree on the probe page within {gap:.3f} of the logit range"
                else:
                    why = f"f16 disagrees with bf16 on the probe page by {gap:.2f} of the logit range"
                del hip16
            else:
                hip, why = Sam2Hip(sd, config, device=self.device, dtype=abi.F16 if storage == "f16" else abi.BF16), "ModelManager.sam_storage"
            if want == "high":
                hip = Sam2Hip(sd, config, device=self.device, dtype=hip.dtype, precision="high")
            self.models[slot] = (_Sam2ProcessorShim(), _Sam2ModelShim(hip, self.dtype))
            log_message(f"SAM 2.1 model loaded: {'f16' if hip.dtype == abi.F16 else 'bf16'} storage ({why}), precision {want!r}.", always_print=True)
            return self.models[slot]

    def load_flux_kontext_sdnq(self, low_vram: bool = False, verbose: bool = False):
        """FLUX.1-Kontext as libmtx_hip graphs behind the diffusers call shape (reference :1176-1252).

        Expected under models/flux/kontext (diffusers layout, bf16 — 24 GB of transformer weights are
        resident in the 288 GB of HBM, so the reference's SDNQ uint4 packing and cpu-offload shuffling are
        not used):  transformer/*.safetensors, vae/*.safetensors, and prompt_embeds.safetensors holding
        `prompt_embeds [512, 4096]` / `pooled_prompt_embeds [768]` of the fixed prompt "Remove all text."
        (exported once with the T5 / CLIP encoders; the text encoders are not on the per-page path —
        the reference caches the same two tensors, inpainting.py:846-873).
        Returns None when nothing is staged: the inpainter then skips, like the reference's
        "pipeline not available" branch."""
        mt = ModelType.FLUX_KONTEXT_SDNQ_PIPELINE
        with self._lock:
            if self.is_loaded(mt):
                return self.models[mt]
            root = self.model_paths[mt]
            if not (root / "transformer").is_dir():
                return None
            from .flux import (KONTEXT_DIT_CFG, KONTEXT_VAE_CFG, FluxDiTHip, FluxKontextHip, FluxVAEHip,
                               dit_param_shapes, vae_param_shapes)
            dcfg, vcfg = dict(KONTEXT_DIT_CFG), dict(KONTEXT_VAE_CFG)
            import json
            if (root / "transformer" / "config.json").exists():     # diffusers FluxTransformer2DModel config
                c = json.loads((root / "transformer" / "config.json").read_text())
                dcfg.update(d=c["num_attention_heads"] * c["attention_head_dim"], heads=c["num_attention_heads"], layers=c["num_layers"],
                            single_layers=c["num_single_layers"], in_channels=c["in_channels"], joint_dim=c["joint_attention_dim"],
                            pooled_dim=c["pooled_projection_dim"], axes_dim=tuple(c["axes_dims_rope"]))
            if (root / "vae" / "config.json").exists():             # diffusers AutoencoderKL config
                c = json.loads((root / "vae" / "config.json").read_text())
                vcfg.update(ch=tuple(c["block_out_channels"]), groups=c["norm_num_groups"], scaling_factor=c["scaling_factor"], shift_factor=c["shift_factor"])
            dit = FluxDiTHip(_ShardedProvider(root / "transformer", dit_param_shapes(dcfg), self.device), dcfg, self.device)
            vae = FluxVAEHip(_ShardedProvider(root / "vae", vae_param_shapes(vcfg), self.device), vcfg, self.device)
            pipe = FluxKontextHip(dit, vae)
            emb = root / "prompt_embeds.safetensors"
            if self._prompt_embeds_ready(root, "kontext"):
                e = self._read_safetensors(emb)
                pipe.set_prompt_embeds(e["prompt_embeds"], e["pooled_prompt_embeds"])
            self.models[mt] = pipe
            log_message("Flux Kontext pipeline loaded (libmtx_hip graphs, bf16).", verbose=verbose)
            return pipe


    # FLUX.2-Klein (the reference's default inpainter) ------------------------------------------------------------
    flux_klein_fp8 = True      # block linears on the MX fp8 matrix path (BASELINE.json config 5); False = all bf16

    def _load_flux_klein(self, model_type: ModelType, variant: str, low_vram: bool = False, verbose: bool = False):
        """FLUX.2-Klein as libmtx_hip graphs behind the diffusers call shape (reference :1254-1337, which loads Disty0's SDNQ 4-bit
        pack of the same network and turns on its quantised matmul — this build's low-precision route is OCP fp8 on the CDNA4 scaled
        matrix instructions instead, quantised here from bf16 weights).

        Expected under models/flux/klein-{4b,9b} (diffusers layout, bf16): transformer/*.safetensors (+ config.json),
        vae/*.safetensors (+ config.json), and prompt_embeds.safetensors holding `prompt_embeds [512, joint_dim]` of
        FluxKleinInpainter.KLEIN_PROMPT (exported once with the Qwen3 text encoder, which is not on the per-page path — the reference
        caches the same tensor, inpainting.py:1110-1124).  Returns None when nothing is staged: the inpainter then skips, like the
        reference's "pipeline unavailable" branch."""
        with self._lock:
            if self.is_loaded(model_type):
                return self.models[model_type]
            root = self.model_paths[model_type]
            import torch.distributed as dist
            rank0 = not _dist_on() or dist.get_rank() == 0
            staged = [bool(rank0 and (root / "transformer").is_dir())]
            if _dist_on():
                dist.broadcast_object_list(staged, src=0)
            if not staged[0]:
                return None
            log_message(f"Loading Flux.2 Klein {variant.upper()} model...", verbose=verbose)
            import json
            from . import flux2
            dcfg = dict(flux2.KLEIN_9B_DIT_CFG if variant == "9b" else flux2.KLEIN_4B_DIT_CFG)
            vcfg = dict(flux2.KLEIN_VAE_CFG)
            cfgs = [None, None]
            if rank0:
                for i, sub in enumerate(("transformer", "vae")):
                    f = root / sub / "config.json"
                    cfgs[i] = json.loads(f.read_text()) if f.exists() else None
            if _dist_on():
                dist.broadcast_object_list(cfgs, src=0)
            if cfgs[0]:                                           # diffusers Flux2Transformer2DModel config
                c = cfgs[0]
                dcfg.update(d=c["num_attention_heads"] * c["attention_head_dim"], heads=c["num_attention_heads"], layers=c["num_layers"],
                            single_layers=c["num_single_layers"], in_channels=c["in_channels"], joint_dim=c["joint_attention_dim"],
                            mlp_ratio=c.get("mlp_ratio", 3.0), axes_dim=tuple(c["axes_dims_rope"]), rope_theta=c.get("rope_theta", 2000.0),
                            guidance_embeds=c.get("guidance_embeds", False))
            if cfgs[1]:                                           # diffusers AutoencoderKLFlux2 config
                c = cfgs[1]
                vcfg.update(ch=tuple(c["block_out_channels"]), groups=c["norm_num_groups"], latent=c["latent_channels"],
                            bn_eps=c.get("batch_norm_eps", 1e-4))
            try:
                dit = flux2.Flux2DiTHip(_ShardedProvider(root / "transformer", flux2.dit_param_shapes(dcfg), self.device), dcfg, self.device,
                                        fp8=self.flux_klein_fp8)
                vae = flux2.Flux2VAEHip(_ShardedProvider(root / "vae", flux2.vae_param_shapes(vcfg), self.device, keep_dtype=("bn.",)), vcfg, self.device)
            except ModelError:
                raise
            except Exception as e:
                raise ModelError(f"Failed to load Flux.2 Klein {variant.upper()} model: {e}") from e
            pipe = flux2.Flux2KleinHip(dit, vae)
            emb = root / "prompt_embeds.safetensors"
            if self._prompt_embeds_ready(root, "klein"):
                pipe.set_prompt_embeds(self._read_safetensors(emb)["prompt_embeds"])
            self.models[model_type] = pipe
            log_message(f"Flux.2 Klein {variant.upper()} model loaded successfully (libmtx_hip graphs, {'fp8 + ' if self.flux_klein_fp8 else ''}bf16).", verbose=verbose)
            return pipe

    def load_flux_klein_9b(self, low_vram: bool = False, verbose: bool = False):
        return self._load_flux_klein(ModelType.FLUX_KLEIN_9B_PIPELINE, "9b", low_vram=low_vram, verbose=verbose)

    def load_flux_klein_4b(self, low_vram: bool = False, verbose: bool = False):
        return self._load_flux_klein(ModelType.FLUX_KLEIN_4B_PIPELINE, "4b", low_vram=low_vram, verbose=verbose)


class _ShardedProvider:
    """name -> tensor over the *.safetensors shards of one diffusers sub-folder.  Rank 0 reads; with more
    than one rank the tensors are handed to the others in flat RCCL broadcasts of up to 1 GiB (start-up only).  What rank 0 finds wrong (no shards,
    missing or mis-shaped parameters) is shared through `broadcast_status` first, so every rank raises the same ModelError."""

    def __init__(self, folder: Path, shapes: dict, device, keep_dtype=()):
        import torch.distributed as dist
        self.shapes, self.device, self.keep_dtype = shapes, device, tuple(keep_dtype)
        self.multi = _dist_on()
        self.rank0 = not self.multi or dist.get_rank() == 0
        self.where = {}
        self.sdnq = None
        error = None
        if self.rank0:
            from safetensors import safe_open
            files = sorted(folder.glob("*.safetensors"))
            if not files:
                error = f"no safetensors shards under {folder}"
            else:
                try:
                    self.handles = [safe_open(str(f), framework="pt", device="cpu") for f in files]
                    for h in self.handles:
                        for k in h.keys():
                            self.where[k] = h
                    # SDNQ-packed parameters (the reference's Disty0/*-SDNQ-* repositories: `<base>.weight` in packed uint8 beside
                    # `<base>.scale` [+ zero_point, svd_up, svd_down]) are expanded to bf16 at load by core/ml/sdnq.py
                    packed = {k for k in shapes if k.endswith(".weight") and k[:-len(".weight")] + ".scale" in self.where}
                    if packed:
                        from .sdnq import SdnqReader
                        self.sdnq = SdnqReader(folder)
                    missing = [k for k in shapes if k not in self.where]
                    bad = [k for k in shapes if k in self.where and k not in packed and tuple(self.where[k].get_slice(k).get_shape()) != tuple(shapes[k])]
                    if missing:
                        error = (f"{folder}: {len(missing)} parameters missing (first: {missing[0]}); a bf16 or SDNQ-packed diffusers "
                                 "checkpoint is expected — nunchaku / GGUF packed weights are not read")
                    elif bad:
                        error = f"{bad[0]}: shape {tuple(self.where[bad[0]].get_slice(bad[0]).get_shape())} != expected {tuple(shapes[bad[0]])}"
                except Exception as e:
                    error = f"cannot read the shards under {folder}: {e}"
        broadcast_status(error)
        self._ready = None
        if self.multi:
            # several ranks: the whole sub-folder travels now, in flat buckets of <= 1 GiB per dtype (24 GB of Kontext weights: ~25
            # broadcasts; tensor by tensor it was ~1 000) — the model's constructor then picks views out of the buckets
            from .flux import broadcast_in_buckets
            self._ready = broadcast_in_buckets([(n, shapes[n], self._dtype_of(n)) for n in sorted(shapes)],
                                               lambda n, v: v.copy_(self._read(n, v.dtype)), self.device)

    def _dtype_of(self, name: str):
        return torch.float32 if self.keep_dtype and name.startswith(self.keep_dtype) else torch.bfloat16

    def _read(self, name: str, dt) -> torch.Tensor:
        if self.sdnq is not None and self.sdnq.is_packed(name):
            return self.sdnq.get(name, self.shapes[name]).to(self.device, dt)
        return self.where[name].get_tensor(name).to(self.device, dt, copy=True)   # never a view of the shard's mapping (it goes away with the handle)

    def __call__(self, name: str) -> torch.Tensor:
        if self._ready is not None:
            return self._ready[name]
        return self._read(name, self._dtype_of(name))


_model_manager = None


def get_model_manager() -> ModelManager:
    global _model_manager
    if _model_manager is None:
        _model_manager = ModelManager()
    return _model_manager
