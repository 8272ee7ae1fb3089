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
from ..device import empty_cache, get_best_device, get_best_dtype


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


def broadcast_state_dict(sd: Optional[dict], template: Optional[dict] = None, src: int = 0) -> dict:
    """One flat collective for a whole checkpoint.  Rank `src` passes `sd`; the others pass a
    `template` (name -> shape) or the same-shaped dict.  No-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sd
    rank = dist.get_rank()
    shapes = {k: tuple(v.shape) for k, v in (sd if rank == src else (template or sd)).items()}
    keys = sorted(shapes)
    sizes = [max(1, int(torch.tensor(shapes[k]).prod().item())) if len(shapes[k]) else 1 for k in keys]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    if rank == src:
        flat.copy_(torch.cat([sd[k].detach().float().reshape(-1) for k in keys]))
    dist.broadcast(flat, src=src)
    flat = flat.cpu()
    out, off = {}, 0
    for k, n in zip(keys, sizes):
        out[k] = flat[off:off + n].view(shapes[k]).clone()
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
                ModelType.YOLO_SPEECH_BUBBLE: model_dir / "yolo" / "yolov8m_seg-speech-bubble.safetensors",
                ModelType.YOLO_SPEECH_BUBBLE_2: model_dir / "yolo" / "manga109-segmentation-bubble.safetensors",
                ModelType.SAM2: model_dir / "sam" / "sam2.1-hiera-large",
                ModelType.RTDETR_CONJOINED_BUBBLE: model_dir / "rtdetr" / "comic-text-and-bubble-detector",
                ModelType.YOLO_OSBTEXT: model_dir / "yolo" / "animetext_yolov12x.safetensors",
                ModelType.YOLO_PANEL: model_dir / "yolo" / "manga109_panel_yolo11l.safetensors",
                ModelType.FLUX_KONTEXT_SDNQ_PIPELINE: model_dir / "flux" / "kontext",
            }
            self.hf_token = None
            self.flux_hf_token = None
            self.flux_inference_lock = threading.Lock()
            self._initialized = True
            log_message(f"Model Manager initialized on device: {self.device}", always_print=True)

    # ---- bookkeeping -------------------------------------------------------------------------------
    def is_loaded(self, model_type: ModelType) -> bool:
        with self._lock:
            return self.models.get(model_type) is not None

    def set_hf_token(self, token):
        self.hf_token = token or None

    def set_flux_hf_token(self, token):
        self.flux_hf_token = token or None

    def clear_cache(self):
        empty_cache(self.device)

    def unload_model(self, model_type: ModelType, force_gc: bool = True, verbose: bool = False):
        with self._lock:
            if not self.is_loaded(model_type):
                return
            log_message(f"Unloading {model_type.value}...", verbose=verbose)
            self.models[model_type] = None
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

    # ---- loaders -----------------------------------------------------------------------------------
    def _read_safetensors(self, path: Path) -> dict:
        import torch.distributed as dist
        rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        if rank0 and not path.exists():
            raise ModelError(f"checkpoint not found: {path} (stage it under ./models; this build never downloads)")
        sd = None
        if rank0:
            from safetensors import safe_open
            with safe_open(str(path), framework="pt", device="cpu") as f:
                sd = {k: f.get_tensor(k) for k in f.keys()}
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            meta = [{k: tuple(v.shape) for k, v in sd.items()}] if rank0 else [None]
            dist.broadcast_object_list(meta, src=0)
            sd = broadcast_state_dict(sd, template={k: torch.empty(s) for k, s in meta[0].items()})
        return sd

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
        (reference :711-743; same `model_path` meaning, see `_resolve_speech_bubble_model`).  The checkpoint is the ultralytics state
        dict exported to safetensors (tools/export_ultralytics_state_dict.py, run once where ultralytics is installed)."""
        with self._lock:
            mt, path = self._resolve_speech_bubble_model(model_path)
            if self.is_loaded(mt):
                return self.models[mt]
            from .yolo import YoloSegHip
            sd = self._read_safetensors(path)
            model = YoloSegHip(sd, device=self.device, names={0: "speech_bubble"})
            self.models[mt] = model
            log_message(f"YOLO bubble detector loaded ({mt.value}).", verbose=verbose)
            return model

    def unload_all(self, verbose: bool = False):
        """every slot emptied, device cache released (reference :1482-1493)"""
        log_message("Unloading all models...", verbose=verbose)
        with self._lock:
            for model_type in list(self.models):
                self.models[model_type] = None
        self.clear_cache()
        log_message("All models unloaded.", verbose=verbose)

    def load_yolo_osbtext(self, token: Optional[str] = None, verbose: bool = False):
        """OSB text detector (YOLO12x "AnimeText", reference :780-808).  The YOLO12 graph (A2C2f area attention) is not built in
        this round: the loader raises ModelError, and the callers degrade exactly as the reference does when the
        gated checkpoint cannot be fetched — `detect_outside_text` falls back to the secondary detector's text_free boxes
        (ocr_detection.py:447-468) and `detect_speech_bubbles` skips OSB text verification (detection.py:196-198)."""
        with self._lock:
            if self.is_loaded(ModelType.YOLO_OSBTEXT):
                return self.models[ModelType.YOLO_OSBTEXT]
            raise ModelError("OSB text detector (YOLO12x) is not available in this build; using the text_free fallback")

    def load_yolo_panel(self, verbose: bool = False):
        """Panel detector (YOLO11-L, reference :810-838).  The YOLO11 graph (C3k2 / C2PSA blocks, depthwise head) is not built in this
        round: the loader hands out whatever object a deployment put in the slot and otherwise raises ModelError, which
        `detect_panels` turns into the reference's ModelError and the page flow into "Panel detection failed ... Using global sorting"."""
        with self._lock:
            if self.is_loaded(ModelType.YOLO_PANEL):
                return self.models[ModelType.YOLO_PANEL]
            raise ModelError("panel detector (YOLO11-L) is not available in this build")

    def load_rtdetr_conjoined_bubble(self, verbose: bool = False):
        """RT-DETR-v2 secondary detector as libmtx_hip graphs with the YOLO-shaped call of the reference's adapter
        (reference :745-778; core/ml/rtdetr_adapter.py).  Expects the HF repo layout (config.json + model.safetensors)."""
        mt = ModelType.RTDETR_CONJOINED_BUBBLE
        with self._lock:
            if self.is_loaded(mt):
                return self.models[mt]
            from .rtdetr import RTDetrHip
            root = self.model_paths[mt]
            if not (root / "config.json").exists():
                raise ModelError(f"RT-DETR config not found: {root / 'config.json'} (stage it under ./models; this build never downloads)")
            try:
                from transformers import RTDetrV2Config
                config = RTDetrV2Config.from_pretrained(str(root))
                sd = self._read_safetensors(root / "model.safetensors")
                model = RTDetrHip(sd, config, device=self.device, names=getattr(config, "id2label", None))
            except ModelError:
                raise
            except Exception as e:
                raise ModelError(f"Failed to load RT-DETR conjoined model: {e}") from e
            self.models[mt] = model
            log_message("RT-DETR conjoined bubble model loaded.", verbose=verbose)
            return model

    def load_sam2(self, verbose: bool = False):
        """-> (processor, model) like the reference (:982-1010), backed by the HIP graph."""
        with self._lock:
            if self.is_loaded(ModelType.SAM2):
                return self.models[ModelType.SAM2]
            from .sam2 import Sam2Hip
            root = self.model_paths[ModelType.SAM2]
            weights, cfg = root / "model.safetensors", root / "config.json"
            if not cfg.exists():
                raise ModelError(f"SAM-2.1 config not found: {cfg}")
            from transformers import Sam2Config
            config = Sam2Config.from_pretrained(str(root))
            sd = self._read_safetensors(weights)
            hip = Sam2Hip(sd, config, device=self.device)
            self.models[ModelType.SAM2] = (_Sam2ProcessorShim(), _Sam2ModelShim(hip, self.dtype))
            log_message("SAM 2.1 model loaded.", verbose=verbose)
            return self.models[ModelType.SAM2]

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
            if emb.exists():
                e = self._read_safetensors(emb)
                pipe.set_prompt_embeds(e["prompt_embeds"], e["pooled_prompt_embeds"])
            self.models[mt] = pipe
            log_message("Flux Kontext pipeline loaded (libmtx_hip graphs, bf16).", verbose=verbose)
            return pipe


class _ShardedProvider:
    """name -> tensor over the *.safetensors shards of one diffusers sub-folder.  Rank 0 reads; with more
    than one rank every tensor is handed to the others by an RCCL broadcast (start-up only)."""

    def __init__(self, folder: Path, shapes: dict, device):
        import torch.distributed as dist
        self.shapes, self.device = shapes, device
        self.multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank0 = not self.multi or dist.get_rank() == 0
        self.where = {}
        if self.rank0:
            from safetensors import safe_open
            files = sorted(folder.glob("*.safetensors"))
            if not files:
                raise ModelError(f"no safetensors shards under {folder}")
            self.handles = [safe_open(str(f), framework="pt", device="cpu") for f in files]
            for h in self.handles:
                for k in h.keys():
                    self.where[k] = h
            missing = [k for k in shapes if k not in self.where]
            if missing:
                raise ModelError(f"{folder}: {len(missing)} parameters missing (first: {missing[0]}); a bf16 diffusers "
                                 "checkpoint is expected — SDNQ / nunchaku / GGUF packed weights are not read")

    def __call__(self, name: str) -> torch.Tensor:
        if self.rank0:
            t = self.where[name].get_tensor(name)
            if tuple(t.shape) != tuple(self.shapes[name]):
                raise ModelError(f"{name}: shape {tuple(t.shape)} != expected {tuple(self.shapes[name])}")
            t = t.to(self.device, torch.bfloat16)
        else:
            t = torch.empty(self.shapes[name], dtype=torch.bfloat16, device=self.device)
        if self.multi:
            import torch.distributed as dist
            dist.broadcast(t, src=0)
        return t


_model_manager = None


def get_model_manager() -> ModelManager:
    global _model_manager
    if _model_manager is None:
        _model_manager = ModelManager()
    return _model_manager
