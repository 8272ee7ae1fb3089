"""Executed in its OWN process by tests/test_reference_dropin.py (sys.modules surgery must not leak into the test session).

Performs the INTEGRATION.md Option A binding under the UNMODIFIED reference at /root/reference (build container only), imports the
reference's `core` and `core.pipeline`, runs the reference's `translate_and_render(page, config)` in cleaning-only mode, and the same page
through this package's `process_page_vision`; prints one JSON line with the verdict.  Third-party wheels the reference imports at module
level but this image lacks (OpenCV, skia, ultralytics, ...) are replaced by inert stand-ins: none of them is touched on the cleaning-only
path once the hot-path modules are served by this package (that is the point of the test).  The primary detector slot holds a canned
ultralytics-shaped model; the bubble-cleaning kernels run on the CPU kernel simulator (tests/emu) in place of libmtx_hip.so.
"""
import importlib.machinery
import importlib.util
import json
import sys
import tempfile
import types
from pathlib import Path
from unittest.mock import MagicMock

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(REF))

stubbed = []
for name in ["cv2", "spandrel", "ultralytics", "oxipng", "diffusers", "sdnq", "sdnq.common", "sdnq.loader", "skia", "uharfbuzz", "manga_ocr", "pythainlp",
             "pythainlp.tokenize", "gradio", "torchvision", "torchvision.transforms", "budoux", "openai", "anthropic", "google", "google.genai"]:
    try:
        if importlib.util.find_spec(name) is not None:
            continue
    except (ImportError, ValueError, ModuleNotFoundError):
        pass
    stub = MagicMock(name=name)
    stub.__spec__ = importlib.machinery.ModuleSpec(name, None)
    stub.__path__ = []
    sys.modules[name] = stub
    stubbed.append(name)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import mangatranslator_amd.integration as amd  # noqa: E402

served = amd.install()

# the CPU kernel simulator stands in for libmtx_hip.so (test-only substitution, as in the other CPU-tier tests)
import mangatranslator_amd.hip.lib as libmod  # noqa: E402
libmod._lib = libmod._open_simulator_for_tests(ROOT / "tests" / "emu" / "libmtx_emu.so")

import core  # noqa: E402  (the REFERENCE's package: its __init__ pulls every re-export through the served modules)
import core.pipeline as ref_pipeline  # noqa: E402
from core.config import MangaTranslatorConfig  # noqa: E402
import utils.exceptions as ref_exc  # noqa: E402

import mangatranslator_amd.core.pipeline as amd_pipeline  # noqa: E402
from mangatranslator_amd.core.ml.model_manager import ModelType, get_model_manager  # noqa: E402
from mangatranslator_amd.utils import exceptions as amd_exc  # noqa: E402

report = dict(stubbed=stubbed, served=served)
report["reference_core_file"] = str(Path(core.__file__).resolve())
report["same_exception_classes"] = amd_exc.ModelError is ref_exc.ModelError
report["core_reexports"] = {n: getattr(getattr(core, n), "__module__", None) for n in
                            ("detect_speech_bubbles", "clean_speech_bubbles", "FluxKleinInpainter", "FluxKontextInpainter", "OutsideTextDetector",
                             "get_model_manager", "translate_and_render", "batch_translate_images")}
report["pipeline_binds"] = {n: getattr(ref_pipeline, n).__module__ for n in
                            ("detect_speech_bubbles", "detect_panels", "clean_speech_bubbles", "retry_cleaning_with_otsu", "upscale_image",
                             "process_outside_text", "prepare_outside_text_work", "finish_outside_text_work", "get_model_manager")}

# ---- a 256 x 384 page with two dark-on-white bubbles -------------------------------------------------------------------------------
H, W = 384, 256
yy, xx = np.mgrid[0:H, 0:W]
page = np.full((H, W, 3), 90, np.uint8)
page[(yy // 6 + xx // 6) % 2 == 0] = 140                                  # screentone background
bubbles = [(70, 90, 52, 40), (170, 270, 60, 50)]                           # cx, cy, a, b
masks = []
for cx, cy, a, b in bubbles:
    m = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0
    page[m] = 250
    masks.append(m)
    for k in range(-2, 3):                                                 # "text": dark strokes inside the bubble
        page[cy + 8 * k - 2: cy + 8 * k + 2, cx - int(a * 0.5): cx + int(a * 0.5)] = 15
tmp = Path(tempfile.mkdtemp())
path = tmp / "p001.png"
Image.fromarray(page).save(path)


class _Boxes:
    def __init__(self, xyxy, conf):
        self.xyxy = torch.tensor(xyxy, dtype=torch.float32)
        self.conf = torch.tensor(conf, dtype=torch.float32)
        self.cls = torch.zeros(len(conf))

    def __len__(self):
        return len(self.xyxy)


class _Masks:
    def __init__(self, data):
        self.data = torch.from_numpy(np.stack(data).astype(np.float32))

    def __len__(self):
        return len(self.data)


class CannedDetector:
    """ultralytics call shape: model(bgr, conf=, device=, imgsz=, retina_masks=, verbose=) -> [Results]"""
    names = {0: "speech_bubble"}
    calls = 0

    def __call__(self, img, **kw):
        CannedDetector.calls += 1
        boxes = [[cx - a, cy - b, cx + a, cy + b] for cx, cy, a, b in bubbles]
        return [types.SimpleNamespace(boxes=_Boxes(boxes, [0.93, 0.88]), masks=_Masks(masks), orig_shape=(H, W), names=self.names)]


mgr = get_model_manager()
mgr.models[ModelType.YOLO_SPEECH_BUBBLE_2] = CannedDetector()
mgr.models[ModelType.YOLO_SPEECH_BUBBLE] = mgr.models[ModelType.YOLO_SPEECH_BUBBLE_2]

cfg = MangaTranslatorConfig(yolo_model_path=None) if "yolo_model_path" in MangaTranslatorConfig.__dataclass_fields__ else MangaTranslatorConfig()
cfg.cleaning_only = True
cfg.verbose = False
cfg.device = torch.device("cpu")
cfg.detection.seg_model = "yolo"
cfg.detection.conjoined_detection = False
cfg.detection.use_osb_text_verification = False
cfg.detection.use_panel_sorting = False
cfg.outside_text.enabled = False
cfg.output.upscale_final_image = False
cfg.output.output_format = "png"
cfg.preprocessing.enabled = False
cfg.translation.send_full_page_context = False
ref_pipeline.validate_config = lambda c: None                              # provider / API-key validation is the LLM side

out_ref = ref_pipeline.translate_and_render(path, cfg, output_path=None)
calls_ref = CannedDetector.calls
page_rgba = amd_pipeline.load_page(path, "RGBA")
out_amd, info = amd_pipeline.process_page_vision(page_rgba, cfg, path)
a, b = np.asarray(out_ref), np.asarray(out_amd)
report.update(ref_mode=out_ref.mode, amd_mode=out_amd.mode, ref_size=list(out_ref.size), same_pixels=bool(a.shape == b.shape and np.array_equal(a, b)),
              changed_pixels=int((np.asarray(out_ref.convert("RGB")) != page).any(-1).sum()), detector_calls_reference=calls_ref,
              bubbles_cleaned=len(info.get("cleaned", [])))
# the text strokes are gone: inside each bubble (away from its outline) everything is the fill colour now
inner_ok = []
rgb = np.asarray(out_ref.convert("RGB"))
for (cx, cy, a_, b_) in bubbles:
    inner = ((xx - cx) / (a_ * 0.6)) ** 2 + ((yy - cy) / (b_ * 0.6)) ** 2 <= 1.0
    inner_ok.append(bool((rgb[inner] > 200).all()))
report["text_removed"] = inner_ok
print("DROPIN_REPORT " + json.dumps(report))
