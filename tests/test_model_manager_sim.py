"""CPU tier: ModelManager loaders (reference core/ml/model_manager.py surface) against staged tiny
checkpoints, with the kernel simulator standing in for libmtx_hip (test-only substitution)."""
import json

import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import save_file

import flux_checks as fc
from oracle import flux_ref as fr
from oracle.rcan_ref import load_ref, make_state_dict


@pytest.fixture()
def manager(emu_lib, tmp_path, monkeypatch):
    import mangatranslator_amd.hip.lib as libmod
    from mangatranslator_amd.core.ml import model_manager as mm
    monkeypatch.setattr(libmod, "_lib", emu_lib)
    monkeypatch.setattr(mm, "_model_manager", None)
    monkeypatch.setattr(mm.ModelManager, "_instance", None)
    m = mm.get_model_manager()
    for k in list(m.model_paths):
        rel = m.model_paths[k].relative_to(m.model_paths[k].parents[1])
        m.model_paths[k] = tmp_path / rel
    yield m
    monkeypatch.setattr(mm.ModelManager, "_instance", None)


def test_singleton_and_missing_checkpoint(manager):
    from mangatranslator_amd.core.ml.model_manager import get_model_manager
    from mangatranslator_amd.utils.exceptions import ModelError
    assert get_model_manager() is manager
    with pytest.raises(ModelError):
        manager.load_upscale()
    assert manager.load_flux_kontext_sdnq() is None      # nothing staged: inpainter skips, as the reference does


def test_other_backend_names_resolve_to_the_native_pipeline(manager):
    """callers configured for the reference's nunchaku / sd.cpp Kontext backends and for SAM 3 (inpainting.py:172-222, detection.py:1661-1666)
    meet methods, not AttributeErrors: the nunchaku trio is the one native pipeline, SAM 3 refuses with ModelError"""
    from mangatranslator_amd.core.ml.model_manager import ModelType
    from mangatranslator_amd.utils.exceptions import ModelError
    manager.set_flux_residual_diff_threshold(1.7)
    assert manager.flux_residual_diff_threshold == 1.0
    assert manager.load_flux_models() == (None, None, None)            # nothing staged: same "no pipeline" answer as the SDNQ loader
    manager.unload_flux_kontext_models(); manager.shutdown_sdcpp_server("flux_kontext"); manager.shutdown_sdcpp_servers()
    with pytest.raises(ModelError):
        manager.load_sam3(token="x")
    for name in ("SAM3", "FLUX_TRANSFORMER", "FLUX_TEXT_ENCODER", "FLUX_PIPELINE", "SDCPP_SERVER"):
        assert not manager.is_loaded(ModelType[name])
    stats = manager.get_memory_stats()
    assert stats == {"device": "cpu", "memory": "N/A"}
    manager.print_memory_stats()
    from mangatranslator_amd.core.device import is_gpu_available
    assert is_gpu_available() is torch.cuda.is_available()


def test_load_upscale_and_unload(manager):
    sd = make_state_dict(n_feats=32, n_resgroups=1, n_resblocks=1, seed=3)
    p = manager.model_paths[list(manager.model_paths)[0]]
    p.parent.mkdir(parents=True)
    save_file(sd, str(p))
    model = manager.load_upscale()
    assert manager.load_upscale() is model
    x = torch.rand(1, 3, 12, 16, generator=torch.Generator().manual_seed(0))
    assert (model(x).cpu() - load_ref(sd)(x)).abs().max() < 2e-2
    manager.unload_upscale_models()
    assert not manager.is_loaded(list(manager.model_paths)[0])


def test_load_flux_kontext_and_inpaint(manager):
    from mangatranslator_amd.core.image.inpainting import FluxKontextInpainter
    from mangatranslator_amd.core.ml.model_manager import ModelType
    t, v = fc.models(seed=4)
    root = manager.model_paths[ModelType.FLUX_KONTEXT_SDNQ_PIPELINE]
    (root / "transformer").mkdir(parents=True); (root / "vae").mkdir()
    tsd = {k: x.to(torch.bfloat16).contiguous() for k, x in t.state_dict().items()}
    keys = sorted(tsd)
    save_file({k: tsd[k] for k in keys[::2]}, str(root / "transformer" / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: tsd[k] for k in keys[1::2]}, str(root / "transformer" / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    save_file({k: x.contiguous() for k, x in v.state_dict().items()}, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    c = t.cfg
    (root / "transformer" / "config.json").write_text(json.dumps(dict(
        num_attention_heads=c["heads"], attention_head_dim=c["d"] // c["heads"], num_layers=c["layers"], num_single_layers=c["single_layers"],
        in_channels=64, joint_attention_dim=c["joint_dim"], pooled_projection_dim=c["pooled_dim"], axes_dims_rope=list(c["axes_dim"]))))
    (root / "vae" / "config.json").write_text(json.dumps(dict(block_out_channels=list(v.cfg["ch"]), norm_num_groups=v.cfg["groups"],
                                                            scaling_factor=v.cfg["scaling_factor"], shift_factor=v.cfg["shift_factor"])))
    g = torch.Generator().manual_seed(9)
    save_file({"prompt_embeds": torch.randn(8, c["joint_dim"], generator=g), "pooled_prompt_embeds": torch.randn(c["pooled_dim"], generator=g)},
              str(root / "prompt_embeds.safetensors"))
    pipe = manager.load_flux_kontext_sdnq()
    assert pipe is not None and manager.load_flux_kontext_sdnq() is pipe
    # the inpainter drives it exactly as it drives the diffusers pipeline; shrink the preferred sizes so the
    # simulator finishes quickly
    inp = FluxKontextInpainter(num_inference_steps=1, backend="sdnq")
    inp.PREFERED_KONTEXT_RESOLUTIONS = [(48, 32), (32, 48), (32, 32)]
    page = Image.fromarray((np.random.default_rng(0).random((96, 128, 3)) * 255).astype(np.uint8))
    mask = np.zeros((96, 128), bool); mask[30:50, 40:80] = True
    out = inp.inpaint_mask(page, mask, seed=1)
    a, b = np.asarray(page).astype(int), np.asarray(out).astype(int)
    assert out.size == page.size and (a != b).any()
    far = np.ones((96, 128), bool); far[10:70, 10:118] = False
    assert (a[far] == b[far]).all()           # pixels far from the mask are untouched by the composite
    assert pipe.residual_diff_threshold == 0.0                     # backend "sdnq": every step runs every block, as in the reference's SDNQ loader
    # the reference's nunchaku backend wraps the pipeline in the first-block cache (model_manager.py:1159-1162): same route here
    cached = FluxKontextInpainter(num_inference_steps=3, backend="nunchaku", residual_diff_threshold=0.4)
    cached.PREFERED_KONTEXT_RESOLUTIONS = [(48, 32), (32, 48), (32, 32)]
    out2 = cached.inpaint_mask(page, mask, seed=1)
    # the threshold travels with the call, not on the shared pipeline object: the sdnq instance above keeps running every block
    assert pipe.residual_diff_threshold == 0.0 and cached._cache_threshold == 0.4 and inp._cache_threshold == 0.0 and manager.flux_residual_diff_threshold == 0.4
    assert out2.size == page.size and len(pipe.last["skipped"]) == 3 and pipe.last["skipped"][0] is False
    out3 = inp.inpaint_mask(page, mask, seed=2)                      # the first instance again, after the cached one used the same pipeline object
    assert out3.size == page.size and pipe.last["skipped"] == []     # no cache decisions: every step ran every block
    manager.unload_flux_kontext_sdnq_models()
    assert not manager.is_loaded(ModelType.FLUX_KONTEXT_SDNQ_PIPELINE)


@pytest.mark.parametrize("fp8", [False, True])
def test_load_flux_klein_and_inpaint(manager, fp8):
    """the reference's default inpainter end to end on a staged tiny checkpoint: loader (config.json geometry, sharded weights, BatchNorm
    statistics kept in fp32, cached prompt embeddings) -> FluxKleinInpainter.inpaint_mask -> composited page"""
    import flux2_checks as f2c
    from mangatranslator_amd.core.image.inpainting import FluxKleinInpainter
    from mangatranslator_amd.core.ml.model_manager import ModelType
    assert manager.load_flux_klein_4b() is None           # nothing staged
    t, v = f2c.models(seed=4)
    root = manager.model_paths[ModelType.FLUX_KLEIN_4B_PIPELINE]
    (root / "transformer").mkdir(parents=True); (root / "vae").mkdir()
    tsd = {k: x.to(torch.bfloat16).contiguous() for k, x in t.state_dict().items()}
    keys = sorted(tsd)
    save_file({k: tsd[k] for k in keys[::2]}, str(root / "transformer" / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: tsd[k] for k in keys[1::2]}, str(root / "transformer" / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    vsd = {k: x.contiguous() for k, x in v.state_dict().items() if not k.endswith("num_batches_tracked")}
    save_file(vsd, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    c = t.cfg
    (root / "transformer" / "config.json").write_text(json.dumps(dict(
        num_attention_heads=c["heads"], attention_head_dim=c["d"] // c["heads"], num_layers=c["layers"], num_single_layers=c["single_layers"],
        in_channels=c["in_channels"], joint_attention_dim=c["joint_dim"], mlp_ratio=c["mlp_ratio"], axes_dims_rope=list(c["axes_dim"]),
        rope_theta=c["rope_theta"], guidance_embeds=False)))
    (root / "vae" / "config.json").write_text(json.dumps(dict(block_out_channels=list(v.cfg["ch"]), norm_num_groups=v.cfg["groups"],
                                                            latent_channels=v.cfg["latent"], batch_norm_eps=v.cfg["bn_eps"])))
    save_file({"prompt_embeds": torch.randn(8, c["joint_dim"], generator=torch.Generator().manual_seed(9))}, str(root / "prompt_embeds.safetensors"))
    manager.flux_klein_fp8 = fp8
    pipe = manager.load_flux_klein_4b()
    assert pipe is not None and manager.load_flux_klein_4b() is pipe
    assert bool(pipe.transformer.fp8) == fp8 and (pipe.transformer.blocks[0]["qkv"].q is not None) == fp8
    assert torch.allclose(pipe.vae.bn_mean.cpu(), v.bn.running_mean)
    inp = FluxKleinInpainter(variant="4b", num_inference_steps=1)
    inp.upscale_small_crops = False                        # keep the simulator's crop at 64 x 64
    page = Image.fromarray((np.random.default_rng(0).random((96, 128, 3)) * 255).astype(np.uint8))
    mask = np.zeros((96, 128), bool); mask[40:50, 50:70] = True
    out = inp.inpaint_mask(page, mask, seed=1)
    a, b = np.asarray(page).astype(int), np.asarray(out).astype(int)
    assert out.size == page.size and (a != b).any() and pipe.calls == 1
    far = np.ones((96, 128), bool); far[10:90, 10:118] = False
    assert (a[far] == b[far]).all()
    inp.unload_models()
    assert not manager.is_loaded(ModelType.FLUX_KLEIN_4B_PIPELINE)


def test_load_yolo11_family_detectors(manager):
    """the default bubble detector (YOLO11-seg), the panel detector (YOLO11) and the OSB text detector (YOLO12) load through the same
    state-dict reader as YOLOv8-seg and are told apart by their blocks; class names come from the file's metadata"""
    from oracle import yolo11_ref as y11
    from mangatranslator_amd.core.ml.model_manager import ModelType
    from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
    from mangatranslator_amd.utils.exceptions import ModelError
    with pytest.raises(ModelError):
        manager.load_yolo_panel()
    with pytest.raises(ModelError):
        manager.load_yolo_osbtext()
    for mt, fam, seg, names in ((ModelType.YOLO_PANEL, "11", False, {0: "body", 1: "frame"}), (ModelType.YOLO_OSBTEXT, "12", False, {0: "text"}),
                                (ModelType.YOLO_SPEECH_BUBBLE_2, "11", True, {0: "speech_bubble"})):
        net = y11.make_model(fam, "n", len(names), seg, seed=1)
        path = manager.model_paths[mt]
        path.parent.mkdir(parents=True, exist_ok=True)
        save_file({k: v.contiguous() for k, v in net.state_dict().items()}, str(path), metadata={"names": repr(names)})
    panel, osb, bubble = manager.load_yolo_panel(), manager.load_yolo_osbtext(), manager.load_yolo_speech_bubble("yolo_2")
    assert isinstance(panel, Yolo11Hip) and panel.a["family"] == "11" and not panel.a["seg"] and panel.names == {0: "body", 1: "frame"}
    assert isinstance(osb, Yolo11Hip) and osb.a["family"] == "12" and osb.names == {0: "text"}
    assert isinstance(bubble, Yolo11Hip) and bubble.a["seg"] and manager.load_yolo_panel() is panel
    page = (np.random.default_rng(0).random((96, 64, 3)) * 255).astype(np.uint8)
    res = panel(page, conf=0.0, imgsz=64, max_det=5)[0]                 # the ultralytics call shape; detect-only: boxes, no masks
    assert res.masks is None and len(res.boxes) == 5 and res.boxes.xyxy.shape == (5, 4)
    res = bubble(page, conf=0.0, imgsz=64, max_det=3)[0]
    assert len(res.masks) == 3 and tuple(res.masks.data.shape[1:]) == (96, 64)


def test_front_replicas_are_separate_instances_of_the_same_checkpoint(manager):
    """`with manager.front_replica(i)`: the front-half loaders (detectors, SAM) hand the thread instance set i — another model object built
    from the same file, same outputs — while every other model type, and every other thread, keeps the plain slot; unloading a type takes
    its replicas with it (core/pipeline.py batch_process_images(front_workers=N) is the caller)"""
    import threading
    from oracle import yolo11_ref as y11
    from oracle.rcan_ref import make_state_dict
    from mangatranslator_amd.core.ml.model_manager import ModelType
    names = {0: "body", 1: "frame"}
    net = y11.make_model("11", "n", len(names), False, seed=1)
    path = manager.model_paths[ModelType.YOLO_PANEL]
    path.parent.mkdir(parents=True, exist_ok=True)
    save_file({k: v.contiguous() for k, v in net.state_dict().items()}, str(path), metadata={"names": repr(names)})
    up = manager.model_paths[ModelType.UPSCALE]
    up.parent.mkdir(parents=True, exist_ok=True)
    save_file(make_state_dict(n_feats=32, n_resgroups=1, n_resblocks=1, seed=3), str(up))
    panel0 = manager.load_yolo_panel()
    with manager.front_replica(1):
        assert manager.current_front_replica() == 1
        panel1 = manager.load_yolo_panel()
        assert manager.load_yolo_panel() is panel1 and panel1 is not panel0
        assert manager.load_upscale() is manager.load_upscale()                     # not a front-half model: one instance for every thread
        rcan = manager.load_upscale()
        seen = []
        th = threading.Thread(target=lambda: seen.append(manager.load_yolo_panel()))    # the setting is the calling thread's only
        th.start(); th.join()
        assert seen == [panel0]
        with manager.front_replica(0):
            assert manager.load_yolo_panel() is panel0
        assert manager.current_front_replica() == 1
    assert manager.current_front_replica() == 0 and manager.load_yolo_panel() is panel0 and manager.load_upscale() is rcan
    assert (ModelType.YOLO_PANEL, 1) in manager.models and (ModelType.UPSCALE, 1) not in manager.models
    page = (np.random.default_rng(0).random((96, 64, 3)) * 255).astype(np.uint8)
    r0, r1 = panel0(page, conf=0.0, imgsz=64, max_det=5)[0], panel1(page, conf=0.0, imgsz=64, max_det=5)[0]
    assert torch.equal(r0.boxes.xyxy.cpu(), r1.boxes.xyxy.cpu()) and torch.equal(r0.boxes.conf.cpu(), r1.boxes.conf.cpu())
    manager.unload_model(ModelType.YOLO_PANEL)
    assert not manager.is_loaded(ModelType.YOLO_PANEL) and (ModelType.YOLO_PANEL, 1) not in manager.models
    with manager.front_replica(1):
        assert manager.load_yolo_panel() is not panel1                               # rebuilt on demand
    manager.unload_all()
    assert all(not isinstance(k, tuple) for k in manager.models)


def test_preload_order_worker_thread_reads_and_legacy_names(manager, monkeypatch):
    """(1) `preload_for_config` asks the loaders in one fixed order whatever fails on the way; (2) inside `thread_local_reads` a
    loader reads its checkpoint itself — no status / tensor broadcast, even with a process group up (the order of worker threads' calls is
    not the same on every rank); (3) a staging directory that still holds the panel detector under its earlier export name is read."""
    import types
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.utils.exceptions import ModelError
    calls = []

    def fake(name, fail=False):
        def load(*a, **k):
            calls.append(name)
            if fail:
                raise ModelError(name + " not staged")
            return object()
        return load
    for attr, fail in (("load_yolo_speech_bubble", False), ("load_rtdetr_conjoined_bubble", True), ("load_sam2", False), ("load_yolo_osbtext", True),
                       ("load_yolo_panel", False), ("load_upscale", False), ("load_flux_klein_4b", False)):
        monkeypatch.setattr(manager, attr, fake(attr, fail))
    cfg = types.SimpleNamespace(
        yolo_model_path=None, detection=types.SimpleNamespace(bubble_detector_model="yolo_2", conjoined_detection=True, seg_model="sam2", use_osb_text_verification=False, use_panel_sorting=True),
        outside_text=types.SimpleNamespace(enabled=True, huggingface_token="", inpainting_method="flux_klein_4b"),
        output=types.SimpleNamespace(upscale_final_image=True, image_upscale_model="model"), preprocessing=types.SimpleNamespace(enabled=False))
    report = manager.preload_for_config(cfg)
    assert calls == ["load_yolo_speech_bubble", "load_rtdetr_conjoined_bubble", "load_sam2", "load_yolo_osbtext", "load_yolo_panel", "load_upscale", "load_flux_klein_4b"]
    assert report["SAM 2.1"] == "loaded" and report["RT-DETR secondary detector"].startswith("ModelError")
    monkeypatch.undo()

    # (2) + (3): the panel detector's earlier export, read from a worker-thread scope while "a process group is up"
    from mangatranslator_amd.core.ml import model_manager as mm2
    path = manager.model_paths[mm2.ModelType.YOLO_PANEL]
    path.parent.mkdir(parents=True, exist_ok=True)
    save_file({"w": torch.arange(6.0).reshape(2, 3)}, str(path.with_name("manga109_panel_yolo11l.safetensors")), metadata={"names": "{0: 'frame'}"})
    monkeypatch.setattr(mm2, "_dist_on", lambda: True)

    def no_collective(*a, **k):
        raise AssertionError("a collective was issued from a worker-thread scope")
    monkeypatch.setattr(mm2, "broadcast_status", no_collective)
    monkeypatch.setattr(mm2, "broadcast_state_dict", no_collective)
    with manager.thread_local_reads():
        sd, md = manager._read_safetensors_with_metadata(path)
        with pytest.raises(ModelError):
            manager._staged(path.with_name("absent.json"), "config")
    assert torch.equal(sd["w"], torch.arange(6.0).reshape(2, 3)) and md["names"] == "{0: 'frame'}"


def test_sam_precision_of_a_batch():
    """`resolve_sam_precision`: "high" for every batch since round 6 (segment-only ones included: there the mask is the product), a pin wins"""
    import types
    from mangatranslator_amd.core.pipeline import resolve_sam_precision
    ns = types.SimpleNamespace
    cfg = lambda osb, up, pin=None: ns(detection=ns(sam_precision=pin), outside_text=ns(enabled=osb), output=ns(upscale_final_image=up))
    assert resolve_sam_precision(cfg(False, False)) == "high"
    assert resolve_sam_precision(cfg(True, False)) == "high"
    assert resolve_sam_precision(cfg(False, True)) == "high"
    assert resolve_sam_precision(cfg(False, False, "high")) == "high"
    assert resolve_sam_precision(cfg(True, True, "fast")) == "fast"


def test_prompt_is_encoded_at_load_time_when_the_encoders_are_staged(manager, tmp_path):
    """reference core/image/inpainting.py:846-873: the fixed prompt is encoded on first use and kept.  Here: `prompt_embeds.safetensors` absent,
    text_encoder/ (CLIP), text_encoder_2/ (T5), tokenizer/, tokenizer_2/ staged next to transformer/ -> the loader encodes "Remove all text."
    once through `transformers`, writes the file, and the pipeline has its embeddings; a second load reads the file (the encoders may be gone)"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import CLIPTextConfig, CLIPTextModel, PreTrainedTokenizerFast, T5Config, T5EncoderModel
    from mangatranslator_amd.core.ml.model_manager import ModelType
    from mangatranslator_amd.core.ml.prompt_embeds import ensure_prompt_embeds
    t, v = fc.models(seed=5)
    c = t.cfg
    root = manager.model_paths[ModelType.FLUX_KONTEXT_SDNQ_PIPELINE]
    (root / "transformer").mkdir(parents=True); (root / "vae").mkdir()
    save_file({k: x.to(torch.bfloat16).contiguous() for k, x in t.state_dict().items()}, str(root / "transformer" / "diffusion_pytorch_model.safetensors"))
    save_file({k: x.contiguous() for k, x in v.state_dict().items()}, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    (root / "transformer" / "config.json").write_text(json.dumps(dict(
        num_attention_heads=c["heads"], attention_head_dim=c["d"] // c["heads"], num_layers=c["layers"], num_single_layers=c["single_layers"],
        in_channels=64, joint_attention_dim=c["joint_dim"], pooled_projection_dim=c["pooled_dim"], axes_dims_rope=list(c["axes_dim"]))))
    (root / "vae" / "config.json").write_text(json.dumps(dict(block_out_channels=list(v.cfg["ch"]), norm_num_groups=v.cfg["groups"],
                                                            scaling_factor=v.cfg["scaling_factor"], shift_factor=v.cfg["shift_factor"])))
    pipe = manager.load_flux_kontext_sdnq()
    assert pipe._embeds is None                                   # nothing to encode with: the inpainter would report the missing embeddings
    manager.unload_flux_kontext_sdnq_models()
    # tiny stand-ins of the snapshot's encoders, written with transformers' own save_pretrained
    words = ["<pad>", "</s>", "<unk>", "remove", "all", "text", ".", "<s>"]
    for name in ("tokenizer", "tokenizer_2"):
        tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
        tk.pre_tokenizer = pre_tokenizers.Whitespace()
        PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", eos_token="</s>", unk_token="<unk>", bos_token="<s>", model_max_length=77).save_pretrained(str(root / name))
    torch.manual_seed(0)
    CLIPTextModel(CLIPTextConfig(vocab_size=len(words), hidden_size=c["pooled_dim"], intermediate_size=32, num_hidden_layers=1, num_attention_heads=2,
                                 max_position_embeddings=77, eos_token_id=1, pad_token_id=0, bos_token_id=7)).save_pretrained(str(root / "text_encoder"))
    T5EncoderModel(T5Config(vocab_size=len(words), d_model=c["joint_dim"], d_kv=8, d_ff=32, num_layers=1, num_heads=2, pad_token_id=0, eos_token_id=1,
                            decoder_start_token_id=0)).save_pretrained(str(root / "text_encoder_2"))
    pipe = manager.load_flux_kontext_sdnq()
    assert (root / "prompt_embeds.safetensors").exists()
    seq, pooled = pipe._embeds
    assert tuple(seq.shape) == (512, c["joint_dim"]) and tuple(pooled.shape) == (c["pooled_dim"],) and torch.isfinite(seq.float()).all() and seq.float().abs().sum() > 0
    manager.unload_flux_kontext_sdnq_models()
    import shutil
    for name in ("text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2"):
        shutil.rmtree(root / name)
    again = manager.load_flux_kontext_sdnq()._embeds
    assert torch.equal(again[0], seq) and torch.equal(again[1], pooled)
    assert ensure_prompt_embeds(tmp_path / "nothing-here", "kontext") is False
