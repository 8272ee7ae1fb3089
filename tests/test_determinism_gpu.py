"""GPU tier: identical calls give identical BYTES.  Every network of the hot path is run several times on the same input — the first call
launches eagerly while its hipGraph is captured, the later ones replay the graph — and on a second instance built from the same weights;
outputs are compared bit for bit.  This is the net that would have caught round 4's memset-node bug (SAM's mask choice depending on
the call before, tests/test_sam2_gpu.py::test_sam2_repeated_calls_are_bit_identical) and GroupNorm's atomically accumulated statistics
(last bits of every VAE output changing from run to run) without a page-level comparison stumbling over them."""
import numpy as np
import pytest
import torch

from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.plan import PlanBuilder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RCAN_HW = (300, 260)


def _same(first, got, what):
    for i, (a, b) in enumerate(zip(first, got)):
        assert torch.equal(a, b), f"{what}: output {i} differs between identical calls"


def test_groupnorm_is_order_independent(hip_lib):
    """many blocks per image (hw = 12 288 -> 12 partial sums per channel): was fp32 atomicAdd, now an ordered reduction"""
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(2, 128, 96, 512, generator=g) * 3 + 1).to(torch.bfloat16)
    outs = []
    for _ in range(2):                                     # two plans: different workspaces, different addresses
        pb = PlanBuilder(hip_lib, torch.device(DEV), abi.BF16)
        xa = pb.act(2, 128, 96, 512)
        xa.t.copy_(x.to(DEV))
        y = pb.groupnorm(xa, pb.const(torch.ones(512)), pb.const(torch.zeros(512)), 32, 1e-6, abi.ACT_SILU)
        plan = pb.build()
        for _ in range(4):
            plan.run(); torch.cuda.synchronize()
            outs.append(y.t.clone())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


def test_long_attention_with_key_split_repeats(hip_lib):
    """FLUX's sequence length: 840 (head, query block) units on 256 CUs, the last partial round goes through the key-split tail + merge"""
    _attention_repeats(hip_lib, 24, 8812)


def _attention_repeats(hip_lib, heads, t, d=128):
    g = torch.Generator(device=DEV).manual_seed(1)
    q = (torch.randn(1, t, heads, d, device=DEV, generator=g) * 0.12).to(torch.bfloat16)
    k = torch.randn(1, t, heads, d, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(1, t, heads, d, device=DEV, generator=g).to(torch.bfloat16)
    pb = PlanBuilder(hip_lib, torch.device(DEV), abi.BF16)
    o = pb.buf((1, t, heads, d), torch.bfloat16, zero=True)
    s = (t * heads * d, heads * d, d)
    pb.attention(pb.hold(q), pb.hold(k), pb.hold(v), o, 1, heads, t, t, d, s, s, s, s, 1.0, q_prescaled=True)
    plan = pb.build()
    first = None
    for _ in range(4):
        o.fill_(float("nan"))
        plan.run(); torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all()
        first = o.clone() if first is None else first
        assert torch.equal(first, o)


def test_detectors_repeat(hip_lib):
    from oracle import rtdetr_ref, yolo11_ref, yolo_ref
    from mangatranslator_amd.core.ml.rtdetr import RTDetrHip
    from mangatranslator_amd.core.ml.yolo import YoloSegHip
    from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
    dev = torch.device(DEV)
    page = (np.random.default_rng(0).random((384, 256, 3)) * 255).astype(np.uint8)
    ynet = yolo_ref.make_model("n", 1, seed=3)
    y11 = yolo11_ref.make_model("11", "n", 1, True, seed=1)
    y12 = yolo11_ref.make_model("12", "n", 1, False, seed=2)
    rmodel, rcfg = rtdetr_ref.make_model("tiny_test", seed=5)
    makers = [("yolov8n-seg", lambda: YoloSegHip(ynet.state_dict(), device=dev, lib=hip_lib), dict(conf=0.0, imgsz=256, max_det=20)),
              ("yolo11n-seg", lambda: Yolo11Hip(y11.state_dict(), device=dev, lib=hip_lib), dict(conf=0.0, imgsz=256, max_det=20)),
              ("yolo12n", lambda: Yolo11Hip(y12.state_dict(), device=dev, lib=hip_lib), dict(conf=0.0, imgsz=256, max_det=20)),
              ("rtdetr", lambda: RTDetrHip(rmodel.state_dict(), rcfg, device=dev, lib=hip_lib), dict(conf=0.0, imgsz=64))]
    for name, make, kw in makers:
        first = None
        for inst in range(2):
            model = make()
            for _ in range(3):
                res = model(page, **kw)[0]
                got = [res.boxes.xyxy.cpu(), res.boxes.conf.cpu(), res.boxes.cls.cpu()]
                if getattr(res, "masks", None) is not None:
                    got.append(res.masks.data.cpu())
                first = got if first is None else first
                _same(first, got, name)


def test_flux_pipelines_repeat(hip_lib):
    """Kontext (bf16) and Klein (MX fp8 linears) at test geometry: the same noise twice through one pipeline and once through a second one"""
    import flux2_checks as f2c
    import flux_checks as fc
    from PIL import Image
    from mangatranslator_amd.core.ml import flux as fx
    from mangatranslator_amd.core.ml import flux2 as f2
    dev = torch.device(DEV)
    h, w = 64, 96
    t, v = fc.models(seed=4)
    img, pe, pooled, noise = fc.inputs(t, h, w, 16)
    first = None
    for inst in range(2):
        pipe = fx.FluxKontextHip(*fc.hip_models(t, v, hip_lib, dev))
        for _ in range(2):
            out = pipe(image=img, width=w, height=h, num_inference_steps=2, guidance_scale=2.5, prompt_embeds=pe[None], pooled_prompt_embeds=pooled[None],
                       latents=noise).images[0]
            first = [out.cpu()] if first is None else first
            _same(first, [out.cpu()], "Kontext")
    t2, v2 = f2c.models(seed=4)
    g = torch.Generator().manual_seed(1)
    page = Image.fromarray((torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy())
    pe2 = torch.randn(16, t2.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    noise2 = torch.randn(1, t2.cfg["in_channels"], h // 16, w // 16, generator=g)
    first = None
    for inst in range(2):
        pipe = f2.Flux2KleinHip(*f2c.hip_models(t2, v2, hip_lib, dev, fp8=True))
        for _ in range(2):
            out = pipe(image=page, width=w, height=h, num_inference_steps=2, guidance_scale=1.0, prompt_embeds=pe2[None], latents=noise2, output_type="pt").images[0]
            first = [out.cpu()] if first is None else first
            _same(first, [out.cpu()], "Klein fp8")


def test_rcan_repeats(hip_lib):
    """the channel attention's 32-way split sums its records in a fixed order whichever workgroup arrives last"""
    from oracle.rcan_ref import make_state_dict
    from mangatranslator_amd.core.ml.rcan import RCANUpscaler
    sd = make_state_dict(n_feats=64, n_resgroups=2, n_resblocks=3, unshuffle=1, seed=7)
    page = torch.from_numpy((np.random.default_rng(1).random((*RCAN_HW, 3)) * 255).astype(np.uint8))
    first = None
    for inst in range(2):
        m = RCANUpscaler(sd, device=torch.device(DEV), lib=hip_lib)
        for _ in range(3):
            out = m.upscale_u8(page.to(DEV)).cpu()
            first = [out] if first is None else first
            _same(first, [out], "RCAN")


def test_bench_geometry_repeats(hip_lib):
    """the networks at the sizes the pages use (the big-tile GEMMs with their K-slice tail behind the 1x1 convolutions and SAM's linears,
    the pipelined conv kernel at thousands of tiles, RCAN at page size): seeded weights of the published architectures, a 1024 x 1536
    page, every result compared over three calls — first eager, then hipGraph replays"""
    from mangatranslator_amd.core.ml.rcan import RCANUpscaler
    from mangatranslator_amd.core.ml.rtdetr import RTDetrHip
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip
    from mangatranslator_amd.core.ml.yolo import YoloSegHip
    from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
    from mangatranslator_amd.utils import synthetic_checkpoints as synth
    from mangatranslator_amd.utils.synthetic_pages import make_page
    dev = torch.device(DEV)
    pg, boxes, _ = make_page(5, 1024, 1536, bubbles=8, osb_regions=0)
    bgr = np.ascontiguousarray(pg[..., ::-1])

    def detector_outputs(res):
        got = [res.boxes.xyxy.cpu(), res.boxes.conf.cpu(), res.boxes.cls.cpu()]
        if getattr(res, "masks", None) is not None:
            got.append(res.masks.data.cpu())
        return got

    v8_shapes, v8_head = synth.yolov8_seg_shapes("m", 1)
    nets = [("yolov8m-seg @1600", lambda: YoloSegHip(synth.seeded_detector(v8_shapes, v8_head, seed=3, class_bias=-1.0, class_gain=0.05, box_gain=0.1), device=dev, lib=hip_lib),
             dict(conf=0.0, imgsz=1600, max_det=50))]
    for tag, fam, scale, seg, seed, imgsz in (("yolo11m-seg @1600", "11", "m", True, 13, 1600), ("yolo11l @640", "11", "l", False, 17, 640), ("yolo12x @640", "12", "x", False, 19, 640)):
        shapes_, head_ = synth.yolo11_shapes(fam, scale, 1, seg)
        sd_ = synth.seeded_detector(shapes_, head_, seed=seed, class_bias=-2.0, class_gain=0.05, box_gain=0.1)
        nets.append((tag, (lambda sd=sd_: Yolo11Hip(sd, device=dev, lib=hip_lib)), dict(conf=0.0, imgsz=imgsz, max_det=50)))
    rcfg = synth.rtdetr_r50_config()
    rsd = synth.rtdetr_state_dict(rcfg, seed=5)
    nets.append(("rtdetr-r50 @640", lambda: RTDetrHip(rsd, rcfg, device=dev, lib=hip_lib), dict(conf=0.0, imgsz=640)))
    for tag, make, kw in nets:
        model = make()
        first = None
        for _ in range(3):
            got = detector_outputs(model(bgr, **kw)[0])
            first = got if first is None else first
            _same(first, got, tag)
        del model
    sam_cfg = synth.sam2_hiera_large_config()
    sam = Sam2Hip(synth.sam2_state_dict(sam_cfg, seed=11), sam_cfg, device=dev, lib=hip_lib, dtype=abi.F16)
    first = None
    for _ in range(3):
        masks, low, iou, sel = sam.segment(pg, np.asarray(boxes, np.float32), return_logits=True)
        got = [t.cpu() for t in (masks, low, iou, sel)]
        first = got if first is None else first
        _same(first, got, "sam2.1 hiera-L")
    del sam
    rcan = RCANUpscaler(synth.rcan_state_dict(seed=7, n_feats=64, n_resgroups=10, n_resblocks=20, unshuffle=1), device=dev, lib=hip_lib)
    page = torch.from_numpy(pg).to(dev)
    first = None
    for _ in range(3):
        out = rcan.upscale_u8(page).cpu()
        first = [out] if first is None else first
        _same(first, [out], "rcan 10x20 @1024x1536")
