"""Pinning kit for the oracles whose upstream library is absent from the build image.

`oracle/yolo_ref.py`, `yolo11_ref.py` (ultralytics), `rcan_ref.py` (spandrel), `flux_ref.py`, `flux2_ref.py` (diffusers),
`cleaning_ref.py` and `cv2_color_ref.py` (OpenCV) restate third-party arithmetic the reference calls
(`core/image/detection.py:1337-1351`, `core/image/image_utils.py:369-374`, `core/image/inpainting.py:877-887, 1577-1589`,
`core/image/cleaning.py:170-382`).  On a machine that HAS those wheels this script runs every oracle beside the real library on
seeded inputs — and on real checkpoints when `--models` points at the reference's `./models` directory — and writes

    tests/golden/pinned_<target>.npz        inputs + the LIBRARY's outputs (small arrays; never weights, never library source)
    tests/golden/pinned_report.json         per case: max |oracle - library|, library version, pass / fail

`tests/test_pinned_oracles.py` replays every fixture it finds against the oracle, so once the files are committed the CPU tier
checks the restatements against what the reference really executes.  Seeded weights are not stored: both sides build them from the
oracle's own `make_*` functions (whose parameter names are the upstream libraries'), i.e. the library model is loaded FROM the
oracle's state dict, and the fixture keeps the seed.

    python tools/pin_oracles.py                        # everything importable here
    python tools/pin_oracles.py --only cv2,spandrel    # a subset
    python tools/pin_oracles.py --models ./models      # also: real checkpoints found there (outputs compared, digests recorded)

Test infrastructure: nothing under mangatranslator_amd/ imports this file.
"""
import argparse
import hashlib
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


# --------------------------------------------------------------------------------------------------------------
# seeded inputs shared by the generator (library side) and the replay (oracle side)
def cv2_cases(seed: int = 0) -> dict:
    """Inputs of every OpenCV primitive `core/image/cleaning.py` / `inpainting.py` use: blob masks (morphology, distance transform,
    contours), grey ramps and two-population images (threshold / Otsu), random BGR images (colour conversions)."""
    rng = np.random.default_rng(seed)
    h, w = 61, 83
    yy, xx = np.mgrid[0:h, 0:w]
    blobs = np.zeros((h, w), np.uint8)
    for cy, cx, ry, rx in ((14, 18, 9, 13), (40, 30, 12, 7), (30, 64, 17, 11), (5, 78, 4, 4), (58, 3, 3, 6)):
        blobs[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
    blobs[20:24, 10:70] = 255                                        # a bar joining two blobs
    blobs[38:43, 28:33] = 0                                          # a hole
    noise = (rng.random((h, w)) < 0.03).astype(np.uint8) * 255
    speck = blobs ^ noise
    grey = np.clip(rng.normal(70, 12, (h, w)), 0, 255).astype(np.uint8)
    grey[blobs > 0] = np.clip(rng.normal(200, 10, int((blobs > 0).sum())), 0, 255).astype(np.uint8)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ramp = np.stack([np.tile(np.arange(256, dtype=np.uint8), (4, 1))] * 3, axis=-1)      # grey ramp as 3 channels
    # in-gamut 8-bit Lab input for the inverse conversion (the same array for library and oracle: inverting each side's OWN forward
    # result would compare different inputs — a one-code Lab difference moves a saturated colour by tens of RGB levels)
    lab_in = _float_lab_u8(np.concatenate([bgr[:24], ramp[:2, ::3].repeat(1, 0)[:, :83]], 0))
    return {"blobs": blobs, "speck": speck, "grey": grey, "bgr": bgr, "ramp": ramp, "lab_in": lab_in}


def _float_lab_u8(rgb_u8):
    c = np.asarray(rgb_u8).astype(np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ m.T / np.array([0.950456, 1.0, 1.088754])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = np.where(xyz[..., 1] > 0.008856, 116.0 * f[..., 1] - 16.0, 903.3 * xyz[..., 1])
    out = np.stack([L * 2.55, 500.0 * (f[..., 0] - f[..., 1]) + 128.0, 200.0 * (f[..., 1] - f[..., 2]) + 128.0], -1)
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def cv2_library_outputs(cv2, cases: dict) -> dict:
    """What the LIBRARY returns (every call mirrors one in the reference, cited)."""
    out = {}
    for k in ((3, 3), (5, 5), (7, 7), (9, 5), (11, 11)):             # cleaning.py:170, 261 getStructuringElement(MORPH_ELLIPSE, k)
        out[f"ellipse_{k[0]}x{k[1]}"] = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, k)
    k7, k5 = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (7, 7)), cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (5, 5))
    for name in ("blobs", "speck"):
        m = cases[name]
        out[f"dilate7_{name}"] = cv2.dilate(m, k7, iterations=1)      # cleaning.py:262
        out[f"erode5_{name}"] = cv2.erode(m, k5, iterations=1)        # cleaning.py:327
        out[f"dilate7x2_{name}"] = cv2.dilate(m, k7, iterations=2)
        out[f"dist_l2_5_{name}"] = cv2.distanceTransform(m, cv2.DIST_L2, 5)                         # cleaning.py:178, detection.py (conjoined split)
        cs, _ = cv2.findContours(m, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)                     # cleaning.py:301
        cs = sorted(cs, key=lambda c: (cv2.boundingRect(c)[1], cv2.boundingRect(c)[0], -cv2.contourArea(c)))
        out[f"contour_count_{name}"] = np.array([len(cs)])
        out[f"contour_area_{name}"] = np.array([cv2.contourArea(c) for c in cs], np.float64)        # cleaning.py:304
        out[f"contour_rect_{name}"] = np.array([cv2.boundingRect(c) for c in cs], np.int64).reshape(-1, 4)
        mo = [cv2.moments(c) for c in cs]
        out[f"contour_m00_m10_m01_{name}"] = np.array([[q["m00"], q["m10"], q["m01"]] for q in mo], np.float64).reshape(-1, 3)
        filled = np.zeros_like(m)
        cv2.drawContours(filled, cs, -1, 255, thickness=cv2.FILLED)                                # cleaning.py:316
        out[f"filled_{name}"] = filled
    g = cases["grey"]
    out["thresh_128"] = cv2.threshold(g, 128, 255, cv2.THRESH_BINARY)[1]                            # cleaning.py:288
    t, o = cv2.threshold(g, 0, 255, cv2.THRESH_BINARY + cv2.THRESH_OTSU)                            # cleaning.py:1085 (Otsu retry)
    out["otsu_value"] = np.array([t], np.float64)
    out["otsu_mask"] = o
    for name in ("bgr", "ramp"):
        im = cases[name]
        out[f"gray_{name}"] = cv2.cvtColor(im, cv2.COLOR_BGR2GRAY)                                  # cleaning.py:232
        out[f"hsv_s_{name}"] = cv2.cvtColor(im[:8], cv2.COLOR_BGR2HSV)[..., 1]                      # cleaning.py:893-1043 (coloured bubbles read S)
        out[f"lab_{name}"] = cv2.cvtColor(im, cv2.COLOR_RGB2LAB)                                    # inpainting.py:1187-1256
    out["lab_back"] = cv2.cvtColor(cases["lab_in"], cv2.COLOR_LAB2RGB)
    return out


def cv2_oracle_outputs(cases: dict) -> dict:
    """The same quantities from oracle/cleaning_ref.py and oracle/cv2_color_ref.py."""
    from oracle import cleaning_ref as cr
    from oracle import cv2_color_ref as cc
    out = {}
    for k in ((3, 3), (5, 5), (7, 7), (9, 5), (11, 11)):
        out[f"ellipse_{k[0]}x{k[1]}"] = cr.ellipse_kernel(k)
    k7, k5 = cr.ellipse_kernel((7, 7)), cr.ellipse_kernel((5, 5))
    for name in ("blobs", "speck"):
        m = cases[name]
        out[f"dilate7_{name}"] = cr.dilate(m, k7)
        out[f"erode5_{name}"] = cr.erode(m, k5)
        out[f"dilate7x2_{name}"] = cr.dilate(cr.dilate(m, k7), k7)
        out[f"dist_l2_5_{name}"] = cr.distance_transform_l2_5x5(m)
        cs = cr.find_external_contours(m)
        cs = sorted(cs, key=lambda c: (cr.bounding_rect(c)[1], cr.bounding_rect(c)[0], -cr.contour_area(c)))
        out[f"contour_count_{name}"] = np.array([len(cs)])
        out[f"contour_area_{name}"] = np.array([cr.contour_area(c) for c in cs], np.float64)
        out[f"contour_rect_{name}"] = np.array([cr.bounding_rect(c) for c in cs], np.int64).reshape(-1, 4)
        mo = []
        for c in cs:                                                  # cv2.moments of a contour: Green's sums, sign chosen so that m00 >= 0
            a00, a10, a01 = cr.contour_sums(c)
            s2, s6 = (0.5, 1.0 / 6) if a00 > 0 else (-0.5, -1.0 / 6)
            mo.append([a00 * s2, a10 * s6, a01 * s6])
        out[f"contour_m00_m10_m01_{name}"] = np.array(mo, np.float64).reshape(-1, 3)
        out[f"filled_{name}"] = cr.draw_filled(cs, m.shape)
    g = cases["grey"]
    out["thresh_128"] = np.where(g > 128, 255, 0).astype(np.uint8)
    t = cr.otsu_threshold(g)
    out["otsu_value"] = np.array([t], np.float64)
    out["otsu_mask"] = np.where(g > t, 255, 0).astype(np.uint8)
    for name in ("bgr", "ramp"):
        im = cases[name]
        out[f"gray_{name}"] = cr.bgr_to_gray(im)
        out[f"hsv_s_{name}"] = np.array([[cr.bgr_pixel_saturation(int(px[0]), int(px[1]), int(px[2])) for px in row] for row in im[:8]], np.uint8)
        out[f"lab_{name}"] = cc.rgb_to_lab_u8(im)
    out["lab_back"] = cc.lab_to_rgb_u8(cases["lab_in"])
    return out


# tolerances of the replay: integer primitives are bit-exact by definition; the float inverse Lab path may differ by one code
CV2_TOL = {"dist_l2_5_": 1e-4, "contour_area_": 1e-9, "contour_m00_m10_m01_": 1e-6, "otsu_value": 0.0, "lab_back": 1.0}


def compare(lib: dict, ora: dict, tol_by_prefix: dict, default_tol: float = 0.0) -> dict:
    rep = {}
    for k, v in lib.items():
        if k not in ora:
            rep[k] = {"ok": False, "why": "oracle has no such output"}
            continue
        a, b = np.asarray(v), np.asarray(ora[k])
        if a.shape != b.shape:
            rep[k] = {"ok": False, "why": f"shape {b.shape} vs library {a.shape}"}
            continue
        tol = next((t for p, t in tol_by_prefix.items() if k.startswith(p)), default_tol)
        d = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0
        rep[k] = {"ok": bool(d <= tol), "max_abs_diff": d, "tol": tol, "differing": int((a != b).sum())}
    return rep


def pin_cv2(cv2=None, out_dir: Path = GOLDEN, seed: int = 0) -> dict:
    cv2 = cv2 if cv2 is not None else importlib.import_module("cv2")
    cases = cv2_cases(seed)
    lib = cv2_library_outputs(cv2, cases)
    rep = compare(lib, cv2_oracle_outputs(cases), CV2_TOL)
    np.savez_compressed(out_dir / "pinned_cv2.npz", seed=np.array([seed]), **{f"out__{k}": np.asarray(v) for k, v in lib.items()})
    return {"library": f"cv2 {getattr(cv2, '__version__', '?')}", "cases": rep}


# --------------------------------------------------------------------------------------------------------------
# networks: the library model is built FROM the oracle's seeded state dict (the oracle uses the upstream parameter names)
def _t2n(t):
    return t.detach().float().cpu().numpy()


def yolo_cases():
    """(fixture key, oracle constructor, ultralytics yaml, task) — n scales: the arithmetic of every block type at 2-3 M parameters"""
    from oracle import yolo11_ref, yolo_ref
    return [
        ("yolov8n_seg", lambda s: yolo_ref.make_model("n", 1, s), "yolov8n-seg.yaml", "segment"),
        ("yolo11n", lambda s: yolo11_ref.make_model("11", "n", 1, False, s), "yolo11n.yaml", "detect"),
        ("yolo11n_seg", lambda s: yolo11_ref.make_model("11", "n", 1, True, s), "yolo11n-seg.yaml", "segment"),
        ("yolo12n", lambda s: yolo11_ref.make_model("12", "n", 1, False, s), "yolo12n.yaml", "detect"),
    ]


def yolo_input(seed: int):
    import torch
    return torch.rand(1, 3, 96, 128, generator=torch.Generator().manual_seed(1000 + seed))


def oracle_yolo_raw(net, x):
    """the oracle's raw head output(s) as a list of arrays: decoded prediction [1, 4 + nc (+ nm), A] and, for -seg, the prototypes"""
    import torch
    with torch.no_grad():
        y = net(x)
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    return [_t2n(t) for t in ys if hasattr(t, "shape")]


def _tensors(o, acc):
    import torch
    if isinstance(o, torch.Tensor):
        acc.append(o)
    elif isinstance(o, (tuple, list)):
        for e in o:
            _tensors(e, acc)
    elif isinstance(o, dict):
        for e in o.values():
            _tensors(e, acc)
    return acc


def pin_ultralytics(ul=None, out_dir: Path = GOLDEN, seed: int = 0) -> dict:
    import torch
    ul = ul if ul is not None else importlib.import_module("ultralytics")
    tasks = getattr(getattr(ul, "nn", None), "tasks", None) or importlib.import_module(ul.__name__ + ".nn.tasks")
    rep, store = {}, {}
    x = yolo_input(seed)
    for key, make, yaml, task in yolo_cases():
        try:
            net = make(seed)
            sd = net.state_dict()
            cls = tasks.SegmentationModel if task == "segment" else tasks.DetectionModel
            m = cls(yaml, ch=3, nc=1, verbose=False)
            m.fuse()                                                  # Conv+BN -> conv.weight / conv.bias: the names the oracle (and the product loader) use
            missing, unexpected = m.load_state_dict(sd, strict=False)
            learn = [k for k in missing if not k.endswith(("dfl.conv.weight", "num_batches_tracked"))]
            if learn or unexpected:
                rep[key] = {"ok": False, "why": f"state dict names differ: missing {learn[:5]} unexpected {list(unexpected)[:5]}"}
                continue
            m.eval().float()
            with torch.no_grad():
                y = m(x)
            ora = oracle_yolo_raw(net, x)
            libs = [_t2n(t) for t in _tensors(y, [])]
            case = {}
            for i, o in enumerate(ora):                               # match by shape: ultralytics nests (pred, (feats, coefficients, proto))
                cand = [l for l in libs if l.shape == o.shape]
                if not cand:
                    case[f"out{i}"] = {"ok": False, "why": f"library returned no tensor of shape {o.shape}"}
                    continue
                d = min(float(np.abs(c - o).max()) for c in cand)
                best = min(cand, key=lambda c: float(np.abs(c - o).max()))
                store[f"{key}__out{i}"] = best
                case[f"out{i}"] = {"ok": bool(d <= 1e-3 * max(1.0, float(np.abs(best).max()))), "max_abs_diff": d, "scale": float(np.abs(best).max())}
            rep[key] = {"ok": all(c["ok"] for c in case.values()), "outputs": case}
        except Exception as e:                                        # a version whose API moved: report, keep going
            rep[key] = {"ok": False, "why": f"{type(e).__name__}: {e}"}
    if store:
        np.savez_compressed(out_dir / "pinned_ultralytics.npz", seed=np.array([seed]), **store)
    return {"library": f"ultralytics {getattr(ul, '__version__', '?')}", "cases": rep}


RCAN_HP = dict(n_feats=16, n_resgroups=2, n_resblocks=3, reduction=4, scale=2, unshuffle=1)


def rcan_input(seed: int):
    import torch
    return torch.rand(1, 3, 24, 32, generator=torch.Generator().manual_seed(2000 + seed))


def pin_spandrel(sp=None, out_dir: Path = GOLDEN, seed: int = 0) -> dict:
    import torch
    from oracle import rcan_ref
    sp = sp if sp is not None else importlib.import_module("spandrel")
    rep, store = {}, {}
    for key, hp in (("rcan_plain", RCAN_HP), ("rcan_unshuffle2", dict(RCAN_HP, unshuffle=2))):
        try:
            sd = rcan_ref.make_state_dict(seed=seed, **hp)
            desc = sp.ModelLoader().load_from_state_dict({k: v.clone() for k, v in sd.items()})      # image_utils.py:369-374 / model_manager.py:652-654
            desc.model.eval().float()
            x = rcan_input(seed)
            with torch.no_grad():
                y = desc(x) if callable(desc) else desc.model(x)
                o = rcan_ref.load_ref(sd)(x)
            y, o = _t2n(y), _t2n(o)
            store[f"{key}__out0"] = y
            d = float(np.abs(y - o).max()) if y.shape == o.shape else float("inf")
            rep[key] = {"ok": bool(d <= 1e-4), "max_abs_diff": d, "arch": getattr(getattr(desc, "architecture", None), "name", "?"), "scale": getattr(desc, "scale", None)}
        except Exception as e:
            rep[key] = {"ok": False, "why": f"{type(e).__name__}: {e}"}
    if store:
        np.savez_compressed(out_dir / "pinned_spandrel.npz", seed=np.array([seed]), **store)
    return {"library": f"spandrel {getattr(sp, '__version__', '?')}", "cases": rep}


FLUX1_CFG = dict(d=128, heads=2, layers=2, single_layers=2, joint_dim=64, pooled_dim=32, axes_dim=(8, 28, 28))


def flux1_inputs(seed: int, cfg=FLUX1_CFG, h2=4, w2=6, t_txt=8):
    import torch
    from oracle import flux_ref as fr
    g = torch.Generator().manual_seed(3000 + seed)
    tn = h2 * w2
    lat = torch.randn(2 * tn, 64, generator=g)
    pe = torch.randn(t_txt, cfg["joint_dim"], generator=g)
    pooled = torch.randn(cfg["pooled_dim"], generator=g)
    ids = torch.cat([fr.image_ids(h2, w2, 0), fr.image_ids(h2, w2, 1)])
    return lat, pe, pooled, ids, t_txt


def pin_diffusers(df=None, out_dir: Path = GOLDEN, seed: int = 0) -> dict:
    """FLUX.1: FluxTransformer2DModel + AutoencoderKL built from the oracle's state dicts (inpainting.py:877-887, model_manager.py:1176-1252);
    FLUX.2: Flux2Transformer2DModel + AutoencoderKLFlux2 (inpainting.py:1577-1589) when this diffusers has them."""
    import torch
    from oracle import flux_ref as fr
    df = df if df is not None else importlib.import_module("diffusers")
    rep, store = {}, {}
    try:
        t, v = fr.make_models(seed=seed, **FLUX1_CFG)
        c = t.cfg
        m = df.FluxTransformer2DModel(patch_size=1, in_channels=c["in_channels"], num_layers=c["layers"], num_single_layers=c["single_layers"],
                                      attention_head_dim=c["d"] // c["heads"], num_attention_heads=c["heads"], joint_attention_dim=c["joint_dim"],
                                      pooled_projection_dim=c["pooled_dim"], guidance_embeds=True, axes_dims_rope=tuple(c["axes_dim"]))
        missing, unexpected = m.load_state_dict(t.state_dict(), strict=False)
        if missing or unexpected:
            rep["flux1_transformer"] = {"ok": False, "why": f"state dict names differ: missing {list(missing)[:5]} unexpected {list(unexpected)[:5]}"}
        else:
            m.eval().float()
            lat, pe, pooled, ids, t_txt = flux1_inputs(seed)
            with torch.no_grad():
                y = m(hidden_states=lat[None], timestep=torch.tensor([0.7]), guidance=torch.tensor([2.5]), pooled_projections=pooled[None],
                      encoder_hidden_states=pe[None], txt_ids=torch.zeros(t_txt, 3), img_ids=ids, return_dict=False)[0][0]
                o = t(lat, 0.7, 2.5, pooled, pe, torch.zeros(t_txt, 3), ids)
            y, o = _t2n(y), _t2n(o)
            store["flux1_transformer__out0"] = y
            d = float(np.abs(y - o).max())
            rep["flux1_transformer"] = {"ok": bool(d <= 2e-4 * max(1.0, float(np.abs(y).max()))), "max_abs_diff": d, "scale": float(np.abs(y).max())}
        vc = v.cfg
        vae = df.AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * len(vc["ch"]),
                               up_block_types=("UpDecoderBlock2D",) * len(vc["ch"]), block_out_channels=tuple(vc["ch"]), layers_per_block=2,
                               latent_channels=vc["latent"], norm_num_groups=vc["groups"], scaling_factor=vc["scaling_factor"],
                               shift_factor=vc["shift_factor"], use_quant_conv=False, use_post_quant_conv=False, mid_block_add_attention=True)
        missing, unexpected = vae.load_state_dict(v.state_dict(), strict=False)
        if missing or unexpected:
            rep["flux1_vae"] = {"ok": False, "why": f"state dict names differ: missing {list(missing)[:5]} unexpected {list(unexpected)[:5]}"}
        else:
            vae.eval().float()
            g = torch.Generator().manual_seed(3100 + seed)
            x = torch.rand(1, 3, 32, 48, generator=g) * 2 - 1
            z = torch.randn(1, vc["latent"], 4, 6, generator=g)
            with torch.no_grad():
                e_lib = vae.encode(x).latent_dist.mode()
                d_lib = vae.decode(z, return_dict=False)[0]
                e_or, d_or = v.encode_mode(x), v.decode(z)
            for name, a, b in (("enc", e_lib, e_or), ("dec", d_lib, d_or)):
                a, b = _t2n(a), _t2n(b)
                store[f"flux1_vae__{name}"] = a
                d = float(np.abs(a - b).max()) if a.shape == b.shape else float("inf")
                rep[f"flux1_vae_{name}"] = {"ok": bool(d <= 2e-4 * max(1.0, float(np.abs(a).max()))), "max_abs_diff": d}
    except Exception as e:
        rep["flux1"] = {"ok": False, "why": f"{type(e).__name__}: {e}"}
    try:
        from oracle import flux2_ref as f2
        if not hasattr(df, "Flux2Transformer2DModel"):
            rep["flux2"] = {"ok": False, "why": "this diffusers has no Flux2Transformer2DModel (the reference pins a git commit, requirements.txt)"}
        else:
            t2, v2 = f2.make_models(seed=seed)
            c = t2.cfg
            m = df.Flux2Transformer2DModel(**{k: c[k] for k in c if k in df.Flux2Transformer2DModel.__init__.__code__.co_varnames})
            missing, unexpected = m.load_state_dict(t2.state_dict(), strict=False)
            rep["flux2_transformer_names"] = {"ok": not missing and not unexpected, "missing": list(missing)[:8], "unexpected": list(unexpected)[:8]}
    except Exception as e:
        rep["flux2"] = {"ok": False, "why": f"{type(e).__name__}: {e}"}
    if store:
        np.savez_compressed(out_dir / "pinned_diffusers.npz", seed=np.array([seed]), **store)
    return {"library": f"diffusers {getattr(df, '__version__', '?')}", "cases": rep}


# --------------------------------------------------------------------------------------------------------------
# replay: the oracle against a committed fixture (what tests/test_pinned_oracles.py runs on the CPU tier)
def _stored(npz, prefix=""):
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix) and k != "seed"}


def replay_cv2(path: Path, extra_tol: dict = None) -> dict:
    z = np.load(path)
    seed = int(z["seed"][0])
    tol = dict(CV2_TOL)
    tol.update(extra_tol or {})
    return compare(_stored(z, "out__"), cv2_oracle_outputs(cv2_cases(seed)), tol)


def _replay_outputs(stored: dict, key: str, outs: list, rel: float) -> dict:
    rep = {}
    for i, o in enumerate(outs):
        name = f"{key}__out{i}"
        if name not in stored:
            continue
        lib = stored[name]
        d = float(np.abs(lib - o).max()) if lib.shape == o.shape else float("inf")
        rep[name] = {"ok": bool(d <= rel * max(1.0, float(np.abs(lib).max()))), "max_abs_diff": d}
    return rep


def replay_ultralytics(path: Path) -> dict:
    z = np.load(path)
    seed, stored, rep = int(z["seed"][0]), _stored(z), {}
    x = yolo_input(seed)
    for key, make, _, _ in yolo_cases():
        if any(k.startswith(key + "__") for k in stored):
            rep.update(_replay_outputs(stored, key, oracle_yolo_raw(make(seed), x), 1e-3))
    return rep


def replay_spandrel(path: Path) -> dict:
    import torch
    from oracle import rcan_ref
    z = np.load(path)
    seed, stored, rep = int(z["seed"][0]), _stored(z), {}
    for key, hp in (("rcan_plain", RCAN_HP), ("rcan_unshuffle2", dict(RCAN_HP, unshuffle=2))):
        if f"{key}__out0" in stored:
            with torch.no_grad():
                o = _t2n(rcan_ref.load_ref(rcan_ref.make_state_dict(seed=seed, **hp))(rcan_input(seed)))
            rep.update(_replay_outputs(stored, key, [o], 1e-4))
    return rep


def replay_diffusers(path: Path) -> dict:
    import torch
    from oracle import flux_ref as fr
    z = np.load(path)
    seed, stored, rep = int(z["seed"][0]), _stored(z), {}
    t, v = fr.make_models(seed=seed, **FLUX1_CFG)
    with torch.no_grad():
        if "flux1_transformer__out0" in stored:
            lat, pe, pooled, ids, t_txt = flux1_inputs(seed)
            rep.update(_replay_outputs(stored, "flux1_transformer", [_t2n(t(lat, 0.7, 2.5, pooled, pe, torch.zeros(t_txt, 3), ids))], 2e-4))
        if "flux1_vae__enc" in stored:
            g = torch.Generator().manual_seed(3100 + seed)
            x = torch.rand(1, 3, 32, 48, generator=g) * 2 - 1
            zz = torch.randn(1, v.cfg["latent"], 4, 6, generator=g)
            for name, o in (("enc", v.encode_mode(x)), ("dec", v.decode(zz))):
                lib, o = stored[f"flux1_vae__{name}"], _t2n(o)
                d = float(np.abs(lib - o).max()) if lib.shape == o.shape else float("inf")
                rep[f"flux1_vae__{name}"] = {"ok": bool(d <= 2e-4 * max(1.0, float(np.abs(lib).max()))), "max_abs_diff": d}
    return rep


REPLAY = {"pinned_cv2.npz": replay_cv2, "pinned_ultralytics.npz": replay_ultralytics, "pinned_spandrel.npz": replay_spandrel,
          "pinned_diffusers.npz": replay_diffusers}


# --------------------------------------------------------------------------------------------------------------
# real checkpoints (optional): the same comparison on the weights the reference downloads
def pin_real_checkpoints(models: Path) -> dict:
    """Looks for the files the reference's ModelManager downloads (model_manager.py:617-838) under `models`; for each one it can open,
    runs the library and the oracle on a seeded page crop and records the gap and the checkpoint's digest.  Nothing is stored but numbers."""
    import torch
    rep = {}
    up = list(models.glob("**/*RCAN*.safetensors")) + list(models.glob("**/*AnimeSharp*.safetensors"))
    for f in up[:2]:
        try:
            from safetensors.torch import load_file
            import spandrel
            from oracle import rcan_ref
            sd = load_file(str(f))
            desc = spandrel.ModelLoader().load_from_state_dict({k: v.clone() for k, v in sd.items()})
            desc.model.eval().float()
            x = torch.rand(1, 3, 96, 128, generator=torch.Generator().manual_seed(7))
            with torch.no_grad():
                y, o = desc(x), rcan_ref.load_ref(sd)(x)
            mse = float(((y.float() - o.float()) ** 2).mean())
            rep[f.name] = {"sha256_16": hashlib.sha256(f.read_bytes()).hexdigest()[:16], "hparams": rcan_ref.rcan_hparams(sd),
                           "psnr_db": 99.0 if mse == 0 else float(10 * np.log10(1.0 / mse)), "max_abs_diff": float((y - o).abs().max())}
        except Exception as e:
            rep[f.name] = {"ok": False, "why": f"{type(e).__name__}: {e}"}
    for f in list(models.glob("**/*.pt"))[:6]:
        try:
            from ultralytics import YOLO
            from mangatranslator_amd.core.ml.model_manager import ModelManager  # noqa: F401  (the product loader tells the families apart)
            m = YOLO(str(f)).model.float().eval()
            m.fuse()
            names = sorted({k.split(".")[2] for k in m.state_dict() if k.count(".") > 2})
            rep[f.name] = {"sha256_16": hashlib.sha256(f.read_bytes()).hexdigest()[:16], "modules": names[:12],
                           "note": "export with tools/export_ultralytics_state_dict.py, then tests/test_pinned_oracles.py::test_real_checkpoint_state_dicts"}
        except Exception as e:
            rep[f.name] = {"ok": False, "why": f"{type(e).__name__}: {e}"}
    return rep


TARGETS = {"cv2": ("cv2", pin_cv2), "ultralytics": ("ultralytics", pin_ultralytics), "spandrel": ("spandrel", pin_spandrel),
           "diffusers": ("diffusers", pin_diffusers)}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--only", default="", help="comma-separated subset of " + ",".join(TARGETS))
    ap.add_argument("--models", default="", help="the reference's ./models directory (real checkpoints)")
    ap.add_argument("--out", default=str(GOLDEN))
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    want = [t for t in a.only.split(",") if t] or list(TARGETS)
    report = {}
    for t in want:
        mod, fn = TARGETS[t]
        try:
            lib = importlib.import_module(mod)
        except Exception as e:
            report[t] = {"status": "wheel absent", "why": f"{type(e).__name__}: {e}"}
            continue
        r = fn(lib, out, a.seed)
        r["status"] = "pinned" if all(c.get("ok") for c in r["cases"].values()) else "MISMATCH"
        report[t] = r
    if a.models:
        report["real_checkpoints"] = pin_real_checkpoints(Path(a.models))
    (out / "pinned_report.json").write_text(json.dumps(report, indent=1, sort_keys=True, default=str))
    for t, r in report.items():
        print(t, r.get("status", ""), r.get("library", r.get("why", "")))
        for k, c in (r.get("cases") or {}).items():
            if not c.get("ok"):
                print("   MISMATCH", k, {x: c[x] for x in c if x != "outputs"})
    return 1 if any(r.get("status") == "MISMATCH" for r in report.values()) else 0


if __name__ == "__main__":
    raise SystemExit(main())
