#!/usr/bin/env python
"""bench.py — pages/sec of the vision hot path on N MI355X (one process per GPU).

Metric (BASELINE.json): pages/sec (detect+segment+inpaint+upscale) 1024x1536 @1/2/4/8 MI355X.
A "step" is one synthetic 1024x1536 page through every hot-path stage that is built so far;
`config.stages` names exactly which stages ran inside the timed region (stages not listed there
are NOT implemented yet and therefore NOT counted — see DESIGN.md "bench scope").
Pages are resident in HBM before the timed region (decode/encode excluded).
Pages are independent units: rank r processes its own pages, no data-path collective; the only
collective is the start-up weight broadcast over RCCL (outside the timed region).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/f16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1536)
    ap.add_argument("--stages", default="all", help="comma list of: segment,upscale (default: every built stage)")
    ap.add_argument("--boxes", type=int, default=8, help="detections per page (generator ground truth, SURVEY.md §8d)")
    ap.add_argument("--upscale-model", default="model", choices=["model", "model_lite"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    return ap.parse_args()


def broadcast_state_dict(sd, rank, world, device):
    """Rank 0 owns the checkpoint; one flat RCCL broadcast over xGMI hands it to every rank."""
    import torch.distributed as dist
    keys = sorted(sd.keys())
    shapes = [tuple(sd[k].shape) for k in keys]
    sizes = [int(torch.tensor(s).prod().item()) if len(s) else 1 for s in shapes]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if rank == 0:
        flat.copy_(torch.cat([sd[k].float().reshape(-1) for k in keys]).to(device))
    dist.broadcast(flat, src=0)
    out, off = {}, 0
    flat = flat.cpu()
    for k, s, n in zip(keys, shapes, sizes):
        out[k] = flat[off:off + n].view(s).clone()
        off += n
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from mangatranslator_amd.hip.lib import get_library
    from mangatranslator_amd.core.ml.rcan import RCANUpscaler
    from mangatranslator_amd.utils.synthetic_pages import make_page
    from oracle.rcan_ref import make_state_dict   # synthetic checkpoint generator (no real weights offline)

    lib = get_library()
    lib.init(local_rank)

    # ---- models: rank 0 "reads" the checkpoints, everyone else receives them over RCCL ----------
    lite = args.upscale_model == "model_lite"
    if lite:   # assumed Fast_RCAN_PU shape (pixel-unshuffle variant); real hyper-parameters come from the file
        rcan_cfg = dict(n_feats=64, n_resgroups=4, n_resblocks=8, unshuffle=2)
    else:      # canonical RCAN: 10 groups x 20 RCAB x 64 feats
        rcan_cfg = dict(n_feats=64, n_resgroups=10, n_resblocks=20, unshuffle=1)
    sd = make_state_dict(seed=7, **rcan_cfg) if (rank == 0 or world == 1) else None
    if world > 1:
        if rank != 0:
            sd = {k: torch.empty_like(v) for k, v in make_state_dict(seed=0, **rcan_cfg).items()}
        sd = broadcast_state_dict(sd, rank, world, device)
    want = ["detect", "segment", "upscale"] if args.stages == "all" else [x.strip() for x in args.stages.split(",")]
    yolo = None
    if "detect" in want:
        from mangatranslator_amd.core.ml.yolo import YoloSegHip
        from oracle.yolo_ref import make_model as make_yolo       # seeded YOLOv8m-seg (the reference's yolo_1 geometry)
        ysd = None
        if rank == 0 or world == 1:
            ynet = make_yolo("m", 1, seed=3)
            with torch.no_grad():
                for l in range(3):
                    ynet.model[22].cv3[l][2].weight.mul_(0.05); ynet.model[22].cv3[l][2].bias.fill_(-1.0)
                    ynet.model[22].cv2[l][2].weight.mul_(0.1)
            ysd = ynet.state_dict()
        if world > 1:
            if rank != 0:
                ysd = {k: torch.empty_like(v) for k, v in make_yolo("m", 1, seed=0).state_dict().items()}
            ysd = broadcast_state_dict(ysd, rank, world, device)
        yolo = YoloSegHip(ysd, device=device, lib=lib, graph=not args.no_graph)
    upscaler = RCANUpscaler(sd, device=device, lib=lib, graph=not args.no_graph) if "upscale" in want else None
    sam = None
    if "segment" in want:
        from mangatranslator_amd.core.ml.sam2 import Sam2Hip
        from oracle.sam2_ref import make_config, make_model
        if rank == 0 or world == 1:
            m, sam_cfg = make_model("hiera_large", seed=11)     # facebook/sam2.1-hiera-large geometry, seeded weights
            sam_sd = {k: v for k, v in m.state_dict().items()}
            del m
        else:
            from transformers import Sam2Model
            sam_cfg = make_config("hiera_large")
            with torch.device("meta"):
                shapes = {k: v.shape for k, v in Sam2Model(sam_cfg).state_dict().items()}
            sam_sd = {k: torch.empty(s) for k, s in shapes.items()}
        if world > 1:
            sam_sd = broadcast_state_dict(sam_sd, rank, world, device)
        sam = Sam2Hip(sam_sd, sam_cfg, device=device, lib=lib, graph=not args.no_graph)
        del sam_sd

    # ---- synthetic pages, resident in HBM ----------------------------------------------------
    W_, H_ = args.width, args.height
    pool = 2
    pages, page_boxes = [], []
    for i in range(pool):
        pg, boxes, regions = make_page(rank * 1000 + i, W_, H_, bubbles=args.boxes)
        pages.append(torch.from_numpy(pg).to(device))
        page_boxes.append(boxes)
    torch.cuda.synchronize()

    stages = [st for st in ("detect", "segment", "upscale") if st in want]
    outs = [None, None, None]
    page_bgr = [pg.flip(-1).contiguous().cpu().numpy() for pg in pages]   # the detector's input is BGR (cv2 layout)
    yolo_conf = 0.6
    if yolo is not None:     # untimed calibration: seeded weights have arbitrary scores; let ~boxes anchors pass
        r0 = yolo(page_bgr[0], conf=0.0, imgsz=1600, max_det=1)[0]
        plan0, _ = yolo._plans[(H_, W_, 1600)]
        sc = plan0.decoded[:, 4].float().sort(descending=True).values
        yolo_conf = float(sc[min(3 * args.boxes, len(sc) - 1)])

    def step(i, timed=False):
        pg = pages[i % pool]
        if yolo is not None:     # letterbox @1600 -> YOLOv8m-seg -> decode -> NMS -> retina masks
            outs[2] = yolo(page_bgr[i % pool], conf=yolo_conf, imgsz=1600)[0]
        if sam is not None:      # detect is not built yet: the generator's ground-truth boxes stand in
            outs[0] = sam.segment(pg, page_boxes[i % pool])
        if upscaler is not None:
            outs[1] = upscaler.upscale_u8(pg)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pages_per_s = world * args.steps / dt

    result = {
        "metric": "pages/sec (detect+segment+inpaint+upscale) 1024x1536",
        "value": pages_per_s, "unit": "pages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{W_}x{H_} synthetic pages, one page per step per GPU, HBM-resident input",
                   "stages": stages,
                   "stages_not_built_yet": ["inpaint(FLUX)"],
                   "detector": {"arch": "YOLOv8m-seg @imgsz 1600 (1088x1600 letterbox)", "weights": "seeded random",
                                "note": "detections feed NMS + retina masks; SAM prompts are the generator's ground-truth boxes"} if yolo is not None else None,
                   "boxes_per_page": args.boxes,
                   "segmenter": {"arch": "SAM-2.1 Hiera-L (HF Sam2Model layout)", "weights": "seeded random"} if sam is not None else None,
                   "upscaler": {"arch": "RCAN", **rcan_cfg, "weights": "seeded random (no checkpoint offline)"} if upscaler is not None else None,
                   "parallelism": f"page-sharded x{world}, weights broadcast once over RCCL"},
    }

    if rank == 0 and sam is not None:
        # per-stage GPU time (HIP events on the launch stream), outside the timed region
        pre, enc, dec, post = sam.plans(args.boxes, H_, W_)
        result["config"]["segment_ms"] = {"preprocess": pre.time(5), "encoder": enc.time(5, graph=False),
                                          "decoder": dec.time(5, graph=False), "upsample_threshold": post.time(5)}
    if rank == 0 and yolo is not None:
        yp, _ = yolo._plans[(H_, W_, 1600)]
        result["config"]["detect_net_ms"] = yp.time(5, graph=False)
    if rank == 0 and upscaler is not None:
        result["config"]["upscale_ms"] = upscaler.plan_for(1, H_, W_).time(3, graph=False)
    if rank == 0 and upscaler is None:
        print(json.dumps(result))
    if rank == 0 and upscaler is not None:
        # ---- roofline of the dominant kernel: 3x3 conv 64->64 at page resolution ---------------
        u = upscaler.hp["unshuffle"]
        plan = upscaler.plan_for(1, H_, W_)
        idx = plan.labels.index("g0b0.conv1")
        iters = 20
        plan.time_range(idx, idx, 3)
        ms = plan.time_range(idx, idx, iters)
        work = upscaler.work(H_, W_)
        gbs = work["conv64_bytes"] / (ms * 1e-3) / 1e9
        tfs = work["conv64_flops"] / (ms * 1e-3) / 1e12
        result["roofline"] = {
            "kernel": "conv3x3_c64_kernel<f16> 64->64 @%dx%d" % (W_ // u, H_ // u),
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": ms, "launches_per_page": work["n_conv64"],
            "algorithmic_bytes_per_launch": work["conv64_bytes"],
            "mfma_tflops": tfs, "mfma_frac": tfs / MFMA_PEAK_TFLOPS,
        }
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, W_, H_)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sd, W_, H_):
    """The oracle (CPU restatement, fp32 torch) timed on a bounded crop of the same page."""
    from oracle.rcan_ref import load_ref
    from mangatranslator_amd.utils.synthetic_pages import make_page
    ref = load_ref(sd)
    cores = min(os.cpu_count() or 1, 32)     # small convolutions stop scaling (and thrash) beyond this
    torch.set_num_threads(cores)
    pg, _, _ = make_page(0, W_, H_)
    ch, cw = 96, 64
    x = torch.from_numpy(pg[:ch, :cw]).permute(2, 0, 1)[None].float() / 255.0
    ref(x[:, :, :16, :16])
    t0 = time.perf_counter()
    ref(x)
    dt = time.perf_counter() - t0
    frac = (ch * cw) / float(W_ * H_)
    return {"value": frac / dt, "unit": "pages/s", "cores": cores, "kind": "port",
            "sample": f"oracle RCAN (torch fp32 CPU) on a {cw}x{ch} crop = {frac:.4f} page, {dt:.2f} s, upscale stage only"}


if __name__ == "__main__":
    main()
