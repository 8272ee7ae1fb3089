"""Start-up weight broadcast of the FLUX.1-Kontext set (11.9 B parameters, 23.8 GB bf16) between TWO ranks that share ONE device (the only
multi-rank arrangement a one-GPU box offers; gloo, staged through host memory — RCCL refuses two ranks on one GPU): bucketed
(core/ml/flux.py broadcast_in_buckets, <= 1 GiB per collective) against tensor by tensor (rounds 1-4).  What it shows is the number of
collectives and the fixed cost per collective; xGMI bandwidth is not in it.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/time_weight_broadcast.py [blocks-fraction]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mangatranslator_amd.core.ml import flux as fx  # noqa: E402


def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    cfg = dict(fx.KONTEXT_DIT_CFG)
    cfg["layers"], cfg["single_layers"] = max(1, int(cfg["layers"] * frac)), max(1, int(cfg["single_layers"] * frac))
    shapes = fx.dit_param_shapes(cfg)
    nbytes = sum(int(torch.Size(s).numel()) * (2 if len(s) >= 2 else 4) for s in shapes.values())
    calls = [0]
    real = dist.broadcast

    def counted(t, src=0, **kw):
        calls[0] += 1
        return real(t, src=src, **kw)
    dist.broadcast = counted
    out = {}
    # bucketed (the provider's form)
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    prov = fx.synthetic_provider(shapes, dev, seed=21, broadcast=True)
    torch.cuda.synchronize(); dist.barrier()
    out["bucketed"] = (time.perf_counter() - t0, calls[0])
    chk = float(prov(next(iter(shapes))).float().abs().sum())
    del prov
    torch.cuda.empty_cache()
    # tensor by tensor (what rounds 1-4 did)
    calls[0] = 0
    gen = fx.synthetic_provider(shapes, dev, seed=21, broadcast=False)
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for name in shapes:
        t = gen(name) if rank == 0 else torch.empty(shapes[name], dtype=torch.bfloat16 if len(shapes[name]) >= 2 else torch.float32, device=dev)
        h = t.cpu()
        dist.broadcast(h, src=0)
        t = h.to(dev)
    torch.cuda.synchronize(); dist.barrier()
    out["per_tensor"] = (time.perf_counter() - t0, calls[0])
    if rank == 0:
        print(f"FLUX.1-Kontext DiT weights x{frac}: {len(shapes)} tensors, {nbytes / 1e9:.1f} GB; two ranks, one device, gloo (host-staged); first tensor |sum| {chk:.1f}")
        for k, (s, n) in out.items():
            print(f"   {k:11s} {s:7.1f} s  {n:5d} collectives  {nbytes / s / 1e9:.2f} GB/s")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
