"""CPU tier, world_size 2 over gloo: the multi-GPU path of the hot path — static page sharding, the start-up
weight broadcast (flat buffer for small checkpoints, per-tensor for FLUX) and the host-side result gather."""
import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mangatranslator_amd.core.ml import flux as fx
    from mangatranslator_amd.core.ml.model_manager import broadcast_state_dict
    from mangatranslator_amd.core.pipeline import process_pages_sharded, shard_pages

    # 1. checkpoint broadcast: rank 0 owns the real tensors, rank 1 only a template
    torch.manual_seed(100 + rank)
    sd = {"a.weight": torch.randn(4, 3, 3, 3), "a.bias": torch.randn(4), "scalar": torch.tensor(float(rank + 5))}
    got = broadcast_state_dict(sd if rank == 0 else None, template=None if rank == 0 else {k: torch.empty_like(v) for k, v in sd.items()})
    torch.manual_seed(100)
    want = {"a.weight": torch.randn(4, 3, 3, 3), "a.bias": torch.randn(4), "scalar": torch.tensor(5.0)}
    assert all(torch.equal(got[k], want[k]) for k in want), "flat broadcast mismatch"

    # 2. FLUX per-tensor broadcast: different seeds per rank, rank 0's values must win
    shapes = {"x.weight": (6, 5), "x.bias": (6,), "norm_q.weight": (8,)}
    prov = fx.synthetic_provider(shapes, "cpu", seed=7 + rank, broadcast=True)
    mine = {k: prov(k) for k in shapes}
    ref = fx.synthetic_provider(shapes, "cpu", seed=7, broadcast=False)
    assert all(torch.equal(mine[k], ref(k)) for k in shapes), "per-tensor broadcast mismatch"

    # 2b. the same set in buckets: several tensors per collective, a tensor larger than the bucket alone in its own, both dtypes
    import mangatranslator_amd.core.ml.flux as fxm
    calls = []
    real_bcast = dist.broadcast
    dist.broadcast = lambda t, src=0: (calls.append(t.numel()), real_bcast(t, src=src))[1]
    try:
        big = {"w0": (40, 16), "w1": (8, 16), "w2": (8, 16), "b0": (40,), "b1": (8,), "w3": (8, 16)}
        got = fxm.broadcast_in_buckets([(n, s, torch.bfloat16 if len(s) == 2 else torch.float32) for n, s in big.items()],
                                       lambda n, v: v.copy_(torch.full(v.shape, float(len(n) + sum(map(ord, n)) % 7 + (0 if rank == 0 else 100)))), "cpu", bucket_bytes=600)
    finally:
        dist.broadcast = real_bcast
    assert calls == [640, 256, 128, 128], calls                     # bf16: 1280 B alone, 2 x 256 B together (a third would pass 600 B), the last alone; fp32: one bucket of two 256-byte entries
    assert all(float(got[n].float().max()) == float(len(n) + sum(map(ord, n)) % 7) and tuple(got[n].shape) == s for n, s in big.items())
    # 2b'. an odd-length 16-bit entry ahead of a matrix (the real VAE's bf16 decoder.conv_out.bias, 6 bytes: ADVICE r05) must not shift the matrix
    # off a 16-byte boundary — the GEMM kernels take 16-byte vector loads and LDS-DMA from these views
    odd = [("conv_out.bias", (3,), torch.bfloat16), ("mid.to_q.weight", (16, 16), torch.bfloat16), ("x", (5,), torch.bfloat16), ("mid.to_k.weight", (16, 16), torch.bfloat16)]
    got = fxm.broadcast_in_buckets(odd, lambda n, v: v.copy_(torch.full(v.shape, float(len(n)))), "cpu")
    assert all(got[n].data_ptr() % 16 == 0 for n, _, _ in odd), [got[n].data_ptr() % 256 for n, _, _ in odd]
    assert all(float(got[n].float().min()) == float(len(n)) == float(got[n].float().max()) for n, _, _ in odd)

    # 2c. worker threads under a process group: the ranks' threads ask the loaders in DIFFERENT orders; inside
    # `thread_local_reads` every call reads its file itself — no collective, so nothing can hang or cross-wire — while the same calls on
    # the main thread (one order on every rank) go through the status + tensor broadcasts
    import threading
    from safetensors.torch import save_file
    from mangatranslator_amd.core.ml import model_manager as mm
    mm.ModelManager._instance = None; mm._model_manager = None
    man = mm.get_model_manager()
    files = []
    for i in range(3):
        f = Path(out_dir) / f"ckpt{i}.safetensors"
        if rank == 0:
            save_file({"w": torch.full((4, 4), float(i))}, str(f))
        files.append(f)
    dist.barrier()
    main_thread = [man._read_safetensors(f)["w"][0, 0].item() for f in files]          # collective path, same order everywhere
    assert main_thread == [0.0, 1.0, 2.0]
    got_t = {}

    def worker():
        order = files if rank == 0 else files[::-1]
        with man.thread_local_reads():
            for f in order:
                got_t[f.name] = man._read_safetensors(f)["w"][0, 0].item()
    th = threading.Thread(target=worker); th.start(); th.join(timeout=60)
    assert not th.is_alive(), "a worker-thread load entered a collective"
    assert got_t == {"ckpt0.safetensors": 0.0, "ckpt1.safetensors": 1.0, "ckpt2.safetensors": 2.0}
    dist.barrier()

    # 3. sharded batch loop + gather
    pages = [f"ch2/010.jpg", "ch2/001.jpg", "ch10/001.jpg", "P1.png", "p10.png", "p2.png", "bad_3.png"]
    mine_pages = shard_pages(pages, rank, world)
    done = []

    def process(p):
        if "bad" in p.name:
            raise ValueError("decode failed")
        done.append(str(p))

    merged = process_pages_sharded(pages, process)
    assert merged["success_count"] == 6 and merged["error_count"] == 1
    assert merged["failed_image_paths"] == ["bad_3.png"] and merged["errors"] == {"bad_3.png": "decode failed"}
    assert len(done) + (1 if any("bad" in str(p) for p in mine_pages) else 0) == len(mine_pages)
    (Path(out_dir) / f"rank{rank}.txt").write_text("\n".join(done))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "rank0.txt").read_text().split("\n")
    r1 = (tmp_path / "rank1.txt").read_text().split("\n")
    assert not set(r0) & set(r1) and len(r0) + len(r1) == 6          # disjoint cover of the good pages
    from mangatranslator_amd.core.pipeline import shard_pages
    pages = ["ch2/010.jpg", "ch2/001.jpg", "ch10/001.jpg", "P1.png", "p10.png", "p2.png", "bad_3.png"]
    order = shard_pages(pages, 0, 1)
    assert order[0::2] == shard_pages(pages, 0, 2) and order[1::2] == shard_pages(pages, 1, 2)


def _worker_failed_read(rank, world, port, out_dir):
    """a checkpoint that is missing: the first ask is a collective (every rank raises the same ModelError), later asks raise at once —
    counted by wrapping the status broadcast"""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.utils.exceptions import ModelError
    man = mm.get_model_manager()
    calls = []
    real = mm.broadcast_status
    mm.broadcast_status = lambda err, src=0: (calls.append(1), real(err, src))[1]
    missing = Path(out_dir) / "absent.safetensors"
    for _ in range(3):
        try:
            man._read_safetensors(missing)
            raise AssertionError("a missing checkpoint loaded")
        except ModelError as e:
            assert "not found" in str(e)
    assert len(calls) == 1, calls
    if rank == 0:
        from safetensors.torch import save_file
        save_file({"w": torch.ones(2)}, str(missing))
    dist.barrier()
    man.forget_failed_loads()
    assert float(man._read_safetensors(missing)["w"].sum()) == 2.0 and len(calls) == 2
    dist.destroy_process_group()


def test_failed_collective_load_is_remembered(tmp_path):
    mp.spawn(_worker_failed_read, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
