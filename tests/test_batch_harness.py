"""The batch harness's page I/O vs goldens produced by running the REFERENCE `batch_translate_images`
(core/pipeline.py:2481-2733, sequential branch, `translate_and_render` replaced by a recorder): page list and order, output naming,
results dict, failed_paths.txt — for three output formats and both directory modes; plus the page writer and a 2-rank gloo run."""
import json
import os
import socket
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "batch_harness.json").read_text())


def _make_tree(root: Path):
    for rel in GOLD["tree"]:
        f = root / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        if f.suffix.lower() in (".png", ".jpg", ".jpeg", ".webp"):
            Image.new("RGB", (8, 8), (200, 10, 10)).save(f)
        else:
            f.write_text("x")


def _cfg(fmt):
    return types.SimpleNamespace(verbose=False, output=types.SimpleNamespace(output_format=fmt, jpeg_quality=95, png_compression=2))


@pytest.mark.parametrize("tag", list(GOLD["runs"]))
def test_batch_matches_reference(tmp_path, tag):
    from mangatranslator_amd.core.pipeline import batch_process_images
    g = GOLD["runs"][tag]
    root, odir = tmp_path / "in", tmp_path / "out"
    _make_tree(root)
    seen = []

    def process(page, path):
        seen.append(str(Path(path).relative_to(root)))
        assert page.mode == ("RGB" if g["fmt"] == "jpeg" or (g["fmt"] == "auto" and path.suffix.lower() in (".jpg", ".jpeg")) else "RGBA")
        if path.name in GOLD["fail"]:
            raise RuntimeError(f"boom: {path.name}")
        return page

    res = batch_process_images(root, _cfg(g["fmt"]), odir, preserve_structure=g["preserve"], process_image=process)
    assert seen == [s[0] for s in g["seen"]]
    assert res["success_count"] == g["success_count"] and res["error_count"] == g["error_count"] and res["errors"] == g["errors"]
    assert [str(Path(p).relative_to(root.resolve())) for p in res["failed_image_paths"]] == g["failed"]
    ff = Path(res["failed_paths_file"])
    assert ff.name == g["failed_file_name"] and ff.parent == odir
    assert [str(Path(l).relative_to(root.resolve())) for l in ff.read_text().split()] == g["failed_file_lines"]
    for rel_in, rel_out in g["seen"]:
        ok = Path(rel_in).name not in GOLD["fail"]
        assert (odir / rel_out).exists() == ok
        if ok:
            with Image.open(odir / rel_out) as im:
                assert im.size == (8, 8)


def test_save_image_with_compression(tmp_path):
    from mangatranslator_amd.core.image.image_utils import save_image_with_compression
    rgba = Image.fromarray(np.dstack([np.full((6, 5, 3), 40, np.uint8), np.zeros((6, 5, 1), np.uint8)]), "RGBA")
    save_image_with_compression(rgba, tmp_path / "a" / "x.jpg", jpeg_quality=500)
    with Image.open(tmp_path / "a" / "x.jpg") as im:          # transparent pixels are composited on white
        assert im.mode == "RGB" and min(im.getpixel((2, 2))) > 245
    save_image_with_compression(rgba, tmp_path / "x.png", png_compression=9)
    with Image.open(tmp_path / "x.png") as im:                # lossless; the colour type is reduced like oxipng reduces it (grey + alpha here)
        assert im.mode in ("RGBA", "LA") and np.array_equal(np.asarray(im.convert("RGBA")), np.asarray(rgba))
    save_image_with_compression(rgba, tmp_path / "x.webp")
    save_image_with_compression(rgba.convert("RGB"), tmp_path / "x.tiff")
    assert (tmp_path / "x.webp").exists() and (tmp_path / "x.png").exists() and not (tmp_path / "x.tiff").exists()


def test_empty_and_missing_dirs(tmp_path):
    from mangatranslator_amd.core.pipeline import batch_process_images, collect_image_files
    empty = {"success_count": 0, "error_count": 0, "errors": {}, "failed_image_paths": []}
    assert batch_process_images(tmp_path / "nope", _cfg("png"), tmp_path / "o") == empty
    (tmp_path / "e").mkdir()
    assert batch_process_images(tmp_path / "e", _cfg("png"), tmp_path / "o") == empty
    assert collect_image_files(tmp_path / "e") == []


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mangatranslator_amd.core.pipeline import batch_process_images
    root, odir = Path(tmp) / "in", Path(tmp) / "out"
    g = GOLD["runs"]["tree_png"]
    mine = []

    def process(page, path):
        mine.append(str(Path(path).relative_to(root)))
        if path.name in GOLD["fail"]:
            raise RuntimeError(f"boom: {path.name}")
        return page

    res = batch_process_images(root, _cfg("png"), odir, preserve_structure=True, process_image=process)
    order = [s[0] for s in g["seen"]]
    assert mine == order[rank::world]                                          # rank r takes pages r, r + G, ... of the batch order
    assert res["success_count"] == g["success_count"] and res["error_count"] == g["error_count"] and res["errors"] == g["errors"]
    assert [str(Path(p).relative_to(root.resolve())) for p in res["failed_image_paths"]] == g["failed"]
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_the_batch(tmp_path):
    _make_tree(tmp_path / "in")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g = GOLD["runs"]["tree_png"]
    written = sorted(str(p.relative_to(tmp_path / "out")) for p in (tmp_path / "out").rglob("*_translated.png"))
    assert written == sorted(s[1] for s in g["seen"] if Path(s[0]).name not in GOLD["fail"])
    assert (tmp_path / "out" / "failed_paths.txt").exists()


def test_save_queue_is_bounded_and_io_is_accounted(tmp_path, monkeypatch):
    """finished pages never pile up behind the codec threads: at most 2 x io_threads saves are pending (a 512-page batch of
    4096x6144 results could exhaust host RAM), and the results carry the decode / encode / process times of the run"""
    import time
    from mangatranslator_amd.core import pipeline
    from mangatranslator_amd.core.image import image_utils
    root, odir = tmp_path / "in", tmp_path / "out"
    root.mkdir()
    for i in range(14):
        Image.new("RGB", (16, 16), (i, 2 * i, 3 * i)).save(root / f"p{i:02d}.png")
    real = image_utils.save_image_with_compression
    live, peak = [0], [0]

    def slow_save(*a, **k):
        live[0] += 1
        peak[0] = max(peak[0], live[0])
        time.sleep(0.05)                                # the encoder is far slower than the page loop
        try:
            return real(*a, **k)
        finally:
            live[0] -= 1

    monkeypatch.setattr(image_utils, "save_image_with_compression", slow_save)
    res = pipeline.batch_process_images(root, _cfg("png"), odir, process_image=lambda page, path: page, io_threads=2)
    assert res["success_count"] == 14 and res["error_count"] == 0
    io = res["io"]
    assert io["pages"] == 14 and 1 <= io["max_pending_saves"] <= 4
    assert io["encode_s"] >= 14 * 0.05 and io["wait_for_save_slot_s"] > 0.05          # the loop did wait for save slots
    assert io["decode_ms_per_page"] > 0 and io["encode_ms_per_page"] >= 50 and "process_ms_per_page" in io
    assert sorted(p.name for p in odir.iterdir()) == [f"p{i:02d}_translated.png" for i in range(14)]
    for i in (0, 13):                                                                  # decoded pixels survive the writer
        assert Image.open(odir / f"p{i:02d}_translated.png").convert("RGB").getpixel((3, 3)) == (i, 2 * i, 3 * i)


def test_two_pages_in_flight_give_the_sequential_results(tmp_path):
    """`batch_process_images(process_front=, process_back=)` (round 4): page i + 1's front half runs on a worker thread while page i's back
    half runs on the caller's; the saved files, their order in the results and the failed pages equal those of the sequential
    `process_image = back(front(.))` run — incl. a page whose FRONT half raises and one whose BACK half raises — and the overlap is real
    (a back half sees the next page's front half already started)."""
    import threading
    import time
    from mangatranslator_amd.core.pipeline import batch_process_images
    root = tmp_path / "in"
    root.mkdir()
    n = 7
    for i in range(n):
        Image.new("RGB", (16, 12), (10 * i, 20, 30)).save(root / f"p{i:02d}.png")
    started, overlapped = {}, []
    lock = threading.Lock()

    def front(page, path):
        i = int(path.stem[1:])
        with lock:
            started[i] = True
        if i == 2:
            raise RuntimeError("front of page 2")
        time.sleep(0.02)
        return {"i": i, "page": page, "thread": threading.get_ident()}

    def back(state):
        i = state["i"]
        time.sleep(0.05)
        with lock:
            overlapped.append(bool(started.get(i + 1)) or i + 1 >= n)
        if i == 4:
            raise ValueError("back of page 4")
        out = state["page"].copy()
        out.putpixel((0, 0), (i, i, i))
        return out

    pipelined = batch_process_images(root, _cfg("png"), tmp_path / "out_p", process_front=front, process_back=back, io_threads=2)
    main = threading.get_ident()
    sequential = batch_process_images(root, _cfg("png"), tmp_path / "out_s", process_image=lambda page, path: back(front(page, path)), io_threads=2)
    for k in ("success_count", "error_count", "errors"):
        assert pipelined[k] == sequential[k], k
    assert pipelined["success_count"] == n - 2 and set(pipelined["errors"]) == {"p02.png", "p04.png"}
    assert [Path(p).name for p in pipelined["failed_image_paths"]] == ["p02.png", "p04.png"]
    names = sorted(f.name for f in (tmp_path / "out_p").glob("*.png"))
    assert names == sorted(f.name for f in (tmp_path / "out_s").glob("*.png")) and len(names) == n - 2
    for name in names:
        assert np.array_equal(np.asarray(Image.open(tmp_path / "out_p" / name)), np.asarray(Image.open(tmp_path / "out_s" / name)))
    assert sum(overlapped[: n - 2]) >= n - 3                      # (first run only) nearly every back half ran beside the next front half
    assert pipelined["io"]["pages_in_flight"] == 2 and sequential["io"]["pages_in_flight"] == 1
    with pytest.raises(ValueError):
        batch_process_images(root, _cfg("png"), tmp_path / "o", process_front=front)
    assert main == threading.get_ident()


def test_several_front_halves_in_flight(tmp_path):
    """`front_workers=3`: up to three front halves run at once, each inside `front_context(slot)` with a slot no other running front half
    holds; files, order, failures equal the sequential run's"""
    import contextlib
    import threading
    import time
    from mangatranslator_amd.core.pipeline import batch_process_images
    root = tmp_path / "in"
    root.mkdir()
    n = 10
    for i in range(n):
        Image.new("RGB", (16, 12), (10 * i, 20, 30)).save(root / f"p{i:02d}.png")
    lock = threading.Lock()
    tls = threading.local()
    held, peak, slots_seen, clashes = set(), [0], set(), []

    @contextlib.contextmanager
    def ctx(slot):
        with lock:
            if slot in held:
                clashes.append(slot)
            held.add(slot)
            slots_seen.add(slot)
            peak[0] = max(peak[0], len(held))
        tls.slot = slot
        try:
            yield
        finally:
            tls.slot = None
            with lock:
                held.discard(slot)

    def front(page, path):
        i = int(path.stem[1:])
        assert getattr(tls, "slot", None) is not None          # the front half runs INSIDE its context, on the thread that entered it
        time.sleep(0.06)
        if i == 3:
            raise RuntimeError("front of page 3")
        return {"i": i, "page": page}

    def back(state):
        i = state["i"]
        if i == 6:
            raise ValueError("back of page 6")
        out = state["page"].copy()
        out.putpixel((0, 0), (i, i, i))
        return out

    t0 = time.perf_counter()
    res = batch_process_images(root, _cfg("png"), tmp_path / "out_p", process_front=front, process_back=back, io_threads=2, front_workers=3,
                               front_context=ctx)
    wall = time.perf_counter() - t0
    seq = batch_process_images(root, _cfg("png"), tmp_path / "out_s", io_threads=2,
                               process_image=lambda page, path: back({"i": int(path.stem[1:]), "page": page}) if int(path.stem[1:]) != 3 else (_ for _ in ()).throw(RuntimeError("front of page 3")))
    for k in ("success_count", "error_count", "errors"):
        assert res[k] == seq[k], k
    assert [Path(p).name for p in res["failed_image_paths"]] == ["p03.png", "p06.png"]
    names = sorted(f.name for f in (tmp_path / "out_p").glob("*.png"))
    assert names == sorted(f.name for f in (tmp_path / "out_s").glob("*.png")) and len(names) == n - 2
    for name in names:
        assert np.array_equal(np.asarray(Image.open(tmp_path / "out_p" / name)), np.asarray(Image.open(tmp_path / "out_s" / name)))
    assert not clashes and slots_seen == {0, 1, 2} and peak[0] == 3
    assert res["io"]["pages_in_flight"] == 4
    assert wall < 0.06 * n * 1.5, wall                            # (loose: `peak` above is the proof of overlap; ten 60 ms front halves one at a time take 0.6 s before any I/O)


@pytest.mark.parametrize("workers", [1, 2, 4])
def test_harness_under_random_delays_and_failures(tmp_path, workers):
    """seeded stress: front / back halves with random durations, random failures in either half and io_threads 1..3 — whatever the
    interleaving, success / error counts, the failed list IN BATCH ORDER and every written page equal the sequential run's"""
    import random
    import time
    from mangatranslator_amd.core.pipeline import batch_process_images
    root = tmp_path / "in"
    root.mkdir()
    n = 17
    for i in range(n):
        Image.new("RGB", (10, 6), (i, 255 - i, 7)).save(root / f"q{i:02d}.png")
    for seed in range(4):
        rng = random.Random(1000 * workers + seed)
        plan = {i: (rng.random() * 0.012, rng.random() * 0.012, rng.random() < 0.15, rng.random() < 0.15) for i in range(n)}

        def front(page, path):
            i = int(path.stem[1:])
            time.sleep(plan[i][0])
            if plan[i][2]:
                raise RuntimeError(f"front {i}")
            return i, page

        def back(state):
            i, page = state
            time.sleep(plan[i][1])
            if plan[i][3]:
                raise ValueError(f"back {i}")
            out = page.copy()
            out.putpixel((1, 1), (i, seed, workers))
            return out

        tag = f"{workers}_{seed}"
        got = batch_process_images(root, _cfg("png"), tmp_path / f"p{tag}", process_front=front, process_back=back, io_threads=1 + seed % 3, front_workers=workers,
                                   front_context=None if workers == 1 else (lambda slot: __import__("contextlib").nullcontext()))
        want = batch_process_images(root, _cfg("png"), tmp_path / f"s{tag}", process_image=lambda page, path: back(front(page, path)), io_threads=2)
        for k in ("success_count", "error_count", "errors", "failed_image_paths"):
            assert got[k] == want[k], (tag, k)
        names = sorted(f.name for f in (tmp_path / f"p{tag}").glob("*.png"))
        assert names == sorted(f.name for f in (tmp_path / f"s{tag}").glob("*.png")) and len(names) == got["success_count"]
        for name in names:
            assert np.array_equal(np.asarray(Image.open(tmp_path / f"p{tag}" / name)), np.asarray(Image.open(tmp_path / f"s{tag}" / name))), (tag, name)
