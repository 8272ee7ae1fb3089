"""Do two detectors' hipGraph replays, submitted on their own streams, share the chip?  Wall time of (submit A, submit B, sync) against
(submit A, sync, submit B, sync) for the panel (YOLO11-L) and outside-text (YOLO12x) networks at imgsz 640."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
from oracle import yolo11_ref as y11

lib = get_library(); lib.init(0)
page = (np.random.default_rng(0).random((1536, 1024, 3)) * 255).astype(np.uint8)
A = Yolo11Hip(y11.make_model("11", "l", 1, False, seed=1).state_dict(), device="cuda:0", lib=lib)
B = Yolo11Hip(y11.make_model("12", "x", 1, False, seed=2).state_dict(), device="cuda:0", lib=lib)
for m in (A, B):
    m(page, conf=0.9, imgsz=640)
torch.cuda.synchronize()


def wall(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def together():
    ta, tb = A.submit(page, conf=0.9, imgsz=640), B.submit(page, conf=0.9, imgsz=640)
    torch.cuda.synchronize()
    A.collect(ta); B.collect(tb)


def apart():
    ta = A.submit(page, conf=0.9, imgsz=640); torch.cuda.synchronize(); A.collect(ta)
    tb = B.submit(page, conf=0.9, imgsz=640); torch.cuda.synchronize(); B.collect(tb)


def only(m):
    def f():
        t = m.submit(page, conf=0.9, imgsz=640); torch.cuda.synchronize(); m.collect(t)
    return f


print(f"A alone {wall(only(A)):.2f} ms, B alone {wall(only(B)):.2f} ms, one after the other {wall(apart):.2f} ms, submitted together {wall(together):.2f} ms", flush=True)
pa, pb = A._plans[(1536, 1024, 640)][0], B._plans[(1536, 1024, 640)][0]
print(f"graph replays timed alone: A {pa.time(10, graph=True):.2f} ms, B {pb.time(10, graph=True):.2f} ms; eager A {pa.time(10):.2f} ms, B {pb.time(10):.2f} ms")
# raw: the two graphs on two streams, no uploads, no post-processing
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def raw_together():
    with torch.cuda.stream(sa):
        pa.run(graph=True)
    with torch.cuda.stream(sb):
        pb.run(graph=True)


def raw_apart():
    with torch.cuda.stream(sa):
        pa.run(graph=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        pb.run(graph=True)
    torch.cuda.synchronize()


def raw_eager_together():
    with torch.cuda.stream(sa):
        pa.run()
    with torch.cuda.stream(sb):
        pb.run()


print(f"graphs only: apart {wall(raw_apart):.2f} ms, together {wall(raw_together):.2f} ms; eager launches on two streams together {wall(raw_eager_together):.2f} ms")
