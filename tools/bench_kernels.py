"""GPU micro-benchmarks of the two MFMA-bound kernels at FLUX shapes (HIP-event timing via mtx_plan_time).
usage: python tools/bench_kernels.py attn T [heads]   |   gemm M N K"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mangatranslator_amd.hip import abi
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.hip.plan import PlanBuilder

lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")


def attn(T, heads=24, d=128, iters=10):
    pb = PlanBuilder(lib, dev, abi.BF16)
    D = heads * d
    qkv = pb.buf((T, 3 * D), torch.bfloat16); qkv.normal_()
    o = pb.buf((T, D), torch.bfloat16)
    # the FLUX graph's form: q pre-multiplied by scale * log2(e) (MTX_ATTN_Q_PRESCALED)
    qkv[:, :D] *= d ** -0.5 * 1.4426950408889634
    pb.attention(qkv, qkv, qkv, o, 1, heads, T, T, d, (0, 3 * D, d), (0, 3 * D, d), (0, 3 * D, d), (0, D, d), d ** -0.5, k_off=D, v_off=2 * D,
                 q_prescaled=True)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    ms = plan.time(iters)
    print(f"attn T={T} heads={heads}: {ms:.3f} ms  {4 * T * T * D / ms / 1e9:.0f} TFLOP/s")


def gemm(M, N, K, iters=20):
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((M, K), torch.bfloat16); a.normal_()
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5)
    pb.gemm(a, w, M, N, K)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    plan.time(5)
    ms = min(plan.time(iters) for _ in range(3))
    print(f"gemm M={M} N={N} K={K}: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s")


def attn_ab(T, heads=24, d=128, rounds=5, iters=20):
    import os
    pb = PlanBuilder(lib, dev, abi.BF16)
    D = heads * d
    qkv = pb.buf((T, 3 * D), torch.bfloat16); qkv.normal_()
    o = pb.buf((T, D), torch.bfloat16)
    pb.attention(qkv, qkv, qkv, o, 1, heads, T, T, d, (0, 3 * D, d), (0, 3 * D, d), (0, 3 * D, d), (0, D, d), d ** -0.5, k_off=D, v_off=2 * D)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    res = {"mma32": [], "bias": [], "nomax": []}
    for r in range(rounds):
        for mode in res:
            os.environ["MTX_ATTN_KERNEL"] = mode
            plan.time(3)
            res[mode].append(plan.time(iters))
    for mode, v in res.items():
        print(f"attn T={T} {mode}: best {min(v):.3f} ms ({4 * T * T * D / min(v) / 1e9:.0f} TF/s), median {sorted(v)[len(v) // 2]:.3f} ms")


def gemm_abl(M, N, K, rounds=4, iters=20):
    """DMA placement variants of the one-barrier loop (MTX_GEMM_ABL), interleaved in one process"""
    import os
    os.environ["MTX_GEMM256_SCHED"] = "lockstep"
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((M, K), torch.bfloat16); a.normal_()
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5)
    pb.gemm(a, w, M, N, K)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    res = {"0": [], "5": [], "1": []}
    for r in range(rounds):
        for mode in res:
            os.environ["MTX_GEMM_ABL"] = mode
            plan.time(3)
            res[mode].append(plan.time(iters))
    os.environ["MTX_GEMM_ABL"] = "0"
    for mode, v in res.items():
        print(f"gemm {M}x{N}x{K} abl={mode}: best {min(v):.3f} ms ({2 * M * N * K / min(v) / 1e9:.0f} TF/s)")


def gemm_ab(M, N, K, rounds=5, iters=20):
    """interleaved A/B of the two 256-tile schedules in one process (run-to-run clock drift is ~10 %)"""
    import os
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((M, K), torch.bfloat16); a.normal_()
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5)
    pb.gemm(a, w, M, N, K)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    res = {"buf": [], "pingpong": [], "lockstep": []}
    for r in range(rounds):
        for mode in res:
            os.environ["MTX_GEMM256_SCHED"] = mode
            plan.time(3)
            res[mode].append(plan.time(iters))
    for mode, v in res.items():
        print(f"gemm {M}x{N}x{K} {mode}: best {min(v):.3f} ms ({2 * M * N * K / min(v) / 1e9:.0f} TF/s), median {sorted(v)[len(v) // 2]:.3f} ms")


def gemm_clamp(M, N, K, rounds=5, iters=20):
    """interleaved A/B of the zero-block select (MTX_GEMM_CLAMP=0) vs clamped rows in the 256-tile loops"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((M, K), torch.bfloat16); a.normal_()
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5)
    pb.gemm(a, w, M, N, K)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    res = {"0": [], "9": [], "10": []}
    for r in range(rounds):
        for mode in res:
            os.environ["MTX_GEMM_ABL"] = mode
            plan.time(3)
            res[mode].append(plan.time(iters))
    for mode, v in res.items():
        print(f"gemm {M}x{N}x{K} abl={mode}: best {min(v):.3f} ms ({2 * M * N * K / min(v) / 1e9:.0f} TF/s), median {sorted(v)[len(v) // 2]:.3f} ms")


def gemm_mintiles(M, N, K, rounds=5, iters=20):
    """128-tile kernel (default below 160 tiles) vs the 256-tile descriptor-DMA kernel on few-tile problems"""
    pb = PlanBuilder(lib, dev, abi.BF16)
    a = pb.buf((M, K), torch.bfloat16); a.normal_()
    w = pb.buf((N, K), torch.bfloat16); w.normal_(0, K ** -0.5)
    pb.gemm(a, w, M, N, K)
    plan = pb.build(); plan.run(); torch.cuda.synchronize()
    res = {"160": [], "1": []}
    for r in range(rounds):
        for mode in res:
            os.environ["MTX_GEMM256_MIN_TILES"] = mode
            plan.time(3)
            res[mode].append(plan.time(iters))
    os.environ.pop("MTX_GEMM256_MIN_TILES")
    for mode, v in res.items():
        print(f"gemm {M}x{N}x{K} min_tiles={mode}: best {min(v):.4f} ms ({2 * M * N * K / min(v) / 1e9:.0f} TF/s)")


if __name__ == "__main__":
    args = sys.argv[1:]
    while args:
        if args[0] == "attn":
            attn(int(args[1])); args = args[2:]
        elif args[0] == "attn_ab":
            attn_ab(int(args[1])); args = args[2:]
        elif args[0] == "abl":
            gemm_abl(int(args[1]), int(args[2]), int(args[3])); args = args[4:]
        elif args[0] == "clamp":
            gemm_clamp(int(args[1]), int(args[2]), int(args[3])); args = args[4:]
        elif args[0] == "mintiles":
            gemm_mintiles(int(args[1]), int(args[2]), int(args[3])); args = args[4:]
        elif args[0] == "ab":
            gemm_ab(int(args[1]), int(args[2]), int(args[3])); args = args[4:]
        else:
            gemm(int(args[1]), int(args[2]), int(args[3])); args = args[4:]

