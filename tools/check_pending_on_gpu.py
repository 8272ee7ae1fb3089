"""Hardware checks of what round 3 built and verified on the CPU simulator only (the round's GPU minutes had run out):

  * gemm256_f8_glu_kernel   — fp8 GEMM with SwiGLU + MX quantisation in its epilogue (mtx_gemm_args.glu_*)
  * attn_mma32_q8_kernel / attn_merge_q8_kernel — long-sequence attention with MX fp8 output (mtx_attn_args.q8)
  * a Klein-geometry denoising step with both (no mtx_quantize_mx launch left) against the step with separate quantiser launches

Each check compares the fused form with the launches it replaces, byte for byte, at FLUX.2-Klein's real shapes.  Run it FIRST next round
(`python tools/check_pending_on_gpu.py`), then A/B `python bench.py --config 5 [--glu-epilogue]`; only then make the fusions the default.
Test infrastructure: nothing under mangatranslator_amd/ imports this file."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import torch  # noqa: E402

from mangatranslator_amd.hip import abi  # noqa: E402
from mangatranslator_amd.hip.lib import get_library  # noqa: E402
import flux2_checks as fc  # noqa: E402
import op_checks as oc  # noqa: E402


def main():
    assert torch.cuda.is_available(), "needs the GPU"
    lib = get_library()
    lib.init(0)
    steps = [
        ("gated GEMM epilogue, double-block MLP-in (8000 x 18432 x 3072)", lambda: oc.check_gemm_f8_glu(lib, abi.BF16, m=8000, col0=0, hid=9216, k=3072, row_off=512)),
        ("gated GEMM epilogue, single-block fused projection (8512 x 27648 x 3072, qkv columns in front)",
         lambda: oc.check_gemm_f8_glu(lib, abi.BF16, m=8512, col0=9216, hid=9216, k=3072, q_col_off=3072, seed=1)),
        ("gated GEMM epilogue, ragged text stream (512 rows)", lambda: oc.check_gemm_f8_glu(lib, abi.BF16, m=512, col0=0, hid=9216, k=3072, seed=2)),
        ("attention with MX fp8 output, T = 8512, 24 heads (key-split tail blocks included)", lambda: oc.check_attention_q8(lib, abi.BF16, heads=24, sq=8512, sk=8512)),
        ("attention with MX fp8 output into the single blocks' concatenation (column offset 0 of a wider buffer)",
         lambda: oc.check_attention_q8(lib, abi.BF16, heads=24, sq=8652, sk=8652, extra_cols=9216, seed=1)),
        ("Klein-geometry step (d = 3072, 1 + 2 blocks, T = 2064) without a quantiser launch",
         lambda: fc.check_no_quantiser_step(lib, "cuda:0", h2=32, w2=32, t_txt=16, d=3072, heads=24, axes_dim=(32, 32, 32, 32), layers=1, single_layers=2,
                                            joint_dim=7680)),
    ]
    failed = 0
    for name, fn in steps:
        t = time.perf_counter()
        try:
            fn()
            torch.cuda.synchronize()
            print(f"ok      {name}   ({time.perf_counter() - t:.1f} s)")
        except Exception as e:      # noqa: BLE001 — report every check
            failed += 1
            print(f"FAILED  {name}: {str(e)[:300]}")
    raise SystemExit(1 if failed else 0)


if __name__ == "__main__":
    main()
