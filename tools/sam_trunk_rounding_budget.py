"""SAM-2.1 Hiera trunk: which 16-bit roundings are left once the weights are exact?  (CPU, HF transformers' fp32 Sam2Model = the oracle.)
The block forward below is transformers' `Sam2MultiScaleBlock.forward` with f16 roundings injected where the HIP graph rounds: GEMM operands
(LayerNorm outputs, qkv, attention output, MLP hidden — `ops`) and the residual stream after each branch (`stream`).  Everything outside the
trunk stays fp32.  DESIGN.md §3 quotes the Hiera-L figures (profiles/r04_sam_trunk_rounding_budget.log):
    python tools/sam_trunk_rounding_budget.py [hiera_large | small_test | tiny_test]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np, torch
from oracle import sam2_ref as sr
import sam2_checks as sc
from transformers.models.sam2 import modeling_sam2 as ms
torch.set_num_threads(8)
size = sys.argv[1] if len(sys.argv) > 1 else "hiera_large"
model, cfg = sr.make_model(size, 2)
h, w, nb = (1536, 1024, 8) if size == "hiera_large" else (300, 200, 3)
page = sc.make_page(h, w, 2)
rng = np.random.default_rng(7)
x0 = rng.uniform(0, 0.5 * w, nb); y0 = rng.uniform(0, 0.5 * h, nb)
boxes = np.stack([x0, y0, x0 + rng.uniform(0.2, 0.45, nb) * w, y0 + rng.uniform(0.2, 0.45, nb) * h], 1).astype(np.float32)
t = time.time(); ref = sr.run(model, page, boxes); print("oracle", round(time.time() - t, 1), "s")
sc.calibrate_logits(model, ref, 9.0)
ref = sr.run(model, page, boxes)
rl = ref["pred_masks"].float()
print("logit std", rl.std().item())
MODE = {"stream": False, "ops": False}
r16 = lambda t: t.to(torch.float16).float()
orig_fwd = ms.Sam2MultiScaleBlock.forward
def fwd(self, hidden_states, **kwargs):
    residual = hidden_states
    hidden_states = self.layer_norm1(hidden_states)
    if MODE["ops"]: hidden_states = r16(hidden_states)
    if self.dim != self.dim_out:
        residual = ms.do_pool(self.proj(hidden_states), self.query_stride)
        if MODE["stream"]: residual = r16(residual)
    window_size = self.window_size
    if self.window_size > 0:
        H, W = hidden_states.shape[1], hidden_states.shape[2]
        hidden_states, pad_hw = ms.window_partition(hidden_states, window_size)
    attn_output = self.attn(hidden_states=hidden_states, **kwargs)
    hidden_states = attn_output
    if self.query_stride:
        window_size = self.window_size // self.query_stride[0]
        H, W = residual.shape[1:3]
        pad_h = (-H) % window_size; pad_w = (-W) % window_size
        pad_hw = (H + pad_h, W + pad_w)
    if self.window_size > 0:
        hidden_states = ms.window_unpartition(hidden_states, window_size, pad_hw, (H, W))
    hidden_states = residual + hidden_states
    if MODE["stream"]: hidden_states = r16(hidden_states)
    ln = self.layer_norm2(hidden_states)
    if MODE["ops"]: ln = r16(ln)
    hidden_states = hidden_states + self.mlp(ln)
    if MODE["stream"]: hidden_states = r16(hidden_states)
    return hidden_states
ms.Sam2MultiScaleBlock.forward = fwd
hooks = []
for name, mod in model.named_modules():
    if "vision_encoder.backbone.blocks." in name:
        if name.endswith("attn.qkv"):
            hooks.append(mod.register_forward_hook(lambda m, i, o: r16(o) if MODE["ops"] else o))
        if name.endswith("attn.proj") or name.endswith("mlp.proj_out"):
            hooks.append(mod.register_forward_pre_hook(lambda m, i: (r16(i[0]),) if MODE["ops"] else None))
for label, mode in (("fp32 (sanity)", dict(stream=False, ops=False)), ("operands f16, stream f16 (hi + lo weights alone)", dict(stream=True, ops=True)),
                    ("operands f16, stream fp32", dict(stream=False, ops=True)), ("stream f16 only", dict(stream=True, ops=False))):
    MODE.update(mode)
    out = sr.run(model, page, boxes)
    d = out["pred_masks"].float() - rl
    mism = (out["masks"] != ref["masks"]).float().mean().item()
    print(f"{label:50s} logit rms err {d.pow(2).mean().sqrt().item():.5f} max {d.abs().max().item():.4f}  mask mismatch {mism:.2e}")
