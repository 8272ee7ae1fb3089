"""What stands between this package and the reference's real checkpoints: SDNQ-packed FLUX weights, the prompt
embeddings the pipelines take as an input, and upscaler hyper-parameters read from the file instead of assumed.

None of the real files is reachable from the build image, so these tests are about self-consistency and plumbing: a writer that follows
the SAME published SDNQ layout as the reader (core/ml/sdnq.py says what that proves and what it does not), tiny seeded HF encoders for
the exporter, a header written by safetensors itself for the RCAN reader."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from mangatranslator_amd.core.ml import sdnq
from mangatranslator_amd.utils.exceptions import ModelError


def _quantize_like_sdnq(w, bits, group, asymmetric, rank):
    """the published scheme, written independently of the reader: optional rank-r SVD part taken out first, then per-group affine
    (asymmetric: zero_point = group minimum, scale = range / (2^bits - 1)) or symmetric (scale = max |w| / (2^(bits-1) - 1)) rounding"""
    n, k = w.shape
    out = {}
    if rank:
        u, s, vh = torch.linalg.svd(w.double(), full_matrices=False)
        up, down = (u[:, :rank] * s[:rank]).float(), vh[:rank].float()
        out["svd_up"], out["svd_down"] = up.to(torch.bfloat16), down.to(torch.bfloat16)
        w = w - out["svd_up"].float() @ out["svd_down"].float()
    g = w.reshape(n, k // group, group)
    if asymmetric:
        lo, hi = g.amin(-1, keepdim=True), g.amax(-1, keepdim=True)
        scale = ((hi - lo) / (2 ** bits - 1)).clamp_min(1e-8)
        q = torch.round((g - lo) / scale).clamp(0, 2 ** bits - 1)
        out["zero_point"] = lo.to(torch.bfloat16)
        scale = scale.to(torch.bfloat16)
        q = torch.round((g - out["zero_point"].float()) / scale.float()).clamp(0, 2 ** bits - 1)
    else:
        scale = (g.abs().amax(-1, keepdim=True) / (2 ** (bits - 1) - 1)).clamp_min(1e-8).to(torch.bfloat16)
        q = torch.round(g / scale.float()).clamp(-(2 ** (bits - 1)), 2 ** (bits - 1) - 1) + 2 ** (bits - 1)
    out["scale"] = scale
    out["weight"] = sdnq.pack_bits(q.to(torch.uint8), bits)
    return out


@pytest.mark.parametrize("bits,group,asym,rank", [(4, 32, True, 8), (4, 64, True, 0), (4, 128, False, 0), (8, 256, True, 4), (2, 16, True, 0)])
def test_sdnq_round_trip(bits, group, asym, rank):
    g = torch.Generator().manual_seed(bits * 100 + group)
    w = torch.randn(48, 256, generator=g) * torch.exp(0.5 * torch.randn(48, 1, generator=g))
    parts = _quantize_like_sdnq(w, bits, group, asym, rank)
    back = sdnq.dequantize(parts["weight"], parts["scale"], (48, 256), parts.get("zero_point"), parts.get("svd_up"), parts.get("svd_down"))
    err = ((back - w).norm() / w.norm()).item()
    bound = {8: 0.01, 4: 0.16, 2: 0.6}[bits]
    assert back.shape == (48, 256) and err < bound, err
    # the packing is exact: unpack(pack(q)) == q for every width, including counts that do not fill the last byte
    q = torch.randint(0, 2 ** bits, (1001,), generator=g, dtype=torch.uint8)
    assert torch.equal(sdnq.unpack_bits(sdnq.pack_bits(q, bits), bits, 1001), q)


def test_sdnq_bit_width_from_the_config_must_match_the_tensor():
    """a config that names uint4 for a module stored in 8 bits (a `modules_dtype_dict` pattern that did not match) must raise,
    not return the first nibbles"""
    w = torch.randn(16, 64, generator=torch.Generator().manual_seed(3))
    parts = _quantize_like_sdnq(w, 8, 32, True, 0)
    ok = sdnq.dequantize(parts["weight"], parts["scale"], (16, 64), parts.get("zero_point"), bits=8)
    assert ok.shape == (16, 64)
    for wrong in (4, 2):
        with pytest.raises(ModelError):
            sdnq.dequantize(parts["weight"], parts["scale"], (16, 64), parts.get("zero_point"), bits=wrong)
    p4 = _quantize_like_sdnq(w, 4, 32, True, 0)
    with pytest.raises(ModelError):
        sdnq.dequantize(p4["weight"], p4["scale"], (16, 64), p4.get("zero_point"), bits=8)
    with pytest.raises(ModelError):
        sdnq.unpack_bits(torch.zeros(9, dtype=torch.uint8), 4, 16)          # one byte too many


def test_sdnq_shards_through_the_flux_provider(tmp_path):
    """a diffusers sub-folder with one packed linear, one plain bf16 linear and a bias: the manager's provider hands out bf16 tensors
    of the logical shapes, the packed one within quantisation error of the original; a folder whose packed element count does not fit
    the expected shape raises ModelError instead of loading garbage"""
    from mangatranslator_amd.core.ml.model_manager import _ShardedProvider
    g = torch.Generator().manual_seed(3)
    w1, w2, b1 = torch.randn(64, 128, generator=g), torch.randn(32, 64, generator=g), torch.randn(64, generator=g)
    parts = _quantize_like_sdnq(w1, 4, 32, True, 4)
    parts = {k: v.contiguous() for k, v in parts.items()}
    sd = {"blk.lin.weight": parts["weight"], "blk.lin.scale": parts["scale"], "blk.lin.zero_point": parts["zero_point"],
          "blk.lin.svd_up": parts["svd_up"], "blk.lin.svd_down": parts["svd_down"], "blk.lin.bias": b1.to(torch.bfloat16),
          "blk.out.weight": w2.to(torch.bfloat16)}
    folder = tmp_path / "transformer"
    folder.mkdir()
    save_file(sd, str(folder / "diffusion_pytorch_model.safetensors"))
    (folder / "config.json").write_text(json.dumps({"quantization_config": {"quant_method": "sdnq", "weights_dtype": "uint4", "group_size": 32, "use_svd": True, "svd_rank": 4}}))
    shapes = {"blk.lin.weight": (64, 128), "blk.lin.bias": (64,), "blk.out.weight": (32, 64)}
    prov = _ShardedProvider(folder, shapes, torch.device("cpu"))
    got = prov("blk.lin.weight")
    assert got.dtype == torch.bfloat16 and got.shape == (64, 128)
    assert ((got.float() - w1).norm() / w1.norm()).item() < 0.12
    assert torch.equal(prov("blk.out.weight"), w2.to(torch.bfloat16)) and torch.equal(prov("blk.lin.bias"), b1.to(torch.bfloat16))
    with pytest.raises(ModelError):
        _ShardedProvider(folder, dict(shapes, **{"blk.lin.weight": (64, 100)}), torch.device("cpu"))("blk.lin.weight")


def _np_pack_nibbles(q: np.ndarray) -> np.ndarray:
    """independent of core/ml/sdnq.py: two 4-bit values per byte, the EARLIER value in the low nibble (sdnq's packed_int layout)"""
    flat = q.reshape(-1).astype(np.uint8)
    if flat.size % 2:
        flat = np.concatenate([flat, np.zeros(1, np.uint8)])
    return (flat[0::2] | (flat[1::2] << 4)).astype(np.uint8)


def test_kontext_uint4_svd_r32_folder(tmp_path):
    """A transformer folder laid out like `Disty0/FLUX.1-Kontext-dev-SDNQ-uint4-svd-r32` (reference model_manager.py:231-233): diffusers'
    FluxTransformer2DModel parameter names, every block linear stored as `<base>.weight` (uint8, two nibbles per byte) + `.scale` +
    `.zero_point` + `.svd_up` / `.svd_down` of rank 32, embedders / norms / biases in bf16, `quantization_config` in config.json.  The
    writer is numpy code that shares nothing with the reader; the provider must hand out every parameter of `dit_param_shapes` in its
    logical shape, equal (to bf16 rounding) to the dequantisation computed here in float64."""
    from mangatranslator_amd.core.ml import flux as fx
    from mangatranslator_amd.core.ml.model_manager import _ShardedProvider
    cfg = dict(d=64, heads=2, layers=1, single_layers=1, in_channels=64, joint_dim=96, pooled_dim=48, axes_dim=(8, 12, 12))
    shapes = fx.dit_param_shapes(cfg)
    rng = np.random.default_rng(7)
    group, rank = 32, 32
    sd, expect = {}, {}
    packed_bases = []
    for name, shp in shapes.items():
        w = rng.standard_normal(shp).astype(np.float32) * 0.05
        base = name[:-len(".weight")] if name.endswith(".weight") else None
        is_block_linear = base is not None and len(shp) == 2 and ("transformer_blocks." in name) and ".norm" not in name
        if not is_block_linear:
            t = torch.from_numpy(w).to(torch.bfloat16)
            sd[name], expect[name] = t, t.float().numpy().astype(np.float64)
            continue
        n, k = shp
        u, sv, vh = np.linalg.svd(w.astype(np.float64), full_matrices=False)
        up = torch.from_numpy((u[:, :rank] * sv[:rank]).astype(np.float32)).to(torch.bfloat16)
        down = torch.from_numpy(vh[:rank].astype(np.float32)).to(torch.bfloat16)
        resid = w.astype(np.float64) - up.float().numpy().astype(np.float64) @ down.float().numpy().astype(np.float64)
        g = resid.reshape(n, k // group, group)
        zp = torch.from_numpy(g.min(-1).astype(np.float32)).to(torch.bfloat16)
        sc = torch.from_numpy(np.maximum((g.max(-1) - g.min(-1)) / 15.0, 1e-8).astype(np.float32)).to(torch.bfloat16)
        zp64, sc64 = zp.float().numpy().astype(np.float64), sc.float().numpy().astype(np.float64)
        q = np.clip(np.rint((g - zp64[..., None]) / sc64[..., None]), 0, 15).astype(np.uint8)
        sd[base + ".weight"] = torch.from_numpy(_np_pack_nibbles(q)).reshape(n, k // 2)
        sd[base + ".scale"], sd[base + ".zero_point"] = sc.reshape(n, k // group, 1), zp.reshape(n, k // group, 1)
        sd[base + ".svd_up"], sd[base + ".svd_down"] = up, down
        expect[name] = (q.astype(np.float64) * sc64[..., None] + zp64[..., None]).reshape(n, k) + up.float().numpy().astype(np.float64) @ down.float().numpy().astype(np.float64)
        packed_bases.append(base)
    assert len(packed_bases) == 17 and "transformer_blocks.0.attn.to_q" in packed_bases and "single_transformer_blocks.0.proj_out" in packed_bases
    folder = tmp_path / "transformer"
    folder.mkdir()
    keys = sorted(sd)
    save_file({k: sd[k].contiguous() for k in keys[::2]}, str(folder / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[1::2]}, str(folder / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    (folder / "config.json").write_text(json.dumps({"_class_name": "FluxTransformer2DModel", "quantization_config": {
        "quant_method": "sdnq", "weights_dtype": "uint4", "group_size": group, "use_svd": True, "svd_rank": rank, "use_quantized_matmul": True,
        "modules_to_not_convert": ["x_embedder", "context_embedder", "proj_out", "time_text_embed", "norm_out"]}}))
    prov = _ShardedProvider(folder, shapes, torch.device("cpu"))
    for name, shp in shapes.items():
        got = prov(name)
        assert tuple(got.shape) == tuple(shp) and got.dtype == torch.bfloat16, name
        want = torch.from_numpy(expect[name].astype(np.float32)).to(torch.bfloat16)
        assert torch.equal(got, want) or (got.float() - want.float()).abs().max() <= 2.0 ** -8 * want.float().abs().max(), name
    # the packed weights really are within 4-bit + rank-32 error of what was quantised (the writer is not vacuous)
    w_q = prov("transformer_blocks.0.attn.to_q.weight").float()
    assert w_q.abs().max() > 0.05


def test_rcan_hyper_parameters_come_from_the_file_header(tmp_path):
    """the "PU" (pixel-unshuffle) fast variant and the plain one, from the safetensors header alone"""
    from oracle import rcan_ref
    from mangatranslator_amd.core.ml.rcan import derive_rcan_hparams, rcan_hparams_from_header, read_safetensors_header
    for name, hp in (("fast_pu", dict(n_feats=48, n_resgroups=3, n_resblocks=5, reduction=8, scale=2, unshuffle=2)),
                     ("plain", dict(n_feats=64, n_resgroups=2, n_resblocks=4, reduction=16, scale=2, unshuffle=1))):
        sd = rcan_ref.make_state_dict(seed=1, **hp)
        f = tmp_path / f"{name}.safetensors"
        save_file({k: v.contiguous() for k, v in sd.items()}, str(f))
        head = read_safetensors_header(f)
        assert head == {k: tuple(v.shape) for k, v in sd.items()}
        got = rcan_hparams_from_header(f)
        assert got == derive_rcan_hparams(sd)
        assert (got["n_feats"], got["n_resgroups"], got["n_resblocks"], got["unshuffle"], got["scale"]) == (hp["n_feats"], hp["n_resgroups"], hp["n_resblocks"], hp["unshuffle"], hp["scale"])
        assert got["cr"] == hp["n_feats"] // hp["reduction"]
    (tmp_path / "junk.safetensors").write_bytes(b"\x00" * 64)
    with pytest.raises(ModelError):
        rcan_hparams_from_header(tmp_path / "junk.safetensors")


def _tiny_tokenizer(folder, chat=False):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {w: i for i, w in enumerate(["<pad>", "<unk>", "</s>", "remove", "all", "text", ".", ",", "user", "assistant", "including", "sound", "effects"])}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", unk_token="<unk>", eos_token="</s>", model_max_length=16)
    if chat:
        fast.chat_template = "{% for m in messages %}{{ m['role'] }} {{ m['content'] }} {% endfor %}{% if add_generation_prompt %}assistant{% endif %}"
    fast.save_pretrained(str(folder))


def test_prompt_embedding_exporter(tmp_path):
    """tools/export_prompt_embeds.py on tiny seeded encoders: the files hold what `ModelManager.load_flux_*` reads — Kontext: T5 sequence
    [512, d_model] + CLIP pooled [hidden]; Klein: three Qwen3 layers side by side [512, 3 * hidden]"""
    import export_prompt_embeds as ex
    from safetensors import safe_open
    from transformers import CLIPTextConfig, CLIPTextModel, Qwen3Config, Qwen3ForCausalLM, T5Config, T5EncoderModel
    torch.manual_seed(0)
    repo = tmp_path / "kontext"
    _tiny_tokenizer(repo / "tokenizer")
    _tiny_tokenizer(repo / "tokenizer_2")
    CLIPTextModel(CLIPTextConfig(vocab_size=16, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=16,
                                 eos_token_id=2, pad_token_id=0, bos_token_id=1)).save_pretrained(str(repo / "text_encoder"))
    T5EncoderModel(T5Config(vocab_size=16, d_model=24, d_kv=8, d_ff=48, num_layers=2, num_heads=3)).save_pretrained(str(repo / "text_encoder_2"))
    assert ex.main(["kontext", str(repo)]) == 0
    with safe_open(str(repo / "prompt_embeds.safetensors"), framework="pt") as f:
        assert f.get_tensor("prompt_embeds").shape == (512, 24) and f.get_tensor("pooled_prompt_embeds").shape == (32,)
        assert f.get_tensor("prompt_embeds").dtype == torch.bfloat16 and f.metadata()["prompt"] == "Remove all text."
    repo = tmp_path / "klein"
    _tiny_tokenizer(repo / "tokenizer", chat=True)
    Qwen3ForCausalLM(Qwen3Config(vocab_size=16, hidden_size=16, intermediate_size=32, num_hidden_layers=28, num_attention_heads=2, num_key_value_heads=1,
                                 head_dim=8, max_position_embeddings=1024)).save_pretrained(str(repo / "text_encoder"))
    assert ex.main(["klein", str(repo), "--out", str(tmp_path / "pe.safetensors")]) == 0
    with safe_open(str(tmp_path / "pe.safetensors"), framework="pt") as f:
        t = f.get_tensor("prompt_embeds")
        assert t.shape == (512, 48) and t.dtype == torch.bfloat16 and torch.isfinite(t.float()).all()
        assert f.metadata()["prompt"].startswith("Remove all text, including")
