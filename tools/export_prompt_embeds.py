"""Export the prompt embeddings the FLUX pipelines of this package take as an input — run ONCE per checkpoint, off the page path.

The reference encodes its fixed prompt on first use and keeps the tensors (`core/image/inpainting.py:846-873` Kontext: T5-XXL sequence
embeddings + CLIP-L pooled vector of "Remove all text."; `:1110-1124` Klein: Qwen3 hidden states of `KLEIN_PROMPT`).  This package keeps
the text encoders out of the serving process altogether: `ModelManager.load_flux_kontext_sdnq / load_flux_klein_*` read
`prompt_embeds.safetensors` next to the transformer (core/ml/model_manager.py), and this script writes that file with HF `transformers`
(which IS installed in the serving image; only the encoder checkpoints are needed, any device).

    python tools/export_prompt_embeds.py kontext <diffusers repo dir> [--out models/flux/kontext/prompt_embeds.safetensors]
    python tools/export_prompt_embeds.py klein   <diffusers repo dir> [--out models/flux/klein-4b/prompt_embeds.safetensors]

`<diffusers repo dir>` is the pipeline snapshot the reference downloads (text_encoder/, text_encoder_2/, tokenizer/, tokenizer_2/ for
Kontext; text_encoder/, tokenizer/ for Klein).  SDNQ-packed encoder weights are de-quantised on load by `mangatranslator_amd.core.ml.sdnq`
when `--sdnq` is given (the reference's repos quantise the text encoders too).

What is restated from diffusers (absent here; pinned at the versions in the reference's requirements.txt), with the call sites:
  Kontext  FluxKontextPipeline.encode_prompt: CLIP tokenizer max_length 77 -> text_encoder(...).pooler_output [768];
           T5 tokenizer max_length 512, padding "max_length" -> text_encoder_2(...)[0] [512, 4096]   (both bf16)
  Klein    Flux2KleinPipeline._get_qwen3_prompt_embeds: chat template (user message, add_generation_prompt, thinking off), max_length 512,
           hidden states of layers (9, 18, 27) concatenated on the feature axis -> [512, 3 * hidden]
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

KONTEXT_PROMPT = "Remove all text."                 # reference core/image/inpainting.py:168
QWEN3_LAYERS = (9, 18, 27)


def _load(cls, folder: Path, dtype, sdnq: bool):
    import torch
    if not sdnq:
        return cls.from_pretrained(str(folder), torch_dtype=dtype).eval()
    # SDNQ-packed encoder: build the module from its config, fill it from the de-quantised shards
    from transformers import AutoConfig
    from mangatranslator_amd.core.ml.sdnq import dequantized_state_dict
    cfg = AutoConfig.from_pretrained(str(folder))
    with torch.device("meta"):
        model = cls.from_config(cfg) if hasattr(cls, "from_config") else cls(cfg)
    sd = dequantized_state_dict(folder, {k: tuple(v.shape) for k, v in model.state_dict().items()}, dtype=dtype)
    model = model.to_empty(device="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing:
        raise SystemExit(f"{folder}: {len(missing)} parameters missing after de-quantisation (first: {missing[0]})")
    return model.to(dtype).eval()


def export_kontext(repo: Path, out: Path, device: str, sdnq: bool, prompt: str = KONTEXT_PROMPT):
    import torch
    from safetensors.torch import save_file
    from transformers import AutoTokenizer, CLIPTextModel, T5EncoderModel
    tok, tok2 = AutoTokenizer.from_pretrained(str(repo / "tokenizer")), AutoTokenizer.from_pretrained(str(repo / "tokenizer_2"))
    clip = _load(CLIPTextModel, repo / "text_encoder", torch.bfloat16, sdnq).to(device)
    t5 = _load(T5EncoderModel, repo / "text_encoder_2", torch.bfloat16, sdnq).to(device)
    with torch.no_grad():
        ids = tok([prompt], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids.to(device)
        pooled = clip(ids, output_hidden_states=False).pooler_output[0]
        ids2 = tok2([prompt], padding="max_length", max_length=512, truncation=True, return_tensors="pt").input_ids.to(device)
        seq = t5(ids2, output_hidden_states=False)[0][0]
    out.parent.mkdir(parents=True, exist_ok=True)
    save_file({"prompt_embeds": seq.to(torch.bfloat16).cpu().contiguous(), "pooled_prompt_embeds": pooled.to(torch.bfloat16).cpu().contiguous()}, str(out),
              metadata={"prompt": prompt, "max_sequence_length": "512", "encoders": "CLIP-L pooler_output + T5-XXL last_hidden_state"})
    print(f"wrote {out}: prompt_embeds {tuple(seq.shape)}, pooled_prompt_embeds {tuple(pooled.shape)}")


def export_klein(repo: Path, out: Path, device: str, sdnq: bool, prompt: str = None):
    import torch
    from safetensors.torch import save_file
    from transformers import AutoModelForCausalLM, AutoTokenizer
    if prompt is None:
        from mangatranslator_amd.core.image.inpainting import KLEIN_PROMPT as prompt
    tok = AutoTokenizer.from_pretrained(str(repo / "tokenizer"))
    enc = _load(AutoModelForCausalLM, repo / "text_encoder", torch.bfloat16, sdnq).to(device)
    text = tok.apply_chat_template([{"role": "user", "content": prompt}], tokenize=False, add_generation_prompt=True, enable_thinking=False)
    with torch.no_grad():
        t = tok([text], padding="max_length", max_length=512, truncation=True, return_tensors="pt").to(device)
        hs = enc(input_ids=t.input_ids, attention_mask=t.attention_mask, output_hidden_states=True, use_cache=False).hidden_states
        seq = torch.stack([hs[k] for k in QWEN3_LAYERS], dim=1)[0]                   # [3, L, H]
        seq = seq.permute(1, 0, 2).reshape(seq.shape[1], -1)                          # [L, 3 H]
    out.parent.mkdir(parents=True, exist_ok=True)
    save_file({"prompt_embeds": seq.to(torch.bfloat16).cpu().contiguous()}, str(out),
              metadata={"prompt": prompt, "max_sequence_length": "512", "encoders": f"Qwen3 hidden states of layers {QWEN3_LAYERS}"})
    print(f"wrote {out}: prompt_embeds {tuple(seq.shape)}")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("pipeline", choices=["kontext", "klein"])
    ap.add_argument("repo", help="diffusers pipeline snapshot with the text encoder(s) and tokenizer(s)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--sdnq", action="store_true", help="the encoder weights are SDNQ-packed (the reference's Disty0/*-SDNQ-* repos)")
    ap.add_argument("--prompt", default=None, help="default: the reference's fixed prompt for that pipeline")
    a = ap.parse_args(argv)
    repo = Path(a.repo)
    out = Path(a.out) if a.out else repo / "prompt_embeds.safetensors"
    if a.pipeline == "kontext":
        export_kontext(repo, out, a.device, a.sdnq, a.prompt or KONTEXT_PROMPT)
    else:
        export_klein(repo, out, a.device, a.sdnq, a.prompt)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
