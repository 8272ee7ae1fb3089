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

from mangatranslator_amd.core.ml.prompt_embeds import KONTEXT_PROMPT, export_klein, export_kontext  # noqa: E402


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
        shapes = export_kontext(repo, out, a.device, a.sdnq, a.prompt or KONTEXT_PROMPT)
    else:
        shapes = export_klein(repo, out, a.device, a.sdnq, a.prompt)
    print(f"wrote {out}: {shapes}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
