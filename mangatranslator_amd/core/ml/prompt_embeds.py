"""Prompt embeddings of the FLUX pipelines, computed with HF `transformers` from the text encoders of a staged diffusers snapshot.

The reference encodes its fixed prompt on first use and keeps the tensors (`core/image/inpainting.py:846-873` Kontext: T5-XXL sequence
embeddings + CLIP-L pooled vector of "Remove all text."; `:1110-1124` Klein: Qwen3 hidden states of `KLEIN_PROMPT`).  Here the text
encoders are not part of the page path: the pipelines read `prompt_embeds.safetensors` next to the transformer.  That file is written by
`tools/export_prompt_embeds.py` — or, since round 5, by `ModelManager.load_flux_kontext_sdnq / load_flux_klein_*` themselves on the first
load when it is absent and the snapshot's `text_encoder*/` + `tokenizer*/` folders are staged (`ensure_prompt_embeds`): `transformers` is
installed in the serving image, so the reference's "encode once, keep" happens at load time instead of needing a separate step.

What is restated from diffusers (absent here; pinned at the versions in the reference's requirements.txt), with the call sites:
  Kontext  FluxKontextPipeline.encode_prompt: CLIP tokenizer max_length 77 -> text_encoder(...).pooler_output [768];
           T5 tokenizer max_length 512, padding "max_length" -> text_encoder_2(...)[0] [512, 4096]   (both bf16)
  Klein    Flux2KleinPipeline._get_qwen3_prompt_embeds: chat template (user message, add_generation_prompt, thinking off), max_length 512,
           hidden states of layers (9, 18, 27) concatenated on the feature axis -> [512, 3 * hidden]
"""
from pathlib import Path

KONTEXT_PROMPT = "Remove all text."                 # reference core/image/inpainting.py:168
QWEN3_LAYERS = (9, 18, 27)


def _load(cls, folder: Path, dtype, sdnq: bool):
    import torch
    if not sdnq:
        return cls.from_pretrained(str(folder), torch_dtype=dtype).eval()
    # SDNQ-packed encoder: build the module from its config, fill it from the de-quantised shards
    from transformers import AutoConfig
    from .sdnq import dequantized_state_dict
    cfg = AutoConfig.from_pretrained(str(folder))
    with torch.device("meta"):
        model = cls.from_config(cfg) if hasattr(cls, "from_config") else cls(cfg)
    sd = dequantized_state_dict(folder, {k: tuple(v.shape) for k, v in model.state_dict().items()}, dtype=dtype)
    model = model.to_empty(device="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing:
        raise RuntimeError(f"{folder}: {len(missing)} parameters missing after de-quantisation (first: {missing[0]})")
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
    _save_atomically({"prompt_embeds": seq.to(torch.bfloat16).cpu().contiguous(), "pooled_prompt_embeds": pooled.to(torch.bfloat16).cpu().contiguous()}, out,
              metadata={"prompt": prompt, "max_sequence_length": "512", "encoders": "CLIP-L pooler_output + T5-XXL last_hidden_state"})
    return {"prompt_embeds": tuple(seq.shape), "pooled_prompt_embeds": tuple(pooled.shape)}


def export_klein(repo: Path, out: Path, device: str, sdnq: bool, prompt: str = None):
    import torch
    from safetensors.torch import save_file
    from transformers import AutoModelForCausalLM, AutoTokenizer
    if prompt is None:
        from ..image.inpainting import KLEIN_PROMPT as prompt
    tok = AutoTokenizer.from_pretrained(str(repo / "tokenizer"))
    enc = _load(AutoModelForCausalLM, repo / "text_encoder", torch.bfloat16, sdnq).to(device)
    text = tok.apply_chat_template([{"role": "user", "content": prompt}], tokenize=False, add_generation_prompt=True, enable_thinking=False)
    with torch.no_grad():
        t = tok([text], padding="max_length", max_length=512, truncation=True, return_tensors="pt").to(device)
        hs = enc(input_ids=t.input_ids, attention_mask=t.attention_mask, output_hidden_states=True, use_cache=False).hidden_states
        seq = torch.stack([hs[k] for k in QWEN3_LAYERS], dim=1)[0]                   # [3, L, H]
        seq = seq.permute(1, 0, 2).reshape(seq.shape[1], -1)                          # [L, 3 H]
    out.parent.mkdir(parents=True, exist_ok=True)
    _save_atomically({"prompt_embeds": seq.to(torch.bfloat16).cpu().contiguous()}, out,
              metadata={"prompt": prompt, "max_sequence_length": "512", "encoders": f"Qwen3 hidden states of layers {QWEN3_LAYERS}"})
    return {"prompt_embeds": tuple(seq.shape)}




def _save_atomically(tensors: dict, out: Path, metadata: dict):
    """write beside the target, then rename into place: a crash mid-write leaves no truncated prompt_embeds.safetensors behind that every
    later load would take for the finished file; the temporary name carries the pid so that two writers never share it"""
    import os
    from safetensors.torch import save_file
    tmp = out.with_name(f".{out.name}.{os.getpid()}.tmp")
    try:
        save_file(tensors, str(tmp), metadata=metadata)
        os.replace(tmp, out)
    finally:
        if tmp.exists():
            tmp.unlink()


def _readable(out: Path) -> bool:
    """a finished safetensors file: the header parses and names `prompt_embeds` (an unreadable file counts as absent and is re-encoded)"""
    try:
        from safetensors import safe_open
        with safe_open(str(out), framework="pt") as f:
            return "prompt_embeds" in f.keys()
    except Exception:      # noqa: BLE001
        return False


def _sdnq_packed(folder: Path) -> bool:
    cfg = folder / "config.json"
    try:
        return "sdnq" in cfg.read_text().lower()
    except OSError:
        return False


def encoders_staged(repo: Path, pipeline: str) -> bool:
    need = ("text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2") if pipeline == "kontext" else ("text_encoder", "tokenizer")
    return all((repo / n).is_dir() for n in need)


def ensure_prompt_embeds(repo: Path, pipeline: str, device="cpu", log=None) -> bool:
    """`repo / "prompt_embeds.safetensors"` exists afterwards?  Present already: True.  Absent and the snapshot's text encoder(s) and
    tokenizer(s) are staged: they are loaded once through `transformers`, the fixed prompt is encoded (the reference's first-use encode,
    inpainting.py:846-873 / :1110-1124), the file is written, the encoders are dropped.  Absent and nothing to make it from: False — the
    pipeline's `encode_prompt` then raises the ModelError that names `tools/export_prompt_embeds.py`.  Called by rank 0 only."""
    out = repo / "prompt_embeds.safetensors"
    if out.exists() and _readable(out):
        return True
    if not encoders_staged(repo, pipeline):
        return False
    # one encoder at a time per snapshot: under ModelManager.thread_local_reads every rank counts as rank 0 here, so a lock file (O_EXCL) picks
    # the writer; the others wait for the finished file (the rename above makes its appearance atomic)
    import os, time
    lock = repo / ".prompt_embeds.lock"
    try:
        fd = os.open(str(lock), os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        os.close(fd)
    except FileExistsError:
        for _ in range(1800):
            if out.exists() and _readable(out):
                return True
            if not lock.exists():
                break
            time.sleep(1.0)
        return out.exists() and _readable(out)
    except OSError:
        pass                                    # read-only snapshot folder: encode anyway (the save will say so)
    try:
        sdnq = _sdnq_packed(repo / "text_encoder") or (pipeline == "kontext" and _sdnq_packed(repo / "text_encoder_2"))
        shapes = export_kontext(repo, out, str(device), sdnq) if pipeline == "kontext" else export_klein(repo, out, str(device), sdnq)
        if log is not None:
            log(f"prompt embeddings encoded once from the staged text encoder(s) and kept: {out} {shapes}")
        return True
    except Exception as e:      # noqa: BLE001 — a broken encoder folder must not take the loader down: the inpainter then reports the missing embeddings
        if log is not None:
            log(f"could not encode the prompt from {repo}: {type(e).__name__}: {e}")
        return False
    finally:
        try:
            lock.unlink()
        except OSError:
            pass
