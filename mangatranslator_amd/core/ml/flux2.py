"""FLUX.2-Klein (Flux2 MMDiT + AutoencoderKLFlux2 + flow-match Euler) on libmtx_hip — SURVEY.md §8 rows a7 / g1.

The object `ModelManager.load_flux_klein_4b()` / `load_flux_klein_9b()` hands to `FluxKleinInpainter`; called with the
diffusers pipeline shape the reference uses (core/image/inpainting.py:1577-1589):
    pipeline(prompt_embeds=, image=PIL, height=, width=, guidance_scale=1.0, num_inference_steps=, generator=).images[0] -> PIL
Weights carry diffusers' names (Flux2Transformer2DModel / AutoencoderKLFlux2 state dicts); the architecture followed is
restated in oracle/flux2_ref.py.

Graph design (MI355X-first), on top of what core/ml/flux.py already does for FLUX.1 (one [T, D] token buffer with the text
rows first, fused q|k|v projections, per-head RMSNorm + RoPE in one in-place kernel with the softmax scale folded into q's
rotary table, gated residuals as GEMM epilogues, one hipGraph replay per denoising step):
  * FLUX.2 shares ONE set of modulation vectors per stream type across all blocks, so a step needs 17 vectors in total:
    one [17 D, D] GEMV per schedule step, cached per timestep.
  * single-stream blocks: `to_qkv_mlp_proj` is one [3 D + 6 D, D] GEMM; attention reads q / k / v straight out of its
    output through strides and writes into columns [0, D) of the [T, 4 D] buffer whose columns [D, 4 D) the SwiGLU
    kernel fills, so `to_out(cat(attn, mlp))` is a plain GEMM.
  * fp8 (`fp8=True`, BASELINE.json config 5): every block linear runs on the MX-scaled fp8 matrix instructions
    (gemm.hip gemm256_f8_kernel, 2x the bf16 rate).  Weights are quantised once at load (e4m3 + one E8M0 scale per 32 k);
    activations are quantised by `mtx_quantize_mx` right after the kernel that produces them (LayerNorm / attention /
    SwiGLU), into fp8 twins of the bf16 buffers.  Embedders, modulation GEMVs, norm_out / proj_out and all
    normalisation / softmax / residual arithmetic stay bf16 / fp32.
"""
import math
import threading
from types import SimpleNamespace

import numpy as np
import torch

from ...hip import abi
from ...hip.lib import get_library
from ...hip.plan import Act, PlanBuilder, PlanCache, glu_interleave
from ...utils.exceptions import ModelError
from .flux import FluxVAEHip, _rows, rope_table, sinusoid, synthetic_provider  # noqa: F401  (re-exported for callers)


def compute_empirical_mu(image_seq_len: int, num_steps: int) -> float:
    """time-shift parameter of the FLUX.2 pipelines (restated in oracle/flux2_ref.py)"""
    a1, b1 = 8.73809524e-05, 1.89833333
    a2, b2 = 0.00016927, 0.45666666
    if image_seq_len > 4300:
        return float(a2 * image_seq_len + b2)
    m_200 = a2 * image_seq_len + b2
    m_10 = a1 * image_seq_len + b1
    a = (m_200 - m_10) / 190.0
    b = m_200 - 200.0 * a
    return float(a * num_steps + b)


def flow_sigmas(steps: int, image_seq_len: int) -> np.ndarray:
    s = np.linspace(1.0, 1.0 / steps, steps)
    mu = compute_empirical_mu(image_seq_len, steps)
    s = math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))
    return np.append(s, 0.0).astype(np.float32)


def token_ids(t_txt, h2, w2, rh2, rw2, ref_t=10.0) -> np.ndarray:
    """[T, 4] position ids (t, h, w, l): text tokens on the l axis, noise tokens at t = 0, reference-image tokens at t = 10"""
    txt = np.zeros((t_txt, 4), np.float32)
    txt[:, 3] = np.arange(t_txt)

    def grid(h, w, t):
        g = np.zeros((h, w, 4), np.float32)
        g[..., 0] = t
        g[..., 1] = np.arange(h)[:, None]
        g[..., 2] = np.arange(w)[None, :]
        return g.reshape(-1, 4)
    parts = [txt, grid(h2, w2, 0.0)]
    if rh2 and rw2:
        parts.append(grid(rh2, rw2, ref_t))
    return np.concatenate(parts)


FP8_ALL = ("qkv", "out", "ff_in", "ff_out", "single_in", "single_out")


class _W:
    """one linear's weight: 16-bit [N, K], or its MX fp8 copy (bytes [N, K] + scale plane)"""

    def __init__(self, w16=None, q=None, scale=None, lds=0):
        self.w16, self.q, self.scale, self.lds = w16, q, scale, lds


class Flux2DiTHip:
    def __init__(self, provider, cfg: dict, device, lib=None, fp8=False, fused_quant=True, glu_epilogue=True, attn_q8=True, attn_qk_f8=True, attn_pv_f8=True):
        """provider(name) -> tensor with diffusers' Flux2Transformer2DModel parameter of that name.
        fp8: False, True (= every block linear) or a tuple of names out of FP8_ALL."""
        self.lib = lib if lib is not None else get_library()
        self.device = torch.device(device)
        self.dtype, self.tdt = abi.BF16, torch.bfloat16
        self.cfg = cfg
        D, H = cfg["d"], cfg["heads"]
        self.hd = D // H
        self.hid = int(D * cfg.get("mlp_ratio", 3.0))
        if self.hd not in (64, 128) or sum(cfg["axes_dim"]) != self.hd:
            raise ModelError("FLUX.2 DiT: head dim must be 64 or 128 and equal sum(axes_dims_rope)")
        self.fp8 = FP8_ALL if fp8 is True else tuple(fp8 or ())
        self.fused_quant = fused_quant      # norms and SwiGLU write the fp8 linears' operands themselves; False: separate quantiser passes (round 2's form, for A/Bs)
        # glu_epilogue: the MLP-in linears' fp8 GEMM applies SwiGLU and writes the MX fp8 operand of MLP-out itself (mtx_gemm_args.glu_*): no
        # [T, 2 * hidden] projection, no SwiGLU-quantiser launch.  Bit-identical to the separate launches on the simulator and on gfx950 at Klein's
        # shapes (tests/test_ops_gpu.py::test_gemm_f8_glu_epilogue); default since round 4 (DiT step 52.3 -> 48.9 ms with both fusions).
        # Needs every linear of the MLP on the fp8 path; False = the separate launches, for A/Bs.
        self.glu_epilogue = bool(glu_epilogue) and all(k in self.fp8 for k in ("ff_in", "ff_out", "single_in", "single_out")) and (3 * D) % 256 == 0
        # attn_q8: the joint attention writes the MX fp8 operand of the output projections itself (mtx_attn_args.q8; long-sequence kernel, so only
        # for T >= 1024 and head dim 128) — with glu_epilogue no quantiser launch is left in a step (tests/test_ops_gpu.py::test_attention_mx_fp8_output).
        self.attn_q8 = bool(attn_q8) and all(k in self.fp8 for k in ("out", "single_out")) and self.hd == 128
        # attn_qk_f8: the rotary kernel also leaves q and k as plain e4m3 rows (q times 8: the pre-scaled q is ~N(0, 0.13), below e4m3's normal range) and
        # the joint attention takes its scores from them on the fp8 matrix instruction (mtx_attn_args.q_f8 / k_f8, logits 2^-3 * q k); P V stays 16-bit.
        # Changes the result (3 mantissa bits under the scores): 4 Klein steps at T = 1568 stay at 41.9 dB against the bf16 pipeline (fp8 linears alone: 41.8;
        # tests/test_flux2_gpu.py::test_klein_fp8_attention_scores_psnr), the launch takes 0.603 ms instead of 0.783 at T = 8704 — on with the fp8 linears, never without.
        self.attn_qk_f8 = bool(attn_qk_f8) and bool(self.fp8) and self.hd == 128
        # attn_pv_f8: P V on the fp8 instruction as well — the values as e4m3 V^T in accumulator key order (MTX_EW_V_F8T, one 15 us launch per attention),
        # the tile's probabilities rounded to e4m3 in registers (mtx_attn_args.v_f8t).  Needs attn_qk_f8 and attn_q8.  Measured like the scores: the launch
        # 0.62 -> 0.48 ms at T = 8704, 4 Klein steps at T = 1568 41.8 dB against the bf16 pipeline (fp8 linears alone 41.8, + fp8 scores 41.9) — the rows
        # leave as MX fp8 for the next linear anyway, the values' rounding is of that order.  On with the fp8 linears, never without.
        self.attn_pv_f8 = bool(attn_pv_f8) and self.attn_qk_f8 and self.attn_q8
        if self.fp8 and (D % 128 or self.hid % 128):
            raise ModelError("FLUX.2 DiT fp8 path: d and the MLP width must be multiples of 128")
        g = lambda n, dt=None: provider(n).detach().to(self.device, dt if dt is not None else self.tdt).contiguous()
        f32 = torch.float32
        W = {nm: g(nm + ".weight") for nm in ("x_embedder", "context_embedder", "proj_out")}
        embs = ["timestep_embedder"] + (["guidance_embedder"] if cfg.get("guidance_embeds") else [])
        for e in embs:
            for l in ("linear_1", "linear_2"):
                W[f"{e}.{l}"] = g(f"time_guidance_embed.{e}.{l}.weight")
        self.embs = embs
        # modulation rows: 0-5 image stream (shift, scale, gate of attention; shift, scale, gate of the MLP), 6-11 text stream,
        # 12-14 single stream (shift, scale, gate), 15-16 norm_out (scale, shift)
        W["mods"] = torch.cat([g("double_stream_modulation_img.linear.weight"), g("double_stream_modulation_txt.linear.weight"),
                               g("single_stream_modulation.linear.weight"), g("norm_out.linear.weight")], 0).contiguous()
        self.n_vec = 17
        if W["mods"].shape[0] != self.n_vec * D:
            raise ModelError("FLUX.2 DiT: unexpected modulation parameter shapes")
        self.W = W
        cat = lambda p, names: torch.cat([g(f"{p}.{n}.weight") for n in names], 0).contiguous()
        self.blocks, self.singles = [], []
        for i in range(cfg["layers"]):
            p = f"transformer_blocks.{i}"
            self.blocks.append(dict(
                qkv=self._weight(cat(p + ".attn", ("to_q", "to_k", "to_v")), "qkv"),
                cqkv=self._weight(cat(p + ".attn", ("add_q_proj", "add_k_proj", "add_v_proj")), "qkv"),
                nqk=torch.cat([g(p + ".attn.norm_q.weight", f32), g(p + ".attn.norm_k.weight", f32)]).contiguous(),
                cnqk=torch.cat([g(p + ".attn.norm_added_q.weight", f32), g(p + ".attn.norm_added_k.weight", f32)]).contiguous(),
                out=self._weight(g(p + ".attn.to_out.0.weight"), "out"), cout=self._weight(g(p + ".attn.to_add_out.weight"), "out"),
                ff_in=self._weight(g(p + ".ff.linear_in.weight"), "ff_in", glu_col0=0), ff_out=self._weight(g(p + ".ff.linear_out.weight"), "ff_out"),
                cff_in=self._weight(g(p + ".ff_context.linear_in.weight"), "ff_in", glu_col0=0),
                cff_out=self._weight(g(p + ".ff_context.linear_out.weight"), "ff_out")))
        for i in range(cfg["single_layers"]):
            p = f"single_transformer_blocks.{i}.attn"
            self.singles.append(dict(fused=self._weight(g(p + ".to_qkv_mlp_proj.weight"), "single_in", glu_col0=3 * D),
                                     nqk=torch.cat([g(p + ".norm_q.weight", f32), g(p + ".norm_k.weight", f32)]).contiguous(),
                                     out=self._weight(g(p + ".to_out.weight"), "single_out")))
        self._plans = PlanCache(6)           # a plan pins ~T x 40 D bytes of activations: keep a few resolutions only
        self._mod_plan = None
        self._mod_cache = {}

    def _weight(self, w16: torch.Tensor, kind: str, glu_col0=None) -> _W:
        if kind not in self.fp8:
            return _W(w16=w16)
        if glu_col0 is not None and self.glu_epilogue:       # rows in [32 a | 32 b] runs from glu_col0 on (hip/plan.py glu_interleave)
            w16 = w16[glu_interleave(glu_col0, self.hid).to(w16.device)].contiguous()
        n, k = w16.shape
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        q, scale, lds = pb.quantize(w16, n, k)
        pb.build().run()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return _W(q=q, scale=scale, lds=lds)            # the 16-bit copy is dropped: half the resident bytes

    # ---- modulation vectors of one timestep ---------------------------------------------------------------
    def _build_mod_plan(self):
        D, W = self.cfg["d"], self.W
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        tin = pb.buf((2, 256), self.tdt)                     # sinusoids of timestep * 1000 (and guidance * 1000)
        a = lambda t: Act(t.view(1, 1, 1, D), 1, 1, 1, D)
        temb = None
        for row, e in enumerate(self.embs):
            h = pb.gemm(tin, W[f"{e}.linear_1"], 1, D, 256, act=abi.ACT_SILU, a_off=row * 256, label=f"temb.{e}.1")
            o = pb.gemm(h, W[f"{e}.linear_2"], 1, D, D, label=f"temb.{e}.2")
            temb = o if temb is None else pb.ew(abi.EW_ADD, a(temb), b=a(o), label="temb.sum").t.view(1, D)
        st = pb.ew(abi.EW_ACT, a(temb), act=abi.ACT_SILU, label="temb.silu")                      # silu(temb) feeds every modulation
        mods = pb.gemm(st.t.view(1, D), W["mods"], 1, self.n_vec * D, D, label="modulation.all")
        plan = pb.build()
        plan.tin, plan.mods = tin, mods
        return plan

    def modulation(self, timestep: float, guidance=None) -> torch.Tensor:
        """[17, D] bf16 modulation rows of one denoising step, cached per (timestep, guidance)"""
        key = (round(float(timestep), 7), None if guidance is None else round(float(guidance), 5))
        if len(self._mod_cache) > 256:
            self._mod_cache.clear()
        if key not in self._mod_cache:
            if self._mod_plan is None:
                self._mod_plan = self._build_mod_plan()
            mp = self._mod_plan
            tin = np.stack([sinusoid(timestep * 1000.0), sinusoid((guidance or 0.0) * 1000.0)])
            mp.tin.copy_(torch.from_numpy(tin).to(self.device, self.tdt))
            mp.run()
            self._mod_cache[key] = mp.mods.view(self.n_vec, self.cfg["d"]).clone()
        return self._mod_cache[key]

    # ---- one denoising step as a plan ------------------------------------------------------------------------
    def _build(self, t_txt, h2, w2, rh2, rw2):
        cfg, W = self.cfg, self.W
        D, H, hd, hid = cfg["d"], cfg["heads"], self.hd, self.hid
        t_noise, t_ref = h2 * w2, rh2 * rw2
        t_img = t_noise + t_ref
        T = t_txt + t_img
        FW = 3 * D + 2 * hid                                     # width of the single blocks' fused projection
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        lat = pb.buf((t_img, cfg["in_channels"]), self.tdt)       # [noise tokens ; reference tokens]
        ctx_in = pb.buf((t_txt, cfg["joint_dim"]), self.tdt)      # prompt embeddings
        mod = pb.buf((self.n_vec, D), self.tdt)
        tab = torch.from_numpy(rope_table(token_ids(t_txt, h2, w2, rh2, rw2), cfg["axes_dim"], theta=cfg.get("rope_theta", 2000.0)))
        q_fold = (1.0 / math.sqrt(hd)) * 1.4426950408889634       # softmax scale * log2(e), folded into q's rotary table
        cs2 = pb.hold(torch.stack([tab, tab * q_fold]).to(self.device).contiguous())
        cs = cs2[0]
        x = pb.buf((T, D), self.tdt)
        nrm = pb.buf((T, D), self.tdt)
        qkv = pb.buf((T, 3 * D), self.tdt)
        o = pb.buf((T, D), self.tdt)
        ffh = pb.buf((T, 2 * hid), self.tdt)
        ffa = pb.buf((T, hid), self.tdt)
        big = pb.buf((T, FW), self.tdt)
        cat = pb.buf((T, D + hid), self.tdt)
        f8 = bool(self.fp8)
        lds = (T + 63) // 64 * 64
        if f8:      # fp8 twins of the GEMM inputs
            twin = lambda k: (pb.buf((T, k), torch.uint8), pb.buf((k // 128, lds), torch.int32, zero=True))
            nrm8, o8, ffa8, cat8 = twin(D), twin(D), twin(hid), twin(D + hid)

        def quant(src, k, dst, r0, r1, label):
            pb.quantize(src, r1 - r0, k, x_off=r0 * k, q=dst[0], scale=dst[1], row_off=r0, lds=lds, label=label)

        def linear(src, src8, w: _W, r0, r1, n, k, out, ldc=None, c_off=0, label="linear", **epi):
            """out[r0:r1, c_off : c_off + n] = epilogue(src[r0:r1, :k] W^T) on the 16-bit or the fp8 kernel, as the weight says"""
            m = r1 - r0
            ldc = ldc or n
            if w.q is not None:
                pb.gemm(src8[0], w.q, m, n, k, out=out, ldc=ldc, a_off=r0 * k, c_off=r0 * ldc + c_off,
                        f8=(src8[1], lds, w.scale, w.lds, r0, 0), label=label + ".f8", **epi)
            else:
                pb.gemm(src, w.w16, m, n, k, out=out, ldc=ldc, a_off=r0 * k, c_off=r0 * ldc + c_off, label=label, **epi)

        pb.gemm(ctx_in, W["context_embedder"], t_txt, D, cfg["joint_dim"], out=x, label="context_embedder")
        pb.gemm(lat, W["x_embedder"], t_img, D, cfg["in_channels"], out=x, c_off=t_txt * D, label="x_embedder")

        def adaln(r0, r1, shift_i, scale_i, label, consumers=()):
            """adaLN LayerNorm of rows [r0, r1).  With fp8 consumers the kernel writes their MX fp8 operand itself (mtx_norm_args.q:
            bit-identical to a quantiser pass over its 16-bit output, which is then only written if some consumer still reads 16-bit)"""
            to8 = f8 and self.fused_quant and any(w.q is not None for w in consumers)
            need16 = not to8 or any(w.q is None for w in consumers)
            pb.norm(x, nrm if need16 else None, r1 - r0, D, eps=1e-6, kind=0, mod_scale=mod[scale_i], mod_shift=mod[shift_i], rows_per=r1 - r0, ldmod=D,
                    x_off=r0 * D, y_off=r0 * D, label=label, q8=nrm8 if to8 else None, q_row_off=r0, lds_q=lds)
            if f8 and not to8:
                quant(nrm, D, nrm8, r0, r1, label + ".q")

        def rope(buf, r0, r1, gamma_qk, ld, label):
            v = _rows(buf, r0, r1, 0, 2 * D)
            e = abi.EwArgs()
            e.a, e.b, e.s, e.y = v.ptr, cs[r0:].data_ptr(), gamma_qk.data_ptr(), v.ptr
            e.n, e.h, e.w, e.c = 1, 1, r1 - r0, 2 * D
            e.lda, e.ldb, e.ldy, e.lds = ld, T * hd, ld, 0
            e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_QK_NORM_ROPE, 0, 1e-6, hd, H, self.dtype
            if qk8 is not None:
                e.y8, e.ldy8, e.y8_mul = qk8.data_ptr() + r0 * 2 * D, 2 * D, 8.0
            pb._add(abi.OP_EW, e, label)

        aq8 = self.attn_q8 and f8 and T >= 1024
        qk8 = pb.buf((T, 2 * D), torch.uint8, zero=True) if (self.attn_qk_f8 and T >= 1024) else None      # [token][q heads | k heads] e4m3

        vt8 = pb.buf((D, (T + 63) // 64 * 64), torch.uint8, zero=True) if (self.attn_pv_f8 and qk8 is not None and aq8) else None      # e4m3 V^T, reused by every block

        def attention(src, ld, out_t, out_ld, label, q8=None):
            pv = None
            if vt8 is not None and q8 is not None:
                pv = pb.v_f8t(src, T, H, ld, v_off=2 * D, out=vt8, label=label + ".v_f8t")
            pb.attention(src, src, src, None if q8 is not None else out_t, 1, H, T, T, hd, (0, ld, hd), (0, ld, hd), (0, ld, hd), (0, out_ld, hd),
                         1.0 / math.sqrt(hd), k_off=D, v_off=2 * D, label=label, q_prescaled=True, q8=q8,
                         qk_f8=(qk8, 0, D, 2 * D, -3) if qk8 is not None else None, pv_f8=pv)

        def swiglu(src, ld, c0, r0, r1, dst, dst_ld, dst_c0, label, dst8=None, consumers=()):
            """silu(a) * b of the two halves of a fused projection.  With fp8 consumers: one pass that writes their MX fp8 operand
            (MTX_QUANT_SWIGLU) and the 16-bit result only if somebody still reads it"""
            to8 = dst8 is not None and self.fused_quant and any(w.q is not None for w in consumers)
            need16 = not to8 or any(w.q is None for w in consumers)
            if to8:
                pb.quantize(src, r1 - r0, hid, ldx=ld, x_off=r0 * ld + c0, q=dst8[0], scale=dst8[1], row_off=r0, lds=lds, ldq=dst_ld, q_col_off=dst_c0,
                            swiglu_b=src, b_off=r0 * ld + c0 + hid, ldb=ld, y=dst if need16 else None, y_off=r0 * dst_ld + dst_c0, ldy=dst_ld,
                            label=label + ".q")
                return
            a_ = _rows(src, r0, r1, c0, hid)
            b_ = _rows(src, r0, r1, c0 + hid, hid)
            pb.ew(abi.EW_SWIGLU, a_, b=b_, out=_rows(dst, r0, r1, dst_c0, hid), label=label)
            if dst8 is not None:
                pb.quantize(dst, r1 - r0, hid, ldx=dst_ld, x_off=r0 * dst_ld + dst_c0, q=dst8[0], scale=dst8[1], row_off=r0, lds=lds, ldq=dst_ld,
                            q_col_off=dst_c0, label=label + ".q")

        res_gate = lambda i, rows: dict(gate=mod[i], gate_rows_per=rows, res=x)
        # double-stream blocks: norms, quantisers and SwiGLU run over both streams at once; the text stream's four linears (512 rows, a
        # fraction of one wave of tiles) ride the plan's side lane beside the image stream's
        for i, B in enumerate(self.blocks):
            tag = f"dbl{i}"
            pb.join()
            adaln(t_txt, T, 0, 1, tag + ".norm1", (B["qkv"],))
            adaln(0, t_txt, 6, 7, tag + ".norm1_ctx", (B["cqkv"],))
            with pb.side():
                linear(nrm, nrm8 if f8 else None, B["cqkv"], 0, t_txt, 3 * D, D, qkv, label=tag + ".qkv_ctx")
            linear(nrm, nrm8 if f8 else None, B["qkv"], t_txt, T, 3 * D, D, qkv, label=tag + ".qkv")
            pb.join()
            rope(qkv, t_txt, T, B["nqk"], 3 * D, tag + ".rope_qk")
            rope(qkv, 0, t_txt, B["cnqk"], 3 * D, tag + ".rope_qk_ctx")
            attention(qkv, 3 * D, o, D, tag + ".attn", q8=(o8[0], o8[1], D, lds, 0) if aq8 else None)
            if f8 and not aq8:
                quant(o, D, o8, 0, T, tag + ".attn.q")
            with pb.side():
                linear(o, o8 if f8 else None, B["cout"], 0, t_txt, D, D, x, label=tag + ".to_add_out", **res_gate(8, t_txt))
            linear(o, o8 if f8 else None, B["out"], t_txt, T, D, D, x, res_off=t_txt * D, label=tag + ".to_out", **res_gate(2, t_img))
            pb.join()
            adaln(t_txt, T, 3, 4, tag + ".norm2", (B["ff_in"],))
            adaln(0, t_txt, 9, 10, tag + ".norm2_ctx", (B["cff_in"],))
            glu = (lambda r0: dict(glu=(ffa8[0], ffa8[1], hid, lds, 0, r0, 0))) if self.glu_epilogue else (lambda r0: {})
            with pb.side():
                linear(nrm, nrm8 if f8 else None, B["cff_in"], 0, t_txt, 2 * hid, D, ffh, label=tag + ".ff_in_ctx", **glu(0))
            linear(nrm, nrm8 if f8 else None, B["ff_in"], t_txt, T, 2 * hid, D, ffh, label=tag + ".ff_in", **glu(t_txt))
            pb.join()
            if not self.glu_epilogue:
                swiglu(ffh, 2 * hid, 0, t_txt, T, ffa, hid, 0, tag + ".swiglu", ffa8 if f8 else None, (B["ff_out"],))
                swiglu(ffh, 2 * hid, 0, 0, t_txt, ffa, hid, 0, tag + ".swiglu_ctx", ffa8 if f8 else None, (B["cff_out"],))
            with pb.side():
                linear(ffa, ffa8 if f8 else None, B["cff_out"], 0, t_txt, D, hid, x, label=tag + ".ff_out_ctx", **res_gate(11, t_txt))
            linear(ffa, ffa8 if f8 else None, B["ff_out"], t_txt, T, D, hid, x, res_off=t_txt * D, label=tag + ".ff_out", **res_gate(5, t_img))
        pb.join()
        for i, S in enumerate(self.singles):
            tag = f"sgl{i}"
            adaln(0, T, 12, 13, tag + ".norm", (S["fused"],))
            linear(nrm, nrm8 if f8 else None, S["fused"], 0, T, FW, D, big, label=tag + ".to_qkv_mlp",
                   **(dict(glu=(cat8[0], cat8[1], D + hid, lds, 3 * D, 0, D)) if self.glu_epilogue else {}))
            rope(big, 0, T, S["nqk"], FW, tag + ".rope_qk")
            attention(big, FW, cat, D + hid, tag + ".attn", q8=(cat8[0], cat8[1], D + hid, lds, 0) if aq8 else None)
            if not self.glu_epilogue:
                swiglu(big, FW, 3 * D, 0, T, cat, D + hid, D, tag + ".swiglu", cat8 if f8 else None, (S["out"],))
            if f8 and S["out"].q is not None and not aq8:          # the attention half of the concatenation: its own quantiser pass over columns [0, D)
                pb.quantize(cat, T, D, ldx=D + hid, q=cat8[0], scale=cat8[1], lds=lds, ldq=D + hid, label=tag + ".attn.q")
            linear(cat, cat8 if f8 else None, S["out"], 0, T, D, D + hid, x, label=tag + ".to_out", **res_gate(14, T))
        pb.norm(x, nrm, t_noise, D, eps=1e-6, kind=0, mod_scale=mod[15], mod_shift=mod[16], rows_per=t_noise, ldmod=D,
                x_off=t_txt * D, y_off=t_txt * D, label="norm_out")
        vel = pb.gemm(nrm, W["proj_out"], t_noise, cfg["in_channels"], D, a_off=t_txt * D, out_f32=True, label="proj_out")
        plan = pb.build()
        plan.lat, plan.ctx_in, plan.mod, plan.vel, plan.x = lat, ctx_in, mod, vel, x
        plan.t_noise, plan.t_img, plan.T = t_noise, t_img, T
        return plan

    def plan_for(self, t_txt, h2, w2, rh2=None, rw2=None):
        rh2 = h2 if rh2 is None else rh2
        rw2 = w2 if rw2 is None else rw2
        key = (t_txt, h2, w2, rh2, rw2)
        if key not in self._plans:
            self._plans[key] = self._build(t_txt, h2, w2, rh2, rw2)
        return self._plans[key]

    def flops_per_step(self, t_txt, h2, w2, rh2=None, rw2=None):
        cfg = self.cfg
        D, hid = cfg["d"], self.hid
        t_img = h2 * w2 + (h2 if rh2 is None else rh2) * (w2 if rw2 is None else rw2)
        T = t_txt + t_img
        dbl = cfg["layers"] * (2 * T * D * 3 * D + 2 * T * D * D + 2 * T * D * 2 * hid + 2 * T * hid * D)
        sgl = cfg["single_layers"] * (2 * T * D * (3 * D + 2 * hid) + 2 * T * (D + hid) * D)
        attn = (cfg["layers"] + cfg["single_layers"]) * 4 * T * T * D
        return dict(gemm=dbl + sgl, attention=attn, attention_per_layer=4 * T * T * D, tokens=T)


class Flux2VAEHip(FluxVAEHip):
    """AutoencoderKLFlux2: the AutoencoderKL graphs of core/ml/flux.py with 32 latent channels and the 1x1 quant convs; the
    BatchNorm over the patchified latents (running statistics, no affine) is applied where the tokens are packed."""

    def __init__(self, provider, cfg: dict, device, lib=None):
        cfg = dict(cfg)
        cfg.setdefault("latent", 32)
        cfg.setdefault("quant_conv", True)
        cfg.setdefault("bn_eps", 1e-4)
        super().__init__(provider, cfg, device, lib=lib)
        self.bn_mean = provider("bn.running_mean").detach().float().to(self.device).contiguous()
        self.bn_std = torch.sqrt(provider("bn.running_var").detach().float().to(self.device) + cfg["bn_eps"]).contiguous()


class Flux2KleinHip:
    """diffusers-pipeline-shaped callable (Flux2KleinPipeline) built from the two graphs above."""

    def __init__(self, dit: Flux2DiTHip, vae: Flux2VAEHip, graph: bool = True):
        self.transformer, self.vae = dit, vae
        self.device = dit.device
        self._execution_device = dit.device
        self._graph = graph and not dit.lib.is_simulator
        self._lock = threading.Lock()
        self._embeds = None
        self.calls = 0                # pipeline invocations
        self.completed = 0            # ... that returned an image (a call that raised is caught by the OSB stage and becomes a flat fill)

    def set_prompt_embeds(self, prompt_embeds: torch.Tensor):
        """Qwen3 hidden states of the (fixed) prompt — computed once per process by the caller"""
        self._embeds = prompt_embeds.reshape(-1, prompt_embeds.shape[-1])

    def encode_prompt(self, prompt=None, device=None, **kw):
        if self._embeds is None:
            raise ModelError("no prompt embeddings: the loader encodes the fixed prompt once when the snapshot's text_encoder/ and tokenizer/ folders are staged next to "
                             "transformer/ (core/ml/prompt_embeds.py); otherwise run `python tools/export_prompt_embeds.py klein <pipeline snapshot>` (writes "
                             "prompt_embeds.safetensors) or hand the tensor to set_prompt_embeds()")
        n = self._embeds.shape[0]
        text_ids = torch.zeros(1, n, 4)
        text_ids[0, :, 3] = torch.arange(n)
        return self._embeds[None], text_ids

    @staticmethod
    def _reference_image(image):
        """what Flux2KleinPipeline does to the conditioning image: cap the area at 1024^2 (LANCZOS), floor both sides to multiples of 16"""
        from PIL import Image
        img = image.convert("RGB") if hasattr(image, "convert") else Image.fromarray(np.asarray(image, dtype=np.uint8))
        w, h = img.size
        if w * h > 1024 * 1024:
            sc = math.sqrt(1024 * 1024 / (w * h))
            img = img.resize((int(w * sc), int(h * sc)), Image.Resampling.LANCZOS)
            w, h = img.size
        w16, h16 = w // 16 * 16, h // 16 * 16
        if (w16, h16) != (w, h):
            l, t = (w - w16) // 2, (h - h16) // 2
            img = img.crop((l, t, l + w16, t + h16))
        return np.asarray(img, dtype=np.uint8)

    @torch.no_grad()
    def __call__(self, image=None, height=None, width=None, num_inference_steps=4, guidance_scale=1.0, generator=None,
                 prompt_embeds=None, output_type="pil", latents=None, **kw):
        self.calls += 1
        if prompt_embeds is None:
            prompt_embeds, _ = self.encode_prompt()
        pe = prompt_embeds.reshape(-1, prompt_embeds.shape[-1])
        ref_dev = None
        if torch.is_tensor(image) and image.dtype == torch.uint8 and image.dim() == 3 and image.shape[0] % 16 == 0 and image.shape[1] % 16 == 0 \
                and image.shape[0] * image.shape[1] <= 1024 * 1024:
            ref_dev, ref = image, image            # a conditioning image that is already on the device and needs none of the host-side fitting
        else:
            ref = self._reference_image(image.cpu().numpy() if torch.is_tensor(image) else image)
        RH, RW = ref.shape[:2]
        H = int(height) if height is not None else RH
        W = int(width) if width is not None else RW
        if RH < 16 or RW < 16 or H < 16 or W < 16:
            raise ModelError(f"FLUX.2 Klein needs at least 16x16 pixels, got image {RW}x{RH}, output {W}x{H}")
        h2, w2, rh2, rw2 = H // 16, W // 16, RH // 16, RW // 16
        dit, vae = self.transformer, self.vae
        C = dit.cfg["in_channels"]
        L = C // 4
        with self._lock:
            enc = vae.encoder_plan(RH, RW)
            enc.src.copy_((ref_dev if ref_dev is not None else torch.from_numpy(ref.copy()).to(self.device)).reshape(1, RH, RW, 3))
            enc.run(graph=self._graph)
            mean = enc.moments.t[0, :, :, :L].float()                                          # [RH/8, RW/8, L]
            tok = mean.view(rh2, 2, rw2, 2, L).permute(0, 2, 4, 1, 3).reshape(rh2 * rw2, C)    # channel = c*4 + dy*2 + dx
            ref_tok = (tok - vae.bn_mean) / vae.bn_std
            if latents is None:
                latents = torch.randn((1, C, h2, w2), generator=generator, dtype=torch.float32,
                                      device=generator.device if generator is not None else "cpu")
            lat = latents.to(self.device, torch.float32)[0].flatten(1).t().contiguous()        # [h2*w2, C]
            plan = dit.plan_for(pe.shape[0], h2, w2, rh2, rw2)
            plan.ctx_in.copy_(pe.to(self.device, dit.tdt))
            plan.lat[plan.t_noise:].copy_(ref_tok.to(dit.tdt))
            sig = flow_sigmas(num_inference_steps, h2 * w2)
            g = float(guidance_scale) if dit.cfg.get("guidance_embeds") else None
            for i in range(num_inference_steps):
                plan.mod.copy_(dit.modulation(float(sig[i]), g))
                plan.lat[: plan.t_noise].copy_(lat.to(dit.tdt))
                plan.run(graph=self._graph)
                lat = lat + (float(sig[i + 1]) - float(sig[i])) * plan.vel
            z = (lat * vae.bn_std + vae.bn_mean).view(h2, w2, L, 2, 2).permute(0, 3, 1, 4, 2).reshape(1, h2 * 2, w2 * 2, L)
            dec = vae.decoder_plan(h2 * 2, w2 * 2)
            dec.z.t.copy_(z.to(dit.tdt))
            dec.run(graph=self._graph)
            out = dec.out[0].clamp(0, 1).clone()
            self.last = dict(latents=lat, sigmas=sig, ref_tokens=ref_tok)
        self.completed += 1
        if output_type == "pt":
            return SimpleNamespace(images=[out])
        from PIL import Image
        u8 = out.mul(255).round().to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
        return SimpleNamespace(images=[Image.fromarray(u8)])


# ---- parameter inventories (diffusers names) ----------------------------------------------------------------
KLEIN_4B_DIT_CFG = dict(d=3072, heads=24, layers=5, single_layers=20, in_channels=128, joint_dim=7680, mlp_ratio=3.0,
                        axes_dim=(32, 32, 32, 32), rope_theta=2000.0, guidance_embeds=False)
KLEIN_9B_DIT_CFG = dict(d=4096, heads=32, layers=8, single_layers=24, in_channels=128, joint_dim=12288, mlp_ratio=3.0,
                        axes_dim=(32, 32, 32, 32), rope_theta=2000.0, guidance_embeds=False)
KLEIN_VAE_CFG = dict(ch=(128, 256, 512, 512), groups=32, latent=32, quant_conv=True, bn_eps=1e-4)


def dit_param_shapes(cfg: dict) -> dict:
    D, hd, hid = cfg["d"], cfg["d"] // cfg["heads"], int(cfg["d"] * cfg.get("mlp_ratio", 3.0))
    s = {}

    def lin(name, dout, din):
        s[name + ".weight"] = (dout, din)

    lin("x_embedder", D, cfg["in_channels"]); lin("context_embedder", D, cfg["joint_dim"]); lin("proj_out", cfg["in_channels"], D)
    for e in ["timestep_embedder"] + (["guidance_embedder"] if cfg.get("guidance_embeds") else []):
        lin(f"time_guidance_embed.{e}.linear_1", D, 256); lin(f"time_guidance_embed.{e}.linear_2", D, D)
    lin("double_stream_modulation_img.linear", 6 * D, D); lin("double_stream_modulation_txt.linear", 6 * D, D)
    lin("single_stream_modulation.linear", 3 * D, D); lin("norm_out.linear", 2 * D, D)
    for i in range(cfg["layers"]):
        p = f"transformer_blocks.{i}"
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(f"{p}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[f"{p}.attn.{n}.weight"] = (hd,)
        for ff in ("ff", "ff_context"):
            lin(f"{p}.{ff}.linear_in", 2 * hid, D); lin(f"{p}.{ff}.linear_out", D, hid)
    for i in range(cfg["single_layers"]):
        p = f"single_transformer_blocks.{i}.attn"
        lin(p + ".to_qkv_mlp_proj", 3 * D + 2 * hid, D); lin(p + ".to_out", D, D + hid)
        s[p + ".norm_q.weight"] = s[p + ".norm_k.weight"] = (hd,)
    return s


def vae_param_shapes(cfg: dict) -> dict:
    from .flux import vae_param_shapes as base
    L = cfg.get("latent", 32)
    s = base(cfg)
    c_last = cfg["ch"][-1]
    s["encoder.conv_out.weight"], s["encoder.conv_out.bias"] = (2 * L, c_last, 3, 3), (2 * L,)
    s["decoder.conv_in.weight"] = (c_last, L, 3, 3)
    s["quant_conv.weight"], s["quant_conv.bias"] = (2 * L, 2 * L, 1, 1), (2 * L,)
    s["post_quant_conv.weight"], s["post_quant_conv.bias"] = (L, L, 1, 1), (L,)
    s["bn.running_mean"] = s["bn.running_var"] = (4 * L,)
    return s
