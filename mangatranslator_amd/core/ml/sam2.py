"""SAM-2.1 (Hiera image encoder + box-prompted mask decoder) on libmtx_hip.

This is the model object behind `ModelManager.load_sam2()` (reference
core/ml/model_manager.py:982-1010 loads HF `Sam2Model`/`Sam2Processor`; the operator
`_process_simple_bubbles`, reference core/image/detection.py:475-511, calls
processor -> model(multimask_output=False) -> post_process_masks -> boolean masks).
Weights use the HF `Sam2Model.state_dict()` key names; the network definition followed is
transformers' modeling_sam2.py (cited per function below as `hf:<line>`).

Graph design (MI355X-first):
  * tokens are kept in WINDOW-MAJOR order for a whole Hiera stage: every op except attention is
    per-token, global attention is permutation-equivariant, so window partition / unpartition
    (hf:397-452) vanish — attention reads each window as a contiguous [ws*ws, C] slab through
    strides.  Layouts change only where the window size does (one row-gather per change).
  * patch embedding = im2col (emitted directly in window-major order) + GEMM with the
    bicubic-interpolated position table fused as the GEMM residual (hf:106-137, 638-644).
  * q-pooling = NHWC 2x2 max-pool over [windows, ws, ws, C] views (hf:295-303, 337-341).
  * FPN neck 1x1 convs are GEMMs; conv_s0/conv_s1 (hf:1608-1609) are folded into the level-0/1
    lateral convs at load time (two chained linear maps), no_memory_embedding and the
    no-mask dense prompt are folded into the level-2 bias.
  * mask decoder: k/q projections of (keys + image PE) are split as W*keys + (W*PE): the PE part is
    a per-layer constant table fused as a batch-broadcast GEMM residual.  ConvTranspose2d(k=2,s=2)
    = 1x1 conv + pixel-shuffle store addressing.  Mask logits stay fp32; the stability rule
    (hf:1265-1311) picks a channel per box on the device and the bilinear upsample to page size is
    fused with the > 0 threshold, writing the page-resolution bitmask directly.
"""
import math
import threading

import numpy as np
import torch
import torch.nn.functional as F

from ...hip import abi
from ...hip.lib import get_library
from ...hip.plan import Act, AsyncLane, LaneTicket, PlanBuilder, PlanCache, result_tensors
from ...utils.exceptions import ModelError

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def hiera_hparams(config) -> dict:
    """Pull the numbers the graph needs out of an HF Sam2Config (or an equivalent dict tree)."""
    vc = _cfg_get(config, "vision_config")
    bb = _cfg_get(vc, "backbone_config")
    pe = _cfg_get(config, "prompt_encoder_config")
    md = _cfg_get(config, "mask_decoder_config")
    img = _cfg_get(bb, "image_size")
    img = img[0] if isinstance(img, (list, tuple)) else img
    return dict(
        image_size=int(img), embed=list(_cfg_get(bb, "embed_dim_per_stage")), blocks=list(_cfg_get(bb, "blocks_per_stage")),
        heads=list(_cfg_get(bb, "num_attention_heads_per_stage")), windows=list(_cfg_get(bb, "window_size_per_stage")),
        global_blocks=list(_cfg_get(bb, "global_attention_blocks")), q_pool_stages=int(_cfg_get(bb, "num_query_pool_stages")),
        mlp_ratio=float(_cfg_get(bb, "mlp_ratio")), ln_eps=float(_cfg_get(bb, "layer_norm_eps")),
        fpn_dim=int(_cfg_get(vc, "fpn_hidden_size")), top_down=list(_cfg_get(vc, "fpn_top_down_levels")),
        dec_dim=int(_cfg_get(md, "hidden_size")), dec_heads=int(_cfg_get(md, "num_attention_heads")),
        dec_layers=int(_cfg_get(md, "num_hidden_layers")), dec_mlp=int(_cfg_get(md, "mlp_dim")),
        dec_down=int(_cfg_get(md, "attention_downsample_rate")),
        stab_delta=float(_cfg_get(md, "dynamic_multimask_stability_delta")),
        stab_thresh=float(_cfg_get(md, "dynamic_multimask_stability_thresh")),
        patch=int(_cfg_get(pe, "patch_size")),
    )


def window_order(hs: int, ws_: int, win: int) -> np.ndarray:
    """raster index of the token at each window-major position (windows row-major, row-major inside)."""
    assert hs % win == 0 and ws_ % win == 0, f"feature map {hs}x{ws_} not divisible by window {win}"
    idx = np.arange(hs * ws_, dtype=np.int64).reshape(hs // win, win, ws_ // win, win)
    return idx.transpose(0, 2, 1, 3).reshape(-1)


class Sam2Hip:
    def __init__(self, state_dict: dict, config, device="cuda", lib=None, graph: bool = True, dtype: int = abi.BF16, precision: str = "fast"):
        """precision = "high" (what `ModelManager` serves; measured on MI355X, Hiera-L at 1024x1536, logits at a trained model's spread, against
        the fp32 reference: mask pixels that differ after `> 0` 5.1e-5 against 1.8e-4 for "fast", logit rms error 0.0051 against 0.016, no pixel
        wrong outside the error band — profiles/r05_parity.json, r06_sam_frontier.json) is three independent parts, each selectable on its own
        for pricing ("hilo", "stream32", "dec32", joined with "+"; "high" = all three, "fast" = none):
          hilo      the trunk's and the neck's weights as PAIRS of the storage type, W = W_hi + W_lo, both multiplied with the same operand tile
                    into one fp32 accumulator (mtx_gemm_args.w_lo): the weights' rounding — the largest term of the error budget
                    — goes away at twice the trunk's matrix work (round 5 ran it as a GEMM over K' = 2K against a copied [x | x] operand);
          stream32  the residual stream stays fp32 from the patch embedding to the neck: the GEMMs that close a branch add the fp32 residual
                    in their epilogue and write fp32, the LayerNorms read fp32 and round once to the next linear's operand type;
          dec32     prompt encoder / two-way transformer / mask head with fp32 operands and arithmetic (csrc/f32ops.hip)."""
        parts = {"fast": set(), "high": {"hilo", "stream32", "dec32"}}.get(precision)
        if parts is None:
            parts = set(precision.split("+"))
            if not parts <= {"hilo", "stream32", "dec32"}:
                raise ModelError("SAM-2: precision must be 'fast', 'high' or a '+'-joined subset of hilo / stream32 / dec32")
        self.precision = precision
        self.hilo, self.stream32, self.dec32 = "hilo" in parts, "stream32" in parts, "dec32" in parts
        self.high = self.hilo and self.stream32 and self.dec32
        self.lib = lib if lib is not None else get_library()
        self.device = torch.device(device)
        self.hp = hiera_hparams(config)
        if dtype not in (abi.BF16, abi.F16):
            raise ModelError("SAM-2: storage dtype must be bf16 or f16")
        self.dtype = dtype
        self.tdt = torch.bfloat16 if dtype == abi.BF16 else torch.float16
        # the prompt encoder / two-way transformer / mask head: the storage type, or fp32 operands and arithmetic under precision "high"
        self.ddtype, self.ddt = (abi.F32, torch.float32) if self.dec32 else (self.dtype, self.tdt)
        self._graph = graph and not self.lib.is_simulator
        self._lane = AsyncLane(self.device, self.lib.is_simulator)
        self._enc = None
        self._dec = PlanCache(6)
        self._post = PlanCache(8)
        self._pre = PlanCache(6)
        hp = self.hp
        if hp["dec_dim"] != hp["fpn_dim"]:
            raise ModelError("SAM-2: decoder hidden size must equal the FPN width")
        self._pack({k: v.detach().float().cpu() for k, v in state_dict.items()})

    # ------------------------------------------------------------------------------------------
    def _c(self, t, dtype=None):
        return t.to(device=self.device, dtype=dtype if dtype is not None else self.tdt).contiguous()

    def _cd(self, t):
        """a matrix of the mask decoder in ITS operand type"""
        return self._c(t, self.ddt)

    def _cw(self, w):
        """matrix [N, K] of a trunk / neck linear: (W, None) in the storage type, or with "hilo" the pair (W_hi, W_lo), W_lo = round(W - W_hi)"""
        if not self.hilo:
            return self._c(w), None
        hi = w.to(self.tdt)
        lo = (w - hi.float()).to(self.tdt)
        return self._c(hi), self._c(lo)

    def _pack(self, sd):
        hp, W = self.hp, {}
        S = hp["image_size"]
        g0 = S // 4
        C0 = hp["embed"][0]
        bbp = "vision_encoder.backbone."
        # patch embed as GEMM over im2col rows: K = 49 taps x 8 channels (3 real)
        wpe = sd[bbp + "patch_embed.projection.weight"]                      # [C0,3,7,7]
        wk = torch.zeros(C0, 49, 8)
        wk[:, :, :3] = wpe.permute(0, 2, 3, 1).reshape(C0, 49, 3)
        W["pe"] = (*self._cw(wk.reshape(C0, 392)), self._c(sd[bbp + "patch_embed.projection.bias"], torch.float32))
        # position table (hf:638-644), stored in the first stage's window-major order
        pos = F.interpolate(sd[bbp + "pos_embed"], size=(g0, g0), mode="bicubic")
        win = sd[bbp + "pos_embed_window"]
        pos = pos + win.tile([1, 1, g0 // win.shape[2], g0 // win.shape[3]])
        pos = pos[0].permute(1, 2, 0).reshape(g0 * g0, C0)
        order0 = torch.from_numpy(window_order(g0, g0, hp["windows"][0]))
        W["pos"] = self._c(pos[order0])
        # blocks
        self.blocks = []
        total = 0
        for si, nb in enumerate(hp["blocks"]):
            for bi in range(nb):
                p = f"{bbp}blocks.{total}."
                dim = hp["embed"][si - 1] if (si > 0 and bi == 0) else hp["embed"][si]
                dim_out = hp["embed"][si]
                win_sz = hp["windows"][si - 1] if (si > 0 and bi == 0) else hp["windows"][si]
                if total in hp["global_blocks"]:
                    win_sz = 0
                pool = 0 < si <= hp["q_pool_stages"] and bi == 0
                blk = dict(dim=dim, dim_out=dim_out, win=win_sz, pool=pool, heads=hp["heads"][si], idx=total,
                           stage_end=(bi == nb - 1))
                for nm in ("layer_norm1", "layer_norm2"):
                    blk[nm] = (self._c(sd[p + nm + ".weight"], torch.float32), self._c(sd[p + nm + ".bias"], torch.float32))
                for nm, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.proj_in"), ("fc2", "mlp.proj_out")):
                    blk[nm] = (*self._cw(sd[p + key + ".weight"]), self._c(sd[p + key + ".bias"], torch.float32))
                    blk[nm + "_n"] = sd[p + key + ".weight"].shape[0]
                if dim != dim_out:
                    blk["skip"] = (*self._cw(sd[p + "proj.weight"]), self._c(sd[p + "proj.bias"], torch.float32))
                self.blocks.append(blk)
                total += 1
        # neck (hf:216-265): convs[n-i] serves level i; levels 0/1 are folded with conv_s0/conv_s1
        n = len(hp["embed"]) - 1
        D = hp["fpn_dim"]
        lat = {}
        for i in range(n + 1):
            lat[i] = (sd[f"vision_encoder.neck.convs.{n - i}.weight"].reshape(D, -1), sd[f"vision_encoder.neck.convs.{n - i}.bias"])
        ws0, bs0 = sd["mask_decoder.conv_s0.weight"].reshape(-1, D), sd["mask_decoder.conv_s0.bias"]
        ws1, bs1 = sd["mask_decoder.conv_s1.weight"].reshape(-1, D), sd["mask_decoder.conv_s1.bias"]
        W["neck0"] = (*self._cw(ws0 @ lat[0][0]), self._c(ws0 @ lat[0][1] + bs0, torch.float32))
        W["neck1"] = (*self._cw(ws1 @ lat[1][0]), self._c(ws1 @ lat[1][1] + bs1, torch.float32))
        fold = sd["no_memory_embedding"].reshape(-1) + sd["prompt_encoder.no_mask_embed.weight"].reshape(-1)
        if 2 in hp["top_down"]:
            W["neck2"] = (*self._cw(lat[2][0]), self._c(lat[2][1] + fold, torch.float32))
            W["neck3"] = (*self._cw(lat[3][0]), self._c(lat[3][1], torch.float32))
        else:
            W["neck2"] = (*self._cw(lat[2][0]), self._c(lat[2][1] + fold, torch.float32))
            W["neck3"] = None
        # dense image positional encoding (hf:1341-1353) and prompt tables
        ge = S // hp["patch"]
        G = sd["shared_image_embedding.positional_embedding"]               # [2, D/2]
        ys = (torch.arange(ge, dtype=torch.float32) + 0.5) / ge
        coords = torch.stack([ys[None, :].expand(ge, ge), ys[:, None].expand(ge, ge)], dim=-1)   # (x, y)
        proj = (2 * coords - 1) @ G * (2 * math.pi)
        img_pe = torch.cat([proj.sin(), proj.cos()], dim=-1).reshape(ge * ge, D)
        self.img_pe_f32 = img_pe
        self.prompt_G = sd["prompt_encoder.shared_embedding.positional_embedding"].numpy().astype(np.float32)
        self.point_embed = sd["prompt_encoder.point_embed.weight"].numpy().astype(np.float32)
        self.not_a_point = sd["prompt_encoder.not_a_point_embed.weight"].numpy().astype(np.float32)
        self.out_tokens = torch.cat([sd["mask_decoder.obj_score_token.weight"], sd["mask_decoder.iou_token.weight"],
                                     sd["mask_decoder.mask_tokens.weight"]], 0).numpy().astype(np.float32)
        # decoder
        md = "mask_decoder.transformer."

        def lin(key):
            return (self._cd(sd[key + ".weight"]), self._c(sd[key + ".bias"], torch.float32))

        def attn(prefix, pe_side):
            a = {k: lin(f"{prefix}.{k}_proj") for k in ("q", "k", "v", "o")}
            if pe_side:   # projection of the dense PE, added as a batch-broadcast residual
                a["pe"] = self._cd(img_pe @ sd[f"{prefix}.{pe_side}_proj.weight"].t())
            return a

        self.dec_layers = []
        for li in range(hp["dec_layers"]):
            p = f"{md}layers.{li}."
            L = dict(self_attn=attn(p + "self_attn", None), t2i=attn(p + "cross_attn_token_to_image", "k"),
                     i2t=attn(p + "cross_attn_image_to_token", "q"),
                     fc1=lin(p + "mlp.proj_in"), fc2=lin(p + "mlp.proj_out"))
            sa = L["self_attn"]
            sa["qk"] = (torch.cat([sa["q"][0], sa["k"][0]], 0).contiguous(), torch.cat([sa["q"][1], sa["k"][1]], 0).contiguous())
            for i in (1, 2, 3, 4):
                L[f"ln{i}"] = (self._c(sd[f"{p}layer_norm{i}.weight"], torch.float32), self._c(sd[f"{p}layer_norm{i}.bias"], torch.float32))
            self.dec_layers.append(L)
        self.final_attn = attn(md + "final_attn_token_to_image", "k")
        self.ln_final = (self._c(sd[md + "layer_norm_final_attn.weight"], torch.float32), self._c(sd[md + "layer_norm_final_attn.bias"], torch.float32))
        # ConvTranspose2d(k=2,s=2) [Cin, Cout, 2, 2] -> 1x1 conv rows ((dy*2+dx)*Cout + co)
        for nm in ("upscale_conv1", "upscale_conv2"):
            wt = sd[f"mask_decoder.{nm}.weight"]
            ci, co = wt.shape[:2]
            W[nm] = (self._cd(wt.permute(2, 3, 1, 0).reshape(4 * co, 1, ci)), self._c(sd[f"mask_decoder.{nm}.bias"].repeat(4), torch.float32))
        W["up_ln"] = (self._c(sd["mask_decoder.upscale_layer_norm.weight"], torch.float32), self._c(sd["mask_decoder.upscale_layer_norm.bias"], torch.float32))
        W["hyper"] = [[lin(f"mask_decoder.output_hypernetworks_mlps.{i}.proj_in"), lin(f"mask_decoder.output_hypernetworks_mlps.{i}.layers.0"),
                       lin(f"mask_decoder.output_hypernetworks_mlps.{i}.proj_out")] for i in range(4)]
        W["iou"] = [lin("mask_decoder.iou_prediction_head.proj_in"), lin("mask_decoder.iou_prediction_head.layers.0"),
                    lin("mask_decoder.iou_prediction_head.proj_out")]
        self.W = W

    # ------------------------------------------------------------------------------------------
    def _gather_map(self, hs, ws_, win_from, win_to):
        """index map for row_gather: token order `win_to` reading from token order `win_from` (0 = raster)."""
        raster_to = window_order(hs, ws_, win_to) if win_to else np.arange(hs * ws_)
        if win_from:
            pos_from = np.empty(hs * ws_, dtype=np.int64)
            pos_from[window_order(hs, ws_, win_from)] = np.arange(hs * ws_)
            idx = pos_from[raster_to]
        else:
            idx = raster_to
        return torch.from_numpy(idx.astype(np.int32)).to(self.device)

    def _build_encoder(self):
        hp, W = self.hp, self.W
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        S = hp["image_size"]
        g = S // 4
        img = pb.act(1, S, S, 8)
        C0 = hp["embed"][0]
        win0 = hp["windows"][0]
        T = g * g
        # "hilo": every trunk / neck linear carries its weight pair (w, w_lo, bias) — one launch, one operand.  "stream32": the residual stream x
        # stays fp32 from the patch embedding to the neck — the projections that close a branch (attn.proj, mlp.proj_out) take the fp32 residual
        # in their epilogue and write fp32; the LayerNorms read fp32 and round once, to the storage type of the GEMM that follows.  Measured on
        # the fp32 oracle with the roundings injected (Hiera-L): with hi + lo weights the stream's 96 roundings are 90 % of what
        # is left of the trunk's error.  Neither part adds a launch to the "fast" op list (round 5's form added six per block).
        s32 = self.stream32
        f32 = torch.float32

        def layer_norm(x, rows, c, wb, label):
            """the operand of the linears that follow: [rows, c] of the storage type"""
            y = pb.buf((rows, c), self.tdt)
            if not s32:
                return pb.norm(x, y, rows, c, ldy=c, gamma=wb[0], beta=wb[1], eps=eps, label=label)
            return pb.norm(x, y, rows, c, gamma=wb[0], beta=wb[1], eps=eps, dtype=abi.F32, out_dtype=self.dtype, label=label)

        def linear(a, wb, rows, n_out, k, label, **kw):
            return pb.gemm(a, wb[0], rows, n_out, k, bias=wb[2], w_lo=wb[1], label=label, **kw)

        def close_branch(a, wb, rows, n_out, k, res, label):
            """stream <- res + a @ W^T + b"""
            if not s32:
                return linear(a, wb, rows, n_out, k, label, res=res)
            return linear(a, wb, rows, n_out, k, label, res=res, res_f32=True, out_f32=True)

        cols = pb.buf((T, 392), self.tdt)
        order0 = pb.hold(torch.from_numpy(window_order(g, g, win0).astype(np.int32)).to(self.device))
        pb.im2col(img, cols, 7, 4, 392, row_map=order0, label="patch_im2col")
        eps = hp["ln_eps"]
        x = close_branch(cols, W["pe"], T, C0, 392, pb.const(W["pos"].float()) if s32 else W["pos"], "patch_embed")
        layout, hs = win0, g
        feats = {}
        stage = 0
        for blk in self.blocks:
            dim, dout, heads = blk["dim"], blk["dim_out"], blk["heads"]
            d = dout // heads
            win = blk["win"]
            tag = f"blk{blk['idx']}"
            if win and win != layout:
                m = pb.hold(self._gather_map(hs, hs, layout, win))
                x = pb.row_gather(x, pb.buf((T, dim), f32 if s32 else self.tdt), m, T, dim, label=tag + ".relayout", dtype=abi.F32 if s32 else None)
                layout = win
            ln1 = layer_norm(x, T, dim, blk["layer_norm1"], tag + ".ln1")
            qkv = linear(ln1, blk["qkv"], T, 3 * dout, dim, tag + ".qkv")
            wtok = win * win if win else T
            nwin = T // wtok
            if dim != dout and not blk["pool"]:
                raise ModelError("SAM-2: a channel change without q-pooling is not built")
            if blk["pool"]:
                if not win:
                    raise ModelError("SAM-2: q-pooling inside a global-attention block is not supported")
                res_full = linear(ln1, blk["skip"], T, dout, dim, tag + ".skip")
                res = pb.ew(abi.EW_MAXPOOL, Act(res_full.view(nwin, win, win, dout), nwin, win, win, dout), i0=2, i1=2, label=tag + ".skip_pool")
                qv = Act(qkv.view(nwin, win, win, 3 * dout), nwin, win, win, dout, 0)
                qp = pb.ew(abi.EW_MAXPOOL, qv, i0=2, i1=2, label=tag + ".q_pool")
                Tq, sq = T // 4, wtok // 4
                q_t, q_str = qp.t, (sq * dout, dout, d)
                res_t = pb.cvt_f32(res, self.dtype, label=tag + ".skip_f32").t.view(Tq, dout) if s32 else res.t
            else:
                Tq, sq = T, wtok
                q_t, q_str = qkv, (wtok * 3 * dout, 3 * dout, d)
                res_t = x
            o = pb.buf((Tq, dout), self.tdt)
            kv_str = (wtok * 3 * dout, 3 * dout, d)
            pb.attention(q_t, qkv, qkv, o, nwin, heads, sq, wtok, d, q_str, kv_str, kv_str, (sq * dout, dout, d),
                         1.0 / math.sqrt(d), k_off=dout, v_off=2 * dout, label=tag + ".attn")
            x1 = close_branch(o, blk["proj"], Tq, dout, dout, res_t, tag + ".proj")
            ln2 = layer_norm(x1, Tq, dout, blk["layer_norm2"], tag + ".ln2")
            hdim = blk["fc1_n"]
            h = linear(ln2, blk["fc1"], Tq, hdim, dout, tag + ".fc1", act=abi.ACT_GELU)
            x = close_branch(h, blk["fc2"], Tq, dout, hdim, x1, tag + ".fc2")
            if blk["pool"]:
                T, hs, layout = Tq, hs // 2, win // 2
            if blk["stage_end"]:
                feats[stage] = (x, T, hs, layout, dout)
                stage += 1
        # ---- neck ---------------------------------------------------------------------------------
        D = hp["fpn_dim"]

        def lateral(level, wb, cout):
            xt, Tn, hn, lay, cin = feats[level]
            if s32:                 # the stage output is the fp32 stream: rounded once, to the lateral projection's operand type
                xt = pb.cvt16(xt, pb.buf((Tn, cin), self.tdt), Tn, cin, copies=1, label=f"neck{level}.cvt")
            y = linear(xt, wb, Tn, cout, cin, f"neck{level}")
            m = pb.hold(self._gather_map(hn, hn, lay, 0))
            r = pb.act(1, hn, hn, cout)
            pb.row_gather(y, r.t, m, Tn, cout, label=f"neck{level}.to_raster")
            return r

        c0 = W["neck0"][0].shape[0]
        c1 = W["neck1"][0].shape[0]
        feat_s0 = lateral(0, W["neck0"], c0)
        feat_s1 = lateral(1, W["neck1"], c1)
        lat2 = lateral(2, W["neck2"], D)
        if W["neck3"] is not None:
            lat3 = lateral(3, W["neck3"], D)
            src = pb.ew(abi.EW_UPSAMPLE2X, lat3, b=lat2, label="fpn_top_down")
        else:
            src = lat2
        plan = pb.build()
        plan.img, plan.feat_s0, plan.feat_s1, plan.src = img, feat_s0, feat_s1, src
        return plan

    # ------------------------------------------------------------------------------------------
    def _dec_attention(self, pb, A, q_in, k_in, v_in, n, sq, sk, internal, q_pe=None, k_pe=None, tag=""):
        """Sam2Attention (hf:896-957) for n boxes: q_in [n*sq, D], k_in/v_in [n*sk, D]."""
        D, heads = self.hp["dec_dim"], self.hp["dec_heads"]
        d = internal // heads
        if q_pe is not None:
            q = pb.gemm(q_in, A["q"][0], sq, internal, D, bias=A["q"][1], res=q_pe, batch=n, a_bs=sq * D, c_bs=sq * internal, res_bs=0, label=tag + ".q")
        else:
            q = pb.gemm(q_in, A["q"][0], n * sq, internal, D, bias=A["q"][1], label=tag + ".q")
        if k_pe is not None:
            k = pb.gemm(k_in, A["k"][0], sk, internal, D, bias=A["k"][1], res=k_pe, batch=n, a_bs=sk * D, c_bs=sk * internal, res_bs=0, label=tag + ".k")
        else:
            k = pb.gemm(k_in, A["k"][0], n * sk, internal, D, bias=A["k"][1], label=tag + ".k")
        v = pb.gemm(v_in, A["v"][0], n * sk, internal, D, bias=A["v"][1], label=tag + ".v")
        o = pb.buf((n * sq, internal), self.ddt)
        pb.attention(q, k, v, o, n, heads, sq, sk, d, (sq * internal, internal, d), (sk * internal, internal, d),
                     (sk * internal, internal, d), (sq * internal, internal, d), 1.0 / math.sqrt(d), label=tag + ".attn")
        return o

    def _build_decoder(self, n):
        hp, W = self.hp, self.W
        enc = self._encoder()
        pb = PlanBuilder(self.lib, self.device, self.ddtype)
        D = hp["dec_dim"]
        ge = hp["image_size"] // hp["patch"]
        P = ge * ge
        NT = 9
        inner = D // hp["dec_down"]
        tok0 = pb.buf((n * NT, D), self.ddt)                 # point embeddings (input, also the query PE)
        bidx = pb.hold((torch.arange(n * P, dtype=torch.int32) % P).to(self.device))
        src, feat_s0, feat_s1 = enc.src, enc.feat_s0, enc.feat_s1
        if self.dec32:                                       # the encoder's three outputs (16-bit) enter the fp32 plan through one conversion each
            src, feat_s0, feat_s1 = (pb.cvt_f32(t, self.dtype, label=f"dec.{nm}_f32") for t, nm in ((src, "src"), (feat_s0, "feat_s0"), (feat_s1, "feat_s1")))
        keys = pb.row_gather(src.t, pb.buf((n * P, D), self.ddt), bidx, n * P, D, label="dec.keys_bcast")
        queries = tok0

        def ln(x, rows, wb, label):
            return pb.norm(x, pb.buf((rows, D), self.ddt), rows, D, gamma=wb[0], beta=wb[1], eps=1e-5, label=label)

        def up(x, wb, cout, skip, label):
            """ConvTranspose2d(k=2, s=2) + the skip feature: the conv kernel's pixel-shuffle store in the 16-bit plans, a GEMM over the
            pixels and a shuffle-add map in the fp32 plan"""
            if not self.dec32:
                return pb.conv2d(x, wb[0], wb[1], 4 * cout, ksize=1, pixel_shuffle=2, res=skip, res_broadcast=True, label=label)
            rows = x.n * x.h * x.w
            cols = pb.gemm(x.t, wb[0], rows, 4 * cout, x.c, bias=wb[1], label=label + ".gemm")
            return pb.shuffle2_add(cols, x.n, x.h, x.w, cout, skip=skip, label=label)

        def add_pe(x, label):
            xa = Act(x.view(1, 1, n * NT, D), 1, 1, n * NT, D)
            return pb.ew(abi.EW_ADD, xa, b=Act(tok0.view(1, 1, n * NT, D), 1, 1, n * NT, D), label=label).t.view(n * NT, D)

        for li, L in enumerate(self.dec_layers):
            tag = f"dec{li}"
            sa = L["self_attn"]
            if li == 0:      # skip_first_layer_pe: q = k = v = queries, output replaces queries (hf:985-987)
                qin = queries
            else:
                qin = add_pe(queries, tag + ".sa_pe")
            qk = pb.gemm(qin, sa["qk"][0], n * NT, 2 * D, D, bias=sa["qk"][1], label=tag + ".sa_qk")
            v = pb.gemm(queries, sa["v"][0], n * NT, D, D, bias=sa["v"][1], label=tag + ".sa_v")
            o = pb.buf((n * NT, D), self.ddt)
            dh = D // hp["dec_heads"]
            pb.attention(qk, qk, v, o, n, hp["dec_heads"], NT, NT, dh, (NT * 2 * D, 2 * D, dh), (NT * 2 * D, 2 * D, dh),
                         (NT * D, D, dh), (NT * D, D, dh), 1.0 / math.sqrt(dh), k_off=D, label=tag + ".sa_attn")
            queries = pb.gemm(o, sa["o"][0], n * NT, D, D, bias=sa["o"][1], res=None if li == 0 else queries, label=tag + ".sa_o")
            queries = ln(queries, n * NT, L["ln1"], tag + ".ln1")
            # tokens -> image
            qin = add_pe(queries, tag + ".t2i_pe")
            o = self._dec_attention(pb, L["t2i"], qin, keys, keys, n, NT, P, inner, k_pe=L["t2i"]["pe"], tag=tag + ".t2i")
            queries = pb.gemm(o, L["t2i"]["o"][0], n * NT, D, inner, bias=L["t2i"]["o"][1], res=queries, label=tag + ".t2i_o")
            queries = ln(queries, n * NT, L["ln2"], tag + ".ln2")
            # MLP (ReLU, hf:367-392 with activation="relu")
            h = pb.gemm(queries, L["fc1"][0], n * NT, hp["dec_mlp"], D, bias=L["fc1"][1], act=abi.ACT_RELU, label=tag + ".fc1")
            queries = pb.gemm(h, L["fc2"][0], n * NT, D, hp["dec_mlp"], bias=L["fc2"][1], res=queries, label=tag + ".fc2")
            queries = ln(queries, n * NT, L["ln3"], tag + ".ln3")
            # image -> tokens
            kin = add_pe(queries, tag + ".i2t_pe")
            o = self._dec_attention(pb, L["i2t"], keys, kin, queries, n, P, NT, inner, q_pe=L["i2t"]["pe"], tag=tag + ".i2t")
            keys2 = pb.gemm(o, L["i2t"]["o"][0], n * P, D, inner, bias=L["i2t"]["o"][1], res=keys, label=tag + ".i2t_o")
            keys = ln(keys2, n * P, L["ln4"], tag + ".ln4")
        qin = add_pe(queries, "dec.final_pe")
        o = self._dec_attention(pb, self.final_attn, qin, keys, keys, n, NT, P, inner, k_pe=self.final_attn["pe"], tag="dec.final")
        queries = pb.gemm(o, self.final_attn["o"][0], n * NT, D, inner, bias=self.final_attn["o"][1], res=queries, label="dec.final_o")
        queries = ln(queries, n * NT, self.ln_final, "dec.ln_final")

        # ---- upscaling (hf:1203-1209) ----------------------------------------------------------------
        ka = Act(keys.view(n, ge, ge, D), n, ge, ge, D)
        c1 = W["upscale_conv1"][0].shape[0] // 4
        up1 = up(ka, W["upscale_conv1"], c1, feat_s1, "dec.up1")
        rows1 = n * up1.h * up1.w
        up1n = pb.norm(up1.t, pb.buf((rows1, c1), self.ddt), rows1, c1, gamma=W["up_ln"][0], beta=W["up_ln"][1], eps=1e-6,
                       act=abi.ACT_GELU, label="dec.up_ln_gelu")
        c2 = W["upscale_conv2"][0].shape[0] // 4
        up2 = up(Act(up1n.view(n, up1.h, up1.w, c1), n, up1.h, up1.w, c1), W["upscale_conv2"], c2, feat_s0, "dec.up2")
        up2a = pb.ew(abi.EW_ACT, up2, act=abi.ACT_GELU, label="dec.up2_gelu")
        hl = up2.h
        # ---- hypernetwork MLPs + IoU head on single token rows -------------------------------------------
        hyper = pb.buf((n, 4, c2), self.ddt)
        for i in range(4):
            (w1, b1), (w2, b2), (w3, b3) = W["hyper"][i]
            t = pb.gemm(queries, w1, n, D, D, lda=NT * D, a_off=(2 + i) * D, bias=b1, act=abi.ACT_RELU, label=f"dec.hyper{i}.0")
            t = pb.gemm(t, w2, n, D, D, bias=b2, act=abi.ACT_RELU, label=f"dec.hyper{i}.1")
            pb.gemm(t, w3, n, c2, D, bias=b3, out=hyper, ldc=4 * c2, c_off=i * c2, label=f"dec.hyper{i}.2")
        (w1, b1), (w2, b2), (w3, b3) = W["iou"]
        t = pb.gemm(queries, w1, n, D, D, lda=NT * D, a_off=1 * D, bias=b1, act=abi.ACT_RELU, label="dec.iou.0")
        t = pb.gemm(t, w2, n, D, D, bias=b2, act=abi.ACT_RELU, label="dec.iou.1")
        iou = pb.gemm(t, w3, n, 4, D, bias=b3, act=abi.ACT_SIGMOID, out_f32=True, label="dec.iou.2")
        # ---- masks = hyper_in @ upscaled (hf:1218-1220), fp32 [n, pix, 4] --------------------------------
        pix = hl * hl
        logits = pb.gemm(up2a.t, hyper, pix, 4, c2, batch=n, a_bs=pix * c2, w_bs=4 * c2, c_bs=pix * 4, out_f32=True, label="dec.masks")
        counts = pb.buf((n, 2), torch.int32, zero=True)
        sel = pb.buf((n,), torch.int32, zero=True)
        pb.mask_select(logits, iou, counts, sel, n, pix, hp["stab_delta"], hp["stab_thresh"])
        plan = pb.build()
        plan.tok0, plan.logits, plan.iou, plan.sel, plan.hl = tok0, logits, iou, sel, hl
        return plan

    # ------------------------------------------------------------------------------------------
    def _encoder(self):
        if self._enc is None:
            self._enc = self._build_encoder()
        return self._enc

    def _decoder(self, n):
        if n not in self._dec:
            self._dec[n] = self._build_decoder(n)
        return self._dec[n]

    def _pre_plan(self, h, w):
        if (h, w) not in self._pre:
            enc = self._encoder()
            pb = PlanBuilder(self.lib, self.device, self.dtype)
            page = pb.buf((h, w, 3), torch.uint8)
            # Sam2ImageProcessorFast resizes the uint8 page with torchvision's antialiased bilinear `resize`, i.e. ATen's fixed-point
            # uint8 kernel (horizontal pass, uint8 intermediate, vertical pass), then rescales and normalises in fp32: the two passes run
            # with ATen's own taps (core/image/device_tail.py aten_aa_bilinear_tables, pinned against the installed torch), the
            # normalisation in the preprocess kernel at identity size
            from ..image.device_tail import aten_aa_bilinear_tables
            S = enc.img.h
            cur, cur_h, cur_w = page, h, w
            if w != enc.img.w:
                b, t, k, bits = aten_aa_bilinear_tables(w, enc.img.w)
                tmp = pb.buf((h, enc.img.w, 3), torch.uint8)
                pb.resample_u8(cur, tmp, h, enc.img.w, 3, cur_w * 3, enc.img.w * 3, pb.const(torch.from_numpy(b.reshape(-1))), pb.const(torch.from_numpy(t.reshape(-1))),
                               k, 0, bits, label="pre.resize_h")
                cur, cur_w = tmp, enc.img.w
            if h != S:
                b, t, k, bits = aten_aa_bilinear_tables(h, S)
                tmp = pb.buf((S, cur_w, 3), torch.uint8)
                pb.resample_u8(cur, tmp, S, cur_w, 3, cur_w * 3, cur_w * 3, pb.const(torch.from_numpy(b.reshape(-1))), pb.const(torch.from_numpy(t.reshape(-1))),
                               k, 1, bits, label="pre.resize_v")
                cur, cur_h = tmp, S
            pb.preprocess(cur, enc.img, S, enc.img.w, IMAGENET_MEAN, IMAGENET_STD, label="pre.normalise")
            plan = pb.build()
            plan.page = page
            self._pre[(h, w)] = plan
        return self._pre[(h, w)]

    def _post_plan(self, n, h, w):
        key = (n, h, w)
        if key not in self._post:
            dec = self._decoder(n)
            pb = PlanBuilder(self.lib, self.device, self.dtype)
            masks = pb.buf((n, h, w), torch.uint8)
            pb.resize_threshold(dec.logits, masks, n, dec.hl, dec.hl, h, w, 0.0, abi.F32, pix_stride=4, sel=dec.sel)
            plan = pb.build()
            plan.masks = masks
            self._post[key] = plan
        return self._post[key]

    def embed_boxes(self, boxes_xyxy: np.ndarray, h: int, w: int) -> np.ndarray:
        """Prompt tokens [N, 9, D] fp32: 6 output tokens + 2 box corners + 1 pad point
        (Sam2Processor box scaling processing_sam2.py:175-203; Sam2PromptEncoder._embed_boxes hf:819-829)."""
        S = self.hp["image_size"]
        b = np.asarray(boxes_xyxy, dtype=np.float32).reshape(-1, 2, 2).copy()
        b[..., 0] = b[..., 0] * np.float32(S / w)
        b[..., 1] = b[..., 1] * np.float32(S / h)
        c = (b + np.float32(0.5)) / np.float32(S)
        proj = (2 * c - 1).astype(np.float32) @ self.prompt_G * np.float32(2 * np.pi)
        pe = np.concatenate([np.sin(proj), np.cos(proj)], axis=-1).astype(np.float32)     # [N,2,D]
        pe[:, 0] += self.point_embed[2]
        pe[:, 1] += self.point_embed[3]
        n = pe.shape[0]
        pad = np.broadcast_to(self.not_a_point, (n, 1, pe.shape[-1]))
        out = np.broadcast_to(self.out_tokens[None], (n,) + self.out_tokens.shape)
        return np.concatenate([out, pe, pad], axis=1).astype(np.float32)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def submit_image(self, page_u8):
        """First half of `segment`, prompt-independent: upload, antialiased resize and the Hiera image encoder are queued on this model's
        own stream (hip/plan.py `AsyncLane`) and the call returns at once.  The reference computes the embedding inside
        `Sam2Model(pixel_values, input_boxes)` after the detectors have finished (core/image/detection.py:494-509); the encoder does not
        depend on the boxes, so a caller that knows the page will be segmented submits it together with the page's detectors and the
        1.6 TFLOP encoder runs beside them.  The ticket goes to `segment(..., ticket=)`; the model is busy until then."""
        page = torch.as_tensor(page_u8)
        h, w = int(page.shape[0]), int(page.shape[1])
        self._lane.acquire()
        try:
            self._lane.adopt(page)
            with self._lane.enter():
                pre = self._pre_plan(h, w)
                enc = self._encoder()
                pre.page.copy_(page.to(self.device))
                pre.run()
                enc.run(graph=self._graph)
        except BaseException:
            self._lane.release()
            raise
        return LaneTicket(self._lane, hw=(h, w), page=page)

    def segment(self, page_u8, boxes_xyxy, return_logits: bool = False, ticket=None):
        """page uint8 [H,W,3] (numpy or tensor) + boxes [N,4] in page pixels -> uint8 masks [N,H,W] (0/1)
        on the device.  One encoder pass per page, all boxes decoded together (as the reference does).  `ticket`: the page was
        already encoded by `submit_image` (same page object); the ticket is closed here (a dropped ticket frees the model by itself)."""
        boxes = np.asarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
        n = boxes.shape[0]
        page = torch.as_tensor(page_u8)
        h, w = int(page.shape[0]), int(page.shape[1])
        close = (lambda: ticket.close()) if isinstance(ticket, LaneTicket) else self._lane.release
        if ticket is not None and ticket["hw"] != (h, w):
            close()
            raise ModelError("SAM ticket belongs to a page of another size")
        if n == 0:
            if ticket is not None:
                close()
            return torch.zeros((0, h, w), dtype=torch.uint8, device=self.device)
        if ticket is None:
            self._lane.acquire()
        try:
            if ticket is None:
                self._lane.adopt(page)
            with self._lane.enter() if ticket is None else self._lane.resume():
                pre = self._pre_plan(h, w)
                enc = self._encoder()
                dec = self._decoder(n)
                post = self._post_plan(n, h, w)
                if ticket is None:
                    pre.page.copy_(page.to(self.device))
                dec.tok0.copy_(torch.from_numpy(self.embed_boxes(boxes, h, w).reshape(n * 9, -1)).to(self.device, self.ddt))
                if ticket is None:
                    pre.run()
                    enc.run(graph=self._graph)
                dec.run(graph=self._graph)
                post.run()
                masks = post.masks.clone()
                out = masks
                if return_logits:
                    sel = dec.sel.long()
                    lg = dec.logits.view(n, dec.hl, dec.hl, 4)
                    low = lg[torch.arange(n, device=self.device), :, :, sel].clone()
                    out = (masks, low, dec.iou.clone(), dec.sel.clone())
            self._lane.hand_over(*result_tensors(out))
            return out
        finally:
            close()

    @torch.no_grad()
    def probe_logits(self, size: int = 256) -> torch.Tensor:
        """low-resolution logits [2, hl, hl] (fp32, host) of two boxes on one synthetic page (white paper, black strokes, a noise band).
        `ModelManager.load_sam2` compares the f16 model's with the bf16 model's: f16 storage has 3 more mantissa bits (mask mismatch
        against the fp32 reference 1.7e-4 instead of 1.55e-3 at Hiera-L, profiles/r04_sam_dtype_probe.json) but a 65504 ceiling at which
        this library's conversions SATURATE (csrc/mtx_device.h from_f32<_Float16>) — an overflow therefore shows as a gross disagreement
        with the bf16 model, not as NaN; it comes from a few weight-driven channels and shows on any input."""
        rng = np.random.default_rng(0)
        page = np.full((size, size, 3), 255, np.uint8)
        page[size // 4: size // 4 + 3, :] = 0
        page[:, size // 3: size // 3 + 2] = 0
        page[size // 2: size // 2 + size // 8] = rng.integers(0, 256, (size // 8, size, 3), dtype=np.uint8)
        boxes = np.array([[size * 0.1, size * 0.1, size * 0.6, size * 0.7], [0, 0, size - 1, size - 1]], np.float32)
        _, low, _, _ = self.segment(page, boxes, return_logits=True)
        return low.float().cpu()

    def plans(self, n, h, w):
        return self._pre_plan(h, w), self._encoder(), self._decoder(n), self._post_plan(n, h, w)
