"""RT-DETR-v2 (secondary / conjoined-bubble detector) on libmtx_hip — SURVEY.md §8 row f1.

The reference wraps HF `RTDetrV2ForObjectDetection` + `RTDetrImageProcessor` in an ultralytics-shaped adapter
(core/ml/rtdetr_adapter.py:61-113) and calls it like the YOLO models (core/image/detection.py:1401-1407):
    model(image, conf=0.35, device=..., imgsz=640, verbose=False)[0].boxes.{xyxy, conf, cls}, .names
`RTDetrHip` keeps that call shape.  Weights carry the HF parameter names.

Graph (two plans, both replayable as hipGraphs):
  A  image -> ResNet-vd backbone (BN folded; bottleneck residual + ReLU fused into the last 1x1 conv; the "vd" shortcut is
     a 2x2 average-pool kernel + 1x1 conv) -> 1x1 projections written straight into the FPN concat buffers -> AIFI
     transformer layer on the stride-32 map (NHWC rows ARE the token matrix: no transposes) -> CCFM top-down / bottom-up
     fusion (RepVGG blocks re-parameterised into single 3x3 convs, every concat a channel slice) -> decoder input
     projections written into one [sum HW, d] memory buffer -> encoder heads (scores, boxes + anchors).
  -- top-k query selection (300 of 8400) on the device --
  B  6 decoder layers: self-attention, multi-scale deformable attention (`mtx_detr` kernel), FFN, iterative box refinement.
Post-processing (sigmoid, top-k over queries x classes, box scaling) follows HF `post_process_object_detection`.
"""
import threading
from types import SimpleNamespace

import numpy as np
import torch
from PIL import Image

from ...hip import abi
from ...hip.lib import get_library
from ...hip.plan import Act, AsyncLane, LaneTicket, PlanBuilder, PlanCache, result_tensors
from ...utils.exceptions import ModelError


def _fold(w, bn, eps):
    g, b, mu, var = (bn[k].float() for k in ("weight", "bias", "running_mean", "running_var"))
    s = g / torch.sqrt(var + eps)
    return w.float() * s.view(-1, 1, 1, 1), b - mu * s


def sincos_2d(h, w, dim, temperature=10000.0):
    pos_dim = dim // 4
    omega = 1.0 / temperature ** (torch.arange(pos_dim, dtype=torch.float64) / pos_dim)
    gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    eh, ew = gy.flatten().outer(omega), gx.flatten().outer(omega)
    return torch.cat([eh.sin(), eh.cos(), ew.sin(), ew.cos()], 1).float()


def make_anchors(shapes, grid_size=0.05, eps=1e-2):
    """[sum HW, 4] logit-space anchors and their validity (HF RTDetrV2Model.generate_anchors)"""
    out = []
    for lvl, (h, w) in enumerate(shapes):
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        xy = torch.stack([gx, gy], -1) + 0.5
        xy[..., 0] /= w
        xy[..., 1] /= h
        wh = torch.ones_like(xy) * grid_size * (2.0 ** lvl)
        out.append(torch.cat([xy, wh], -1).reshape(h * w, 4))
    a = torch.cat(out, 0)
    valid = ((a > eps) & (a < 1 - eps)).all(-1, keepdim=True)
    return torch.log(a / (1 - a)), valid


class RTDetrHip:
    def __init__(self, state_dict: dict, config, device, lib=None, graph: bool = True, names=None):
        self.lib = lib if lib is not None else get_library()
        self.device = torch.device(device)
        self.dtype, self.tdt = abi.F16, torch.float16
        self.cfg = config
        id2label = getattr(config, "id2label", None) or {}
        self.names = {int(k): str(v) for k, v in (names or id2label).items()}
        self._graph = graph and not self.lib.is_simulator
        self._lane = AsyncLane(self.device, self.lib.is_simulator)
        self._plans = PlanCache(4)
        if config.decoder_method != "default" or config.num_feature_levels != len(config.decoder_in_channels) or config.normalize_before:
            raise ModelError("RT-DETR: unsupported configuration (decoder_method / extra feature levels / pre-norm)")
        if config.learn_initial_query:
            raise ModelError("RT-DETR: learn_initial_query is not supported")
        self._pack({k: v.detach().cpu() for k, v in state_dict.items()})

    # ---- weights -------------------------------------------------------------------------------------------
    def _pack(self, sd):
        self.W = {}
        dev, tdt = self.device, self.tdt

        def conv(name, w, b, act):
            co, ci, kh, kw = w.shape
            ci_p = (ci + 7) // 8 * 8
            wt = torch.zeros(co, kh * kw, ci_p)
            wt[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
            self.W[name] = (wt.to(dev, tdt).contiguous(), b.float().to(dev).contiguous(), co, kh, act)

        def bn(prefix):
            return {k: sd[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}

        def lin(name, src, pad_out=0, pad_in=0):
            w, b = sd[src + ".weight"].float(), sd[src + ".bias"].float()
            n, k = w.shape
            n_p, k_p = max(n, pad_out), max(k, pad_in)
            wp, bp = torch.zeros(n_p, k_p), torch.zeros(n_p)
            wp[:n, :k], bp[:n] = w, b
            self.W[name] = (wp.to(dev, tdt).contiguous(), bp.to(dev).contiguous(), n_p, k_p)

        def ln(name, src):
            self.W[name] = (sd[src + ".weight"].float().to(dev).contiguous(), sd[src + ".bias"].float().to(dev).contiguous())

        relu = abi.ACT_RELU
        act_enc = {"silu": abi.ACT_SILU, "relu": abi.ACT_RELU, "gelu": abi.ACT_GELU}[self.cfg.activation_function]
        bb = "model.backbone.model"
        for i in range(3):
            p = f"{bb}.embedder.embedder.{i}"
            conv(f"stem{i}", *_fold(sd[p + ".convolution.weight"], bn(p + ".normalization"), 1e-5), relu)
        self.stages = []
        bcfg = self.cfg.backbone_config
        for s, depth in enumerate(bcfg.depths):
            blocks = []
            for l in range(depth):
                p = f"{bb}.encoder.stages.{s}.layers.{l}"
                for j, a in ((0, relu), (1, relu), (2, abi.ACT_NONE)):
                    conv(f"s{s}b{l}c{j}", *_fold(sd[f"{p}.layer.{j}.convolution.weight"], bn(f"{p}.layer.{j}.normalization"), 1e-5), a)
                sc = None
                if f"{p}.shortcut.convolution.weight" in sd:
                    conv(f"s{s}b{l}sc", *_fold(sd[f"{p}.shortcut.convolution.weight"], bn(f"{p}.shortcut.normalization"), 1e-5), abi.ACT_NONE)
                    sc = "conv"
                elif f"{p}.shortcut.1.convolution.weight" in sd:
                    conv(f"s{s}b{l}sc", *_fold(sd[f"{p}.shortcut.1.convolution.weight"], bn(f"{p}.shortcut.1.normalization"), 1e-5), abi.ACT_NONE)
                    sc = "pool"
                stride = 2 if (l == 0 and (s > 0 or bcfg.downsample_in_first_stage)) else 1
                blocks.append((sc, stride))
            self.stages.append(blocks)
        if bcfg.layer_type != "bottleneck" or bcfg.downsample_in_bottleneck:
            raise ModelError("RT-DETR backbone: only the bottleneck ResNet-vd layout is supported")
        eps = self.cfg.batch_norm_eps
        for i in range(3):
            conv(f"enc_proj{i}", *_fold(sd[f"model.encoder_input_proj.{i}.0.weight"], bn(f"model.encoder_input_proj.{i}.1"), 1e-5), abi.ACT_NONE)
            conv(f"dec_proj{i}", *_fold(sd[f"model.decoder_input_proj.{i}.0.weight"], bn(f"model.decoder_input_proj.{i}.1"), eps), abi.ACT_NONE)
        e = "model.encoder"
        for i in range(2):
            conv(f"lateral{i}", *_fold(sd[f"{e}.lateral_convs.{i}.conv.weight"], bn(f"{e}.lateral_convs.{i}.norm"), eps), act_enc)
            conv(f"down{i}", *_fold(sd[f"{e}.downsample_convs.{i}.conv.weight"], bn(f"{e}.downsample_convs.{i}.norm"), eps), act_enc)
            for blk in ("fpn", "pan"):
                p = f"{e}.{blk}_blocks.{i}"
                conv(f"{blk}{i}.conv1", *_fold(sd[p + ".conv1.conv.weight"], bn(p + ".conv1.norm"), eps), act_enc)
                conv(f"{blk}{i}.conv2", *_fold(sd[p + ".conv2.conv.weight"], bn(p + ".conv2.norm"), eps), act_enc)
                if p + ".conv3.conv.weight" in sd:
                    raise ModelError("RT-DETR: hidden_expansion != 1 is not supported")
                for k in range(3):      # RepVGG: 3x3 + 1x1 branches -> one 3x3 conv
                    w3, b3 = _fold(sd[f"{p}.bottlenecks.{k}.conv1.conv.weight"], bn(f"{p}.bottlenecks.{k}.conv1.norm"), eps)
                    w1, b1 = _fold(sd[f"{p}.bottlenecks.{k}.conv2.conv.weight"], bn(f"{p}.bottlenecks.{k}.conv2.norm"), eps)
                    w3 = w3.clone()
                    w3[:, :, 1, 1] += w1[:, :, 0, 0]
                    conv(f"{blk}{i}.rep{k}", w3, b3 + b1, act_enc)
        a = f"{e}.aifi.0.layers.0"
        self.W["aifi.qk"] = (torch.cat([sd[a + ".self_attn.q_proj.weight"], sd[a + ".self_attn.k_proj.weight"]]).to(dev, tdt).contiguous(),
                             torch.cat([sd[a + ".self_attn.q_proj.bias"], sd[a + ".self_attn.k_proj.bias"]]).float().to(dev).contiguous())
        lin("aifi.v", a + ".self_attn.v_proj"); lin("aifi.o", a + ".self_attn.o_proj")
        lin("aifi.fc1", a + ".mlp.fc1"); lin("aifi.fc2", a + ".mlp.fc2")
        ln("aifi.ln1", a + ".self_attn_layer_norm"); ln("aifi.ln2", a + ".final_layer_norm")
        lin("enc_out", "model.enc_output.0"); ln("enc_out_ln", "model.enc_output.1")
        lin("enc_score", "model.enc_score_head", pad_out=(self.cfg.num_labels + 7) // 8 * 8)
        for j in range(3):
            lin(f"enc_bbox{j}", f"model.enc_bbox_head.layers.{j}", pad_out=8 if j == 2 else 0)
        d = "model.decoder"
        lin("qpos0", d + ".query_pos_head.layers.0", pad_in=8); lin("qpos1", d + ".query_pos_head.layers.1")
        for l in range(self.cfg.decoder_layers):
            p = f"{d}.layers.{l}"
            self.W[f"d{l}.qk"] = (torch.cat([sd[p + ".self_attn.q_proj.weight"], sd[p + ".self_attn.k_proj.weight"]]).to(dev, tdt).contiguous(),
                                  torch.cat([sd[p + ".self_attn.q_proj.bias"], sd[p + ".self_attn.k_proj.bias"]]).float().to(dev).contiguous())
            lin(f"d{l}.v", p + ".self_attn.v_proj"); lin(f"d{l}.o", p + ".self_attn.o_proj")
            ln(f"d{l}.ln1", p + ".self_attn_layer_norm"); ln(f"d{l}.ln2", p + ".encoder_attn_layer_norm"); ln(f"d{l}.ln3", p + ".final_layer_norm")
            # sampling offsets and attention weights share their input: one GEMM
            wo, bo = sd[p + ".encoder_attn.sampling_offsets.weight"], sd[p + ".encoder_attn.sampling_offsets.bias"]
            wa, ba = sd[p + ".encoder_attn.attention_weights.weight"], sd[p + ".encoder_attn.attention_weights.bias"]
            self.n_off, self.n_aw = wo.shape[0], wa.shape[0]
            self.W[f"d{l}.offaw"] = (torch.cat([wo, wa]).to(dev, tdt).contiguous(), torch.cat([bo, ba]).float().to(dev).contiguous())
            lin(f"d{l}.value", p + ".encoder_attn.value_proj"); lin(f"d{l}.out", p + ".encoder_attn.output_proj")
            lin(f"d{l}.fc1", p + ".mlp.fc1"); lin(f"d{l}.fc2", p + ".mlp.fc2")
            for j in range(3):
                lin(f"d{l}.bbox{j}", f"{d}.bbox_embed.{l}.layers.{j}", pad_out=8 if j == 2 else 0)
        lin("cls", f"{d}.class_embed.{self.cfg.decoder_layers - 1}", pad_out=(self.cfg.num_labels + 7) // 8 * 8)

    # ---- graph helpers -----------------------------------------------------------------------------------------
    def _conv(self, pb, x, name, stride=1, out=None, res=None, after=False):
        w, b, co, k, act = self.W[name]
        return pb.conv2d(x, w, b, co, ksize=k, stride=stride, act=act if not after else abi.ACT_RELU, out=out, res=res,
                         act_after_res=after, label=name)

    def _lin(self, pb, a, name, m, act=abi.ACT_NONE, res=None, out=None, k=None):
        w, b, n, kk = self.W[name]
        return pb.gemm(a, w, m, n, k or kk, bias=b, act=act, res=res, out=out, label=name)

    def _mha(self, pb, x, pos, tag, rows, heads, D, images=1):
        """post-norm transformer self-attention block: LN(x + O(attn((x+pos)Wq, (x+pos)Wk, xWv))).  images > 1 (the batched encoder plan): x and
        pos hold `images` sequences of rows / images positions back to back; every sequence attends to itself only (the launch's batch dimension)"""
        a2 = lambda t: Act(t.view(1, 1, rows, D), 1, 1, rows, D)
        qk_in = pb.ew(abi.EW_ADD, a2(x), b=a2(pos), label=tag + ".addpos")
        wqk, bqk = self.W[tag + ".qk"]
        qk = pb.gemm(qk_in.t.view(rows, D), wqk, rows, 2 * D, D, bias=bqk, label=tag + ".qk")
        v = self._lin(pb, x, tag + ".v", rows)
        o = pb.buf((rows, D), self.tdt)
        hd = D // heads
        per = rows // images
        pb.attention(qk, qk, v, o, images, heads, per, per, hd, (per * 2 * D, 2 * D, hd), (per * 2 * D, 2 * D, hd), (per * D, D, hd), (per * D, D, hd), hd ** -0.5,
                     k_off=D, label=tag + ".attn")
        y = self._lin(pb, o, tag + ".o", rows, res=x)
        g, b = self.W[tag + ".ln1"]
        return pb.norm(y, pb.buf((rows, D), self.tdt), rows, D, gamma=g, beta=b, eps=self.cfg.layer_norm_eps, label=tag + ".ln1")

    def _csp(self, pb, x, tag):
        c1 = self._conv(pb, x, tag + ".conv1")
        c2 = self._conv(pb, x, tag + ".conv2")
        h = self._conv(pb, c1, tag + ".rep0")
        h = self._conv(pb, h, tag + ".rep1")
        return self._conv(pb, h, tag + ".rep2", res=c2)          # act(conv) + conv2(x): the CSP sum

    def _build(self, H, W, batch=1):
        """backbone + hybrid encoder + the encoder's proposal head.  batch = B > 1 (core/ml/detector_batch.py RTDetrBatcher): B resized images through one
        graph — every activation carries the image index outermost, AIFI attends per image, the decoder memory / proposal scores / boxes come out as
        B blocks of S rows (image b: rows b * S .. ) — with the same arithmetic per image as the one-image plan"""
        cfg = self.cfg
        D, heads = cfg.d_model, cfg.encoder_attention_heads
        B = int(batch)
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        src = pb.buf((B, H, W, 3), torch.uint8)
        x = pb.act(B, H, W, 8)
        pb.image_convert(abi.IMG_HWC_U8_TO_NHWC, src, x.t, B, H, W, 8, mul=1.0, label="rescale")
        x = self._conv(pb, x, "stem0", stride=2)
        x = self._conv(pb, x, "stem1")
        x = self._conv(pb, x, "stem2")
        x = pb.ew(abi.EW_MAXPOOL, x, i0=3, i1=2, label="stem.pool")
        feats = []
        for s, blocks in enumerate(self.stages):
            for l, (sc, stride) in enumerate(blocks):
                t = self._conv(pb, x, f"s{s}b{l}c0")
                t = self._conv(pb, t, f"s{s}b{l}c1", stride=stride)
                if sc == "pool":
                    r = self._conv(pb, pb.ew(abi.EW_AVGPOOL2, x, label=f"s{s}b{l}.avgpool"), f"s{s}b{l}sc")
                elif sc == "conv":
                    r = self._conv(pb, x, f"s{s}b{l}sc")
                else:
                    r = x
                x = self._conv(pb, t, f"s{s}b{l}c2", res=r, after=True)
            if s >= 1:
                feats.append(x)
        shapes = [(f.h, f.w) for f in feats]
        # concat buffers of the top-down pass: [upsampled top | backbone projection]
        cat_fpn = [pb.act(B, h, w, 2 * D) for (h, w) in shapes[:2]]
        self._conv(pb, feats[0], "enc_proj0", out=cat_fpn[0].slice(D, D))
        self._conv(pb, feats[1], "enc_proj1", out=cat_fpn[1].slice(D, D))
        top = self._conv(pb, feats[2], "enc_proj2")
        # AIFI on the stride-32 map
        h5, w5 = shapes[2]
        rows = B * h5 * w5
        pos = pb.const(sincos_2d(h5, w5, D, cfg.positional_encoding_temperature).repeat(B, 1), self.tdt)
        t = top.t.view(rows, D)
        y = self._mha(pb, t, pos, "aifi", rows, heads, D, images=B)
        f1 = self._lin(pb, y, "aifi.fc1", rows, act={"gelu": abi.ACT_GELU, "relu": abi.ACT_RELU, "silu": abi.ACT_SILU}[cfg.encoder_activation_function])
        f2 = self._lin(pb, f1, "aifi.fc2", rows, res=y)
        g, b = self.W["aifi.ln2"]
        top_t = pb.norm(f2, pb.buf((rows, D), self.tdt), rows, D, gamma=g, beta=b, eps=cfg.layer_norm_eps, label="aifi.ln2")
        top = Act(top_t.view(B, h5, w5, D), B, h5, w5, D)
        # CCFM top-down
        cat_pan = [pb.act(B, h, w, 2 * D) for (h, w) in shapes[1:]]        # [downsampled | lateral output]
        lat0 = self._conv(pb, top, "lateral0", out=cat_pan[1].slice(D, D))
        pb.ew(abi.EW_UPSAMPLE2X, lat0, out=cat_fpn[1].slice(0, D), label="fpn0.up")
        p4 = self._csp(pb, cat_fpn[1], "fpn0")
        lat1 = self._conv(pb, p4, "lateral1", out=cat_pan[0].slice(D, D))
        pb.ew(abi.EW_UPSAMPLE2X, lat1, out=cat_fpn[0].slice(0, D), label="fpn1.up")
        p3 = self._csp(pb, cat_fpn[0], "fpn1")
        # bottom-up
        self._conv(pb, p3, "down0", stride=2, out=cat_pan[0].slice(0, D))
        n4 = self._csp(pb, cat_pan[0], "pan0")
        self._conv(pb, n4, "down1", stride=2, out=cat_pan[1].slice(0, D))
        n5 = self._csp(pb, cat_pan[1], "pan1")
        # decoder memory: the three projected maps back to back as token rows
        S = sum(h * w for h, w in shapes)
        R = B * S                                   # rows of the proposal head: image b's S tokens at rows b * S ..
        mem = pb.buf((R, D), self.tdt)
        start = 0
        for i, (f, (h, w)) in enumerate(zip((p3, n4, n5), shapes)):
            if B == 1:
                self._conv(pb, f, f"dec_proj{i}", out=Act(mem[start:start + h * w].view(1, h, w, D), 1, h, w, D))
            else:           # the convolution writes [B, h, w, D] densely; an image's level then moves to its place among that image's S rows
                lvl = self._conv(pb, f, f"dec_proj{i}")
                for b_ in range(B):
                    pb.ew(abi.EW_COPY, Act(lvl.t[b_:b_ + 1], 1, h, w, D), out=Act(mem[b_ * S + start:b_ * S + start + h * w].view(1, h, w, D), 1, h, w, D),
                          label=f"dec_proj{i}.place{b_}")
            start += h * w
        anchors, valid = make_anchors(shapes)
        mask = pb.const(valid.float().expand(S, D).repeat(B, 1).contiguous(), self.tdt)
        a2 = lambda t_, c: Act(t_.view(1, 1, R, c), 1, 1, R, c)
        masked = pb.ew(abi.EW_MUL, a2(mem, D), b=a2(mask, D), label="valid_mask")
        eo = self._lin(pb, masked.t.view(R, D), "enc_out", R)
        g, b = self.W["enc_out_ln"]
        om = pb.norm(eo, pb.buf((R, D), self.tdt), R, D, gamma=g, beta=b, eps=cfg.layer_norm_eps, label="enc_out_ln")
        w, bias, ncp, _ = self.W["enc_score"]
        scores = pb.gemm(om, w, R, ncp, D, bias=bias, out_f32=True, label="enc_score")
        bx = self._lin(pb, om, "enc_bbox0", R, act=abi.ACT_RELU)
        bx = self._lin(pb, bx, "enc_bbox1", R, act=abi.ACT_RELU)
        w, bias, _, _ = self.W["enc_bbox2"]
        boxes = pb.gemm(bx, w, R, 8, D, bias=bias, out_f32=True, label="enc_bbox2")
        plan = pb.build()
        plan.src, plan.mem, plan.om, plan.scores, plan.boxes = src, mem, om, scores, boxes
        plan.shapes, plan.S, plan.images = shapes, S, B
        a8 = torch.zeros(S, 8)
        a8[:, :4] = torch.where(valid, anchors, torch.full((), torch.finfo(torch.float32).max))
        plan.anchors = a8.to(self.device)
        return plan

    def _build_decoder(self, shapes, S):
        cfg = self.cfg
        D, heads, Q, L = cfg.d_model, cfg.decoder_attention_heads, cfg.num_queries, cfg.decoder_layers
        hd = D // heads
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        mem = pb.buf((S, D), self.tdt)
        h = pb.buf((Q, D), self.tdt)                         # query embeddings (gathered encoder rows)
        h_in = h
        ref_logit = pb.buf((Q, 8), torch.float32)            # gathered box logits (cols 4..7 unused)
        ref = [pb.buf((Q, 8), torch.float32) for _ in range(2)]
        ref_t = pb.buf((Q, 8), self.tdt)
        pb.box_refine(ref_logit, ref[0], ref_t, Q, label="ref.init")
        act_dec = {"relu": abi.ACT_RELU, "gelu": abi.ACT_GELU, "silu": abi.ACT_SILU}[cfg.decoder_activation_function]
        cur = 0
        for l in range(L):
            tag = f"d{l}"
            qp = self._lin(pb, ref_t, "qpos0", Q, act=abi.ACT_RELU, k=8)
            qpos = self._lin(pb, qp, "qpos1", Q)
            y = self._mha(pb, h, qpos, tag, Q, heads, D)
            # cross attention over the encoder memory
            a2 = lambda t: Act(t.view(1, 1, Q, D), 1, 1, Q, D)
            q_in = pb.ew(abi.EW_ADD, a2(y), b=a2(qpos), label=tag + ".xq")
            w, b = self.W[tag + ".offaw"]
            n_tot = self.n_off + self.n_aw
            offaw = pb.gemm(q_in.t.view(Q, D), w, Q, n_tot, D, bias=b, label=tag + ".offaw")
            value = self._lin(pb, mem, tag + ".value", S)
            samp = pb.buf((Q, D), self.tdt)
            pb.deform_attention(value, offaw, offaw[:, self.n_off:], ref[cur], samp, Q, heads, hd, shapes, cfg.decoder_n_points,
                                cfg.decoder_offset_scale, ld_off=n_tot, ld_aw=n_tot, label=tag + ".deform")
            z = self._lin(pb, samp, tag + ".out", Q, res=y)
            g, b = self.W[tag + ".ln2"]
            z = pb.norm(z, pb.buf((Q, D), self.tdt), Q, D, gamma=g, beta=b, eps=cfg.layer_norm_eps, label=tag + ".ln2")
            f1 = self._lin(pb, z, tag + ".fc1", Q, act=act_dec)
            f2 = self._lin(pb, f1, tag + ".fc2", Q, res=z)
            g, b = self.W[tag + ".ln3"]
            h = pb.norm(f2, pb.buf((Q, D), self.tdt), Q, D, gamma=g, beta=b, eps=cfg.layer_norm_eps, label=tag + ".ln3")
            bx = self._lin(pb, h, tag + ".bbox0", Q, act=abi.ACT_RELU)
            bx = self._lin(pb, bx, tag + ".bbox1", Q, act=abi.ACT_RELU)
            delta = self._lin(pb, bx, tag + ".bbox2", Q)
            pb.box_refine(ref[cur], ref[cur ^ 1], ref_t, Q, delta=delta, ld_delta=8, label=tag + ".refine")
            cur ^= 1
        w, bias, ncp, _ = self.W["cls"]
        logits = pb.gemm(h, w, Q, ncp, D, bias=bias, out_f32=True, label="class_embed")
        plan = pb.build()
        plan.mem, plan.h0, plan.ref_logit, plan.logits, plan.boxes = mem, h_in, ref_logit, logits, ref[cur]
        return plan

    def plans(self, H, W):
        key = (H, W)
        if key not in self._plans:
            a = self._build(H, W)
            self._plans[key] = (a, self._build_decoder(a.shapes, a.S))
        return self._plans[key]

    # ---- inference ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_raw(self, img_u8: np.ndarray):
        """resized RGB uint8 [H, W, 3] -> (logits [Q, C] fp32, boxes cxcywh [Q, 4] fp32 in 0..1)"""
        with self._lane.busy:
            with self._lane.enter():
                out = self._enqueue(img_u8)
            self._lane.hand_over(*result_tensors(out))
            return out

    def _enqueue(self, img_u8: np.ndarray):
        H, W = img_u8.shape[:2]
        if H % 32 or W % 32:
            raise ModelError("RT-DETR input must be a multiple of 32")
        cfg = self.cfg
        a, b = self.plans(H, W)
        a.src.copy_(torch.from_numpy(np.array(img_u8, dtype=np.uint8)).to(self.device).view(1, H, W, 3))
        a.run(graph=self._graph)
        nc, Q = cfg.num_labels, cfg.num_queries
        top = a.scores[:, :nc].max(-1).values.topk(Q, dim=0).indices
        b.mem.copy_(a.mem)
        b.h0.copy_(a.om.index_select(0, top))
        b.ref_logit.copy_((a.boxes + a.anchors).index_select(0, top))
        b.run(graph=self._graph)
        return b.logits[:, :nc].clone(), b.boxes[:, :4].clone()

    def __call__(self, source, conf: float = 0.35, device=None, verbose: bool = False, imgsz=None, **_kw):
        return self.collect(self.submit(source, conf=conf, imgsz=imgsz))

    @torch.no_grad()
    def submit(self, source, conf: float = 0.35, imgsz=None, **_kw):
        """first half of a call (see hip/plan.py `AsyncLane`): the host-side resize, then upload, backbone + encoder graph, query
        selection and decoder graph queued on this model's own stream; nothing here waits for the GPU"""
        if isinstance(source, Image.Image):
            pil = source.convert("RGB") if source.mode != "RGB" else source
        elif isinstance(source, np.ndarray):
            arr = source
            if arr.ndim == 2:
                arr = np.stack([arr] * 3, -1)
            pil = Image.fromarray(np.ascontiguousarray(arr[..., :3][..., ::-1]))      # cv2 BGR -> RGB, as the adapter does
        else:
            pil = Image.open(source).convert("RGB")
        ow, oh = pil.size
        size = int(imgsz) if imgsz is not None else 640
        img = np.asarray(pil.resize((size, size), resample=Image.Resampling.BILINEAR))        # RTDetrImageProcessor: resize + 1/255
        self._lane.acquire()
        try:
            with self._lane.enter():
                logits, boxes = self._enqueue(img)
                nc = self.cfg.num_labels
                scores = logits.sigmoid()
                k = min(self.cfg.num_queries, scores.numel())
                top_s, idx = scores.flatten().topk(k)
                labels, qi = idx % nc, idx // nc
                cx, cy, w, h = boxes[qi].unbind(-1)
                scale = torch.tensor([ow, oh, ow, oh], dtype=boxes.dtype).to(boxes.device, non_blocking=True)
                xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1) * scale
                keep = top_s > float(conf)
        except BaseException:
            self._lane.release()
            raise
        return LaneTicket(self._lane, xyxy=xyxy, top_s=top_s, labels=labels, keep=keep, hw=(oh, ow))

    @torch.no_grad()
    def collect(self, t):
        try:
            with self._lane.resume():
                keep = t["keep"]
                res = [SimpleNamespace(boxes=_Boxes(t["xyxy"][keep].float(), t["top_s"][keep].float(), t["labels"][keep].float()), names=self.names,
                                       orig_shape=t["hw"], masks=None)]
            self._lane.hand_over(*result_tensors(res))
            return res
        finally:
            t.close() if isinstance(t, LaneTicket) else self._lane.release()


class _Boxes:
    """the slice of ultralytics `Boxes` the detection operator reads (reference core/ml/rtdetr_adapter.py:18-30)"""

    def __init__(self, xyxy, conf, cls):
        self.xyxy, self.conf, self.cls = xyxy, conf, cls

    def __len__(self):
        return int(self.xyxy.shape[0])
