"""Static-graph builder: Python describes a network once, libmtx_hip executes it natively.

A `PlanBuilder` allocates activation buffers as torch tensors (PyTorch is only the allocator /
stream owner here) and records ops as `mtx_op` structs; `build()` hands the array to
`mtx_plan_create`.  `Plan.run()` is then one C call that launches every kernel of the network on
the caller's HIP stream (optionally as a hipGraph replay) — no Python in the per-layer loop.
"""
import contextlib
import threading
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import torch

from ..utils.exceptions import ModelError
from . import abi

_TORCH_DT = {abi.BF16: torch.bfloat16, abi.F16: torch.float16, abi.F32: torch.float32}


def _ptr(t, offset_elems: int = 0) -> Optional[int]:
    if t is None:
        return None
    if isinstance(t, Act):
        return t.ptr
    return t.data_ptr() + offset_elems * t.element_size()


@dataclass
class Act:
    """Channels-last activation view: channels [c0, c0+c) of a [N, H, W, LD] buffer."""
    t: torch.Tensor
    n: int
    h: int
    w: int
    c: int
    c0: int = 0

    @property
    def ld(self) -> int:
        return self.t.shape[-1]

    @property
    def ptr(self) -> int:
        return self.t.data_ptr() + self.c0 * self.t.element_size()

    def slice(self, c0: int, c: int) -> "Act":
        assert c0 % 8 == 0 and c % 8 == 0 and c0 + c <= self.c
        return Act(self.t, self.n, self.h, self.w, c, self.c0 + c0)

    def torch(self) -> torch.Tensor:
        return self.t[..., self.c0:self.c0 + self.c]


class Plan:
    def __init__(self, lib, handle, keep, n_ops):
        self.lib = lib
        self._h = C.c_void_p(handle)
        self._keep = keep
        self.n_ops = n_ops

    def _dev(self):
        """the plan's own device (the HIP device is per thread and defaults to 0 — a worker thread of rank k must not launch this
        plan on device 0's stream); None = the thread's current device (plans built before `device` existed, CPU simulator)"""
        d = getattr(self, "device", None)
        return d if d is not None and getattr(d, "type", None) == "cuda" else None

    def _stream(self, stream):
        if stream is not None:
            return C.c_void_p(stream)
        if torch.cuda.is_available() and not self.lib.is_simulator:
            return C.c_void_p(torch.cuda.current_stream(self._dev()).cuda_stream)
        return C.c_void_p(0)

    def run(self, stream=None, graph: bool = False) -> None:
        if not graph or self.lib.is_simulator:
            self.lib.check(self.lib.mtx_plan_run(self._h, self._stream(stream)), "mtx_plan_run")
            return
        # hipGraph capture/replay needs a non-default stream: use a side stream owned by the
        # plan, ordered after the caller's current stream and joined back into it.
        if stream is not None:
            self.lib.check(self.lib.mtx_plan_run_graph(self._h, C.c_void_p(stream)), "mtx_plan_run_graph")
            return
        cur = torch.cuda.current_stream(self._dev())
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=cur.device)
        self._side.wait_stream(cur)
        self.lib.check(self.lib.mtx_plan_run_graph(self._h, C.c_void_p(self._side.cuda_stream)), "mtx_plan_run_graph")
        cur.wait_stream(self._side)

    def run_range(self, first: int, last: int, stream=None) -> None:
        self.lib.check(self.lib.mtx_plan_run_range(self._h, first, last, self._stream(stream)), "mtx_plan_run_range")

    def time(self, iters: int, graph: bool = False, stream=None) -> float:
        ms = C.c_float(0.0)
        if graph and stream is None and not self.lib.is_simulator:
            torch.cuda.synchronize()
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=self._dev())
            stream = self._side.cuda_stream
        self.lib.check(self.lib.mtx_plan_time(self._h, self._stream(stream), iters, int(graph), C.byref(ms)), "mtx_plan_time")
        return float(ms.value)

    def time_range(self, first: int, last: int, iters: int, stream=None) -> float:
        ms = C.c_float(0.0)
        self.lib.check(self.lib.mtx_plan_time_range(self._h, first, last, self._stream(stream), iters, C.byref(ms)), "mtx_plan_time_range")
        return float(ms.value)

    def time_ops(self, op_indices, iters: int = 1, stream=None) -> float:
        """in-context duration (ms) of the listed ops summed over `iters` measurements: graph replay of the whole plan minus graph
        replay without them, or (MTX_TIME_OPS=stamp) device wall-clock stamps around them inside one replay (include/mtx_hip.h)"""
        idx = (C.c_int * len(op_indices))(*[int(i) for i in op_indices])
        ms = C.c_float(0.0)
        self.lib.check(self.lib.mtx_plan_time_ops(self._h, self._stream(stream), idx, len(op_indices), iters, C.byref(ms)), "mtx_plan_time_ops")
        return float(ms.value)

    def close(self):
        """Destroys the native plan (its hipGraph exec, lane side stream and events) and releases the activation buffers.  The last
        replay may still be in flight on the plan's side stream when a cache evicts the plan, so the streams the plan ran on are
        drained first — destroying a graph exec / stream under running work relies on the runtime deferring it."""
        if self._h:
            if not self.lib.is_simulator and torch.cuda.is_available():
                side = getattr(self, "_side", None)
                if side is not None:
                    side.synchronize()
                torch.cuda.current_stream().synchronize()
            self.lib.mtx_plan_destroy(self._h)
            self._h = None
            self._keep = None          # the activation buffers go back to the allocator

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlanCache:
    """shape-keyed plans with a bound: every plan pins its own activation buffers, and bubble crops / intermediate upscale passes come in
    arbitrary sizes, so an unbounded dict grows device memory with every new size.  Least recently used plans are destroyed."""

    def __init__(self, capacity: int = 8):
        from collections import OrderedDict
        self.capacity, self._d = max(1, int(capacity)), OrderedDict()

    def __contains__(self, key):
        return key in self._d

    def __len__(self):
        return len(self._d)

    def __getitem__(self, key):
        self._d.move_to_end(key)
        return self._d[key]

    def __setitem__(self, key, value):
        self._d[key] = value
        self._d.move_to_end(key)
        while len(self._d) > self.capacity:
            _, old = self._d.popitem(last=False)
            for p in (old if isinstance(old, (tuple, list)) else (old,)):
                if hasattr(p, "close"):
                    p.close()

    def get(self, key, default=None):
        return self[key] if key in self._d else default

    def items(self):
        return self._d.items()

    def values(self):
        return self._d.values()

    def clear(self):
        while self._d:
            _, old = self._d.popitem(last=False)
            for p in (old if isinstance(old, (tuple, list)) else (old,)):
                if hasattr(p, "close"):
                    p.close()


class PlanBuilder:
    def __init__(self, lib, device, dtype: int = abi.BF16, lanes: bool = True):
        self.lib = lib
        self._lanes = lanes                     # False: `with pb.side()` records on the main lane (one in-order stream)
        self.device = torch.device(device)
        self.dtype = dtype
        self.tdtype = _TORCH_DT[dtype]
        self.ops: List[abi.Op] = []
        self._side, self._join_next = False, False
        self.labels: List[str] = []
        self.keep: list = []
        self.conv1x1_as_gemm = True             # 1x1 convolutions of >= 1024 pixels go to the GEMM kernels (round 4: detector graphs -15 %); False = the conv kernel, for A/Bs

    # ---- memory ---------------------------------------------------------------------------
    def hold(self, t):
        self.keep.append(t)
        return t

    def act(self, n, h, w, c, ld=None, zero=False) -> Act:
        ld = c if ld is None else ld
        assert ld % 8 == 0, "pixel stride must be a multiple of 8 elements"
        t = (torch.zeros if zero else torch.empty)((n, h, w, ld), dtype=self.tdtype, device=self.device)
        self.keep.append(t)
        return Act(t, n, h, w, c)

    def buf(self, shape, dtype=torch.float32, zero=False):
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    def const(self, t: torch.Tensor, dtype=None):
        t = t.detach().to(device=self.device, dtype=dtype if dtype is not None else t.dtype).contiguous()
        self.keep.append(t)
        return t

    # ---- op recording ---------------------------------------------------------------------
    # ---- lanes (include/mtx_hip.h, MTX_LANE_*): ops recorded inside `with pb.side():` go to the plan's side stream and run beside the
    # main ops recorded after the block; `pb.join()` makes the next main op wait for them
    def side(self):
        pb = self

        class _Side:
            def __enter__(self_):
                pb._side = pb._lanes

            def __exit__(self_, *exc):
                pb._side = False
        return _Side()

    def join(self):
        self._join_next = True

    def _add(self, kind: int, args, label: str) -> int:
        op = abi.Op()
        op.kind = kind
        op.lane = (abi.LANE_SIDE if self._side else 0) | (abi.LANE_JOIN if (self._join_next and not self._side) else 0)
        if self._join_next and not self._side:
            self._join_next = False
        setattr(op.u, abi.UNION_FIELD[kind], args)
        self.ops.append(op)
        self.labels.append(label)
        return len(self.ops) - 1

    def conv2d(self, x: Act, w_packed, bias, cout: int, ksize: int = 3, stride: int = 1,
               act: int = abi.ACT_NONE, act_param: float = 0.0, res: Optional[Act] = None,
               res_scale: float = 1.0, out: Optional[Act] = None, pixel_shuffle: int = 0,
               chan_sum=None, res_broadcast: bool = False, pad_mode: int = 0, act_after_res: bool = False,
               label: str = "conv", valid_hw=None, out_scale=None) -> Act:
        pad_total = 1 if pad_mode == 1 else 2 * (ksize // 2)
        ho = (x.h + pad_total - ksize) // stride + 1
        wo = (x.w + pad_total - ksize) // stride + 1
        if out is None:
            if pixel_shuffle:
                out = self.act(x.n, ho * pixel_shuffle, wo * pixel_shuffle, cout // (pixel_shuffle ** 2))
            else:
                out = self.act(x.n, ho, wo, cout)
        if (self.conv1x1_as_gemm and ksize == 1 and stride == 1 and not pixel_shuffle and chan_sum is None and not res_broadcast and valid_hw is None
                and out_scale is None and not act_after_res and res_scale == 1.0 and pad_mode == 0 and act_param == 0.0 and x.c % 8 == 0 and cout % 8 == 0
                and x.n * x.h * x.w >= 1024):
            # a 1x1 convolution over NHWC pixels IS a GEMM of the pixel rows ([n h w, ld] with K = Cin) against the packed filter [Cout, Cin]: the
            # MFMA GEMM kernels (fused bias / act / residual, output into a channel slice through ldc) serve it 2-3x faster than the halo-tile
            # conv kernel at detector shapes — YOLO11m-seg @1600 7.16 -> 6.10 ms, YOLO12x @640 10.6 -> 8.8 ms, config 2 34.0 -> 35.7 pages/s on
            # one box (profiles/r04_visit_n_conv1x1_as_gemm_ab.log); parity tests unchanged
            self.gemm(x, w_packed, x.n * x.h * x.w, cout, x.c, lda=x.ld, ldw=int(w_packed.shape[-1]), out=out, ldc=out.ld, bias=bias, act=act,
                      res=res, ldres=(res.ld if res is not None else None), label=label)
            return out
        a = abi.ConvArgs()
        a.x, a.w, a.bias = x.ptr, _ptr(w_packed), _ptr(bias)
        a.res = res.ptr if res is not None else None
        a.y = out.ptr
        a.chan_sum = _ptr(chan_sum)
        a.n, a.h, a.w_in, a.cin, a.cout = x.n, x.h, x.w, x.c, cout
        a.ksize, a.stride = ksize, stride
        a.ldx, a.ldy, a.ldres = x.ld, out.ld, (res.ld if res is not None else 0)
        a.act, a.act_param, a.res_scale = act, act_param, res_scale
        a.pixel_shuffle, a.dtype = pixel_shuffle, self.dtype
        a.res_broadcast_n = 1 if res_broadcast else 0
        a.pad_mode = pad_mode
        a.act_after_res = 1 if act_after_res else 0
        a.valid_hw = _ptr(valid_hw)
        a.out_scale = _ptr(out_scale)        # device [n][cout] f32: y = out_scale * act(conv + bias) + res_scale * res
        self._add(abi.OP_CONV2D, a, label)
        return out

    def conv_tiles(self, x: Act, ksize=3, stride=1, cout=None, with_res=False, with_scale=False, act=abi.ACT_NONE, pixel_shuffle=0) -> int:
        """rows per image of the chan_sum buffer the conv with these arguments fills (which kernel runs — hence how many partial rows it
        writes — depends on the channel counts, on whether a residual or output factors come with the sums, on the activation and the store form:
        pass what the conv2d call will pass)"""
        a = abi.ConvArgs()
        a.n, a.h, a.w_in, a.cin, a.cout, a.ksize, a.stride = x.n, x.h, x.w, x.c, (cout if cout is not None else x.c), ksize, stride
        a.ldx, a.ldy, a.dtype = x.ld, (cout if cout is not None else x.c), self.dtype
        a.chan_sum = x.ptr                       # non-null markers: only their presence matters here
        a.res = x.ptr if with_res else None
        a.ldres = a.ldy
        a.out_scale = x.ptr if with_scale else None
        a.act, a.pixel_shuffle = act, pixel_shuffle
        t = self.lib.mtx_conv2d_tiles(C.byref(a))
        if t < 0:
            raise ModelError(f"mtx_conv2d_tiles: {self.lib.last_error()}")
        return t

    def gemm(self, a_t, w_t, m, n, k, lda=None, ldw=None, out=None, ldc=None, bias=None, act=abi.ACT_NONE,
             res=None, ldres=None, gate=None, ldgate=None, gate_rows_per=1, alpha=1.0, batch=1,
             a_bs=0, w_bs=0, c_bs=0, res_bs=0, out_f32=False, a_off=0, w_off=0, c_off=0, res_off=0,
             label="gemm", f8=None, flags=0, glu=None, w_lo=None, res_f32=False):
        """w_lo: the low half of a weight pair W = w_t + w_lo (same layout as w_t; mtx_gemm_args.w_lo).  res_f32: `res` is an fp32 matrix
        (needs out_f32; the fp32 residual stream of SAM's precision "high").
        f8 = (a_scale, lds_a, w_scale, lds_w, a_scale_off, w_scale_off): a_t / w_t are e4m3 byte matrices from `quantize` (offsets in
        bytes), the epilogue operands and the output stay in the builder's 16-bit type (include/mtx_hip.h, in_dtype == MTX_F8).
        glu = (q, scale, ldq, lds, col0, row_off, q_col_off): the columns from col0 on are [32 a | 32 b] spans (`glu_interleave` order of
        w's rows) and land as the MX fp8 matrix silu(a) * b in rows [row_off, row_off + m) of q / scale from byte column q_col_off on
        (mtx_gemm_args.glu_*); with col0 == 0 no 16-bit output exists at all"""
        g = abi.GemmArgs()
        g.flags = flags
        if glu is not None:
            gq, gsc, gldq, glds, gcol0, grow, gqcol = glu
            assert f8 is not None and gqcol % 128 == 0 and gcol0 % 256 == 0
            g.glu_q, g.glu_scale = gq.data_ptr() + grow * gldq + gqcol, gsc.data_ptr() + 4 * (grow + (gqcol // 128) * glds)
            g.glu_ldq, g.glu_lds, g.glu_col0 = gldq, glds, gcol0
            if out is None and gcol0 == 0:
                out = self.buf((8,), self.tdtype)          # never written: every tile takes the gated epilogue
        if f8 is not None:
            a_sc, lds_a, w_sc, lds_w, a_sc_off, w_sc_off = f8
            g.a_scale, g.w_scale = a_sc.data_ptr() + 4 * a_sc_off, w_sc.data_ptr() + 4 * w_sc_off
            g.lds_a, g.lds_w, g.in_dtype = lds_a, lds_w, abi.F8
        if out is None:
            out = self.buf((batch, m, n) if batch > 1 else (m, n), torch.float32 if out_f32 else self.tdtype)
        g.a, g.w, g.c = _ptr(a_t, a_off), _ptr(w_t, w_off), _ptr(out, c_off)
        g.bias, g.res, g.gate = _ptr(bias), _ptr(res, res_off), _ptr(gate)
        g.m, g.n, g.k = m, n, k
        g.lda, g.ldw, g.ldc = (lda or k), (ldw or k), (ldc or n)
        g.ldres, g.ldgate = (ldres or n), (ldgate or n)
        g.batch, g.a_bstride, g.w_bstride, g.c_bstride = batch, a_bs, w_bs, c_bs
        g.res_bstride = res_bs
        g.gate_rows_per = gate_rows_per
        g.act, g.act_param, g.alpha = act, 0.0, alpha
        g.dtype, g.out_dtype = self.dtype, (abi.F32 if out_f32 else self.dtype)
        if w_lo is not None:
            g.w_lo = _ptr(w_lo, w_off)
        if res_f32:
            assert out_f32 and res is not None
            g.res_dtype = abi.F32
        if ((m >= 2048 and n >= 1024 and k >= 512) or (m >= 256 and k >= 8192) or (flags & abi.GEMM_FORCE_TILE256)) and batch == 1 and not out_f32:
            # large problems: one shared scratch per plan for the K-slice tail of the 256-tile kernel (ops of a plan run in order)
            # (side-lane ops run beside main-lane ops: they get a scratch of their own)
            ws_name = "_gemm_ws_side" if self._side else "_gemm_ws"
            if getattr(self, ws_name, None) is None:
                setattr(self, ws_name, self.buf((abi.GEMM_WORKSPACE_BYTES,), torch.uint8, zero=True))      # the tickets in its last 4 KiB start at zero
            g.workspace, g.workspace_bytes = getattr(self, ws_name).data_ptr(), abi.GEMM_WORKSPACE_BYTES
        self._add(abi.OP_GEMM, g, label)
        return out

    def attention(self, q, k, v, o, batch, heads, sq, sk, d, q_str, k_str, v_str, o_str, scale,
                  q_off=0, k_off=0, v_off=0, o_off=0, label="attn", q_prescaled=False, q8=None, qk_f8=None, pv_f8=None):
        """q8 = (q bytes [sq, ldq], scale plane [ldq / 128, lds], ldq, lds, byte column offset): the rows leave as the MX fp8 operand of the
        next linear instead of (o None) or beside 16-bit values — long-sequence kernel only (mtx_attn_args.q8).
        qk_f8 = (e4m3 bytes [rows, ld], q byte column, k byte column, ld, exponent): the scores are 2^exponent * q_f8 k_f8^T on the fp8
        matrix instruction (mtx_attn_args.q_f8 / k_f8; the rotary kernel's y8) — pre-scaled q, long-sequence kernel only"""
        a = abi.AttnArgs()
        a.flags = abi.ATTN_Q_PRESCALED if q_prescaled else 0
        a.q, a.k, a.v, a.o = _ptr(q, q_off), _ptr(k, k_off), _ptr(v, v_off), (_ptr(o, o_off) if o is not None else None)
        if q8 is not None:
            q8q, q8s, ldq, lds8, col = q8
            assert col % 128 == 0 and o is None
            a.q8, a.q8_scale, a.ldq8, a.lds_q8 = q8q.data_ptr() + col, q8s.data_ptr() + 4 * (col // 128) * lds8, ldq, lds8
        if qk_f8 is not None:
            t8, qc, kc, ld8, ex = qk_f8
            assert q_prescaled and t8.dtype == torch.uint8
            a.q_f8, a.k_f8, a.qf8_ss, a.kf8_ss, a.qk_f8_exp = t8.data_ptr() + qc, t8.data_ptr() + kc, ld8, ld8, ex
        if pv_f8 is not None:            # (e4m3 V^T [heads * 128, ld] as `v_f8t` writes it, ld): P V on the fp8 instruction too (mtx_attn_args.v_f8t)
            vt8, ld_v = pv_f8
            assert qk_f8 is not None and q8 is not None and vt8.dtype == torch.uint8
            a.v_f8t, a.vf8_ld = vt8.data_ptr(), ld_v
        a.batch, a.heads, a.sq, a.sk, a.d = batch, heads, sq, sk, d
        a.q_bs, a.q_ss, a.q_hs = q_str
        a.k_bs, a.k_ss, a.k_hs = k_str
        a.v_bs, a.v_ss, a.v_hs = v_str
        a.o_bs, a.o_ss, a.o_hs = o_str
        a.scale, a.dtype = scale, self.dtype
        if d == 128 and sq >= 1024 and sk >= 256:      # long sequences: scratch for the key-split tail (include/mtx_hip.h)
            ws = self.buf((abi.ATTN_WORKSPACE_BYTES,), torch.uint8)
            a.workspace, a.workspace_bytes = ws.data_ptr(), abi.ATTN_WORKSPACE_BYTES
        self._add(abi.OP_ATTN, a, label)
        return o

    def v_f8t(self, v, rows, heads, ldv, v_off=0, out=None, label="v_f8t"):
        """MTX_EW_V_F8T: the value rows [rows, heads * 128] (16-bit, row stride ldv) as e4m3 V^T [heads * 128, ld] with ld = rows padded to 64, keys in
        accumulator order inside every 64-key tile — the operand of `attention(pv_f8=...)`.  -> (tensor, ld)"""
        ld = (rows + 63) // 64 * 64
        if out is None:
            out = self.buf((heads * 128, ld), torch.uint8, zero=True)
        e = abi.EwArgs()
        e.a, e.y8, e.ldy8 = _ptr(v, v_off), out.data_ptr(), ld
        e.n, e.h, e.w, e.c = 1, 1, rows, heads * 128
        e.lda, e.kind, e.dtype = ldv, abi.EW_V_F8T, self.dtype
        self._add(abi.OP_EW, e, label)
        return out, ld

    def norm(self, x, y, rows, c, ldx=None, ldy=None, gamma=None, beta=None, eps=1e-6, kind=0,
             mod_scale=None, mod_shift=None, rows_per=0, ldmod=0, x_off=0, y_off=0, act=abi.ACT_NONE,
             label="norm", q8=None, q_row_off=0, lds_q=0, dtype=None, out_dtype=None):
        """out_dtype (dtype = abi.F32 only): the type y is written in — None = fp32, abi.BF16 / abi.F16 = rounded once to the next linear's operand type.
        q8 = (q bytes [R, c], scale plane [c / 128, lds_q]): also (or, with y None, only) the MX fp8 twin of the result, rows landing
        at q_row_off (include/mtx_hip.h mtx_norm_args.q)"""
        a = abi.NormArgs()
        a.x, a.y = _ptr(x, x_off), (_ptr(y, y_off) if y is not None else None)
        if q8 is not None:
            a.q, a.q_scale = q8[0].data_ptr() + q_row_off * c, q8[1].data_ptr() + 4 * q_row_off
            a.ldq, a.lds_q = c, lds_q
        a.gamma, a.beta = _ptr(gamma), _ptr(beta)
        a.mod_scale, a.mod_shift = _ptr(mod_scale), _ptr(mod_shift)
        a.rows, a.c, a.ldx, a.ldy = rows, c, (ldx or c), (ldy or c)
        a.rows_per, a.ldmod = rows_per, ldmod
        a.eps, a.kind, a.dtype, a.act = eps, kind, (self.dtype if dtype is None else dtype), act      # dtype = abi.F32: an fp32 op inside a 16-bit plan
        a.out_dtype = a.dtype if out_dtype is None else out_dtype
        self._add(abi.OP_NORM, a, label)
        return y

    def groupnorm(self, x: Act, gamma, beta, groups=32, eps=1e-6, act=abi.ACT_NONE, out=None, label="gn") -> Act:
        assert x.c == x.ld and x.c0 == 0, "groupnorm needs a dense NHWC tensor"
        if out is None:
            out = self.act(x.n, x.h, x.w, x.c)
        ws = self.buf((abi.groupnorm_ws_floats(x.n, x.h * x.w, x.c, groups),), torch.float32)
        a = abi.GroupNormArgs()
        a.x, a.y, a.gamma, a.beta, a.workspace = x.ptr, out.ptr, _ptr(gamma), _ptr(beta), _ptr(ws)
        a.n, a.hw, a.c, a.groups = x.n, x.h * x.w, x.c, groups
        a.eps, a.act, a.dtype = eps, act, self.dtype
        self._add(abi.OP_GROUPNORM, a, label)
        return out

    def ew(self, kind, a_: Act, b: Optional[Act] = None, s=None, out: Optional[Act] = None, lds=0,
           act=abi.ACT_NONE, act_param=0.0, i0=0, i1=0, label="ew", dtype=None) -> Act:
        if out is None:
            if kind == abi.EW_UPSAMPLE2X:
                out = self.act(a_.n, a_.h * 2, a_.w * 2, a_.c)
            elif kind == abi.EW_AVGPOOL2:
                out = self.act(a_.n, (a_.h + 1) // 2, (a_.w + 1) // 2, a_.c)
            elif kind == abi.EW_MAXPOOL:
                pd = i0 // 2 if i0 % 2 else 0
                out = self.act(a_.n, (a_.h + 2 * pd - i0) // i1 + 1, (a_.w + 2 * pd - i0) // i1 + 1, a_.c)
            else:
                out = self.act(a_.n, a_.h, a_.w, a_.c)
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = a_.ptr, (b.ptr if b is not None else None), _ptr(s), out.ptr
        e.n, e.h, e.w, e.c = a_.n, a_.h, a_.w, a_.c
        e.lda, e.ldb, e.ldy, e.lds = a_.ld, (b.ld if b is not None else 0), out.ld, lds
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = kind, act, act_param, i0, i1, (self.dtype if dtype is None else dtype)
        self._add(abi.OP_EW, e, label)
        return out

    def residual_dist(self, after, before, prev, rows, c, parts=None, after_off=0, before_off=0, ld=None, label="residual_dist"):
        """first-block cache probe (include/mtx_hip.h MTX_EW_RESIDUAL_DIST): r = after - before (rounded to the storage type) against `prev`;
        -> fp32 [RESDIST_PARTS, 2]: per part (sum |prev - r|, sum |prev|); `residual_distance(parts)` adds them in index order"""
        if parts is None:
            parts = self.buf((abi.RESDIST_PARTS, 2), torch.float32, zero=True)
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = _ptr(after, after_off), _ptr(before, before_off), _ptr(prev), parts.data_ptr()
        e.n, e.h, e.w, e.c = 1, 1, rows, c
        e.lda = e.ldb = (ld or c)
        e.lds, e.ldy = c, 2
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_RESIDUAL_DIST, 0, 0.0, 0, 0, self.dtype
        self._add(abi.OP_EW, e, label)
        return parts

    def dwconv(self, x: Act, w_taps, bias, ksize: int, act=abi.ACT_NONE, out: Optional[Act] = None, label="dwconv") -> Act:
        """depthwise k x k, stride 1 (include/mtx_hip.h MTX_EW_DWCONV): w_taps T [k*k, C], bias fp32 [C] or None"""
        if out is None:
            out = self.act(x.n, x.h, x.w, x.c)
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = x.ptr, _ptr(bias), _ptr(w_taps), out.ptr
        e.n, e.h, e.w, e.c = x.n, x.h, x.w, x.c
        e.lda, e.ldb, e.ldy, e.lds = x.ld, 0, out.ld, 0
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_DWCONV, act, 0.0, ksize, 0, self.dtype
        self._add(abi.OP_EW, e, label)
        return out

    def channel_attention(self, chan_sum, w1, b1, w2, b2, s_out, n, tiles, c, cr, inv_hw, label="ca", inv_hw_dev=None,
                          before_conv=None, valid_hw=None, split=True):
        """before_conv = (t: Act, conv_w_packed, conv_bias): chan_sum holds the sums of t and the factors are those of
        mean(conv3x3(t) + bias) — available before that conv runs (include/mtx_hip.h, mtx_ca_args.t)"""
        a = abi.CaArgs()
        a.inv_hw_dev = _ptr(inv_hw_dev)
        a.chan_sum, a.w1, a.b1, a.w2, a.b2, a.s = (_ptr(chan_sum), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(s_out))
        a.n, a.tiles, a.c, a.cr, a.inv_hw = n, tiles, c, cr, inv_hw
        if before_conv is not None:
            t, cw, cb = before_conv
            a.t, a.conv_w, a.conv_b = t.ptr, _ptr(cw), _ptr(cb)
            a.h, a.w, a.ldt, a.dtype = t.h, t.w, t.ld, self.dtype
            a.valid_hw = _ptr(valid_hw)
            if split:
                # MTX_CA_SPLIT workgroups per image: partial records + arrival counters (zero once; every launch leaves them zero).
                # One scratch per builder and image count: the ops of a plan run one after the other.
                key = f"_ca_scratch_{n}"
                if getattr(self, key, None) is None:
                    setattr(self, key, self.buf((abi.ca_scratch_bytes(n) // 4,), torch.float32, zero=True))
                a.scratch = getattr(self, key).data_ptr()
        self._add(abi.OP_CA, a, label)
        return s_out

    def image_convert(self, kind, src, dst, n, h, w, c_pad, unshuffle=1, mul=1.0, add=(0.0, 0.0, 0.0), label="img", valid_hw=None):
        a = abi.ImgArgs()
        a.valid_hw = _ptr(valid_hw)
        a.src, a.dst = _ptr(src), _ptr(dst)
        a.n, a.h, a.w, a.c_pad, a.unshuffle, a.mul = n, h, w, c_pad, unshuffle, mul
        for i in range(3):
            a.add[i] = float(add[i])
        a.add[3] = 0.0
        a.kind, a.dtype = kind, self.dtype
        self._add(abi.OP_IMG, a, label)
        return dst

    def row_gather(self, src, dst, index_i32, rows, c, lda=None, ldy=None, label="row_gather", dtype=None):
        """dst[r, :c] = src[index[r], :c]  (token re-ordering between window layouts)"""
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = _ptr(src), None, _ptr(index_i32), _ptr(dst)
        e.n, e.h, e.w, e.c = 1, 1, rows, c
        e.lda, e.ldb, e.ldy, e.lds = (lda or c), 0, (ldy or c), 0
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_ROW_GATHER, 0, 0.0, 0, 0, (self.dtype if dtype is None else dtype)
        self._add(abi.OP_EW, e, label)
        return dst

    # ---- fp32 ops (csrc/f32ops.hip) --------------------------------------------------------------------------------------------------
    def cvt16(self, src_f32, dst16, rows, c, copies=1, lda=None, label="cvt16"):
        """dst16 [rows, >= copies * c] (this builder's 16-bit type) <- src_f32 [rows, c] rounded, written `copies` times side by side"""
        assert self.dtype in (abi.F16, abi.BF16) and src_f32.dtype == torch.float32 and dst16.dtype == self.tdtype
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = _ptr(src_f32), None, None, _ptr(dst16)
        e.n, e.h, e.w, e.c = 1, 1, rows, c
        e.lda, e.ldb, e.ldy, e.lds = (lda or c), 0, dst16.shape[-1], 0
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_CVT_16, 0, 0.0, self.dtype, copies, abi.F32
        self._add(abi.OP_EW, e, label)
        return dst16

    def cvt_f32(self, x: Act, src_dtype: int, label="cvt_f32") -> Act:
        """fp32 copy of a 16-bit activation (`src_dtype` = its abi dtype)"""
        out = self.act(x.n, x.h, x.w, x.c) if self.dtype == abi.F32 else Act(self.buf((x.n, x.h, x.w, x.c), torch.float32), x.n, x.h, x.w, x.c)
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = x.ptr, None, None, out.ptr
        e.n, e.h, e.w, e.c = x.n, x.h, x.w, x.c
        e.lda, e.ldb, e.ldy, e.lds = x.ld, 0, out.ld, 0
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_CVT_F32, 0, 0.0, src_dtype, 0, abi.F32
        self._add(abi.OP_EW, e, label)
        return out

    def shuffle2_add(self, cols, n, h, w, c, skip: Optional[Act] = None, skip_broadcast: bool = True, label="shuffle2_add") -> Act:
        """`cols` [n*h*w, 4c]: the four output pixels of every input pixel side by side (a ConvTranspose2d(k=2, s=2) computed as a GEMM) ->
        [n, 2h, 2w, c] plus an optional skip feature at output resolution (one image for every n when `skip_broadcast`)"""
        assert self.dtype == abi.F32
        out = self.act(n, 2 * h, 2 * w, c)
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = _ptr(cols), (skip.ptr if skip is not None else None), None, out.ptr
        e.n, e.h, e.w, e.c = n, h, w, c
        e.lda, e.ldb, e.ldy = 4 * c, (skip.ld if skip is not None else 0), out.ld
        e.lds = 0 if (skip is None or skip_broadcast) else 4 * h * w * skip.ld
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_SHUFFLE2_ADD, 0, 0.0, 0, 0, abi.F32
        self._add(abi.OP_EW, e, label)
        return out

    def im2col(self, x: Act, dst, k, stride, ldy, row_map=None, label="im2col"):
        e = abi.EwArgs()
        e.a, e.b, e.s, e.y = x.ptr, None, _ptr(row_map), _ptr(dst)
        e.n, e.h, e.w, e.c = x.n, x.h, x.w, x.c
        e.lda, e.ldb, e.ldy, e.lds = x.ld, 0, ldy, 0
        e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_IM2COL, 0, 0.0, k, stride, self.dtype
        self._add(abi.OP_EW, e, label)
        return dst

    def mask_select(self, logits, iou, counts, sel, n, pix, delta=0.05, thresh=0.98, label="mask_select"):
        a = abi.MaskSelectArgs()
        a.logits, a.iou, a.counts, a.sel = _ptr(logits), _ptr(iou), _ptr(counts), _ptr(sel)
        a.n, a.pix, a.delta, a.thresh = n, pix, delta, thresh
        self._add(abi.OP_MASK_SELECT, a, label)

    def preprocess(self, src_u8, dst: "Act", h, w, mean, std, label="preprocess"):
        a = abi.PreprocArgs()
        a.src, a.dst = _ptr(src_u8), dst.ptr
        a.h, a.w, a.oh, a.ow, a.c_pad = h, w, dst.h, dst.w, dst.ld
        for i in range(3):
            a.mean[i], a.std[i] = float(mean[i]), float(std[i])
        a.dtype = self.dtype
        self._add(abi.OP_PREPROC, a, label)
        return dst

    def resample_u8(self, src_u8, dst_u8, out_h, out_w, c, ld_src, ld_dst, bounds_i32, taps_i32, ksize, axis, coeff_bits=0, src_row0=0, label="resample"):
        """one axis of a fixed-point 8-bit resampling pass (mtx_tail_args, MTX_TAIL_RESAMPLE): Pillow's (coeff_bits 0 = 22) or ATen's"""
        a = abi.TailArgs()
        a.kind, a.src, a.dst = abi.TAIL_RESAMPLE, _ptr(src_u8), _ptr(dst_u8)
        a.out_h, a.out_w, a.c, a.ld_src, a.ld_dst = out_h, out_w, c, ld_src, ld_dst
        a.bounds, a.coeff, a.ksize, a.axis, a.src_row0, a.coeff_bits = _ptr(bounds_i32), _ptr(taps_i32), ksize, axis, src_row0, coeff_bits
        self._add(abi.OP_TAIL, a, label)
        return dst_u8

    def letterbox(self, src_u8, dst: "Act", h, w, new_h, new_w, pad_top, pad_left, pad_value=114.0, label="letterbox"):
        a = abi.PreprocArgs()
        a.src, a.dst = _ptr(src_u8), dst.ptr
        a.h, a.w, a.oh, a.ow, a.c_pad = h, w, dst.h, dst.w, dst.ld
        for i in range(3):
            a.mean[i], a.std[i] = 0.0, 1.0
        a.dtype, a.mode = self.dtype, 1
        a.new_h, a.new_w, a.pad_top, a.pad_left, a.pad_value = new_h, new_w, pad_top, pad_left, pad_value
        self._add(abi.OP_PREPROC, a, label)
        return dst

    def yolo_decode(self, levels, strides, nc, nm, reg_max, out, cls_off=0, mc_off=0, label="yolo_decode", box_f32=None, image=0):
        """box_f32: per level an fp32 [n * h * w, 4 * reg_max] matrix of DFL logits (mtx_yolo_decode_args.box_f32) read instead of the level's box channels;
        image: which image of a batched head [n, h, w, ld] this launch decodes (one launch per image)"""
        a = abi.YoloDecodeArgs()
        for i, (lv, st) in enumerate(zip(levels, strides)):
            a.level[i], a.lh[i], a.lw[i], a.lld[i], a.lstride[i] = lv.ptr + image * lv.h * lv.w * lv.ld * lv.t.element_size(), lv.h, lv.w, lv.ld, st
            if box_f32 is not None:
                a.box_f32[i] = _ptr(box_f32[i]) + image * lv.h * lv.w * 4 * reg_max * 4
        a.n_levels, a.nc, a.nm, a.reg_max, a.out, a.dtype = len(levels), nc, nm, reg_max, _ptr(out), self.dtype
        a.cls_off, a.mc_off = cls_off, mc_off
        self._add(abi.OP_YOLO_DECODE, a, label)
        return out

    def resize_threshold(self, src, dst, n, hs, ws, hd, wd, thresh=0.0, src_dtype=abi.F32, pix_stride=1,
                         sel=None, batch_stride=-1, roi=None, crop_xyxy=None, label="resize_thresh"):
        a = abi.ResizeThreshArgs()
        a.src, a.dst = _ptr(src), _ptr(dst)
        a.pix_stride, a.sel = pix_stride, _ptr(sel)
        a.batch_stride = batch_stride
        if roi is not None:
            a.roi_y, a.roi_x, a.roi_h, a.roi_w = roi
        a.crop_xyxy = _ptr(crop_xyxy)
        a.n, a.hs, a.ws, a.hd, a.wd, a.thresh, a.dtype = n, hs, ws, hd, wd, thresh, src_dtype
        self._add(abi.OP_RESIZE_THRESH, a, label)
        return dst

    def deform_attention(self, value, off, aw, ref_f32, out, rows, heads, d, level_shapes, points, offset_scale,
                         ld_value=None, ld_off=None, ld_aw=None, ld_out=None, label="deform_attn"):
        """RT-DETR multi-scale deformable attention (include/mtx_hip.h mtx_detr_args kind 0)"""
        a = abi.DetrArgs()
        a.value, a.off, a.aw, a.ref, a.out = _ptr(value), _ptr(off), _ptr(aw), _ptr(ref_f32), _ptr(out)
        a.kind, a.rows, a.heads, a.d, a.levels, a.points = 0, rows, heads, d, len(level_shapes), points
        start = 0
        for i, (h, w) in enumerate(level_shapes):
            a.lh[i], a.lw[i], a.lstart[i] = h, w, start
            start += h * w
        lp = len(level_shapes) * points
        a.ld_value, a.ld_out = ld_value or heads * d, ld_out or heads * d
        a.ld_off, a.ld_aw = ld_off or heads * lp * 2, ld_aw or heads * lp
        a.offset_scale, a.dtype = offset_scale, self.dtype
        self._add(abi.OP_DETR, a, label)
        return out

    def box_refine(self, ref_in_f32, ref_out_f32, ref_t, rows, delta=None, ld_delta=8, label="box_refine"):
        """ref_out = sigmoid(delta + logit(ref_in)) (delta given) or sigmoid(ref_in) (logits in); ref_t = T copy [rows, 8]"""
        a = abi.DetrArgs()
        a.ref, a.ref_out, a.ref_t, a.delta = _ptr(ref_in_f32), _ptr(ref_out_f32), _ptr(ref_t), _ptr(delta)
        a.kind, a.rows, a.ld_delta, a.dtype = (1 if delta is not None else 2), rows, ld_delta, self.dtype
        self._add(abi.OP_DETR, a, label)

    def quantize(self, x, rows, k, ldx=None, x_off=0, q=None, scale=None, row_off=0, lds=None, label="quantize_mx",
                 ldq=None, q_col_off=0, swiglu_b=None, b_off=0, ldb=None, y=None, y_off=0, ldy=None):
        """MX fp8 copy of rows [0, rows) of a 16-bit [*, ldx] matrix starting x_off elements in; with `q` / `scale` given the result
        lands in rows [row_off, row_off + rows) of those buffers (q [R, ldq] bytes from column q_col_off on, scale [K / 128, lds] uint32
        from plane q_col_off / 128 on).  swiglu_b: the quantised matrix is silu(x) * swiglu_b (MTX_QUANT_SWIGLU; y: its 16-bit copy).
        -> (q, scale, lds)"""
        if q is None:
            lds = (rows + 63) // 64 * 64
            q = self.buf((rows, k), torch.uint8)
            scale = self.buf((k // 128, lds), torch.int32, zero=True)
        ldq = ldq or k
        assert q_col_off % 128 == 0
        a = abi.QuantArgs()
        a.x, a.q, a.scale = _ptr(x, x_off), q.data_ptr() + row_off * ldq + q_col_off, scale.data_ptr() + 4 * (row_off + (q_col_off // 128) * lds)
        a.rows, a.k, a.ldx, a.ldq, a.lds, a.dtype = rows, k, (ldx or k), ldq, lds, self.dtype
        if swiglu_b is not None:
            a.op, a.b, a.ldb = abi.QUANT_SWIGLU, _ptr(swiglu_b, b_off), (ldb or ldx or k)
            if y is not None:
                a.y, a.ldy = _ptr(y, y_off), (ldy or k)
        self._add(abi.OP_QUANT, a, label)
        return q, scale, lds

    def memset(self, t, value=0, label="memset"):
        a = abi.MemsetArgs()
        a.ptr, a.bytes, a.value = _ptr(t), t.numel() * t.element_size(), value
        self._add(abi.OP_MEMSET, a, label)

    # ---- finish ---------------------------------------------------------------------------
    def build(self) -> Plan:
        n = len(self.ops)
        arr = (abi.Op * max(n, 1))(*self.ops)
        handle = C.c_void_p()
        self.lib.check(self.lib.mtx_plan_create(arr, n, C.byref(handle)), "mtx_plan_create")
        plan = Plan(self.lib, handle.value, list(self.keep), n)
        plan.device = self.device          # where its buffers live: `run` takes streams of THIS device, whatever the calling thread's current device is
        plan.labels = list(self.labels)
        plan.ops = list(self.ops)          # the recorded argument blocks (benchmarks group launches by kernel and shape)
        return plan


def residual_distance(parts: torch.Tensor) -> float:
    """mean |prev - r| / mean |prev| from the parts a `residual_dist` op left (added in index order, in double: the same bytes give the same
    verdict on every run); inf when `prev` is all zero"""
    p = parts.detach().to("cpu", torch.float64)
    d, m = float(p[:, 0].sum()), float(p[:, 1].sum())
    return d / m if m > 0.0 else float("inf")


def glu_interleave(col0: int, hid: int) -> torch.Tensor:
    """Row order of a fused [col0 + 2 * hid, K] projection for the gated epilogue of the fp8 GEMM (mtx_gemm_args.glu_*): rows below col0
    stay, then runs of 32 "a" rows (col0 + 32 u ..) each followed by the 32 "b" rows of the same outputs (col0 + hid + 32 u ..).
    `w[glu_interleave(col0, hid)]` is the weight to hand to the kernel."""
    assert hid % 32 == 0
    u = torch.arange(hid // 32)[:, None, None] * 32 + torch.arange(32)[None, None, :]                # [U, 1, 32]
    pair = torch.cat([col0 + u, col0 + hid + u], 1).reshape(-1)                                      # a-run, b-run per u
    return torch.cat([torch.arange(col0), pair])


class AsyncLane:
    """One model's private HIP stream for `submit` / `collect` style calls.

    The reference runs a page's detectors one after the other, each call returning finished results (`model(img, conf=...)`,
    core/image/detection.py:1337-1351, 1401-1407; core/image/ocr_detection.py:425-431).  On an MI355X the 640-pixel networks are far
    too small to fill 256 CUs, and every call ends in a synchronising read — four detectors cost the SUM of their GPU times plus
    their host post-processing.  With a lane per model, `submit` enqueues the upload and the graph replay on the model's own stream
    and returns at once; the caller submits every detector of the page, then collects: the graphs run side by side on the chip and one
    model's post-processing overlaps the others' kernels.  `__call__` = `collect(submit(...))` keeps the reference's call shape.

    The lane also owns the model's busy lock: a model's plan has ONE set of buffers, so a second submit waits until the first was
    collected.  On the CPU simulator (no streams) everything degenerates to an in-line call."""

    def __init__(self, device, simulator: bool):
        self.on = (not simulator) and torch.cuda.is_available() and torch.device(device).type == "cuda"
        self.device = torch.device(device)
        self._stream = None
        self.busy = threading.Lock()

    @property
    def stream(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        return self._stream

    def enter(self):
        """context for the enqueue half: the lane's stream, ordered after everything the caller has queued so far"""
        if not self.on:
            return contextlib.nullcontext()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def resume(self):
        """context for the collect half (reads the results on the lane's stream)"""
        return torch.cuda.stream(self.stream) if self.on else contextlib.nullcontext()

    def adopt(self, *tensors):
        """device tensors the CALLER allocated and the lane's kernels read (a page uploaded once for several detectors): the allocator
        must not hand their blocks out again while the lane still reads them, whatever the caller does with them after `submit`"""
        if self.on:
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.stream)

    def hand_over(self, *tensors):
        """the caller's stream may touch what the lane produced; `tensors` = results allocated on the lane's stream that leave with the
        caller (their blocks return to the lane's pool when the caller drops them: recorded on the caller's stream so that the lane does
        not reuse them under kernels of the caller that still read them — ADVICE r03)"""
        if self.on:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)

    ACQUIRE_TIMEOUT_S = 120.0      # far beyond any graph of this package; a wait this long is a ticket that was never closed

    def acquire(self):
        """waits for the model's previous ticket to be closed.  A ticket kept alive by a traceback / frame / reference cycle, or a caller that
        re-submits to the same model inside the `except` of a failed collect, never releases the lane by itself (the finaliser only runs when
        the ticket is collected): instead of hanging the thread for good, say so"""
        if not self.busy.acquire(timeout=self.ACQUIRE_TIMEOUT_S):
            raise RuntimeError(f"model lane busy for {self.ACQUIRE_TIMEOUT_S:.0f} s: a ticket of an earlier submit() was neither collected nor closed "
                               "(close tickets in try / finally, or use `with ticket:`)")

    def release(self):
        """idempotent: a ticket may be closed by `collect` and again by its finaliser"""
        try:
            self.busy.release()
        except RuntimeError:
            pass


def result_tensors(obj, _depth=0):
    """the CUDA tensors reachable from a result object (lists / tuples / dicts / namespaces / small result classes, three levels deep):
    what `AsyncLane.hand_over` records on the consumer's stream"""
    if torch.is_tensor(obj):
        return [obj] if obj.is_cuda else []
    if _depth >= 3 or obj is None or isinstance(obj, (str, bytes, int, float, bool)):
        return []
    if isinstance(obj, dict):
        items = obj.values()
    elif isinstance(obj, (list, tuple)):
        items = obj
    elif hasattr(obj, "__dict__"):
        items = vars(obj).values()
    else:
        return []
    out = []
    for v in items:
        out += result_tensors(v, _depth + 1)
    return out


class LaneTicket(dict):
    """What a model's `submit` returns: the arguments of its `collect` half, plus the duty to free the model's lane.  `collect` closes it;
    a ticket that is DROPPED (an exception between submit and collect in the caller) releases the lane when it is garbage-collected, so a
    failed page cannot leave a model busy for good.  The lane's stream is in order, so a later submit that reuses the
    model's buffers still runs behind whatever the dropped ticket had queued."""

    def __init__(self, lane: "AsyncLane", **fields):
        super().__init__(**fields)
        self._lane, self._open = lane, True

    def close(self):
        if self._open:
            self._open = False
            self._lane.release()

    def __enter__(self):
        return self

    def __exit__(self, *exc):          # `with model.submit(...) as ticket:` — the lane is free again however the block ends
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass
