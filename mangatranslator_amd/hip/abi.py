"""ctypes mirror of include/mtx_hip.h (field order and types must match 1:1; checked against
mtx_abi_sizeof() when the library is opened)."""
import ctypes as C

ABI_VERSION = 8

# enums
BF16, F16, F32, U8, I32, F8 = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GELU, ACT_GELU_TANH, ACT_SIGMOID, ACT_LEAKY = range(7)
(EW_SCALE_RES, EW_ADD, EW_MUL, EW_ACT, EW_UPSAMPLE2X, EW_MAXPOOL, EW_COPY, EW_GATE_RES,
 EW_ROW_GATHER, EW_IM2COL, EW_SOFTMAX_ROWS, EW_TRANSPOSE, EW_QK_NORM_ROPE, EW_AVGPOOL2, EW_SWIGLU, EW_DWCONV) = range(16)
EW_SHUFFLE2_ADD, EW_CVT_F32, EW_CVT_16 = 16, 17, 18            # fp32 ops only (csrc/f32ops.hip)
EW_SUB, EW_RESIDUAL_DIST = 19, 20                              # y = a - b; the first-block cache probe (include/mtx_hip.h)
EW_V_F8T = 21                                                  # v -> e4m3 [head][128][keys], keys in accumulator order per 64-key tile (mtx_attn_args.v_f8t)
RESDIST_PARTS = 256
IMG_NCHW_F32_TO_NHWC, IMG_NHWC_TO_NCHW_F32, IMG_NHWC_TO_HWC_U8, IMG_HWC_U8_TO_NHWC = range(4)
(OP_CONV2D, OP_GEMM, OP_ATTN, OP_NORM, OP_GROUPNORM, OP_EW, OP_CA, OP_IMG, OP_RESIZE_THRESH,
 OP_MEMSET, OP_MASK_SELECT, OP_PREPROC, OP_YOLO_DECODE, OP_DETR, OP_QUANT, OP_TAIL) = range(1, 17)

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

GN_PIX_PER_BLOCK = 1024


def groupnorm_ws_floats(n: int, hw: int, c: int, groups: int) -> int:
    """MTX_GROUPNORM_WS_FLOATS of include/mtx_hip.h"""
    return n * groups * 2 + n * ((hw + GN_PIX_PER_BLOCK - 1) // GN_PIX_PER_BLOCK) * c * 2


class ConvArgs(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("bias", vp), ("res", vp), ("y", vp), ("chan_sum", vp),
                ("n", i32), ("h", i32), ("w_in", i32), ("cin", i32), ("cout", i32),
                ("ksize", i32), ("stride", i32),
                ("ldx", i32), ("ldy", i32), ("ldres", i32),
                ("act", i32), ("act_param", f32), ("res_scale", f32),
                ("pixel_shuffle", i32), ("dtype", i32), ("res_broadcast_n", i32), ("pad_mode", i32), ("act_after_res", i32),
                ("valid_hw", vp), ("out_scale", vp)]


class GemmArgs(C.Structure):
    _fields_ = [("a", vp), ("w", vp), ("bias", vp), ("res", vp), ("gate", vp), ("c", vp),
                ("m", i64), ("n", i64), ("k", i64),
                ("lda", i64), ("ldw", i64), ("ldc", i64), ("ldres", i64), ("ldgate", i64),
                ("batch", i64), ("a_bstride", i64), ("w_bstride", i64), ("c_bstride", i64),
                ("res_bstride", i64),
                ("gate_rows_per", i32),
                ("act", i32), ("act_param", f32), ("alpha", f32),
                ("dtype", i32), ("out_dtype", i32), ("workspace", vp), ("workspace_bytes", i64),
                ("a_scale", vp), ("w_scale", vp), ("lds_a", i64), ("lds_w", i64), ("in_dtype", i32), ("flags", i32),
                ("glu_q", vp), ("glu_scale", vp), ("glu_ldq", i64), ("glu_lds", i64), ("glu_col0", i64),
                ("w_lo", vp), ("res_dtype", i32)]


GEMM_FORCE_TILE256, GEMM_NO_SPLIT = 1, 2
GEMM_WORKSPACE_BYTES = 2 * 320 * 256 * 256 * 4


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("o", vp),
                ("batch", i64), ("heads", i64), ("sq", i64), ("sk", i64), ("d", i64),
                ("q_bs", i64), ("q_ss", i64), ("q_hs", i64), ("k_bs", i64), ("k_ss", i64), ("k_hs", i64),
                ("v_bs", i64), ("v_ss", i64), ("v_hs", i64), ("o_bs", i64), ("o_ss", i64), ("o_hs", i64),
                ("scale", f32), ("dtype", i32), ("workspace", vp), ("workspace_bytes", i64), ("flags", i32),
                ("q8", vp), ("q8_scale", vp), ("ldq8", i64), ("lds_q8", i64),
                ("q_f8", vp), ("k_f8", vp), ("qf8_ss", i64), ("kf8_ss", i64), ("qk_f8_exp", i32),
                ("v_f8t", vp), ("vf8_ld", i64)]


ATTN_Q_PRESCALED = 1
ATTN_WORKSPACE_BYTES = 256 * (256 * 128 * 4 + 256 * 2 * 4)


class NormArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("gamma", vp), ("beta", vp), ("mod_scale", vp), ("mod_shift", vp),
                ("rows", i64), ("c", i64), ("ldx", i64), ("ldy", i64), ("rows_per", i64), ("ldmod", i64),
                ("eps", f32), ("kind", i32), ("dtype", i32), ("act", i32),
                ("q", vp), ("q_scale", vp), ("ldq", i64), ("lds_q", i64), ("out_dtype", i32)]


class GroupNormArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("gamma", vp), ("beta", vp), ("workspace", vp),
                ("n", i64), ("hw", i64), ("c", i64), ("groups", i64),
                ("eps", f32), ("act", i32), ("dtype", i32)]


class EwArgs(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("s", vp), ("y", vp),
                ("n", i64), ("h", i64), ("w", i64), ("c", i64),
                ("lda", i64), ("ldb", i64), ("ldy", i64), ("lds", i64),
                ("kind", i32), ("act", i32), ("act_param", f32), ("i0", i32), ("i1", i32), ("dtype", i32),
                ("y8", vp), ("ldy8", i64), ("y8_mul", f32)]


class CaArgs(C.Structure):
    _fields_ = [("chan_sum", vp), ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("s", vp),
                ("n", i32), ("tiles", i32), ("c", i32), ("cr", i32), ("inv_hw", f32), ("inv_hw_dev", vp),
                ("t", vp), ("conv_w", vp), ("conv_b", vp), ("h", i32), ("w", i32), ("ldt", i32), ("dtype", i32), ("valid_hw", vp),
                ("scratch", vp)]


CA_SPLIT, CA_RECORD = 32, 320


def ca_scratch_bytes(n: int) -> int:
    return n * (CA_SPLIT * CA_RECORD * 4 + 4)


class ImgArgs(C.Structure):
    _fields_ = [("src", vp), ("dst", vp),
                ("n", i64), ("h", i64), ("w", i64),
                ("c_pad", i32), ("unshuffle", i32),
                ("mul", f32), ("add", f32 * 4),
                ("kind", i32), ("dtype", i32), ("valid_hw", vp)]


class ResizeThreshArgs(C.Structure):
    _fields_ = [("src", vp), ("dst", vp),
                ("n", i64), ("hs", i64), ("ws", i64), ("hd", i64), ("wd", i64),
                ("thresh", f32), ("dtype", i32), ("pix_stride", i32), ("sel", vp),
                ("batch_stride", i64), ("roi_y", i32), ("roi_x", i32), ("roi_h", i32), ("roi_w", i32), ("crop_xyxy", vp)]


class MaskSelectArgs(C.Structure):
    _fields_ = [("logits", vp), ("iou", vp), ("counts", vp), ("sel", vp),
                ("n", i64), ("pix", i64), ("delta", f32), ("thresh", f32)]


class PreprocArgs(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("h", i64), ("w", i64), ("oh", i64), ("ow", i64),
                ("c_pad", i32), ("mean", f32 * 3), ("std", f32 * 3), ("dtype", i32),
                ("mode", i32), ("new_h", i32), ("new_w", i32), ("pad_top", i32), ("pad_left", i32), ("pad_value", f32)]


class YoloDecodeArgs(C.Structure):
    _fields_ = [("level", vp * 4), ("lh", i32 * 4), ("lw", i32 * 4), ("lld", i32 * 4), ("lstride", i32 * 4),
                ("n_levels", i32), ("nc", i32), ("nm", i32), ("reg_max", i32), ("out", vp), ("dtype", i32),
                ("cls_off", i32), ("mc_off", i32), ("box_f32", vp * 4)]


class MemsetArgs(C.Structure):
    _fields_ = [("ptr", vp), ("bytes", i64), ("value", i32)]


class DetrArgs(C.Structure):
    _fields_ = [("value", vp), ("off", vp), ("aw", vp), ("ref", vp), ("out", vp), ("delta", vp), ("ref_out", vp), ("ref_t", vp),
                ("kind", i32), ("rows", i32), ("heads", i32), ("d", i32), ("levels", i32), ("points", i32),
                ("lh", i32 * 4), ("lw", i32 * 4), ("lstart", i32 * 4),
                ("ld_value", i32), ("ld_off", i32), ("ld_aw", i32), ("ld_out", i32), ("ld_delta", i32),
                ("offset_scale", f32), ("dtype", i32)]


class QuantArgs(C.Structure):
    _fields_ = [("x", vp), ("q", vp), ("scale", vp),
                ("rows", i64), ("k", i64), ("ldx", i64), ("ldq", i64), ("lds", i64), ("dtype", i32),
                ("op", i32), ("b", vp), ("ldb", i64), ("y", vp), ("ldy", i64)]


QUANT_PLAIN, QUANT_SWIGLU = 0, 1


class TailArgs(C.Structure):
    _fields_ = [("kind", i32), ("src", vp), ("dst", vp), ("out_h", i32), ("out_w", i32), ("c", i32), ("ld_src", i64), ("ld_dst", i64),
                ("bounds", vp), ("coeff", vp), ("ksize", i32), ("axis", i32), ("src_row0", i32), ("coeff_bits", i32),
                ("alpha", vp), ("ld_alpha", i64), ("x", i32), ("y", i32), ("page_c", i32), ("src_c", i32),
                ("gamma_tab", vp), ("cbrt_tab", vp), ("lab_coef", vp), ("cbrt_n", i32),
                ("mask", vp), ("ld_mask", i64), ("other", vp), ("ld_other", i64), ("sums", vp), ("params", vp)]


TAIL_RESAMPLE, TAIL_COMPOSITE, TAIL_LAB_STATS, TAIL_LAB_REMAP, TAIL_EDT_COLS, TAIL_EDT_ROWS = 0, 1, 2, 3, 4, 5


class CleanArgs(C.Structure):
    _fields_ = [("page_bgr", vp), ("masks", vp), ("rois", vp), ("offsets", vp),
                ("base", vp), ("roi", vp), ("eroded", vp), ("thresholded", vp), ("shrunk", vp),
                ("dist_a", vp), ("dist_b", vp), ("stats", vp), ("zones", vp),
                ("n", i32), ("page_h", i32), ("page_w", i32), ("max_zones", i32),
                ("dil_r", i32), ("ero_r", i32), ("dil_dx", C.c_int8 * 64), ("ero_dx", C.c_int8 * 64),
                ("threshold", i32), ("use_otsu", i32), ("shrink_fixed", i32), ("junction_fixed", i32),
                ("sweeps", i32), ("max_pixels", i32)]


CLEAN_ARGS_KIND = 100      # mtx_abi_sizeof() key of the op-level-only struct


class _OpUnion(C.Union):
    _fields_ = [("conv", ConvArgs), ("gemm", GemmArgs), ("attn", AttnArgs), ("norm", NormArgs),
                ("gn", GroupNormArgs), ("ew", EwArgs), ("ca", CaArgs), ("img", ImgArgs),
                ("rt", ResizeThreshArgs), ("ms", MemsetArgs), ("sel", MaskSelectArgs), ("pre", PreprocArgs), ("yd", YoloDecodeArgs), ("detr", DetrArgs), ("quant", QuantArgs), ("tail", TailArgs)]


LANE_SIDE, LANE_JOIN = 1, 2


class Op(C.Structure):
    _fields_ = [("kind", i32), ("lane", i32), ("u", _OpUnion)]


ARG_TYPES = {OP_CONV2D: ConvArgs, OP_GEMM: GemmArgs, OP_ATTN: AttnArgs, OP_NORM: NormArgs,
             OP_GROUPNORM: GroupNormArgs, OP_EW: EwArgs, OP_CA: CaArgs, OP_IMG: ImgArgs,
             OP_RESIZE_THRESH: ResizeThreshArgs, OP_MEMSET: MemsetArgs,
             OP_MASK_SELECT: MaskSelectArgs, OP_PREPROC: PreprocArgs, OP_YOLO_DECODE: YoloDecodeArgs, OP_DETR: DetrArgs, OP_QUANT: QuantArgs, OP_TAIL: TailArgs}
UNION_FIELD = {OP_CONV2D: "conv", OP_GEMM: "gemm", OP_ATTN: "attn", OP_NORM: "norm",
               OP_GROUPNORM: "gn", OP_EW: "ew", OP_CA: "ca", OP_IMG: "img",
               OP_RESIZE_THRESH: "rt", OP_MEMSET: "ms", OP_MASK_SELECT: "sel", OP_PREPROC: "pre", OP_YOLO_DECODE: "yd", OP_DETR: "detr", OP_QUANT: "quant", OP_TAIL: "tail"}

# every symbol include/mtx_hip.h declares (tests check the built library exports all of them)
EXPORTS = [
    "mtx_abi_version", "mtx_abi_sizeof", "mtx_last_error", "mtx_init", "mtx_device_info",
    "mtx_conv2d", "mtx_conv2d_tiles", "mtx_gemm", "mtx_gemm_last_split", "mtx_attention", "mtx_norm", "mtx_groupnorm",
    "mtx_elementwise", "mtx_channel_attention", "mtx_image_convert", "mtx_resize_threshold",
    "mtx_mask_select", "mtx_preprocess", "mtx_yolo_decode", "mtx_detr", "mtx_quantize_mx", "mtx_page_tail", "mtx_bubble_clean", "mtx_host_text_mask", "mtx_host_chamfer_l2_5x5", "mtx_host_mask_outline", "mtx_host_png_encode",
    "mtx_plan_create", "mtx_plan_run", "mtx_plan_run_graph", "mtx_plan_num_ops",
    "mtx_plan_run_range", "mtx_plan_destroy", "mtx_plan_time", "mtx_plan_time_range", "mtx_plan_time_ops",
]
