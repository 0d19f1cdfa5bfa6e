"""ctypes binding of libplyolo_hip.so (the C ABI declared in include/plyolo.h).

There is NO fallback: if the shared library is missing or an entry point fails the
caller gets an exception (`PlyoloError`).  The library is built in-tree by
`__graft_entry__.build()` / `make -C pl_yolo_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLYOLO_LIB: another build of the same library (same-box A/Bs of two builds, tools/ab/ab_lib.sh); the default is the in-tree one
LIB_PATH = os.environ.get("PLYOLO_LIB") or os.path.join(_HERE, "libplyolo_hip.so")

BF16, F32 = 0, 1
ACT = {None: 0, "silu": 1, "relu": 2, "lrelu": 3, "hswish": 4, "gelu": 5}


class PlyoloError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    # mirrors plyolo_conv_desc (include/plyolo.h); x_coef / x_coef_ld / x_act describe a lazy input (NULL: x as stored)
    _fields_ = [(n, C.c_int) for n in ("dtype", "N", "H", "W", "Cin", "Cout", "ksize", "stride", "x_ld", "y_ld", "y_f32")] + \
               [("x_coef", C.c_void_p), ("x_coef_ld", C.c_int), ("x_act", C.c_int)]


STAT_SLOTS = 8  # PLYOLO_STAT_SLOTS


class BnStats(C.Structure):
    _fields_ = [("slots", C.c_void_p), ("count", C.c_double), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("eps", C.c_float), ("momentum", C.c_float), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p),
                ("split", C.c_int), ("gamma2", C.c_void_p), ("beta2", C.c_void_p), ("running_mean2", C.c_void_p),
                ("running_var2", C.c_void_p), ("num_batches_tracked2", C.c_void_p)]


class BnParams(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("eps", C.c_float)]


class Split(C.Structure):          # plyolo_split
    _fields_ = [("split", C.c_int), ("p2", C.c_void_p), ("ld2", C.c_int), ("fwd_to", C.c_void_p), ("fwd_ld", C.c_int), ("fwd_acc", C.c_int)]


class BnBwdSplit(C.Structure):     # plyolo_bn_bwd_split
    _fields_ = [("split", C.c_int), ("gamma2", C.c_void_p), ("dgamma2", C.c_void_p), ("dbeta2", C.c_void_p)]


class BnBwdFuse(C.Structure):      # plyolo_bn_bwd_fuse
    _fields_ = [("dout", C.c_void_p), ("dout_ld", C.c_int), ("dout2", C.c_void_p), ("dout2_ld", C.c_int), ("dout_split", C.c_int),
                ("z", C.c_void_p), ("z_ld", C.c_int), ("coef", C.c_void_p), ("bslots", C.c_void_p),
                ("gamma", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("par_split", C.c_int),
                ("gamma2", C.c_void_p), ("dgamma2", C.c_void_p), ("dbeta2", C.c_void_p), ("act", C.c_int),
                ("dz", C.c_void_p), ("dz_ld", C.c_int), ("fwd_to", C.c_void_p), ("fwd_ld", C.c_int)]


class BnRedSeg(C.Structure):       # plyolo_bn_red_seg
    _fields_ = [("c0", C.c_int), ("c1", C.c_int), ("z", C.c_void_p), ("z_ld", C.c_int), ("coef", C.c_void_p), ("coef_ld", C.c_int),
                ("bslots", C.c_void_p), ("slot_ld", C.c_int), ("act", C.c_int)]


BN_RED_SEGS = 3


class BnRed(C.Structure):          # plyolo_bn_red
    _fields_ = [("n", C.c_int), ("seg", BnRedSeg * BN_RED_SEGS)]


class BiasJob(C.Structure):        # plyolo_bias_job
    _fields_ = [("dy", C.c_void_p), ("M", C.c_int), ("C", C.c_int), ("ld", C.c_int), ("db", C.c_void_p), ("nblk", C.c_int)]


class PackEntry(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w", "wp", "wpd", "dwp", "dw", "b", "bp", "dbp", "db")] + [
        (n, C.c_int) for n in ("Cout", "Cin", "Cin_p", "ksize", "Cout_total", "Cout_p8", "co_off", "nslab", "blk0", "nblk")
    ]


class YoloxDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("A", C.c_int), ("C", C.c_int), ("M", C.c_int), ("nlevels", C.c_int),
        ("lvl_h", C.c_int * 8), ("lvl_w", C.c_int * 8), ("lvl_stride", C.c_int * 8),
        ("lvl_off", C.c_int * 8), ("lvl_row", C.c_int * 8), ("use_l1", C.c_int),
    ]


class YoloV7Desc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("M", C.c_int), ("C", C.c_int), ("na", C.c_int), ("nlevels", C.c_int),
        ("lvl_h", C.c_int * 3), ("lvl_w", C.c_int * 3), ("lvl_stride", C.c_int * 3), ("lvl_row", C.c_int * 3),
        ("anchors", C.c_float * 18), ("cand_cap", C.c_int),
    ]


def pack_elems(dtype, Cout_total, Cin_p, ksize):
    """(wp_elems, wpd_elems) of the packed weight buffers of one convolution."""
    a, b = _sz(0), _sz(0)
    call("plyolo_pack_elems", dtype, Cout_total, Cin_p, ksize, C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def yolov7_desc(B, M, num_classes, sizes, strides, anchors):
    """sizes [(h,w)]*3, anchors [3][3][2] pixels -> YoloV7Desc for the level-major raw layout."""
    d = YoloV7Desc()
    d.B, d.M, d.C, d.na, d.nlevels = B, M, num_classes, 3, 3
    r = 0
    for l, ((h, w), s) in enumerate(zip(sizes, strides)):
        d.lvl_h[l], d.lvl_w[l], d.lvl_stride[l], d.lvl_row[l] = h, w, int(s), r
        r += B * h * w
    flat = [float(v) for lv in anchors for a in lv for v in a]
    assert len(flat) == 18, "yolov7: 3 levels x 3 anchors expected"
    for i, v in enumerate(flat):
        d.anchors[i] = v
    d.cand_cap = 45 * M
    return d


class NmsDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("A", C.c_int), ("C", C.c_int), ("conf_thre", C.c_float), ("nms_thre", C.c_float),
        ("class_agnostic", C.c_int), ("max_nms", C.c_int), ("max_det", C.c_int), ("numel_threshold", C.c_int),
    ]


class AugImage(C.Structure):   # plyolo_aug_image
    _fields_ = [("src", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("dh", C.c_int), ("dw", C.c_int), ("flip", C.c_int), ("hsv", C.c_int),
                ("hgain", C.c_double), ("sgain", C.c_double), ("vgain", C.c_double)]


MAX_HOLES = 8


class Rect(C.Structure):         # plyolo_rect
    _fields_ = [("x1", C.c_int), ("y1", C.c_int), ("x2", C.c_int), ("y2", C.c_int)]


class MosaicTile(C.Structure):   # plyolo_mosaic_tile
    _fields_ = [("src", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("dh", C.c_int), ("dw", C.c_int), ("lx1", C.c_int), ("ly1", C.c_int),
                ("lx2", C.c_int), ("ly2", C.c_int), ("sx1", C.c_int), ("sy1", C.c_int)]


class ReduceJob(C.Structure):   # plyolo_reduce_job
    _fields_ = [("dwp", C.c_void_p), ("nslab", C.c_int), ("per", C.c_int), ("groups", C.c_int), ("elems", C.c_size_t)]


class FmtImage(C.Structure):   # plyolo_fmt_image
    _fields_ = [("det", C.c_void_p), ("n", C.c_int), ("ld", C.c_int), ("row0", C.c_int), ("scale", C.c_float)]


_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_P = C.POINTER

ABI_VERSION = 6   # == PLYOLO_ABI_VERSION of include/plyolo.h: the ctypes mirrors below follow THAT header's struct layouts

# name -> (restype, argtypes); every symbol include/plyolo.h declares
SIGNATURES = {
    "plyolo_version": (_i, []),
    "plyolo_arch": (C.c_char_p, []),
    "plyolo_build_flags": (_i, []),
    "plyolo_last_error": (C.c_char_p, []),
    "plyolo_plan_create": (_vp, []),
    "plyolo_plan_destroy": (None, [_vp]),
    "plyolo_plan_begin": (_i, [_vp]),
    "plyolo_plan_end": (_i, [_vp]),
    "plyolo_plan_size": (_i, [_vp]),
    "plyolo_plan_lanes": (_i, [_vp]),
    "plyolo_plan_lane": (_i, [_vp, _i]),
    "plyolo_plan_record": (_i, [_vp, _i]),
    "plyolo_plan_wait": (_i, [_vp, _i, _i]),
    "plyolo_plan_hook": (_i, [_vp, _i, _i]),
    "plyolo_plan_set_hook": (_i, [_vp, _vp, _vp]),
    "plyolo_plan_hooks": (_i, [_vp]),
    "plyolo_rccl_set_library": (_i, [C.c_char_p]),
    "plyolo_rccl_allreduce_bucket": (_i, [_vp, _vp, _sz, _i, _vp]),
    "plyolo_plan_run": (_i, [_vp, _vp]),
    "plyolo_plan_graph_instantiate": (_i, [_vp, _vp]),
    "plyolo_plan_graph_launch": (_i, [_vp, _vp]),
    "plyolo_plan_profile": (_i, [_vp, _vp, _vp, _i]),
    "plyolo_plan_lane_times": (_i, [_vp, _vp, _vp, _i]),
    "plyolo_plan_op_info": (_i, [_vp, _i, C.c_char_p, _i, _P(_d), _P(_d)]),
    "plyolo_plan_op_lane": (_i, [_vp, _i]),
    "plyolo_plan_stamp_times": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i]),
    "plyolo_conv2d_fwd": (_i, [_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "plyolo_conv2d_fwd_bn_act": (_i, [_P(ConvDesc), _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "plyolo_conv2d_dgrad": (_i, [_P(ConvDesc), _vp, _vp, _vp, _i, _vp]),
    "plyolo_conv2d_wgrad": (_i, [_P(ConvDesc), _vp, _vp, _vp, _vp]),
    "plyolo_conv2d_dgrad_bn_fits": (_i, [_P(ConvDesc), _i]),
    "plyolo_conv2d_dgrad_bn": (_i, [_P(ConvDesc), _P(BnBwdFuse), _vp, _vp, _i, _vp]),
    "plyolo_conv2d_dgrad_bn_red": (_i, [_P(ConvDesc), _P(BnBwdFuse), _vp, _vp, _i, _P(BnRed), _vp]),
    "plyolo_conv2d_dgrad_red": (_i, [_P(ConvDesc), _vp, _vp, _vp, _i, _P(BnRed), _vp]),
    "plyolo_conv2d_dgrad_red_fits": (_i, [_P(ConvDesc)]),
    "plyolo_conv2d_bwd_pw_red": (_i, [_P(ConvDesc), _P(BnBwdFuse), _vp, _vp, _vp, _i, _vp, _P(BnRed), _vp]),
    "plyolo_conv2d_bwd_pw_fits": (_i, [_P(ConvDesc), _i]),
    "plyolo_conv2d_wgrad_bn_fits": (_i, [_P(ConvDesc), _i]),
    "plyolo_conv2d_wgrad_bn": (_i, [_P(ConvDesc), _P(BnBwdFuse), _vp, _vp, _vp]),
    "plyolo_conv2d_bwd_pw_slabs": (_i, [_P(ConvDesc)]),
    "plyolo_conv2d_bwd_pw": (_i, [_P(ConvDesc), _P(BnBwdFuse), _vp, _vp, _vp, _i, _vp, _vp]),
    "plyolo_conv2d_wgrad_slabs": (_i, [_P(ConvDesc)]),
    "plyolo_bias_grad": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "plyolo_bias_grad_multi": (_i, [_i, _vp, _i, _vp]),
    "plyolo_pack_weights": (_i, [_vp, _i, _i, _i, _vp]),
    "plyolo_reduce_slabs": (_i, [_vp, _i, _sz, _vp]),
    "plyolo_lnw_act_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f, _i, _vp, _i, _vp, _vp]),
    "plyolo_lnw_act_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "plyolo_mosaic4": (_i, [_vp, _i, _i, _vp, _vp]),
    "plyolo_warp_affine_u8": (_i, [_vp, _i, _i, _P(_d), _vp, _i, _i, _i, _vp]),
    "plyolo_warp_perspective_u8": (_i, [_vp, _i, _i, _P(_d), _vp, _i, _i, _i, _vp]),
    "plyolo_resize_pad_u8": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "plyolo_mixup_blend_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "plyolo_rect_sums_u8": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "plyolo_cutout_holes_u8": (_i, [_vp, _i, _i, _vp, _i, _vp, C.c_double, _vp]),
    "plyolo_reduce_slabs_plan": (_i, [_i, _sz, _P(_i), _P(_i)]),
    "plyolo_reduce_slabs_multi": (_i, [_vp, _i, _i, _i, _d, _vp]),
    "plyolo_pack_elems": (_i, [_i, _i, _i, _i, _P(_sz), _P(_sz)]),
    "plyolo_unpack_wgrads": (_i, [_vp, _i, _i, _i, _vp]),
    "plyolo_pack_plan": (_i, [_vp, _i]),
    "plyolo_pack_weights_flat": (_i, [_vp, _i, _i, _i, _vp]),
    "plyolo_unpack_wgrads_flat": (_i, [_vp, _i, _i, _i, _vp]),
    "plyolo_bn_finalize": (_i, [_P(BnStats), _i, _vp, _vp]),
    "plyolo_bn_eval_coef": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "plyolo_dwconv3x3_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "plyolo_dwconv3x3_dgrad": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "plyolo_dwconv3x3_wgrad_blocks": (_i, [_i, _i, _i, _i, _i]),
    "plyolo_dwconv3x3_wgrad": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "plyolo_bicubic2x_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "plyolo_bicubic2x_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "plyolo_bias_coef": (_i, [_i, _vp, _vp, _vp]),
    "plyolo_fold_conv_bn": (_i, [_vp, _vp, _P(BnParams), _i, _i, _vp, _vp, _vp]),
    "plyolo_repconv_fuse": (_i, [_vp, _P(BnParams), _vp, _P(BnParams), _P(BnParams), _i, _i, _vp, _vp, _vp]),
    "plyolo_bn_eval_coef_at": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "plyolo_bn_act_fwd": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _P(BnStats), _P(Split), _vp]),
    "plyolo_channel_stats": (_i, [_i, _i, _i, _vp, _i, _vp, _vp]),
    "plyolo_act_bwd": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "plyolo_bn_act_bwd_reduce": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _P(Split), _vp]),
    "plyolo_bn_act_bwd_dz": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _P(Split), _P(BnBwdSplit), _vp]),
    "plyolo_focus_s2d": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "plyolo_copy_add": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "plyolo_upsample2x_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "plyolo_upsample2x_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "plyolo_maxpool_s1_fwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "plyolo_maxpool_s1_bwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "plyolo_spp_pools_bwd_fits": (_i, [_i, _i, _i]),
    "plyolo_spp_pools_fwd_fits": (_i, [_i, _i, _i, _i, _i, _vp]),
    "plyolo_spp_pools_fwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "plyolo_spp_pools_bwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "plyolo_maxpool2x2_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "plyolo_maxpool2x2_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "plyolo_implicit_bias": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "plyolo_scale_channels": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "plyolo_implicit_bwd_blocks": (_i, [_sz]),
    "plyolo_implicit_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _vp]),
    "plyolo_implicit_param_grads": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "plyolo_f32_to_act": (_i, [_i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "plyolo_memset_async": (_i, [_vp, _i, _sz, _vp]),
    "plyolo_nhwc_to_nchw_f32": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "plyolo_nchw_f32_to_nhwc": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "plyolo_yolox_workspace": (_sz, [_P(YoloxDesc)]),
    "plyolo_yolox_loss_fwd": (_i, [_P(YoloxDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "plyolo_yolox_loss_bwd": (_i, [_P(YoloxDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "plyolo_yolox_eval_decode": (_i, [_P(YoloxDesc), _vp, _vp, _vp]),
    "plyolo_yolov7_workspace": (_sz, [_P(YoloV7Desc)]),
    "plyolo_yolov7_loss_fwd": (_i, [_P(YoloV7Desc), _vp, _vp, _vp, _vp, _sz, _vp]),
    "plyolo_yolov7_loss_bwd": (_i, [_P(YoloV7Desc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "plyolo_yolov7_matched": (_i, [_P(YoloV7Desc), _vp, _vp, _vp, _vp]),
    "plyolo_yolov7_eval_decode": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "plyolo_postprocess_workspace": (_sz, [_P(NmsDesc)]),
    "plyolo_postprocess": (_i, [_P(NmsDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "plyolo_preproc_batch": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "plyolo_format_detections": (_i, [_vp, _i, _i, _vp, _vp]),
    "plyolo_batched_nms": (_i, [_P(NmsDesc), _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "plyolo_sgd_momentum": (_i, [_vp, _vp, _vp, _sz, _vp, _f, _f, _i, _vp]),
    "plyolo_ema_update": (_i, [_vp, _vp, _sz, _f, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PlyoloError(
                "libplyolo_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pl_yolo_amd/csrc`. There is no CPU fallback." % LIB_PATH
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if l.plyolo_version() != ABI_VERSION:
            raise PlyoloError("libplyolo_hip.so has ABI version %d, this package binds version %d (include/plyolo.h: PLYOLO_ABI_VERSION): "
                              "rebuild the library (`make -C pl_yolo_amd/csrc`)" % (l.plyolo_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().plyolo_last_error()
        raise PlyoloError("%s failed (rc=%d): %s" % (what or "plyolo call", rc, msg.decode() if msg else "?"))


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        check(rc, name)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
