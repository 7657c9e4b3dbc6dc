"""ctypes binding of libhrp_hip.so (C ABI declared in include/hrp.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRP_LIB") or os.path.join(_HERE, "libhrp_hip.so")   # HRP_LIB: development builds (A/B variants)
CSRC = os.path.join(_HERE, "csrc")

HRP_F32, HRP_BF16, HRP_F32X3 = 0, 1, 2
MAX_TAPS = 16
EW_MAX_IN = 4
EW_IDENTITY, EW_AFFINE, EW_BN_TRAIN = 0, 1, 2
FK_MAX_JOINTS, FK_MAX_KP = 32, 24
PIL_KMAX = 12


class HrpError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("res", C.c_void_p),
                ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("stats", C.c_void_p),
                ("dtype", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("x_pitch", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
                ("y_H", C.c_int32), ("y_W", C.c_int32), ("y_pitch", C.c_int32), ("res_pitch", C.c_int32),
                ("out_stride", C.c_int32), ("out_off_y", C.c_int32), ("out_off_x", C.c_int32),
                ("in_stride", C.c_int32), ("ntaps", C.c_int32),
                ("dy", C.c_int32 * MAX_TAPS), ("dx", C.c_int32 * MAX_TAPS), ("wtap", C.c_int32 * MAX_TAPS),
                ("w_ntaps", C.c_int32), ("w_cout_pad", C.c_int32), ("relu", C.c_int32),
                ("bnb_x", C.c_void_p), ("bnb_mask", C.c_void_p), ("bnb_consts", C.c_void_p),
                ("bnb_x_pitch", C.c_int32), ("bnb_mask_pitch", C.c_int32),
                ("bnb_stats", C.c_void_p), ("bnb_gamma", C.c_void_p), ("bnb_beta", C.c_void_p),
                ("bnb_count", C.c_float), ("bnb_eps", C.c_float),
                ("pro_mode", C.c_int32), ("pro_reserved", C.c_int32),
                ("pro_x2", C.c_void_p), ("pro_stats", C.c_void_p), ("pro_bsums", C.c_void_p),
                ("pro_gamma", C.c_void_p), ("pro_beta", C.c_void_p),
                ("pro_count", C.c_float), ("pro_eps", C.c_float), ("pro_side", C.c_void_p),
                ("pro_mask", C.c_void_p), ("pro_side2", C.c_void_p), ("pro_side2_acc", C.c_int32), ("pro_reserved2", C.c_int32),
                ("res_mask", C.c_void_p),
                ("tail_mode", C.c_int32), ("tail_side_acc", C.c_int32), ("tail_stats", C.c_void_p), ("tail_bsums", C.c_void_p),
                ("tail_gamma", C.c_void_p), ("tail_beta", C.c_void_p), ("tail_count", C.c_float), ("tail_eps", C.c_float),
                ("tail_mask", C.c_void_p), ("tail_g", C.c_void_p), ("tail_side", C.c_void_p),
                ("tail_x2", C.c_void_p), ("tail_w2", C.c_void_p), ("tail_stats2", C.c_void_p), ("tail_gamma2", C.c_void_p),
                ("tail_beta2", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p),
                ("dtype", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("x_pitch", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("dy_pitch", C.c_int32),
                ("in_stride", C.c_int32), ("ntaps", C.c_int32),
                ("dy_t", C.c_int32 * MAX_TAPS), ("dx_t", C.c_int32 * MAX_TAPS),
                ("dw_cin", C.c_int32), ("dw_tap_stride", C.c_int32), ("dw_tap_off", C.c_int32), ("accumulate", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("phase", C.c_int32), ("reserved", C.c_int32)]


class WgradFoldDesc(C.Structure):
    _fields_ = [("workspace", C.c_void_p), ("dw", C.c_void_p),
                ("G", C.c_int32), ("pairs", C.c_int32), ("n_cib", C.c_int32), ("nte", C.c_int32), ("nb", C.c_int32),
                ("Cout", C.c_int32), ("dw_cin", C.c_int32), ("ntaps", C.c_int32), ("dw_tap_stride", C.c_int32),
                ("dw_tap_off", C.c_int32), ("accumulate", C.c_int32), ("reserved", C.c_int32)]


class Sim2RealLossDesc(C.Structure):
    _fields_ = [("rendered", C.c_void_p), ("seg", C.c_void_p), ("kp3d", C.c_void_p), ("kp3d_int", C.c_void_p),
                ("B", C.c_int32), ("HW", C.c_int32), ("K", C.c_int32), ("mask_loss", C.c_int32),
                ("w_mask", C.c_float), ("w_iou", C.c_float), ("w_scale", C.c_float), ("w_align", C.c_float),
                ("terms", C.c_void_p), ("d_rendered", C.c_void_p), ("d_kp3d", C.c_void_p), ("d_kp3d_int", C.c_void_p),
                ("workspace", C.c_void_p)]


class SilhouetteDesc(C.Structure):
    _fields_ = [("uv", C.c_void_p), ("xyz", C.c_void_p), ("faces", C.c_void_p),
                ("B", C.c_int32), ("V", C.c_int32), ("F", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("sigma", C.c_float), ("blur_radius", C.c_float), ("alpha", C.c_void_p), ("logp", C.c_void_p), ("count", C.c_void_p)]


BLOCK_MAX = 2


class BlockDesc(C.Structure):
    _fields_ = [("conv1", ConvDesc), ("conv2", ConvDesc)]


class BlockInfo(C.Structure):
    _fields_ = [("n", C.c_int32), ("grid", C.c_int32), ("lds_bytes", C.c_int32), ("reserved", C.c_int32),
                ("first_wg", C.c_int32 * BLOCK_MAX), ("bands", C.c_int32 * BLOCK_MAX)]


class PackEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("ntaps", C.c_int32), ("pad_t", C.c_int32)]


class EwInput(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch", C.c_int32), ("up", C.c_int32), ("mode", C.c_int32),
                ("a", C.c_void_p), ("b", C.c_void_p), ("stats", C.c_void_p),
                ("count", C.c_float), ("eps", C.c_float)]


class EwDesc(C.Structure):
    _fields_ = [("inp", EwInput * EW_MAX_IN), ("nin", C.c_int32), ("out", C.c_void_p),
                ("out_pitch", C.c_int32), ("dtype", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("relu", C.c_int32),
                ("mask", C.c_void_p), ("mask_pitch", C.c_int32), ("consts_out", C.c_void_p)]


class EwBwdDesc(C.Structure):
    _fields_ = [("dout", C.c_void_p), ("out", C.c_void_p), ("dout_pitch", C.c_int32), ("out_pitch", C.c_int32),
                ("inp", EwInput), ("din", C.c_void_p), ("din_pitch", C.c_int32), ("sums", C.c_void_p),
                ("dtype", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("relu", C.c_int32), ("accumulate", C.c_int32),
                ("mask", C.c_void_p), ("mask_pitch", C.c_int32),
                ("din2", C.c_void_p), ("din2_pitch", C.c_int32), ("accumulate2", C.c_int32), ("pooled", C.c_void_p)]


class BnEntry(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("d", C.c_void_p),
                ("out_scale", C.c_void_p), ("out_shift", C.c_void_p), ("counter", C.c_void_p),
                ("C", C.c_int32), ("count", C.c_float), ("momentum", C.c_float), ("eps", C.c_float),
                ("accumulate", C.c_int32)]


# batched launches (include/hrp.h hrp_batch_*)
BATCH_MAX = 32
BATCH_CONV, BATCH_WGRAD, BATCH_EW_FWD, BATCH_EW_BWD_REDUCE, BATCH_EW_BWD_APPLY, BATCH_WGRAD_FOLD = range(6)


class BatchInfo(C.Structure):
    _fields_ = [("family", C.c_int32), ("n", C.c_int32), ("dtype", C.c_int32), ("variant", C.c_int32),
                ("grid", C.c_int32), ("lds_bytes", C.c_int32), ("grid2", C.c_int32),
                ("grid3", C.c_int32), ("lds_bytes3", C.c_int32),
                ("blk0", C.c_int32 * (BATCH_MAX + 1)), ("blk2", C.c_int32 * (BATCH_MAX + 1)),
                ("ws_bytes", C.c_int64 * BATCH_MAX)]


class PoseLossDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pose", "rot", "trans", "root_uv", "depth", "xyz_int", "xyz_fk",
                                          "gt_pose", "gt_root_rot", "gt_root_trans", "gt_root_uv", "gt_kp3d", "gt_kp2d", "mask", "K",
                                          "d_pose", "d_rot", "d_trans", "d_root_uv", "d_depth", "d_xyz_int", "d_xyz_fk", "out")] + \
               [("weights", C.c_float * 10), ("B", C.c_int32), ("P", C.c_int32), ("J", C.c_int32), ("root", C.c_int32),
                ("image_size", C.c_float), ("rot_dim", C.c_int32)]


OPT_CHUNK = 4096


class OptTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64)]


class OptChunk(C.Structure):
    _fields_ = [("tensor", C.c_int32), ("offset", C.c_int32)]


class RegStepDesc(C.Structure):
    """hrp_regressor_step_desc (include/hrp.h): one step of one iterative regressor."""
    _fields_ = [("M", C.c_int32), ("P", C.c_int32), ("K", C.c_int32), ("N", C.c_int32),
                ("u_prev", C.c_void_p), ("u_bias", C.c_void_p), ("z", C.c_void_p), ("zw", C.c_void_p),
                ("z_len", C.c_int32), ("z_pitch", C.c_int32), ("zw_sk", C.c_int32), ("zw_sp", C.c_int32),
                ("u_out", C.c_void_p), ("a", C.c_void_p), ("a_mask", C.c_void_p), ("v", C.c_void_p),
                ("a_pitch", C.c_int32), ("v_sk", C.c_int32), ("v_sp", C.c_int32), ("a2_pitch", C.c_int32),
                ("a_out", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out_mask", C.c_void_p),
                ("a2", C.c_void_p), ("w2", C.c_void_p), ("w_sn", C.c_int64), ("w_sk", C.c_int64),
                ("out_pitch", C.c_int32), ("out_accumulate", C.c_int32), ("out_sum_accumulate", C.c_int32), ("w2_sk", C.c_int32),
                ("out", C.c_void_p), ("out_sum", C.c_void_p)]


class LinWgradDesc(C.Structure):
    """hrp_linear_wgrad_desc (include/hrp.h)."""
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("x_pitch", C.c_int32), ("dy_pitch", C.c_int32), ("dw_ld", C.c_int32), ("M", C.c_int32),
                ("K", C.c_int32), ("N", C.c_int32), ("accumulate", C.c_int32), ("reserved", C.c_int32)]


REG_MAX_P, REG_MAX_PROBLEMS, LIN_WGRAD_MAX = 16, 4, 8
COPY_MAX = 16


class CopyDesc(C.Structure):      # hrp_copy_desc
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_pitch", C.c_int), ("dst_pitch", C.c_int), ("rows", C.c_int),
                ("cols", C.c_int), ("accumulate", C.c_int), ("reserved", C.c_int)]


def copy_cols_batch(items, stream):
    """items: [(src pointer or None, src_pitch, dst pointer, dst_pitch, rows, cols, accumulate)] -> hrp_copy_cols_batch launches of
    at most COPY_MAX problems each."""
    for i0 in range(0, len(items), COPY_MAX):
        grp = items[i0:i0 + COPY_MAX]
        arr = (CopyDesc * len(grp))()
        for d, (src, sp, dst, dp, rows, cols, acc) in zip(arr, grp):
            d.src, d.src_pitch, d.dst, d.dst_pitch, d.rows, d.cols, d.accumulate = src, sp, dst, dp, rows, cols, acc
        call("hrp_copy_cols_batch", arr, len(grp), stream)


class FkChain(C.Structure):
    _fields_ = [("njoints", C.c_int32), ("parent", C.c_int32 * FK_MAX_JOINTS), ("type", C.c_int32 * FK_MAX_JOINTS),
                ("cfg", C.c_int32 * FK_MAX_JOINTS), ("mimic_mul", C.c_float * FK_MAX_JOINTS),
                ("mimic_off", C.c_float * FK_MAX_JOINTS), ("origin", (C.c_float * 12) * FK_MAX_JOINTS),
                ("axis", (C.c_float * 3) * FK_MAX_JOINTS), ("nkp", C.c_int32), ("kp_frame", C.c_int32 * FK_MAX_KP),
                ("kp_offset", (C.c_float * 3) * FK_MAX_KP), ("dof", C.c_int32)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
# name -> argtypes (every function returns int unless noted); mirrors include/hrp.h one to one
PROTOTYPES = {
    "hrp_version": [], "hrp_device_ok": [],
    "hrp_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hrp_nhwc_to_nchw": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hrp_nchw_grad_from_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hrp_pack_weights": [_P, _I, _I, _I, _P],
    "hrp_pack_blocks": [_I, _I, _I, _I, _I, _I],
    "hrp_pack_weights_compact": [_P, _P, _I, _I, _I, _P],
    "hrp_nchw_to_nhwc_s2d": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hrp_u8_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "hrp_gather_f32": [_P, _P, _P, _I, _I, _P],
    "hrp_pil_resize_table": [_I, _I, _P],
    "hrp_pil_resize_normalize": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P],
    "hrp_broadcast_hw": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "hrp_bilinear_nhwc_to_nchw": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "hrp_fill_zero": [_P, _L, _P],
    "hrp_maxpool3x3s2_fwd": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    "hrp_maxpool3x3s2_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "hrp_conv2d_fwd": [C.POINTER(ConvDesc), _P],
    "hrp_conv_rowstrip_channels": [C.POINTER(ConvDesc)],
    "hrp_conv_pointwise": [C.POINTER(ConvDesc)],
    "hrp_conv2d_bwd_weight": [C.POINTER(WgradDesc), _P],
    "hrp_colsum": [_P, _I, _L, _I, _I, _P, _I, _P, _L, _P],
    "hrp_colsum_workspace_bytes": [_L, _I],
    "hrp_ew_fwd": [C.POINTER(EwDesc), _P],
    "hrp_ew_bwd_reduce": [C.POINTER(EwBwdDesc), _P],
    "hrp_ew_pool2": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P],
    "hrp_ew_bwd_apply": [C.POINTER(EwBwdDesc), _P],
    "hrp_bn_running_update": [_P, _I, _P], "hrp_bn_fold": [_P, _I, _P], "hrp_bn_param_grad": [_P, _I, _P],
    "hrp_avgpool_fwd": [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    "hrp_avgpool_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "hrp_softargmax3d_fwd": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "hrp_softargmax3d_bwd": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "hrp_pose_geometry_fwd": [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P],
    "hrp_pose_geometry_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "hrp_fk_project_fwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "hrp_fk_project_bwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "hrp_fk_project_rot_fwd": [_P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P],
    "hrp_fk_project_rot_bwd": [_P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "hrp_copy_cols": [_P, _I, _P, _I, _I, _I, _I, _P],
    "hrp_copy_cols_batch": [_P, _I, _P],
    "hrp_scale_rows": [_P, _I, _I, _I, _P, _F, _P],
    "hrp_mul_f32": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "hrp_opt_grad_sumsq": [_P, _P, _I, _P, _P, _P],
    "hrp_opt_adam_step": [_P, _P, _I, _P, _F, _P, _F, _F, _F, _F, _P],
    "hrp_batch_prepare": [_I, _P, _I, _P, C.POINTER(BatchInfo)],
    "hrp_batch_launch": [_P, C.POINTER(BatchInfo), _P],
    "hrp_softargmax_flat_fwd": [_P, _I, _I, _I, _I, _I, _P, _P, _P],
    "hrp_softargmax_flat_bwd": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "hrp_rot6d_compose_fwd": [_P, _P, _P, _I, _P],
    "hrp_rot6d_compose_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "hrp_wgrad_fold_desc_of": [C.POINTER(WgradDesc), C.POINTER(WgradFoldDesc)],
    "hrp_batch_wgrad_fold_descs": [_P, C.POINTER(BatchInfo), C.POINTER(WgradFoldDesc)],
    "hrp_block_channels": [C.POINTER(BlockDesc)],
    "hrp_block_prepare": [_P, _I, _P, C.POINTER(BlockInfo)],
    "hrp_block_launch": [_P, C.POINTER(BlockInfo), _P],
    "hrp_pose_loss": [C.POINTER(PoseLossDesc), _P],
    "hrp_l1_loss": [_P, _P, _F, _I, _P, _P, _P],
    "hrp_sim2real_loss": [C.POINTER(Sim2RealLossDesc), _P],
    "hrp_mesh_pose": [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P],
    "hrp_mesh_pose_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P],
    "hrp_silhouette_fwd": [C.POINTER(SilhouetteDesc), _P],
    "hrp_silhouette_bwd": [C.POINTER(SilhouetteDesc), _P, _P, _P],
    "hrp_linear_fwd": [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _L, _P],
    "hrp_linear_bwd_data": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P],
    "hrp_linear_workspace_bytes": [_I, _I, _I],
    "hrp_linear_bwd_weight": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "hrp_rng_advance": [_P, _P],
    "hrp_dropout_f32": [_P, _I, _P, _I, _P, _I, _I, _F, _P, C.c_uint32, _P],
    "hrp_dropout_masks": [_P, _L, _F, _P, C.c_uint32, _P],
    "hrp_regressor_step": [_P, _I, _P],
    "hrp_linear_wgrad_batch": [_P, _I, _P],
    "hrp_project_fwd": [_P, _P, _I, _I, _P, _P],
    "hrp_project_bwd": [_P, _P, _P, _I, _I, _P, _P],
}

_lib = None


def source_hash():
    """sha256[:16] over the library's sources, the way csrc/Makefile computes SRC_HASH."""
    import glob
    import hashlib
    files = [os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))]
    names = sorted(files + ["../../include/hrp.h", "Makefile"])      # GNU make $(sort): byte order
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libhrp_hip.so (hipcc cross-compiles without a GPU).  force: from scratch."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise HrpError("building libhrp_hip.so failed:\n" + r.stdout[-4000:])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HrpError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # PyTorch ships its own libamdhip64; it must be the HIP runtime already in the process when our
        # library is dlopen'ed, otherwise two runtimes coexist and stream / device handles do not match
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, args in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.hrp_last_error.restype = C.c_char_p
        L.hrp_last_error.argtypes = []
        L.hrp_source_hash.restype = C.c_char_p
        L.hrp_source_hash.argtypes = []
        L.hrp_wgrad_workspace_bytes.restype = C.c_int64
        L.hrp_linear_workspace_bytes.restype = C.c_int64
        L.hrp_colsum_workspace_bytes.restype = C.c_int64
        L.hrp_wgrad_workspace_bytes.argtypes = [C.POINTER(WgradDesc)]
        L.hrp_batch_table_bytes.restype = C.c_int64
        L.hrp_batch_table_bytes.argtypes = [C.c_int, C.c_int]
        L.hrp_block_table_bytes.restype = C.c_int64
        L.hrp_block_table_bytes.argtypes = []
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise HrpError(f"{what} failed ({rc}): {lib().hrp_last_error().decode()}")


_profile_hook = None  # set by bench.py's instrumented pass: fn(name, args, launch) -> None


def set_profile_hook(fn):
    global _profile_hook
    _profile_hook = fn


def profiling():
    return _profile_hook is not None


# development aid (timing ablations only - results are wrong): HRP_SKIP=name1,name2 turns those launches into no-ops
_skip = frozenset(filter(None, os.environ.get("HRP_SKIP", "").split(",")))


def call(name, *args):
    if _skip and name in _skip:
        return
    if _profile_hook is not None:
        _profile_hook(name, args, lambda: check(getattr(lib(), name)(*args), name))
        return
    check(getattr(lib(), name)(*args), name)


def call_batch(batch, stream):
    """One batched launch (plan.BatchLaunch: .table device tensor, .info BatchInfo, .items the merged launches)."""
    name = "hrp_batch_launch"
    if _skip and (name in _skip or FAMILY_FN[batch.fam] in _skip):
        return
    fn = lambda: check(lib().hrp_batch_launch(batch.table.data_ptr(), C.byref(batch.info), stream), name)   # noqa: E731
    if _profile_hook is not None:
        _profile_hook(name, (batch,), fn)
        return
    fn()


def call_block(batch, stream):
    """One fused inference BasicBlock launch (plan.BlockBatch: .table host launch table, .info BlockInfo, .items)."""
    name = "hrp_block_launch"
    if _skip and name in _skip:
        return
    fn = lambda: check(lib().hrp_block_launch(batch.table, C.byref(batch.info), stream), name)   # noqa: E731
    if _profile_hook is not None:
        _profile_hook(name, (batch,), fn)
        return
    fn()


FAMILY_FN = {"block": "hrp_block_launch", "conv": "hrp_conv2d_fwd", "wgrad": "hrp_conv2d_bwd_weight", "wgrad_fold": "hrp_wgrad_fold", "ew_fwd": "hrp_ew_fwd", "ew_red": "hrp_ew_bwd_reduce",
             "ew_app": "hrp_ew_bwd_apply"}
