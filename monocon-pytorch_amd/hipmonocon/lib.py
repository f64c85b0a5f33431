"""ctypes binding of libmonocon_hip.so (C-ABI declared in include/monocon_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this
module raises.  Tensors cross the boundary as raw device pointers only.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmonocon_hip.so")

MC_F32, MC_I64 = 0, 1
NUM_PREDS = 10

EXPORTS = (
    "mc_create", "mc_destroy", "mc_last_error", "mc_version", "mc_bind_params", "mc_pack_params",
    "mc_forward_infer", "mc_backbone_forward", "mc_neck_forward", "mc_head_forward", "mc_decode", "mc_make_targets", "mc_losses", "mc_losses_backward", "mc_losses_backward_pred", "mc_forward_train", "mc_backward", "mc_train_generation", "mc_head_forward_train", "mc_head_backward", "mc_train_debug_node", "mc_optim_bind", "mc_clip_adamw_step", "mc_op_conv", "mc_op_conv_wgrad", "mc_op_conv_dgrad", "mc_op_stem", "mc_op_maxpool2", "mc_op_deconv4x4",
    "mc_op_nchw_to_nhwc", "mc_op_nhwc_to_nchw", "mc_workspace_bytes", "mc_query_workspace", "mc_forward_cost",
    "mc_preprocess", "mc_preprocess_augmented", "mc_profile_forward", "mc_profile_train", "mc_set_precision", "mc_set_local_maximum_kernel", "mc_set_conv_cfg", "mc_bench_conv", "mc_bench_mfma_peak",
    "mc_comm_unique_id", "mc_comm_init", "mc_comm_destroy", "mc_comm_set_overlap", "mc_comm_info", "mc_comm_exposed_ms", "mc_allreduce_grads",
    "mc_build_train_plan", "mc_tune_export", "mc_tune_import",
    "mc_rotate_iou_eval", "mc_box3d_overlap", "mc_kitti_image_overlap", "mc_kitti_statistics_part",
)


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p), ("numel", C.c_int64), ("dtype", C.c_int32)]


class Labels(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("gt_bboxes", "gt_labels", "gt_bboxes_3d", "depths", "gt_kpts_2d",
                                          "gt_kpts_valid_mask", "mask")]


TARGET_FIELDS = ("center_heatmap_target", "wh_target", "offset_target", "dim_target", "alpha_cls_target",
                 "alpha_offset_target", "depth_target", "center2kpt_offset_target", "kpt_heatmap_target",
                 "kpt_heatmap_offset_target", "indices", "indices_kpt", "mask_target", "mask_center2kpt_offset",
                 "mask_kpt_heatmap_offset")


class Targets(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TARGET_FIELDS]


class MonoconHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MonoconHipError(
            "libmonocon_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C monocon-pytorch_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    # torch first: it ships its own libamdhip64, and the process must end up with ONE HIP runtime -- loaded before torch,
    # this library would bind the system runtime and then see no device next to torch's (measured on the GPU box)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    fp = C.POINTER(C.c_float)
    lib.mc_version.restype = i
    lib.mc_create.argtypes = [i, C.POINTER(vp)]
    lib.mc_destroy.argtypes = [vp]
    lib.mc_last_error.argtypes = [vp]
    lib.mc_last_error.restype = C.c_char_p
    lib.mc_bind_params.argtypes = [vp, C.POINTER(TensorDesc), i]
    lib.mc_pack_params.argtypes = [vp, i, vp]
    lib.mc_forward_infer.argtypes = [vp, vp, i, i, i, C.POINTER(vp), vp, vp]
    lib.mc_backbone_forward.argtypes = [vp, vp, i, i, i, C.POINTER(vp), vp]
    lib.mc_neck_forward.argtypes = [vp, C.POINTER(vp), i, i, i, vp, vp]
    lib.mc_head_forward.argtypes = [vp, vp, i, i, i, C.POINTER(vp), vp]
    lib.mc_decode.argtypes = [vp, C.POINTER(vp), vp, vp, i, i, i, i, i, f, f, f, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mc_make_targets.argtypes = [vp, C.POINTER(Labels), i, i, i, i, i, i, C.POINTER(Targets), vp]
    lib.mc_losses.argtypes = [vp, C.POINTER(vp), C.POINTER(Targets), i, i, i, i, vp, vp]
    lib.mc_losses_backward.argtypes = [vp, C.POINTER(vp), C.POINTER(Targets), i, i, i, i, vp, C.POINTER(vp), vp]
    lib.mc_losses_backward_pred.argtypes = lib.mc_losses_backward.argtypes
    lib.mc_forward_train.argtypes = [vp, vp, C.POINTER(Labels), i, i, i, i, C.POINTER(vp), vp, vp]
    lib.mc_backward.argtypes = [vp, vp, vp]
    lib.mc_head_forward_train.argtypes = [vp, vp, C.POINTER(Labels), i, i, i, i, C.POINTER(vp), vp, vp]
    lib.mc_head_backward.argtypes = [vp, vp, vp, vp]
    lib.mc_train_generation.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    lib.mc_train_debug_node.argtypes = [vp, i, i, vp, C.POINTER(i), vp]
    d = C.c_double
    lib.mc_optim_bind.argtypes = [vp, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.mc_clip_adamw_step.argtypes = [vp, d, d, d, d, d, d, i, vp, vp]
    lib.mc_op_conv.argtypes = [vp, C.POINTER(vp), C.POINTER(i), i, i, i, i, vp, i, i, i, vp, vp, vp, i, vp, vp]
    lib.mc_op_conv_wgrad.argtypes = [vp, C.POINTER(vp), C.POINTER(i), i, i, i, i, vp, i, i, i, vp, vp]
    lib.mc_op_stem.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp]
    lib.mc_op_maxpool2.argtypes = [vp, vp, i, i, i, i, vp, vp]
    lib.mc_op_deconv4x4.argtypes = [vp, vp, i, i, i, i, vp, vp, vp]
    lib.mc_op_nchw_to_nhwc.argtypes = [vp, vp, i, i, i, i, vp, vp]
    lib.mc_op_nhwc_to_nchw.argtypes = [vp, vp, i, i, i, i, vp, vp]
    lib.mc_workspace_bytes.argtypes = [vp]
    lib.mc_query_workspace.argtypes = [vp, i, i, i, i, C.POINTER(C.c_size_t)]
    lib.mc_workspace_bytes.restype = C.c_size_t
    lib.mc_forward_cost.argtypes = [vp, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mc_profile_forward.argtypes = [vp, i, fp, C.POINTER(i), vp]
    lib.mc_bench_mfma_peak.argtypes = [vp, i, i, fp]
    lib.mc_profile_train.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i), vp]
    lib.mc_preprocess.argtypes = [vp, vp, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), i, i, vp, vp]
    lib.mc_preprocess_augmented.argtypes = [vp, vp, vp, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), i, i, vp, vp]
    lib.mc_op_conv_dgrad.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, i, vp, vp]
    lib.mc_comm_unique_id.argtypes = [vp, vp]
    lib.mc_comm_init.argtypes = [vp, i, i, vp]
    lib.mc_comm_destroy.argtypes = [vp]
    lib.mc_comm_set_overlap.argtypes = [vp, i]
    lib.mc_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(C.c_ulonglong), C.c_char_p, i]
    lib.mc_comm_exposed_ms.argtypes = [vp, fp]
    lib.mc_allreduce_grads.argtypes = [vp, vp]
    lib.mc_build_train_plan.argtypes = [vp, i, i, i]
    lib.mc_tune_export.argtypes = [vp, C.POINTER(i), i, C.POINTER(i)]
    lib.mc_tune_import.argtypes = [vp, C.POINTER(i), i]
    ll, dp, llp = C.c_longlong, C.POINTER(C.c_double), C.POINTER(C.c_longlong)
    lib.mc_rotate_iou_eval.argtypes = [vp, vp, vp, ll, ll, i, vp, vp]
    lib.mc_box3d_overlap.argtypes = [vp, vp, vp, ll, ll, i, vp, vp]
    lib.mc_kitti_image_overlap.argtypes = [dp, ll, dp, ll, i, dp]
    lib.mc_kitti_statistics_part.argtypes = [i, dp, ll, llp, llp, llp, dp, dp, dp, llp, llp, i, C.c_double, dp, ll, i, dp, dp, llp]
    lib.mc_set_precision.argtypes = [vp, i]
    lib.mc_set_local_maximum_kernel.argtypes = [vp, i]
    lib.mc_set_conv_cfg.argtypes = [vp, i]
    lib.mc_bench_conv.argtypes = [vp, i, i, i, i, C.POINTER(i), i, i, i, i, i, fp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int or name not in ("mc_last_error", "mc_workspace_bytes"):
            if name not in ("mc_last_error", "mc_workspace_bytes"):
                fn.restype = i
    _lib = lib
    return lib


def check(handle, rc, what):
    if rc != 0:
        msg = load().mc_last_error(handle)
        raise MonoconHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
