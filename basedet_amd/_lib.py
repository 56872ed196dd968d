"""ctypes binding of libbasedet_hip.so (the C ABI declared in include/basedet_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a kernel is asked to run on a
tensor that is not on a HIP device, this module raises.  torch is only the memory carrier (data_ptr / stream).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BASEDET_HIP_LIB") or os.path.join(_HERE, "lib", "libbasedet_hip.so")   # override: A/B of two builds

BD_MAX_SEGS = 8
EPI_RELU, EPI_ADD_BEFORE, EPI_ADD_AFTER, EPI_MASK, EPI_SPARSE = 1, 2, 4, 8, 16


class BasedetHipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("R", C.c_int32), ("S", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("nseg", C.c_int32),
        ("Hi", C.c_int32 * BD_MAX_SEGS), ("Wi", C.c_int32 * BD_MAX_SEGS),
        ("Ho", C.c_int32 * BD_MAX_SEGS), ("Wo", C.c_int32 * BD_MAX_SEGS),
        ("in_off", C.c_int32 * BD_MAX_SEGS), ("out_off", C.c_int32 * BD_MAX_SEGS),
        ("in_pix_per_img", C.c_int32), ("out_pix_per_img", C.c_int32),
        ("route", C.c_int32 * 4), ("sr_seed", C.c_uint32),        # per-call kernel routing / e5m2 rounding seed (0 = the library's choice)
    ]


class PackDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("row_scale", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("Cout", C.c_int32), ("RS", C.c_int32), ("Cin", C.c_int32), ("block_start", C.c_int32)]


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every symbol include/basedet_hip.h declares
SIGNATURES = {
    "bd_last_error_string": (C.c_char_p, []),
    "bd_version": (_I, []),
    "bd_conv_last_kernel": (C.c_char_p, []),
    "bd_probe_mfma_rate": (_I, [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "bd_probe_kernel_clock": (_I, [C.c_char_p, _I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bd_conv2d_fwd": (_I, [_D, _P, _P, _P, _P, _P, _I, _P]),
    "bd_conv2d_dgrad": (_I, [_D, _P, _P, _P, _P, _P, _I, _P]),
    "bd_conv2d_fwd_bits": (_I, [_D, _P, _P, _P, _P, _P, _P, _I, _P]),
    "bd_conv2d_fwd_ex": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "bd_conv2d_dgrad_ex": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "bd_conv2d_dgrad_bits": (_I, [_D, _P, _P, _P, _P, _P, _I, _P]),
    "bd_conv2d_wgrad_workspace_bytes": (_Z, [_D]),
    "bd_conv2d_wgrad": (_I, [_D, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "bd_conv2d_wgrad_bias_workspace_bytes": (_Z, [_D]),
    "bd_conv2d_wgrad_bias": (_I, [_D, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "bd_wgrad_queue_create": (_I, [_P]),
    "bd_wgrad_queue_destroy": (_I, [_P]),
    "bd_wgrad_queue_pending": (_I, [_P]),
    "bd_conv2d_wgrad_queued": (_I, [_P, _D, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "bd_wgrad_queue_flush": (_I, [_P, _P]),
    "bd_stem_conv7x7_fwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "bd_stem_pool_fwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "bd_stem_weight_pack": (_I, [_P, _P, _P, _P]),
    "bd_weight_pack": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "bd_weight_pack_blocks": (_I, [_I, _I, _I]),
    "bd_weight_pack_multi": (_I, [_P, _I, _I, _P]),
    "bd_colsum_workspace_bytes": (_Z, [_I]),
    "bd_colsum_bf16": (_I, [_P, _I, _L, _L, _L, _I, _P, _I, _P, _Z, _P]),
    "bd_pad_normalize": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "bd_bottleneck_fwd_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "bd_bottleneck_fwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bd_h2d_create": (_I, [C.POINTER(C.c_void_p), _I, _I]),
    "bd_h2d_threads": (_I, [_P]),
    "bd_h2d_submit": (_I, [_P, _P, _I, _L, _P, _L, _P]),
    "bd_h2d_destroy": (_I, [_P]),
    "bd_pad_normalize_nchw": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "bd_maxpool3x3s2_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "bd_upsample2x_add_fwd": (_I, [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _P]),
    "bd_upsample2x_add_bwd": (_I, [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "bd_relu_bf16": (_I, [_P, _P, _L, _P]),
    "bd_relu_bwd_bf16": (_I, [_P, _P, _P, _P, _L, _P]),
    "bd_add_bf16": (_I, [_P, _P, _P, _L, _P]),
    "bd_anchors_generate": (_I, [_I, _I, _I, _F, _P, _I, _P, _P]),
    "bd_points_generate": (_I, [_I, _I, _I, _F, _I, _P, _P]),
    "bd_box_pairwise": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "bd_box_encode": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "bd_box_decode": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "bd_retina_assign_encode": (_I, [_P, _I, _P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_fcos_assign": (_I, [_P, _I, _P, _P, _P, _I, _F, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "bd_ota_assign_workspace_bytes": (_Z, [_I, _I]),
    "bd_ota_assign": (_I, [_P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_ota_sinkhorn_workspace_bytes": (_Z, [_I, _I, _I]),
    "bd_ota_assign_sinkhorn": (_I, [_P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _F, _I, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_freeanchor_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "bd_freeanchor_loss_fwd_bwd": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _P, _P, _F, _I, _F, _F, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "bd_atss_assign_workspace_bytes": (_Z, [_I, _I]),
    "bd_atss_assign": (_I, [_P, _I, _P, _P, _I, _I, _F, _P, _P, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_nms_workspace_bytes": (_Z, [_I]),
    "bd_batched_nms": (_I, [_P, _P, _P, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "bd_focal_loss_fwd_bwd": (_I, [_P, _P, _L, _I, _F, _F, _P, _I, _F, _P, _P, _P]),
    "bd_focal_loss_fwd_bwd_general": (_I, [_P, _P, _L, _I, _F, _F, _P, _I, _F, _P, _P, _P]),
    "bd_smooth_l1_fwd_bwd": (_I, [_P, _P, _P, _L, _I, _I, _F, _P, _I, _F, _P, _P, _P]),
    "bd_giou_ltrb_fwd_bwd": (_I, [_P, _P, _P, _P, _L, _P, _F, _P, _P, _P]),
    "bd_bce_logits_fwd_bwd": (_I, [_P, _I, _I, _P, _P, _L, _P, _P, _P, _P]),
    "bd_groupnorm_workspace_bytes": (_Z, [_I, _I, _I, _L]),
    "bd_groupnorm_fwd": (_I, [_P, _P, _P, _I, _I, _P, _P, _L, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "bd_conv2d_fwd_gnstats_bytes": (_Z, [_D]),
    "bd_conv2d_fwd_gnstats": (_I, [_D, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_groupnorm_fwd_parts": (_I, [_D, _P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "bd_groupnorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _L, _I, _I, _P, _P, _P, _I, _P, _Z, _P]),
    "bd_fcos_offsets_fwd": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _L, _P, _P]),
    "bd_fcos_offsets_workspace_bytes": (_Z, []),
    "bd_fcos_offsets_bwd": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_rpn_assign_encode": (_I, [_P, _I, _P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bd_sample_labels": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "bd_segment_topk": (_I, [_P, _I, _I, _L, _I, _I, _I, _I, _P, _P, _I, _F, _I, _P, _P, _P, _P]),
    "bd_nms_batched_workspace_bytes": (_Z, [_I, _I]),
    "bd_nms_batched": (_I, [_P, _P, _P, _I, _I, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "bd_rpn_proposals_workspace_bytes": (_Z, [_I, _I, _P, _I, _I, _I]),
    "bd_rpn_proposals": (_I, [_P, _I, _I, _I, _I, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "bd_rpn_proposals_joint": (_I, [_P, _I, _I, _I, _I, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "bd_rcnn_sample_targets": (_I, [_P, _P, _I, _P, _P, _I, _I, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bd_roi_align_fwd": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "bd_conv1x1_thin_bwd_workspace_bytes": (_Z, []),
    "bd_conv1x1_thin_fwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P]),
    "bd_conv1x1_thin_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _I, _P, _Z, _P]),
    "bd_roi_align_bwd": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "bd_roi_align_bwd_bf16_workspace_bytes": (_Z, [_I, _I, _P, _P, _I]),
    "bd_roi_align_bwd_bf16": (_I, [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "bd_subsample2x_fwd": (_I, [_P, _L, _L, _I, _I, _P, _L, _L, _I, _I, _P]),
    "bd_subsample2x_bwd_add": (_I, [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _P]),
    "bd_f32_to_bf16": (_I, [_P, _P, _L, _P]),
    "bd_f32_to_bf16_add": (_I, [_P, _P, _L, _P]),
    "bd_rpn_loss_fwd_bwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _L, _F, _P, _P, _P, _P]),
    "bd_rcnn_loss_fwd_bwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P]),
    "bd_det_scores": (_I, [_P, _P, _I, _I, _L, _I, _P, _P]),
    "bd_rcnn_predict": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "bd_det_candidates": (_I, [_I, _P, _P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "bd_det_finalize": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "bd_sgd_momentum_step": (_I, [_P, _P, _P, _L, _F, _F, _F, _F, _P]),
    "bd_clip_grad_value": (_I, [_P, _L, _F, _F, _F, _P]),
    "bd_clip_grad_norm_workspace_bytes": (_Z, []),
    "bd_clip_grad_norm": (_I, [_P, _L, _F, _F, _F, _P, _P, _Z, _P]),
    "bd_quantize_fp8": (_I, [_P, _L, _F, _P, _P]),
    "bd_weight_pack_fp8": (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "bd_conv2d_fwd_fp8": (_I, [_D, _P, _P, _P, _P, _P, _P, _I, _P]),
    "bd_conv2d_fwd_fp8_ex": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "bd_quantize_bf8": (_I, [_P, _L, _F, _P, C.c_uint32, _P]),
    "bd_absmax_bf16": (_I, [_P, _L, _P, _P]),
    "bd_weight_pack_fp8_t": (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "bd_conv2d_dgrad_fp8": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "bd_conv2d_wgrad_fp8_workspace_bytes": (_Z, [_D]),
    "bd_conv2d_wgrad_fp8": (_I, [_D, _P, _P, _F, _P, _P, _I, _P, _Z, _P]),
    "bd_conv1x1_fp8": (_I, [_D, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "bd_sigmoid_focal_loss_elem": (_I, [_P, _P, _L, _F, _F, _P, _P, _P, _P]),
    "bd_bce_elem": (_I, [_P, _P, _L, _I, _P, _P, _P, _P]),
    "bd_smooth_l1_elem": (_I, [_P, _P, _L, _F, _P, _P, _P, _P]),
    "bd_iou_loss_ltrb": (_I, [_P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "bd_iou_to_loss": (_I, [_P, _L, _I, _F, _P, _P]),
    "bd_matcher_matrix": (_I, [_P, _I, _L, _P, _P, _I, _I, _P, _P, _P, _P]),
    "bd_assign_roi_levels": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "bd_roi_pool_max_fwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _I, _I, _P, _P]),
    "bd_comm_unique_id": (_I, [_P]),
    "bd_comm_init": (_I, [C.POINTER(C.c_void_p), _P, _I, _I, _I]),
    "bd_comm_rank": (_I, [_P]),
    "bd_comm_world": (_I, [_P]),
    "bd_comm_stream": (_P, [_P]),
    "bd_comm_bcast": (_I, [_P, _P, _Z, _I, _I, _P]),
    "bd_comm_allreduce": (_I, [_P, _P, _Z, _I, _I, _P]),
    "bd_comm_allreduce_async": (_I, [_P, _P, _Z, _I, _I, C.POINTER(C.c_void_p), _I]),
    "bd_comm_allreduce_async_bf16": (_I, [_P, _P, _P, _Z, _I, C.POINTER(C.c_void_p), _I]),
    "bd_comm_wait": (_I, [_P, _P]),
    "bd_comm_destroy": (_I, [_P]),
}

_lib = None


def load():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BasedetHipError(
            f"{LIB_PATH} is missing: build it with `python -m basedet_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the basedet_amd operator path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().bd_last_error_string()
        raise BasedetHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL). Refuses host tensors: no CPU fallback."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise BasedetHipError("basedet_amd kernels need tensors on a HIP device (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise BasedetHipError("basedet_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def f32arr(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def i32arr(vals):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])
