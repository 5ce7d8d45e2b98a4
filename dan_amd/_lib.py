"""ctypes binding of libdanhip.so (the C ABI declared in include/danhip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# 16-bit activation type of this process: DANHIP_DTYPE=bf16 (default) | fp16 selects the library build (include/danhip.h)
ACT_NAME = os.environ.get("DANHIP_DTYPE", "bf16").lower()
if ACT_NAME not in ("bf16", "fp16"):
    raise ValueError("DANHIP_DTYPE must be bf16 or fp16, got %r" % ACT_NAME)
ACT_DTYPE = torch.float16 if ACT_NAME == "fp16" else torch.bfloat16
SO_PATH = os.path.join(_HERE, "libdanhip_f16.so" if ACT_NAME == "fp16" else "libdanhip.so")
if os.environ.get("DANHIP_LIB_PATH"):          # diagnosis: A/B another in-tree build of the same library (tools/build_one.sh variants)
    SO_PATH = os.path.abspath(os.environ["DANHIP_LIB_PATH"])

F32, BF16 = 0, 1      # BF16 = "the build's 16-bit activation type" in out_dtype arguments
SPLIT3 = 3            # fp16 build, forward convolutions: the [hi | lo | hi] half-limb layout of csrc/split_infer.hip


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "H", "W", "Cin", "Ho", "Wo", "Cout", "kh", "kw", "stride")]


class ConvPitch(ctypes.Structure):
    """danhip_conv_pitch: pixel pitches (elements) of a call's input / output operand and of a data gradient's mask (channel-slice views)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("x_pitch", "y_pitch", "aux_pitch")]


class PackEntry(ctypes.Structure):
    _fields_ = [("w_hwio", ctypes.c_void_p), ("wf_packed", ctypes.c_void_p), ("wb_packed", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("kh", "kw", "cin", "cin_real", "cout", "rows_f", "cols_f", "rows_b", "cols_b", "co8", "first_block", "pad_")]


P = ctypes.c_void_p
I32 = ctypes.c_int32
I64 = ctypes.c_int64
FL = ctypes.c_float
DESC = ctypes.POINTER(ConvDesc)

# name -> argtypes (every function returns int status except the two noted below)
SIGNATURES = {
    "danhip_conv_packed_dims": [DESC, ctypes.c_int, ctypes.POINTER(I64), ctypes.POINTER(I64)],
    "danhip_pack_conv_weight": [DESC, P, I32, P, P, P],
    "danhip_conv2d_fwd_pool": [DESC, P, P, P, P, P, P],
    "danhip_pack_entry_init": [ctypes.POINTER(PackEntry), DESC, P, I32, P, P, I32, ctypes.POINTER(ctypes.c_int32)],
    "danhip_pack_conv_weights_batched": [P, I32, I32, P],
    "danhip_conv2d_fwd": [DESC, P, P, P, P, ctypes.c_int, ctypes.c_int, P, P],
    "danhip_conv2d_fwd_ws": [DESC, P, P, P, P, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_size_t, P],
    "danhip_conv2d_fwd_f32": [DESC, P, P, P, P, ctypes.c_int, P, P],
    "danhip_maxpool2x2_fwd_f32": [P, P, I32, I32, I32, I32, P],
    "danhip_split3_f32": [P, P, I64, I32, I32, ctypes.c_int, P],
    "danhip_unsplit3_f32": [P, P, I64, I32, I32, P],
    "danhip_maxpool2x2_split3": [P, P, I32, I32, I32, I32, P],
    "danhip_l2norm_split3": [P, P, P, I64, I32, P],
    "danhip_conv3x3_c3_f32_split3": [P, P, P, P, I32, I32, I32, I32, ctypes.c_int, P],
    "danhip_l2norm_fwd_f32": [P, P, P, I64, I32, P],
    "danhip_resize_bilinear_add_fwd_f32": [P, P, P, I32, I32, I32, I32, I32, I32, P],
    "danhip_avgpool2x2s1_same_fwd_f32": [P, P, I32, I32, I32, I32, P],
    "danhip_deform_sample_fwd_f32": [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P],
    "danhip_conv2d_bwd_data": [DESC, P, P, P, P, ctypes.c_int, P],
    "danhip_conv2d_bwd_data_ws": [DESC, P, P, P, P, ctypes.c_int, P, ctypes.c_size_t, P],
    "danhip_conv2d_bwd_data_bits": [DESC, P, P, P, P, ctypes.c_int, P],
    "danhip_conv2d_bwd_data_bits_first": [DESC, P, P, P, P, I32, P, P, P],
    "danhip_relu_bits": [P, P, I64, I32, P],
    "danhip_conv2d_fwd_relu_bits": [DESC, P, P, P, P, P, P, P, P],
    "danhip_conv2d_bwd_weight": [DESC, P, P, P, P, I32, P],
    "danhip_conv2d_bwd_weight_ws": [DESC, P, P, P, P, I32, P, ctypes.c_size_t, P],
    "danhip_avgpool2x2s1_same_fwd_strided": [P, I32, P, I32, I32, I32, I32, I32, ctypes.c_int, P],
    "danhip_avgpool2x2s1_same_bwd_strided": [P, I32, P, I32, I32, I32, I32, I32, P],
    "danhip_add16": [P, P, P, I64, P],
    "danhip_residual_bwd": [P, P, P, P, P, ctypes.c_int, I64, P],
    "danhip_conv2d_fwd_strided": [DESC, P, P, P, P, ctypes.c_int, I32, ctypes.POINTER(ConvPitch), P, ctypes.c_size_t, P],
    "danhip_conv2d_bwd_data_strided": [DESC, P, P, P, P, ctypes.c_int, ctypes.POINTER(ConvPitch), P, ctypes.c_size_t, P],
    "danhip_conv2d_fwd_concat2": [DESC, P, P, I32, I32, P, P, P, ctypes.c_int, P],
    "danhip_conv2d_bwd_weight_strided": [DESC, P, P, P, P, I32, ctypes.POINTER(ConvPitch), P, ctypes.c_size_t, P],
    "danhip_relu_bwd_bias_grad": [P, P, P, I64, I32, P],
    "danhip_maxpool2x2_fwd": [P, P, I32, I32, I32, I32, P],
    "danhip_maxpool2x2_fwd_arg": [P, P, P, I32, I32, I32, I32, P],
    "danhip_maxpool2x2_bwd_arg": [P, P, P, I32, I32, I32, I32, ctypes.c_int, P],
    "danhip_conv2d_fwd_pool_arg": [DESC, P, P, P, P, P, P, P],
    "danhip_conv2d_fwd_relu_bits_arg": [DESC, P, P, P, P, P, P, P, P, P],
    "danhip_maxpool2x2_bwd": [P, P, P, I32, I32, I32, I32, ctypes.c_int, P],
    "danhip_l2norm_fwd": [P, P, P, I64, I32, P],
    "danhip_l2norm_bwd": [P, P, P, P, P, I64, I32, ctypes.c_int, ctypes.c_int, P],
    "danhip_preprocess_u8": [P, P, I64, P],
    "danhip_resize_bilinear_add_fwd": [P, P, P, I32, I32, I32, I32, I32, I32, P],
    "danhip_resize_bilinear_add_bwd": [P, P, I32, I32, I32, I32, I32, I32, ctypes.c_int, P],
    "danhip_avgpool2x2s1_same_fwd": [P, P, I32, I32, I32, I32, P],
    "danhip_avgpool2x2s1_same_bwd": [P, P, P, I32, I32, I32, I32, ctypes.c_int, P],
    "danhip_slice_deliver": [P, I32, I32, I32, P, I32, P, I32, ctypes.c_int, I64, P],
    "danhip_batchnorm_fwd_train": [P, P, P, P, P, P, P, P, I64, I32, FL, FL, ctypes.c_int, P, P],
    "danhip_batchnorm_fwd_infer": [P, P, P, P, P, P, I64, I32, ctypes.c_int, P],
    "danhip_batchnorm_bwd": [P, P, P, P, P, P, P, P, I64, I32, P],
    "danhip_dynamic_anchor_routing_eval": [P, P, P, P, I64, I32, I32, I32, I32, I32, P, P, P, ctypes.c_size_t, P],
    "danhip_dynamic_anchor_routing_train": [P, P, P, P, I64, I32, I32, I32, I32, I32, FL, FL, ctypes.c_uint64, ctypes.c_uint64, P, P, P, P,
                                            ctypes.c_size_t, P],
    "danhip_nms": [P, I32, I32, I32, FL, P, P, P],
    "danhip_argsort_desc_f32": [P, I64, I32, P, P, ctypes.c_size_t, P],
    "danhip_augment_preprocess": [P, I32, I32, I32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float), I32, I32, I32, I32, I32, P, I32, I32,
                                  P, ctypes.c_size_t, P],
    "danhip_deform_psroi_pool_fwd": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, FL, FL, I32, I32, P],
    "danhip_deform_psroi_pool_bwd": [P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, FL, FL, I32, I32, P],
    "danhip_maxpool3x3s2_same_fwd": [P, P, I32, I32, I32, I32, P],
    "danhip_maxpool3x3s2_same_bwd": [P, P, P, I32, I32, I32, I32, P],
    "danhip_resize_u8_linear": [P, I32, I32, P, I32, I32, I32, ctypes.c_double, ctypes.c_double, P],
    "danhip_bbox_vote": [P, P, I32, I32, ctypes.c_double, I32, P, P, P, ctypes.c_size_t, P],
    "danhip_deform_sample_fwd": [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P],
    "danhip_deform_sample_bwd": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, ctypes.c_int, P, ctypes.c_size_t, P],
    "danhip_deform_conv_fwd": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, ctypes.c_int, P, ctypes.c_size_t, P],
    "danhip_deform_conv_bwd": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, ctypes.c_int, P, ctypes.c_size_t, P],
    "danhip_deform_conv_bwd_with_col": [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, ctypes.c_int, P, ctypes.c_size_t, P],
    "danhip_deform_conv_bwd_deliver": [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, ctypes.c_int, ctypes.c_int, P, ctypes.c_size_t, P],
    "danhip_cast_pad_f32_to_bf16": [P, P, P, I64, I32, I32, P],
    "danhip_head_split_fwd": [P, P, P, I32, I32, I32, I32, I32, I32, I32, P],
    "danhip_head_split_bwd": [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, P],
    "danhip_hard_neg_select": [P, P, P, P, P, P, I32, I32, FL, ctypes.c_int, P],
    "danhip_detection_loss_fwd": [P, P, P, P, P, P, P, P, I32, I32, P],
    "danhip_detection_loss_bwd": [P, P, P, P, P, P, P, FL, FL, I32, I32, P],
    "danhip_sgd_momentum_flat": [P, P, P, P, P, P, I32, I64, FL, FL, FL, P, P],
    "danhip_sgd_momentum_flat_dynamic": [P, P, P, P, P, P, I32, I64, FL, FL, P, P, P],
    "danhip_anchors_generate": [P, P, P, P, P, P, I32, I32, I32, FL, FL, FL, I32, P],
    "danhip_iou_matrix": [P, P, P, P, P, P, P, I32, I32, P],
    "danhip_dual_max_match": [P, I32, I32, FL, FL, ctypes.c_int, P, P, P, ctypes.c_size_t, P],
    "danhip_small_mining_match": [P, I32, I32, FL, FL, FL, I32, FL, P, P, P, ctypes.c_size_t, P],
    "danhip_encode_anchors": [P, P, P, P, P, P, P, P, P, I32, FL, FL, FL, FL, FL, P],
    "danhip_decode_anchors": [P, P, P, P, P, P, I32, I32, FL, FL, FL, FL, P],
    "danhip_face_scores": [P, P, P, FL, I64, P],
    "danhip_comm_load": [ctypes.c_char_p],
    "danhip_comm_rccl_version": [ctypes.POINTER(ctypes.c_int)],
    "danhip_comm_unique_id": [P],
    "danhip_comm_create": [P, I32, I32, I32, ctypes.POINTER(ctypes.c_void_p)],
    "danhip_comm_destroy": [P],
    "danhip_comm_info": [P, ctypes.POINTER(I32), ctypes.POINTER(I32), ctypes.POINTER(I32)],
    "danhip_comm_async_error": [P, ctypes.POINTER(I32)],
    "danhip_comm_abort": [P],
    "danhip_comm_allreduce_sum": [P, P, I64, ctypes.c_int, P],
    "danhip_comm_reduce_scatter_sum": [P, P, P, I64, ctypes.c_int, P],
    "danhip_comm_allgather": [P, P, P, I64, ctypes.c_int, P],
    "danhip_encode_anchors_batched": [P] * 11 + [I32, I32, I32, I32, I32, FL, FL, FL, I32, FL, FL, FL, FL, FL, FL, P, P, P, P, P,
                                                 ctypes.c_size_t, P],
}

_lib = None


class DanhipError(RuntimeError):
    pass


def lib():
    """Loads libdanhip.so; raises loudly when it is missing (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        _lib = _load(SO_PATH, ACT_NAME)
    return _lib


_lib_f16 = None


def lib_f16():
    """The fp16 build (libdanhip_f16.so) whatever DANHIP_DTYPE says: the split-operand evaluation path (csrc/split_infer.hip) runs its
    three-limb-product convolutions on v_mfma_f32_16x16x32_f16 also inside a bf16 process.  The two libraries share nothing (separate
    option tables, error strings, RCCL binding); this one is used for forward convolutions and weight packing only."""
    global _lib_f16
    if _lib_f16 is None:
        _lib_f16 = lib() if ACT_NAME == "fp16" else _load(os.path.join(_HERE, "libdanhip_f16.so"), "fp16")
    return _lib_f16


def call_f16(name, *args):
    L = lib_f16()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise DanhipError("%s (fp16 build) failed (%d): %s" % (name, rc, L.danhip_last_error().decode()))


def _load(so_path, act_name):
    if True:
        if not os.path.exists(so_path):
            raise DanhipError("libdanhip.so not found at %s — run `python -m dan_amd.build` (there is no fallback path)" % so_path)
        L = ctypes.CDLL(so_path)
        L.danhip_last_error.restype = ctypes.c_char_p
        L.danhip_last_error.argtypes = []
        L.danhip_version.restype = ctypes.c_int
        L.danhip_act_dtype.restype = ctypes.c_int
        if L.danhip_act_dtype() != (2 if act_name == "fp16" else 1):
            raise DanhipError("%s was built for another activation dtype than %s" % (so_path, act_name))
        L.danhip_match_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_conv_kernel_label.restype = ctypes.c_char_p
        L.danhip_conv_kernel_label.argtypes = [DESC, ctypes.c_int]
        L.danhip_conv_wgrad_kernel_label.restype = ctypes.c_char_p
        L.danhip_conv_wgrad_kernel_label.argtypes = [DESC]
        L.danhip_match_workspace_bytes.argtypes = [I32, I32]
        L.danhip_encode_anchors_batched_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_encode_anchors_batched_workspace_bytes.argtypes = [I32, I32, I32]
        L.danhip_routing_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_routing_workspace_bytes.argtypes = [I64, I32, ctypes.c_int]
        L.danhip_bbox_vote_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_augment_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_argsort_workspace_bytes.argtypes = [I64]
        L.danhip_argsort_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_augment_workspace_bytes.argtypes = []
        L.danhip_set_option.restype = ctypes.c_int
        L.danhip_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.danhip_get_option.restype = ctypes.c_int
        L.danhip_get_option.argtypes = [ctypes.c_char_p]
        L.danhip_conv2d_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_conv2d_workspace_bytes.argtypes = [DESC, ctypes.c_int]
        L.danhip_conv2d_bwd_weight_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_conv2d_bwd_weight_workspace_bytes.argtypes = [DESC]
        L.danhip_deform_conv_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_deform_conv_workspace_bytes.argtypes = [I32, I32, I32, I32, I32, I32, I32, ctypes.c_int]
        L.danhip_deform_sample_bwd_workspace_bytes.restype = ctypes.c_size_t
        L.danhip_deform_sample_bwd_workspace_bytes.argtypes = [I32, I32, I32, I32]
        L.danhip_conv2d_fwd_pool_only.restype = ctypes.c_int
        L.danhip_conv2d_fwd_pool_only.argtypes = [DESC]
        L.danhip_conv2d_fwd_concat2_supported.restype = ctypes.c_int
        L.danhip_conv2d_fwd_concat2_supported.argtypes = [DESC, I32, I32]
        L.danhip_conv2d_fwd_emits_bits.restype = ctypes.c_int
        L.danhip_conv2d_fwd_emits_bits.argtypes = [DESC, ctypes.c_int]
        L.danhip_conv2d_bwd_data_takes_bits.restype = ctypes.c_int
        L.danhip_conv2d_bwd_data_takes_bits.argtypes = [DESC]
        L.danhip_conv2d_bwd_data_first_supported.restype = ctypes.c_int
        L.danhip_conv2d_bwd_data_first_supported.argtypes = [DESC]
        L.danhip_deform_conv_fused.restype = ctypes.c_int
        L.danhip_deform_conv_fused.argtypes = [I32] * 9
        L.danhip_bbox_vote_workspace_bytes.argtypes = [I32, I32]
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the export is missing
            fn.restype = ctypes.c_int
            fn.argtypes = args
    return L


def check(rc, what):
    if rc != 0:
        raise DanhipError("%s failed (%d): %s" % (what, rc, lib().danhip_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "danhip needs contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
