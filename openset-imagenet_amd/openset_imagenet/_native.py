"""ctypes binding of libosi_hip.so (C ABI declared in include/osi.h).

The library is the product: there is no CPU or eager-PyTorch fallback. Importing this module never needs a GPU
(the .so only links the HIP runtime), but every compute entry point requires device pointers on an MI355X, and
`lib()` raises if the shared object has not been built (`python __graft_entry__.py` / `make -C csrc`).
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.path.join(CSRC_DIR, "libosi_hip.so")
TORCH_LIB_PATH = os.path.join(CSRC_DIR, "libosi_torch.so")   # TORCH_LIBRARY(osi, ...) registration on top of the C ABI

OSI_OK = 0
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x128, TILE_64x64 = 0, 1, 2, 3, 4
LOSS_ENTROPIC, LOSS_SOFTMAX, LOSS_GARBAGE = 0, 1, 2


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "R", "S", "stride", "pad")]

    @classmethod
    def make(cls, B, H, W, Cin, Cout, k, stride, pad):
        Ho = (H + 2 * pad - k) // stride + 1
        Wo = (W + 2 * pad - k) // stride + 1
        return cls(B, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad)


class ConvEpilogue(Structure):
    """osi_conv_epilogue: out = [relu](conv * scale[n] + shift[n] [+ residual])"""
    _fields_ = [("scale", c_void_p), ("shift", c_void_p), ("residual", c_void_p), ("relu", c_int)]


class BnEvalLayer(Structure):
    """osi_bn_eval_layer"""
    _fields_ = [("running_mean", c_void_p), ("running_var", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("scale", c_void_p),
                ("shift", c_void_p), ("C", c_int)]


P = c_void_p
_PD = POINTER(ConvDesc)

# name -> (restype, argtypes). Must list every symbol of include/osi.h (tests/test_abi.py checks the header against it).
_SIGS = {
    "osi_abi_version": (c_int, []),
    "osi_build_arch": (c_char_p, []),
    "osi_strerror": (c_char_p, [c_int]),
    "osi_set_tuning": (c_int, [c_char_p, c_int]),
    "osi_get_tuning": (c_int, [c_char_p, POINTER(c_int)]),
    "osi_conv_fwd": (c_int, [_PD, P, P, P, c_int, P]),
    "osi_conv_fwd_act": (c_int, [_PD, P, P, P, P, P, c_int, P, c_size_t, POINTER(c_int), POINTER(c_int), P]),
    "osi_conv_fwd_act2": (c_int, [_PD, P, P, P, P, P, P, c_int, P, c_size_t, POINTER(c_int), POINTER(c_int), P]),
    "osi_conv_fwd_epilogue_workspace": (c_size_t, [_PD]),
    "osi_conv_fwd_epilogue": (c_int, [_PD, P, P, P, POINTER(ConvEpilogue), P, c_size_t, P]),
    "osi_conv_fwd_wino_epilogue_pre": (c_int, [_PD, P, P, P, POINTER(ConvEpilogue), P, c_size_t, P]),
    "osi_bn_eval_coeffs_multi": (c_int, [POINTER(BnEvalLayer), c_int, c_float, P]),
    "osi_conv_wgrad_act": (c_int, [_PD, P, P, P, P, P, P, c_size_t, P]),
    "osi_conv_wino_eligible": (c_int, [_PD, c_int]),
    "osi_conv_wino_workspace": (c_size_t, [_PD]),
    "osi_conv_fwd_wino": (c_int, [_PD, P, P, P, P, P, P, c_size_t, P, c_size_t, POINTER(c_int), POINTER(c_int), P]),
    "osi_conv_dgrad_fused_wino": (c_int, [_PD, P, P, P, P, P, c_size_t, POINTER(c_int), P]),
    "osi_conv_wino_weights_bytes": (c_size_t, [_PD]),
    "osi_conv_wino_slab_bytes": (c_size_t, []),
    "osi_conv_wino_transform_weights": (c_int, [_PD, P, c_int, P, c_size_t, P]),
    "osi_conv_fwd_wino_pre": (c_int, [_PD, P, P, P, P, P, P, c_size_t, P, c_size_t, POINTER(c_int), POINTER(c_int), P]),
    "osi_conv_dgrad_fused_wino_pre": (c_int, [_PD, P, P, P, P, P, c_size_t, POINTER(c_int), P]),
    "osi_conv_wgrad_wino_workspace": (c_size_t, [_PD]),
    "osi_conv_wgrad_wino": (c_int, [_PD, P, P, P, P, P, P, c_size_t, P]),
    "osi_conv_fwd_bnstats_workspace": (c_size_t, [_PD]),
    "osi_conv_fwd_bnstats": (c_int, [_PD, P, P, P, c_int, P, c_size_t, POINTER(c_int), POINTER(c_int), P]),
    "osi_bn_finalize_stats": (c_int, [P, c_size_t, c_int, c_int, c_int, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P]),
    "osi_conv_dgrad": (c_int, [_PD, P, P, P, c_int, c_int, P]),
    "osi_conv_dgrad_fused_workspace": (c_size_t, [_PD]),
    "osi_conv_dgrad_fused": (c_int, [_PD, P, P, P, P, P, c_int, POINTER(c_int), P]),
    "osi_bn_backward_reduce": (c_int, [P, P, c_int, P, P, c_int, c_int, P, c_size_t, P]),
    "osi_bn_backward_fused": (c_int, [P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, P, c_size_t, P]),
    "osi_conv_wgrad_workspace": (c_size_t, [_PD]),
    "osi_conv_wgrad": (c_int, [_PD, P, P, P, P, c_size_t, P]),
    "osi_stem_weight_pack": (c_int, [P, P, c_int, P]),
    "osi_stem_wgrad_direct_workspace": (c_size_t, [_PD]),
    "osi_stem_wgrad_direct": (c_int, [_PD, P, P, P, P, c_size_t, P]),
    "osi_stem_wgrad_fused_workspace": (c_size_t, [_PD]),
    "osi_stem_wgrad_fused": (c_int, [_PD, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "osi_stem_grad_unpack": (c_int, [P, P, c_int, P]),
    "osi_bn_workspace": (c_size_t, [c_int, c_int]),
    "osi_bn_train_stats": (c_int, [P, c_int, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P, c_size_t, P]),
    "osi_bn_eval_coeffs": (c_int, [P, P, P, P, c_float, c_int, P, P, P]),
    "osi_bn_apply": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "osi_bn_backward_workspace": (c_size_t, [c_int, c_int]),
    "osi_bn_backward": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, c_size_t, P]),
    "osi_bn_relu_mask_bytes": (c_size_t, [c_int, c_int]),
    "osi_bn_apply_relu_mask": (c_int, [P, P, P, P, P, P, c_int, c_int, P]),
    "osi_bn_apply_relu_mask2": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "osi_bn_backward_relu_mask": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, c_size_t, P]),
    "osi_nchw3_to_nhwc4": (c_int, [P, P, c_int, c_int, c_int, P]),
    "osi_u8hwc3_to_nhwc4": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "osi_u8_crop_flip_to_nhwc4": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "osi_resnet50_stage_input_u8": (c_int, [c_void_p, P, P, P, P]),
    "osi_resnet50_bind_input_nhwc4": (c_int, [c_void_p, P]),
    "osi_maxpool3x3s2_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "osi_maxpool3x3s2_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "osi_bn_relu_maxpool_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "osi_bn_relu_maxpool_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "osi_avgpool_fwd": (c_int, [P, P, c_int, c_int, c_int, P]),
    "osi_avgpool_bwd": (c_int, [P, P, c_int, c_int, c_int, P]),
    "osi_linear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "osi_linear_bwd": (c_int, [P, P, P, P, c_int, P, P, c_int, c_int, c_int, P]),
    "osi_loss_fwd_bwd": (c_int, [c_int, P, P, c_int, c_int, c_float, c_longlong, P, P, c_int, c_float, c_float, P, P, P, P]),
    "osi_softmax": (c_int, [P, P, c_int, c_int, P]),
    "osi_confidence_accumulate": (c_int, [P, P, c_int, c_int, c_float, c_longlong, c_int, P, P]),
    "osi_confidence_from_scores": (c_int, [P, P, c_int, c_int, c_float, c_longlong, c_int, P, P]),
    "osi_oscr_workspace": (c_size_t, [c_int]),
    "osi_oscr_f32": (c_int, [P, P, c_int, c_int, c_longlong, P, c_size_t, P, P, P, P, P]),
    "osi_oscr_f64": (c_int, [P, P, c_int, c_int, c_longlong, P, c_size_t, P, P, P, P, P]),
    "osi_adam_step": (c_int, [P, P, P, P, c_size_t, c_double, c_double, c_double, c_double, c_longlong, c_float, P]),
    "osi_sgd_step": (c_int, [P, P, P, c_size_t, c_float, c_float, c_int, c_float, P]),
    "osi_fill_f32": (c_int, [P, c_size_t, c_float, P]),
    "osi_scale_f32": (c_int, [P, c_size_t, c_float, P]),
    "osi_i64_add": (c_int, [P, c_int, c_longlong, P]),
    "osi_resnet50_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int]),
    "osi_resnet50_destroy": (None, [c_void_p]),
    "osi_resnet50_num_tensors": (c_int, [c_void_p]),
    "osi_resnet50_tensor_info": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t)]),
    "osi_resnet50_param_floats": (c_size_t, [c_void_p]),
    "osi_resnet50_num_bn": (c_int, [c_void_p]),
    "osi_resnet50_bn_info": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t)]),
    "osi_resnet50_buffer_floats": (c_size_t, [c_void_p]),
    "osi_resnet50_workspace_bytes": (c_size_t, [c_void_p]),
    "osi_resnet50_num_stages": (c_int, [c_void_p]),
    "osi_resnet50_geometry": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "osi_resnet50_stage_grad_range": (c_int, [c_void_p, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "osi_resnet50_grads_ready": (c_int, [c_void_p, P, P]),
    "osi_resnet50_set_overlap": (c_int, [c_void_p, c_int]),
    "osi_resnet50_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "osi_resnet50_profile": (c_int, [c_void_p, c_int]),
    "osi_resnet50_profile_read": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(c_int)]),
    "osi_resnet50_timeline_read": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(c_int), POINTER(c_int), c_int, POINTER(c_int)]),
    "osi_resnet50_debug_num_gates": (c_int, [c_void_p]),
    "osi_resnet50_debug_gate_shape": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "osi_resnet50_debug_gate": (c_int, [c_void_p, P, c_int, P, P, P]),
    "osi_resnet50_forward": (c_int, [c_void_p, P, P, P, P, P, P, P, c_int, P]),
    "osi_resnet50_backward": (c_int, [c_void_p, P, P, P, P, P, c_int, c_int, P]),
}

_lib = None
DIAGNOSTIC_LIB = None     # path of a diagnostic build selected with OSI_HIP_LIB + OSI_DEV=1 (never in production)


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded shared library; raises NativeLibraryMissing when it has not been built."""
    global _lib, LIB_PATH
    if _lib is None:
        # dev only: a diagnostic build of the same C ABI (csrc `make stamps / diag / ablate`: some of them compute WRONG results by
        # design). Honoured only together with OSI_DEV=1, announced loudly, and refused by train.worker() (DIAGNOSTIC_LIB).
        global DIAGNOSTIC_LIB
        if os.environ.get("OSI_HIP_LIB"):
            if os.environ.get("OSI_DEV") != "1":
                raise NativeLibraryMissing("OSI_HIP_LIB selects a diagnostic build of the native library and is only honoured with OSI_DEV=1")
            LIB_PATH = os.environ["OSI_HIP_LIB"]
            DIAGNOSTIC_LIB = LIB_PATH
            import warnings
            warnings.warn(f"openset_imagenet: DIAGNOSTIC native library {LIB_PATH} (OSI_HIP_LIB + OSI_DEV=1): results may be wrong by design",
                          RuntimeWarning, stacklevel=2)
        if not os.path.isfile(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build the gfx950 HIP library first (python __graft_entry__.py, or make -C {CSRC_DIR}). "
                "There is no CPU fallback for the training hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError here = ABI drift, loud on purpose
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        # development A/B switches: the environment is read HERE, once, and handed to the library explicitly
        for env, knob in (("OSI_WGRAD_TILE", b"wgrad_tile"), ("OSI_WGRAD_BLOCKS", b"wgrad_blocks"), ("OSI_WGRAD_NST", b"wgrad_nst"),
                          ("OSI_WGRAD_GROUP", b"wgrad_group"), ("OSI_BN_GRID", b"bn_grid"), ("OSI_BN_GRID_BWD", b"bn_grid_bwd"), ("OSI_BN_SINGLE_P", b"bn_single_p"), ("OSI_BN_WIDE_P", b"bn_wide_p"), ("OSI_TAIL_GAIN", b"tail_gain"), ("OSI_TAIL_QMAX", b"tail_qmax"), ("OSI_WGRAD3", b"wgrad3"), ("OSI_WGRAD3_BLOCKS", b"wgrad3_blocks"), ("OSI_FWD_WIDE", b"fwd_wide"), ("OSI_DGRAD_WIDE", b"dgrad_wide"), ("OSI_TAIL_SPLIT", b"tail_split"), ("OSI_TAIL_CUS", b"tail_cus"), ("OSI_TAIL_SMAX", b"tail_smax"), ("OSI_TAIL_MINT", b"tail_mint"), ("OSI_STEM_DIRECT", b"stem_direct"), ("OSI_DP_RESERVED_CUS", b"dp_reserved_cus"), ("OSI_FWD_ROWS", b"fwd_rows"), ("OSI_FWD_W3", b"fwd_w3"), ("OSI_DGRAD_W3", b"dgrad_w3"), ("OSI_FWD_WINO", b"fwd_wino"), ("OSI_DGRAD_WINO", b"dgrad_wino"), ("OSI_WINO_STREAMK", b"wino_streamk"), ("OSI_WGRAD_WINO", b"wgrad_wino"), ("OSI_WINO_WIDE", b"wino_wide")):
            if os.environ.get(env):
                check(handle.osi_set_tuning(knob, int(os.environ[env])), f"osi_set_tuning({knob.decode()})")
    return _lib


_ops = None


def ops():
    """`torch.ops.osi` — the PyTorch-ROCm custom ops the Python package calls on the hot path (csrc/osi_torch_ops.cpp). Loading the
    registration library also loads libosi_hip.so (rpath $ORIGIN); raises NativeLibraryMissing when it has not been built."""
    global _ops
    if _ops is None:
        lib()   # the C-ABI library first: same handle for ctypes and for the op library, and the loud failure if it is missing
        if not os.path.isfile(TORCH_LIB_PATH):
            raise NativeLibraryMissing(f"{TORCH_LIB_PATH} not found: build it first (python __graft_entry__.py, or make -C {CSRC_DIR}).")
        import torch
        torch.ops.load_library(TORCH_LIB_PATH)
        _ops = torch.ops.osi
    return _ops


def declared_symbols():
    return sorted(_SIGS)


def check(code, what=""):
    if code != OSI_OK:
        msg = lib().osi_strerror(code).decode()
        raise RuntimeError(f"libosi_hip {what} failed: {msg} (code {code})")


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def stream_of(t):
    """The current HIP stream of the tensor's device as an integer handle."""
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def require_gpu_f32(*tensors):
    import torch
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("openset_imagenet (MI355X build): tensors must live on the GPU — there is no CPU fallback "
                               "(set_device_gpu(index) / model.to('cuda')).")
        if t.dtype != torch.float32 and t.dtype != torch.int64:
            raise RuntimeError(f"unsupported dtype {t.dtype}: the hot path is fp32 (labels int64)")
