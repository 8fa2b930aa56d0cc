"""ctypes binding of libudp_pose_hip.so (include/udp_pose_hip.h).

The library is built in-tree (``__graft_entry__.build()`` or
``make -C udp-pose_amd/csrc``).  There is no CPU fallback: if the library is
missing, ``lib()`` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UDP_POSE_LIB") or os.path.join(_HERE, "libudp_pose_hip.so")   # UDP_POSE_LIB: diagnostic builds

UDP_OK = 0
UDP_F32, UDP_BF16, UDP_F16X2 = 0, 1, 2
DTYPES = {"f32": UDP_F32, "bf16": UDP_BF16, "f16x2": UDP_F16X2}
UDP_OP_STEM, UDP_OP_CONV, UDP_OP_FUSE, UDP_OP_STEM7, UDP_OP_MAXPOOL, UDP_OP_BILINEAR = 0, 1, 2, 3, 4, 5
UDP_OP_PSA_POOL, UDP_OP_PSA_MLP, UDP_OP_PSA_SCALE, UDP_OP_PSA_SP, UDP_OP_BLOCK = 6, 7, 8, 9, 10
UDP_BUF_NONE, UDP_BUF_OUTPUT = -1, -2
ABI_VERSION = 19
MAX_LANES, MAX_WAIT = 4, 8


class BnItem(C.Structure):
    """struct udp_bn_item (include/udp_pose_hip.h): one tensor of a multi-tensor BatchNorm call."""
    _fields_ = [(k, C.c_void_p) for k in ("x", "dy", "y_relu", "res", "y", "dx", "g_out", "gamma", "beta", "running_mean",
                                          "running_var", "save_mean", "save_invstd", "dgamma", "dbeta", "ws")] + [
        ("m", C.c_int64), ("c", C.c_int32), ("rows", C.c_int32), ("relu", C.c_int32), ("reserved", C.c_int32)]


class WgradItem(C.Structure):
    """struct udp_wgrad_item (include/udp_pose_hip.h): one member of udp_conv2d_wgrad_group."""
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)] + [(k, C.c_int32) for k in (
                    "n", "hin", "win", "cin_k", "hout", "wout", "cout_k", "ks", "stride", "cout", "cin", "accumulate")]


class ConvItem(C.Structure):
    """struct udp_conv_item (include/udp_pose_hip.h): one member of udp_conv2d_fused_group."""
    _fields_ = [("op", C.c_void_p), ("inp", C.c_void_p), ("weights", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("out", C.c_void_p), ("bn_ws", C.c_void_p), ("bn_ws_doubles", C.c_size_t), ("bn_rows", C.c_int32),
                ("reserved", C.c_int32)]


class ConvOp(C.Structure):
    """struct udp_conv_op (include/udp_pose_hip.h)."""
    _fields_ = [
        ("kind", C.c_int32), ("ks", C.c_int32), ("stride", C.c_int32), ("relu", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32),
        ("hin", C.c_int32), ("win", C.c_int32), ("hout", C.c_int32), ("wout", C.c_int32),
        ("in_buf", C.c_int32), ("out_buf", C.c_int32), ("res_buf", C.c_int32),
        ("n_up", C.c_int32), ("up_buf", C.c_int32 * 3), ("up_shift", C.c_int32 * 3),
        ("w_off", C.c_int64), ("b_off", C.c_int64),
        ("in_coff", C.c_int32), ("in_pitch", C.c_int32), ("out_coff", C.c_int32), ("out_pitch", C.c_int32),
        ("res_coff", C.c_int32), ("res_pitch", C.c_int32),
        ("lane", C.c_int32), ("n_wait", C.c_int32), ("wait_op", C.c_int32 * 8),
        ("w2_off", C.c_int64), ("b2_off", C.c_int64),
        ("group", C.c_int32), ("wfmt", C.c_int32), ("wexp", C.c_int32), ("in_stuff2", C.c_int32),
        ("n_out2", C.c_int32), ("out2_buf", C.c_int32 * 2), ("out2_coff", C.c_int32 * 2), ("out2_pitch", C.c_int32 * 2),
        ("add2_buf", C.c_int32 * 2), ("add2_coff", C.c_int32 * 2), ("add2_pitch", C.c_int32 * 2),
        ("chain_cout", C.c_int32), ("chain_buf", C.c_int32), ("chain_relu", C.c_int32), ("chain_wexp", C.c_int32),
    ]


class UdpPoseError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("udp_pose_hip error %d: %s" % (code, msg))
        self.code = code


class PackDesc(C.Structure):
    """struct udp_pack_desc."""
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("cout", C.c_int32), ("cin", C.c_int32), ("ks", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
_SIGS = {
    "udp_abi_version": (C.c_int, []),
    "udp_f16x2_overflow": (C.c_int, [C.c_void_p, C.c_int]),
    "udp_last_error": (C.c_char_p, []),
    "udp_hrnet_create": (C.c_int, [C.POINTER(ConvOp), C.c_int, C.POINTER(C.c_int64), C.c_int, _P,
                                   C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "udp_hrnet_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "udp_hrnet_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, C.c_int, _P]),
    "udp_hrnet_profile": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, C.POINTER(C.c_float), _P]),
    "udp_hrnet_destroy": (C.c_int, [_P]),
    "udp_hrnet_num_launches": (C.c_int, [_P]),
    "udp_hrnet_lanes": (C.c_int, [_P, C.c_int]),
    "udp_hrnet_flops_per_image": (C.c_double, [_P]),
    "udp_conv2d_fused": (C.c_int, [C.POINTER(ConvOp), C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "udp_conv2d_fused_bn": (C.c_int, [C.POINTER(ConvOp), C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_size_t,
                                      C.POINTER(C.c_int), _P]),
    "udp_flip_fuse": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "udp_flip_fuse_scaled": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "udp_decode_gaussian": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int,
                                      _P, _P, _P, _P, _P]),
    "udp_decode_offset": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_float,
                                    _P, _P, _P, _P, _P]),
    "udp_gaussian_taps_host": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "udp_warp_affine": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "udp_warp_affine_ex": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, _P, _P]),
    "udp_aid_apply": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int,
                                C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "udp_target_gaussian": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_float, _P, _P, _P]),
    "udp_target_offset": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float, _P, _P, _P]),
    "udp_mse_loss": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "udp_oks_nms": (C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_int32), C.c_int, C.c_int, _P, C.c_double, C.c_int,
                              C.c_double, C.c_int, C.c_double, C.c_int, _P, _P, _P]),
    # training step
    "udp_pack_conv_weights": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "udp_pack_conv_weights_batch": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "udp_zero_stuff2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "udp_conv2d_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "udp_conv2d_wgrad": (C.c_int, [_P, _P] + [C.c_int] * 12 + [_P, C.c_int, _P, C.c_size_t, _P]),
    "udp_bn_workspace_doubles": (C.c_size_t, [C.c_int]),
    "udp_bn_train_fwd": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P,
                                   C.c_int, _P, C.c_int, _P, _P]),
    "udp_bn_rows_max": (C.c_int, []),
    "udp_bn_train_fwd_from_sums": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P,
                                             C.c_int, _P, C.c_int, _P, C.c_int, _P]),
    "udp_bn_train_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "udp_debug_multi_order": (C.c_int, [_P, _P, _P, C.c_int, C.c_uint, _P, _P, _P, _P]),
    "udp_conv2d_wgrad_group": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "udp_conv2d_fused_group": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "udp_bn_train_fwd_multi": (C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_int, _P]),
    "udp_bn_train_bwd_multi": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "udp_ew_accumulate": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, _P]),
    "udp_relu_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P]),
    "udp_upsample_bwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "udp_bias_grad": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, C.c_int, _P]),
    "udp_nchw_to_nhwc": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "udp_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                C.c_float, _P]),
    "udp_adam_coefficients": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_int, _P]),
    "udp_adam_step_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, _P, C.c_float, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """The loaded library.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        if l.udp_abi_version() != ABI_VERSION:
            raise RuntimeError("libudp_pose_hip.so ABI %d != binding %d" % (l.udp_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc):
    if rc != UDP_OK:
        raise UdpPoseError(rc, lib().udp_last_error().decode("utf-8", "replace"))


def f16x2_overflow(reset=True):
    """udp_f16x2_overflow on torch's current stream: True if a split-fp16 store since the last reset left fp16's range."""
    rc = lib().udp_f16x2_overflow(stream_ptr(), int(bool(reset)))
    if rc < 0:
        raise UdpPoseError(rc, lib().udp_last_error().decode("utf-8", "replace"))
    return rc != 0


def ptr(t):
    """Device/host pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
