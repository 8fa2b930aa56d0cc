"""UDP decode with the reference's signature, running on the GPU.

``get_final_preds(config, batch_heatmaps, center, scale)`` mirrors
deep_hrnet/lib/core/inference.py:149-186 (same argument meaning, same 3-tuple,
same dtypes: float32 coordinates without POST_PROCESS / for the offset head,
float64 with POST_PROCESS; same assertion on ndarray / ndim as :35-37).  It
accepts the reference's host ndarray or a device tensor; the arithmetic is done
by udp_decode_gaussian / udp_decode_offset of libudp_pose_hip.so.  Unlike the
reference it does not mutate ``batch_heatmaps`` (the reference overwrites it
with the blurred maps at :80, which its callers avoid by passing copies).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _cfg(config, *path, default=None):
    cur = config
    try:
        for p in path:
            cur = cur[p] if isinstance(cur, dict) else getattr(cur, p)
        return cur
    except (KeyError, AttributeError):
        if default is None:
            raise
        return default


def gaussian_taps(ksize):
    """fp32 taps of cv2.GaussianBlur(ksize, sigma=0) as the kernels use them."""
    buf = (C.c_float * ksize)()
    _lib.check(_lib.lib().udp_gaussian_taps_host(ksize, buf))
    return np.frombuffer(buf, dtype=np.float32).copy()


def decode_device(heatmaps, center, scale, target_type="gaussian", post_process=False, kpd=4.0,
                  cs_is_f32=True, want_idx=True):
    """Device-resident decode.  heatmaps: cuda fp32 [N,C,H,W]; center/scale: cuda fp64 [N,2].
    Returns cuda tensors (preds f64 [N,J,2], maxvals f32 [N,J,1], preds_in f64 [N,J,2], idx i32 [N,J])."""
    if not heatmaps.is_cuda:
        raise RuntimeError("udp-pose_amd has no CPU path: heatmaps must live on the GPU")
    if heatmaps.dim() != 4:
        raise AssertionError("batch_images should be 4-ndim")
    heatmaps = heatmaps.contiguous()
    n, c, h, w = heatmaps.shape
    dev = heatmaps.device
    offset = target_type == "offset"
    if target_type not in ("gaussian", "offset"):
        raise ValueError("TARGET_TYPE %r" % (target_type,))
    if offset and c % 3:
        raise ValueError("offset head needs 3 channels per joint, got %d" % c)
    j = c // 3 if offset else c
    center = center.to(device=dev, dtype=torch.float64).contiguous()
    scale = scale.to(device=dev, dtype=torch.float64).contiguous()
    if tuple(center.shape) != (n, 2) or tuple(scale.shape) != (n, 2):
        raise ValueError("center/scale must be [N,2]")
    preds = torch.empty(n, j, 2, dtype=torch.float64, device=dev)
    pin = torch.empty(n, j, 2, dtype=torch.float64, device=dev)
    maxvals = torch.empty(n, j, 1, dtype=torch.float32, device=dev)
    idx = torch.empty(n, j, dtype=torch.int32, device=dev) if want_idx else None
    lib = _lib.lib()
    if offset:
        _lib.check(lib.udp_decode_offset(_lib.ptr(heatmaps), n, j, h, w, _lib.ptr(center), _lib.ptr(scale),
                                         int(cs_is_f32), float(kpd), _lib.ptr(preds), _lib.ptr(maxvals),
                                         _lib.ptr(pin), _lib.ptr(idx), _lib.stream_ptr()))
    else:
        _lib.check(lib.udp_decode_gaussian(_lib.ptr(heatmaps), n, j, h, w, _lib.ptr(center), _lib.ptr(scale),
                                           int(cs_is_f32), int(bool(post_process)), _lib.ptr(preds),
                                           _lib.ptr(maxvals), _lib.ptr(pin), _lib.ptr(idx), _lib.stream_ptr()))
    return preds, maxvals, pin, idx


def get_final_preds(config, batch_heatmaps, center, scale, return_idx=False):
    """inference.py:149-186.  Returns (preds, maxvals, preds_in_input_space) as host ndarrays."""
    target_type = _cfg(config, "MODEL", "TARGET_TYPE")
    post = bool(_cfg(config, "TEST", "POST_PROCESS", default=False)) and target_type == "gaussian"
    kpd = float(_cfg(config, "LOSS", "KPD", default=4.0))
    if isinstance(batch_heatmaps, torch.Tensor):
        hm = batch_heatmaps
    else:
        assert isinstance(batch_heatmaps, np.ndarray), "batch_heatmaps should be numpy.ndarray"
        assert batch_heatmaps.ndim == 4, "batch_images should be 4-ndim"
        hm = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32))
    if not hm.is_cuda:
        hm = hm.cuda()
    c_np = center.detach().cpu().numpy() if isinstance(center, torch.Tensor) else np.asarray(center)
    s_np = scale.detach().cpu().numpy() if isinstance(scale, torch.Tensor) else np.asarray(scale)
    cs_is_f32 = c_np.dtype == np.float32 and s_np.dtype == np.float32
    c_t = torch.from_numpy(np.ascontiguousarray(c_np, dtype=np.float64))
    s_t = torch.from_numpy(np.ascontiguousarray(s_np, dtype=np.float64))
    preds, maxvals, pin, idx = decode_device(hm, c_t, s_t, target_type, post, kpd, cs_is_f32)
    out_dtype = np.float64 if post else np.float32     # dtype the reference's arrays end up with
    res = (preds.cpu().numpy().astype(out_dtype), maxvals.cpu().numpy(), pin.cpu().numpy().astype(out_dtype))
    return res + (idx.cpu().numpy(),) if return_idx else res
