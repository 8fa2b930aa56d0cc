"""Flip-test fuse on the GPU (reference contracts of lib/utils/transforms.py).

``flip_back`` / ``flip_back_offset`` keep the reference's argument meaning
(deep_hrnet/lib/utils/transforms.py:15-29 / :31-47) and ``flip_fuse`` is the
``(output + output_flipped) * 0.5`` of deep_hrnet/lib/core/function.py:161-171,
all through udp_flip_fuse of libudp_pose_hip.so.
"""
import numpy as np
import torch

from . import _lib

COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # lib/dataset/coco.py:91-92
MPII_FLIP_PAIRS = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]                    # lib/dataset/mpii.py:30


def channel_map(num_channels, matched_parts, is_offset):
    """Source channel / sign per output channel of flip_back (gaussian) or
    flip_back_offset (joint triplets swapped, x-offset channel 3j+1 negated)."""
    src = np.arange(num_channels, dtype=np.int32)
    sign = np.ones(num_channels, dtype=np.float32)
    if is_offset:
        if num_channels % 3:
            raise ValueError("offset maps need 3 channels per joint")
        jsrc = np.arange(num_channels // 3)
        for a, b in matched_parts:
            jsrc[a], jsrc[b] = b, a
        for j in range(num_channels // 3):
            src[3 * j:3 * j + 3] = 3 * jsrc[j] + np.arange(3)
            sign[3 * j + 1] = -1.0
    else:
        for a, b in matched_parts:
            src[a], src[b] = b, a
    return src, sign


def flip_fuse(output, output_flipped, matched_parts, is_offset=False, out=None, divisor=1.0):
    """(output + flip_back*(output_flipped)) * 0.5 [/ divisor] on device; cuda fp32 [N,C,H,W].
    divisor=255 gives the RSN test loop's ``outputs/255.0`` (RSN .../test.py:181-185)."""
    if not (output.is_cuda and output_flipped.is_cuda):
        raise RuntimeError("udp-pose_amd has no CPU path: tensors must live on the GPU")
    assert output.dim() == 4, "output_flipped should be [batch_size, num_joints, height, width]"
    output = output.contiguous()
    output_flipped = output_flipped.contiguous()
    n, c, h, w = output.shape
    src, sign = channel_map(c, matched_parts, is_offset)
    src_t = torch.from_numpy(src).to(output.device)
    sign_t = torch.from_numpy(sign).to(output.device)
    if out is None:
        out = torch.empty_like(output)
    _lib.check(_lib.lib().udp_flip_fuse_scaled(_lib.ptr(output), _lib.ptr(output_flipped), _lib.ptr(src_t),
                                               _lib.ptr(sign_t), n, c, h, w, float(divisor), _lib.ptr(out),
                                               _lib.stream_ptr()))
    return out


def _flip_back_any(output_flipped, matched_parts, is_offset):
    host = not isinstance(output_flipped, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(output_flipped, dtype=np.float32)) if host else output_flipped
    t = t.cuda() if not t.is_cuda else t
    # flip_back(x) = 2 * (0.5 * (0 + flip_back(x))): exact in fp32
    r = flip_fuse(torch.zeros_like(t), t, matched_parts, is_offset) * 2.0
    return r.cpu().numpy() if host else r


def flip_back(output_flipped, matched_parts):
    """transforms.py:15-29 (host ndarray in -> host ndarray out, device tensor -> device tensor)."""
    return _flip_back_any(output_flipped, matched_parts, False)


def flip_back_offset(output_flipped, matched_parts):
    """transforms.py:31-47."""
    return _flip_back_any(output_flipped, matched_parts, True)
