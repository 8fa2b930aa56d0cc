"""Oracle: flip-test fuse (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates deep_hrnet/lib/utils/transforms.py:15-29 (flip_back), :31-47
(flip_back_offset) and the fuse in deep_hrnet/lib/core/function.py:151-171.
"""
import numpy as np

COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
MPII_FLIP_PAIRS = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]


def channel_permutation(num_channels, pairs, is_offset):
    """Source channel and sign for every output channel of flip_back*.

    gaussian: out[c] = in[partner(c)].  offset: channels are (hm,ox,oy)
    triplets per joint; joints are swapped as triplets and the x-offset
    channel (3j+1) is negated (transforms.py:40)."""
    src = np.arange(num_channels)
    sign = np.ones(num_channels, dtype=np.float32)
    if is_offset:
        nj = num_channels // 3
        jsrc = np.arange(nj)
        for a, b in pairs:
            jsrc[a], jsrc[b] = b, a
        for j in range(nj):
            for k in range(3):
                src[3 * j + k] = 3 * jsrc[j] + k
            sign[3 * j + 1] = -1.0
    else:
        for a, b in pairs:
            src[a], src[b] = b, a
    return src, sign


def flip_back(output_flipped, pairs):
    """transforms.py:15-29: reverse W, swap left/right joint channels."""
    assert output_flipped.ndim == 4
    src, _ = channel_permutation(output_flipped.shape[1], pairs, False)
    return np.ascontiguousarray(output_flipped[:, src, :, ::-1])


def flip_back_offset(output_flipped, pairs):
    """transforms.py:31-47: reverse W, negate x-offsets, swap joint triplets."""
    assert output_flipped.ndim == 4
    src, sign = channel_permutation(output_flipped.shape[1], pairs, True)
    return np.ascontiguousarray(output_flipped[:, src, :, ::-1] * sign[None, :, None, None])


def flip_fuse(output, output_flipped, pairs, is_offset):
    """function.py:161-171: (output + flip_back(output_flipped)) * 0.5 in fp32."""
    fb = flip_back_offset(output_flipped, pairs) if is_offset else flip_back(output_flipped, pairs)
    return ((output + fb) * np.float32(0.5)).astype(np.float32)
