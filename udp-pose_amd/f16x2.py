"""Split-fp16 ("f16x2", UDP_F16X2 of include/udp_pose_hip.h) tensors on the host side.

A value x is stored as two fp16 numbers, ``x ~= hi + lo * 2**-11`` with ``hi = fp16(x)`` and
``lo = fp16((x - hi) * 2**11)`` (22 significant bits); an NHWC tensor ``[..., C]`` becomes the fp16
tensor ``[..., 2, C]`` (per pixel the C hi values, then the C lo values).  These helpers exist for tests
and tools: the forward itself never leaves the device format.
"""
import torch

LO_SCALE = 2048.0


def encode(x):
    """fp32 ``[..., C]`` -> fp16 ``[..., 2, C]``."""
    x = x.to(torch.float32)
    hi = x.to(torch.float16)
    lo = ((x - hi.to(torch.float32)) * LO_SCALE).to(torch.float16)
    return torch.stack([hi, lo], dim=-2).contiguous()


def decode(t):
    """fp16 ``[..., 2, C]`` -> fp32 ``[..., C]``."""
    return t[..., 0, :].to(torch.float32) + t[..., 1, :].to(torch.float32) * (1.0 / LO_SCALE)


def pack_weights_ws(wp):
    """[taps][cout_pad][cin] fp32 -> the fragment-major split-fp16 layout of ``udp_conv_op.wfmt == 1``
    (include/udp_pose_hip.h): uint8 bytes, 1 KiB blocks [tap][cin chunk][cout pair][nb][plane], a block = 64 lanes
    x 8 fp16 in MFMA A-operand order, so a wave of conv_ws_h2_kernel fetches a fragment with ONE contiguous
    16-bytes-per-lane load.  cout_pad must be a multiple of 32; cin is zero-padded to a multiple of 32."""
    taps, cp, cin = wp.shape
    if cp % 32:
        raise ValueError("cout_pad %d is not a multiple of 32" % cp)
    nch = (cin + 31) // 32
    w = torch.zeros(taps, cp, nch * 32, dtype=torch.float32)
    w[:, :, :cin] = wp.to(torch.float32)
    pl = encode(w).permute(2, 0, 1, 3)                          # [plane, tap, cout, k]
    pl = pl.reshape(2, taps, cp // 32, 4, 2, 4, nch, 4, 8)      # cout = 32*pair + 8*a + 4*nb + b; k = 32*c + 8*kg + j
    out = pl.permute(1, 6, 2, 4, 0, 7, 3, 5, 8).contiguous()    # tap, c, pair, nb, plane, kg, a, b, j  (lane = kg*16 + a*4 + b)
    return out.view(torch.uint8).reshape(-1)
