"""Split-fp16 ("f16x2", UDP_F16X2 of include/udp_pose_hip.h) tensors on the host side.

A value x is stored as two fp16 numbers, ``x ~= hi + lo * 2**-11`` with ``hi = fp16(x)`` and
``lo = fp16((x - hi) * 2**11)`` (22 significant bits); an NHWC tensor ``[..., C]`` becomes the fp16
tensor ``[..., 2, C]`` (per pixel the C hi values, then the C lo values).  These helpers exist for tests
and tools: the forward itself never leaves the device format.
"""
import math

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


def weight_exponent(wp):
    """``udp_conv_op.wexp`` of a weight tensor: the power of two that puts its largest magnitude into
    [2**13, 2**14) (0 for an all-zero tensor, clamped to +-40)."""
    m = float(wp.abs().max()) if wp.numel() else 0.0
    if not math.isfinite(m):
        raise ValueError("f16x2 storage: non-finite weight")
    if m == 0.0:
        return 0
    return max(-40, min(40, 13 - math.frexp(m)[1] + 1))        # frexp: m = f * 2**e with 0.5 <= f < 1


def pack_weights_ws(wp):
    """[taps][cout_pad][cin] fp32 -> ``(bytes, wexp)``: the fragment-major split-fp16 layout of ``udp_conv_op.wfmt == 1``
    (include/udp_pose_hip.h) -- uint8 bytes, 1 KiB blocks [tap][cin chunk][cout pair][nb][plane], a block = 64 lanes
    x 8 fp16 in MFMA A-operand order, so a wave of conv_ws_h2_kernel fetches a fragment with ONE contiguous
    16-bytes-per-lane load -- of the weights scaled by ``2**wexp`` (``weight_exponent``): plane hi = fp16(w'),
    plane lo = fp16(w' - hi), the plain residual (w' is large enough that it stays a normal fp16 number for every
    weight above 2**-16 of the largest).  cout_pad must be a multiple of 32; cin is zero-padded to a multiple of 32."""
    taps, cp, cin = wp.shape
    if cp % 32:
        raise ValueError("cout_pad %d is not a multiple of 32" % cp)
    nch = (cin + 31) // 32
    wexp = weight_exponent(wp)
    w = torch.zeros(taps, cp, nch * 32, dtype=torch.float32)
    w[:, :, :cin] = torch.ldexp(wp.to(torch.float32), torch.tensor(wexp))
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    pl = torch.stack([hi, lo], dim=-2).permute(2, 0, 1, 3)      # [plane, tap, cout, k]
    pl = pl.reshape(2, taps, cp // 32, 4, 2, 4, nch, 4, 8)      # cout = 32*pair + 8*a + 4*nb + b; k = 32*c + 8*kg + j
    out = pl.permute(1, 6, 2, 4, 0, 7, 3, 5, 8).contiguous()    # tap, c, pair, nb, plane, kg, a, b, j  (lane = kg*16 + a*4 + b)
    return out.view(torch.uint8).reshape(-1), wexp
