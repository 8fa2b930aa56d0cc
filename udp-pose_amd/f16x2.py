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
