#!/usr/bin/env python3
"""Times single fused-conv launches of the HRNet-W32 layer shapes through udp_conv2d_fused, sweeping
environment knobs of the tile choosers (read at every call).

    python tools/bench_conv.py [--dtype f16x2] [--n 128] [--knob NAME=v1,v2 ...]
"""
import argparse
import ctypes as C
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from udp_pose_amd import _lib  # noqa: E402

SHAPES = [  # ks, stride, cin, cout, hout, wout, residual
    (3, 1, 32, 32, 64, 48, True), (3, 1, 64, 64, 32, 24, True), (3, 1, 128, 128, 16, 12, True),
    (3, 1, 256, 256, 8, 6, True), (3, 1, 64, 64, 64, 48, False), (1, 1, 64, 256, 64, 48, True),
    (1, 1, 256, 64, 64, 48, False), (3, 2, 64, 64, 64, 48, False), (3, 2, 32, 64, 32, 24, True),
    (3, 2, 64, 128, 16, 12, True), (3, 2, 256, 64, 32, 24, False),
]


def run(dtype, n, shape, reps=20):
    ks, st, cin, cout, ho, wo, res = shape
    hi, wi = ho * st, wo * st
    esz = 2 if dtype == "bf16" else 4
    g = torch.Generator("cuda").manual_seed(0)
    mk = lambda nbytes: (torch.randn(nbytes // 2, device="cuda", generator=g) * 0.5).to(torch.float16)
    x = mk(n * hi * wi * cin * esz)
    w = (torch.randn(ks * ks * cout * cin * esz // 2, device="cuda", generator=g) * 0.05).to(torch.float16)
    b = torch.zeros(cout, device="cuda")
    r = mk(n * ho * wo * cout * esz) if res else None
    out = torch.empty(n * ho * wo * cout * esz, dtype=torch.uint8, device="cuda")
    op = _lib.ConvOp()
    op.kind, op.ks, op.stride, op.relu = _lib.UDP_OP_CONV, ks, st, 1
    op.cin, op.cout, op.cout_pad = cin, cout, cout
    op.hin, op.win, op.hout, op.wout = hi, wi, ho, wo
    op.wfmt = int(os.environ.get("UDP_POSE_WS", "1") != "0" and dtype == "f16x2")
    call = lambda: _lib.check(_lib.lib().udp_conv2d_fused(C.byref(op), _lib.DTYPES[dtype], n, _lib.ptr(x), _lib.ptr(w),
                                                          _lib.ptr(b), _lib.ptr(r), None, None, None, _lib.ptr(out),
                                                          _lib.stream_ptr()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tf = 2.0 * ks * ks * cin * cout * ho * wo * n / us / 1e6
    return us, tf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16x2")
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--shapes", default="")
    a = ap.parse_args()
    knobs = [(k.split("=")[0], k.split("=")[1].split(",")) for k in a.knob]
    shapes = SHAPES if not a.shapes else [SHAPES[int(i)] for i in a.shapes.split(",")]
    for shape in shapes:
        print("k%d s%d %d->%d %dx%d res=%d" % shape)
        for combo in itertools.product(*[v for _, v in knobs]) if knobs else [()]:
            for (k, _), v in zip(knobs, combo):
                if v == "-":
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            try:
                us, tf = run(a.dtype, a.n, shape)
                print("    %-50s %8.1f us %8.1f TFLOP/s" % (" ".join("%s=%s" % (k[9:], v) for (k, _), v in zip(knobs, combo)), us, tf))
            except Exception as e:                      # a knob combination the kernels do not cover
                print("    %-50s %s" % (" ".join("%s=%s" % (k[9:], v) for (k, _), v in zip(knobs, combo)), str(e)[:60]))


if __name__ == "__main__":
    main()
