#!/usr/bin/env python3
"""Where does a conv workgroup spend its cycles?  Runs ONE fused conv launch of the diagnostic
library (libudp_pose_hip_stamps.so, `make -C udp-pose_amd/csrc stamps`) and prints mean s_memtime
deltas between the in-kernel stamps:
  0 start | 1 prologue done | 2 chunk-0 DMA issued | 3 chunk-0 landed (vmcnt 0) | 4 barrier passed
  5 MFMA loop done | 6 epilogue issued | 7 stores drained
    python tools/stamp_conv.py [cin cout h w ks stride n dtype]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from udp_pose_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "udp-pose_amd", "libudp_pose_hip_stamps.so")


def main():
    a = sys.argv[1:]
    cin, cout, h, w, ks, st, n = [int(x) for x in (a[:7] if len(a) >= 7 else (32, 32, 64, 48, 3, 1, 128))]
    dtype = a[7] if len(a) > 7 else "bf16"
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    h2 = dtype == "f16x2"
    if h2:
        tdt = torch.float32       # 4 bytes per element; the hi/lo bit patterns are filled below
    lib = _lib.lib()
    lib.udp_debug_set_stamps.argtypes = [C.c_void_p]
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // st + 1, (w + 2 * pad - ks) // st + 1
    from udp_pose_amd import f16x2
    enc = (lambda t: f16x2.encode(t)) if h2 else (lambda t: t.to(tdt))
    x = enc(torch.randn(n, h, w, cin, device="cuda"))
    wt = enc(torch.randn(ks * ks, (cout + 31) // 32 * 32, cin, device="cuda") * 0.05)
    b = torch.zeros((cout + 31) // 32 * 32, device="cuda")
    res = enc(torch.randn(n, ho, wo, cout, device="cuda"))
    out = torch.empty(n, ho, wo, cout, device="cuda", dtype=tdt)
    op = _lib.ConvOp()
    op.kind, op.ks, op.stride, op.relu = _lib.UDP_OP_CONV, ks, st, 1
    op.cin, op.cout, op.cout_pad = cin, cout, (cout + 31) // 32 * 32
    op.hin, op.win, op.hout, op.wout = h, w, ho, wo
    nslots = 65536 * 4 * 16
    stamps = torch.zeros(nslots, dtype=torch.int64, device="cuda")

    def run():
        _lib.check(lib.udp_conv2d_fused(C.byref(op), _lib.DTYPES[dtype], n, _lib.ptr(x),
                                        _lib.ptr(wt), _lib.ptr(b), _lib.ptr(res), None, None, None, _lib.ptr(out),
                                        _lib.stream_ptr()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _lib.check(lib.udp_debug_set_stamps(C.c_void_p(stamps.data_ptr())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 16)[:, :8]
    s = s[s[:, 0] != 0]
    print("%d waves, launch %.1f us (with stamps)" % (len(s), e0.elapsed_time(e1) * 1e3))
    names = ["prologue", "issue DMA c0", "wait DMA c0", "barrier", "MFMA loop (+later chunks)", "epilogue issue", "store drain"]
    d = np.diff(s.astype(np.float64), axis=1)
    for k, nm in enumerate(names):
        print("  %-28s mean %8.0f  p10 %8.0f  p90 %8.0f cycles" % (nm, d[:, k].mean(), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
    life = s[:, 7] - s[:, 0]
    print("  wave lifetime mean %.0f cycles; kernel span (first start -> last end) %.0f cycles" % (life.mean(), s[:, 7].max() - s[:, 0].min()))
    st0 = np.sort(s[:, 0] - s[:, 0].min())
    print("  wave start offsets: p25 %.0f p50 %.0f p75 %.0f p100 %.0f" % tuple(np.percentile(st0, [25, 50, 75, 100])))


if __name__ == "__main__":
    main()
