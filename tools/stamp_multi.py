#!/usr/bin/env python3
"""Timeline of ONE merged weight-stationary launch (conv_ws_multi<3>) inside the real W32 forward.

Runs the split-fp16 program truncated after the first merged 3x3 group of the given stage with the diagnostic
library (`make -C udp-pose_amd/csrc stamps`), reads the per-wave s_memtime stamps of that last launch and prints,
per member of the group: workgroup lifetimes, when they started / ended inside the launch, and how lifetimes
depend on what the co-resident workgroup of the same CU was doing (members are dispatched deepest first).
    python tools/stamp_multi.py [stage (2|3|4)] [batch]
"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from udp_pose_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "udp-pose_amd", "libudp_pose_hip_stamps_ws.so")
from udp_pose_amd import hrnet_plan  # noqa: E402
import bench  # noqa: E402


def main():
    stage = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    full = hrnet_plan.HRNetProgram.ops_array
    cut = {}

    def truncated(self):
        arr = full(self)
        want = "stage%d" % stage
        k = None
        for i, op in enumerate(self._ops):
            if op["name"].startswith(want) and op.get("group", 0) > 0 and op["ks"] == 3 and op["stride"] == 1:
                g = op["group"] + 1      # the block's second conv (carries the residual)
                k = max(j for j, o in enumerate(self._ops) if o.get("group", 0) == g) + 1
                break
        cut["ops"] = [(o["name"], o["cin"], o["cout"], o["hout"], o["wout"]) for o in self._ops[:k] if o.get("group", 0) == g]
        # the library wants the output op last: keep it (its kernel is not a weight-stationary one and leaves
        # no stamps in the _ws_only build; it reads a buffer nothing wrote in this truncated program)
        keep = list(range(k)) + [len(self._ops) - 1]
        out = (_lib.ConvOp * len(keep))(*[arr[i] for i in keep])
        for i in range(len(keep)):      # one lane: launches strictly in program order, the group's launch is the last
            out[i].lane = 0
            out[i].n_wait = 0
        return out
    hrnet_plan.HRNetProgram.ops_array = truncated
    _, net = bench.build_net("f16x2")
    net.use_graph = False
    hp = bench.HotPath(net, batch, torch.device("cuda", 0), seed=1)
    lib = _lib.lib()
    set_stamps = lib.udp_debug_set_stamps_ws
    set_stamps.argtypes = [C.c_void_p]
    stamps = torch.zeros(8192 * 4 * 16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        try:
            hp.step()
        except Exception as e:          # the truncated program has no output op: the decode after it may complain
            print("step raised", type(e).__name__, e)
    torch.cuda.synchronize()
    _lib.check(set_stamps(C.c_void_p(stamps.data_ptr())))
    try:
        hp.step()
    except Exception as e:
        print("step raised", type(e).__name__, e)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4, 16)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "stamp_multi_stage%d_raw.npy" % stage), s[: 4096])
    print("group members:", cut.get("ops"))
    analyse(s)


def analyse(s, ghz=2.4):
    """Per member of the last launch: where a workgroup's lifetime goes.  s_memtime counts shader cycles and every XCD
    has its own counter (offsets of 1e11 between XCDs), so only differences inside one wave / one XCD are used; a
    stamp slot belongs to the last launch iff it lies between the wave's first and last stamp.  us at `ghz`."""
    s = s.astype(np.float64)
    xcc = (s[:, 0, 8].astype(np.int64) >> 32) & 0xF
    t0 = s[:, 0, 0]
    live = np.zeros(len(s), bool)
    for x in range(8):                               # the newest 170 us on each XCD's own clock = the last launch
        m = (xcc == x) & (t0 > 0)
        if m.any():
            live |= m & (t0 > t0[m].max() - 400000)
    w = s[live] / (ghz * 1e3)
    ok = lambda k: (w[:, 0, k] >= w[:, 0, 0]) & (w[:, 0, k] <= w[:, 0, 7])
    nch = 1 + ok(9) + ok(11) * 2 + ok(13) * 4        # K chunks stamped: 1, 2, 4, 8
    life = w[:, :, 7].max(1) - w[:, :, 0].min(1)
    print("%d workgroups of the last launch (us at %.1f GHz)" % (live.sum(), ghz))
    for k in (1, 2, 4, 8):
        m = nch == k
        if not m.any():
            continue
        v = w[m]
        d = lambda a, b: (v[:, :, b] - v[:, :, a]).mean()
        print(" member with %d K chunks: %d workgroups, life %.1f us | start->DMA issue %.2f  index math + load issue %.2f  "
              "first wait %.2f  barrier %.2f  loop %.2f  epilogue %.2f  store drain %.2f" % (
                  k, m.sum(), life[m].mean(), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7)))
        if k >= 2:
            print("    chunk 0 (9 steps) %.2f | wait + barrier before chunk 1 %.2f | arrival spread of the 4 waves there %.2f" % (
                d(4, 9), d(9, 10), (v[:, :, 9].max(1) - v[:, :, 9].min(1)).mean()))
        if k == 2:
            print("    chunk 1 %.2f" % d(10, 5))
        if k >= 4:
            print("    chunk 1 %.2f | wait + barrier before chunk 2 %.2f | %s %.2f" % (
                d(10, 11), d(11, 12), "chunks 2-3" if k == 4 else "chunks 2-3", d(12, 5) if k == 4 else d(12, 13)))
        if k == 8:
            print("    wait + barrier before chunk 4 %.2f | chunks 4-7 %.2f" % (d(13, 14), d(14, 5)))


if __name__ == "__main__":
    main()
