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
    lib.udp_debug_set_stamps.argtypes = [C.c_void_p]
    stamps = torch.zeros(8192 * 4 * 16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        try:
            hp.step()
        except Exception as e:          # the truncated program has no output op: the decode after it may complain
            print("step raised", type(e).__name__, e)
    torch.cuda.synchronize()
    _lib.check(lib.udp_debug_set_stamps(C.c_void_p(stamps.data_ptr())))
    try:
        hp.step()
    except Exception as e:
        print("step raised", type(e).__name__, e)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4, 16)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "stamp_multi_stage%d_raw.npy" % stage), s[: 4096])
    print("group members:", cut.get("ops"))
    # earlier launches of the truncated program stamped the same buffer: the last launch's entries are the newest
    newest = s[:, 0, 0].max()
    live = s[:, 0, 0] > newest - 30000
    nwg = int(np.argmin(live)) if not live.all() else len(live)
    s = s[:nwg]
    t0 = s[:, :, 0].min()
    start = (s[:, :, 0].min(axis=1) - t0) / 100.0      # s_memtime ticks at 100 MHz -> us
    end = (s[:, :, 7].max(axis=1) - t0) / 100.0
    hw = s[:, 0, 8]
    cu = ((hw >> 32) & 0xF) * 1024 + ((hw >> 13) & 7) * 16 * 2 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
    print("%d workgroups, launch span %.1f us, %d distinct CUs" % (nwg, end.max(), len(set(cu.tolist()))))
    # member boundaries: members are laid out back to back; a member's workgroups have the same number of
    # MFMA steps -> recover the boundaries from the jumps in (stamp5 - stamp4) is fragile; print deciles instead
    life = end - start
    order = np.arange(nwg)
    for lo in range(0, nwg, max(1, nwg // 14)):
        hi = min(nwg, lo + max(1, nwg // 14))
        print("  wg %4d-%4d: start %6.1f..%6.1f us  life mean %5.1f (p10 %5.1f p90 %5.1f)  end max %6.1f" % (
            lo, hi - 1, start[lo:hi].min(), start[lo:hi].max(), life[lo:hi].mean(),
            np.percentile(life[lo:hi], 10), np.percentile(life[lo:hi], 90), end[lo:hi].max()))
    # phase breakdown of a wave (cycles of the 100 MHz counter -> us)
    d = np.diff(s[:, :, :8].astype(np.float64), axis=2) / 100.0
    names = ["prologue", "issue DMA c0", "wait DMA c0", "barrier", "MFMA loop", "epilogue issue", "store drain"]
    for lo in range(0, nwg, max(1, nwg // 7)):
        hi = min(nwg, lo + max(1, nwg // 7))
        print("  wg %4d-%4d: " % (lo, hi - 1) + "  ".join("%s %.1f" % (nm, d[lo:hi, :, k].mean()) for k, nm in enumerate(names)))
    np.save(os.path.join(ROOT, "gpurun_out", "stamp_multi_stage%d.npy" % stage), s)
    # per member (by the number of K chunks a workgroup stamped: chunk-boundary stamps 9/10 (c=1), 11/12 (c=2), 13/14 (c=4))
    f = s.astype(np.float64) / 100.0
    nch = 1 + (s[:, 0, 9] > 0) + (s[:, 0, 11] > 0) * 2 + (s[:, 0, 13] > 0) * 4      # 1, 2, 4, 8 chunks
    for k in (1, 2, 4, 8):
        m = nch == k
        if not m.any():
            continue
        w = f[m]
        print("  member with %d K chunks: %d workgroups, life %.1f us" % (k, m.sum(), life[m].mean()))
        print("     prologue %.2f  dma-issue %.2f  first wait %.2f  barrier %.2f  loop %.2f  epilogue %.2f  drain %.2f" % tuple(
            (w[:, :, j + 1] - w[:, :, j]).mean() for j in range(7)))
        if k >= 2:
            print("     chunk0 compute %.2f us (9 steps)   wait+barrier before chunk1 %.2f" % (
                (w[:, :, 9] - w[:, :, 4]).mean(), (w[:, :, 10] - w[:, :, 9]).mean()))
        if k >= 4:
            print("     chunk1 compute %.2f   wait+barrier before chunk2 %.2f   chunks2-3 compute+waits %.2f" % (
                (w[:, :, 11] - w[:, :, 10]).mean(), (w[:, :, 12] - w[:, :, 11]).mean(), (w[:, :, 5] - w[:, :, 12]).mean() if k == 4 else (w[:, :, 13] - w[:, :, 12]).mean()))
        if k == 8:
            print("     wait+barrier before chunk4 %.2f   chunks4-7 %.2f" % ((w[:, :, 14] - w[:, :, 13]).mean(), (w[:, :, 5] - w[:, :, 14]).mean()))
        # per-wave spread inside a workgroup at the barrier before chunk 1 (who waits for whom)
        if k >= 2:
            arr = w[:, :, 9]
            print("     arrival spread at the chunk-1 barrier (max-min over the 4 waves): mean %.2f us" % (arr.max(1) - arr.min(1)).mean())
    # concurrency per CU
    per = collections.defaultdict(list)
    for i in range(nwg):
        per[int(cu[i])].append((start[i], end[i], i))
    print("workgroups per CU: min %d max %d" % (min(len(v) for v in per.values()), max(len(v) for v in per.values())))


if __name__ == "__main__":
    main()
