#!/usr/bin/env python3
"""Headline benchmark: images/s of the UDP-Pose hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic person crops that
are already resident in HBM: HRNet-W32 256x192 forward on the batch AND on its
W-mirrored copy (flip-test, one 2N-image launch sequence) -> flip fuse -> UDP decode
(DARK / Taylor) -> keypoints [N,17,2] + maxvals on device.  Workload = BASELINE.json
configs[1]: pose_hrnet_w32 256x192, batch 64 per GPU, flip-test on.

The timed mode (`value`, `dtype`) is the PARITY-GRADE one: split-fp16 storage ("f16x2": 22-bit hi+lo
operands, three fp16 MFMAs per product, fp32 accumulate), which holds the north-star contract (heat-maps
within 1e-3 of the fp32 reference, arg-max identical).  The reduced-precision bf16-storage mode that
configs[1] names is timed right after it and reported under `other_modes` with its own parity block -- it
does NOT meet the contract and is never the headline.

    python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no torch.distributed environment the script starts N ranks of itself
(python -m torch.distributed.run, one per GPU) before anything touches the GPU and relays rank 0's line.
A batch of the split-fp16 mode runs as two sub-batch lanes inside udp_hrnet_forward (two hipGraphs on two streams, joined
before the flip fuse; include/udp_pose_hip.h: udp_hrnet_lanes) -- still one step = one batch of 64 crops.
Prints ONE JSON line (rank 0).  `value` is whole-job images/s; multi-GPU is weak scaling over
independent replicas (every rank decodes its own batch; no data-path collective).  The line also
carries `roofline` (dominant kernel class: algorithmic FLOPs / measured launch time vs the dense
MFMA peak, from a hipEvent-instrumented pass of the same K steps) and `cpu_baseline` (the CPU
oracle -- stock torch fp32 conv graph + NumPy decode, proven equal to the reference in
tests/test_oracle_golden.py -- timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from udp_pose_amd import synth  # noqa: E402
from udp_pose_amd.inference import decode_device  # noqa: E402
from udp_pose_amd.model import MODELS  # noqa: E402
from udp_pose_amd.transforms import COCO_FLIP_PAIRS, channel_map  # noqa: E402
from udp_pose_amd import _lib  # noqa: E402

# dense MFMA peaks, MI355X_MICROARCH.md.  f16x2 (split fp16) issues three fp16 MFMAs per algorithmic
# product, so its ceiling in ALGORITHMIC FLOP/s is the fp16 peak / 3
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "f16x2": 2500.0 / 3}
PEAK_HBM_GBS = 8000.0


# model -> (MODEL.EXTRA, input H, input W, weight seed); w32 = BASELINE.json configs[1], w48 = configs[3]
# rsn18 = configs[4]: RSN-18 backbone with the 51-channel offset head + UDP offset decode
MODELS_CFG = {"w32": (synth.W32_EXTRA, 256, 192, 0), "w48": (synth.scaled_extra(48), 384, 288, 2),
              "rsn18": (None, 256, 192, 4)}


def build_net(dtype, target_type="gaussian", model="w32"):
    extra, _, _, seed = MODELS_CFG[model]
    if model == "rsn18":
        from udp_pose_amd.model import RSN18Hip
        och = 51 if target_type == "offset" else 17
        cp = os.path.join(ROOT, "tests", "golden", "bn_calib_rsn18_%d.npz" % och)
        sd = synth.synth_rsn18_state_dict(och, seed=seed, bn_calib=dict(np.load(cp)) if os.path.exists(cp) else None)
        return sd, RSN18Hip(och, dtype=dtype).load_state_dict(sd)
    calib_path = os.path.join(ROOT, "tests", "golden", "bn_calib_%s_%s.npz" % (model, target_type))
    calib = dict(np.load(calib_path)) if os.path.exists(calib_path) else None
    sd = synth.synth_state_dict(extra, 17, target_type, seed=seed, bn_calib=calib)
    cfg = {"MODEL": {"NAME": "pose_hrnet", "EXTRA": extra, "NUM_JOINTS": 17, "TARGET_TYPE": target_type}}
    net = MODELS["pose_hrnet"](cfg, is_train=False, dtype=dtype)
    net.load_state_dict(sd, strict=True)
    return sd, net


class HotPath:
    """forward(+mirrored) -> flip fuse -> decode, all on device, fixed buffers."""

    def __init__(self, net, batch, device, seed, h=256, w=192, target_type="gaussian"):
        self.net, self.n, self.h, self.w = net.to(device), batch, h, w
        self.tt = target_type
        self.ch = 51 if target_type == "offset" else 17
        crops = synth.synth_crops(min(batch, 8), h, w, seed=seed)
        reps = (batch + crops.shape[0] - 1) // crops.shape[0]
        x = torch.from_numpy(np.tile(crops, (reps, 1, 1, 1))[:batch]).to(device)
        x += 0.01 * torch.randn(x.shape, device=device, generator=torch.Generator(device).manual_seed(seed))
        self.xin, _ = self.net.io_buffers(batch, h, w, True)
        self.xin.copy_(x)
        c, s = synth.synth_center_scale(batch, seed=seed)
        self.center = torch.from_numpy(c.astype(np.float64)).to(device)
        self.scale = torch.from_numpy(s.astype(np.float64)).to(device)
        src, sign = channel_map(self.ch, COCO_FLIP_PAIRS, target_type == "offset")
        self.src = torch.from_numpy(src).to(device)
        self.sign = torch.from_numpy(sign).to(device)
        self.fused = torch.empty(batch, self.ch, h // 4, w // 4, device=device)

    def step(self):
        raw = self.net.raw_forward(self.xin, flip_test=True)
        n = self.n
        _lib.check(_lib.lib().udp_flip_fuse(_lib.ptr(raw), C_ptr(raw, n), _lib.ptr(self.src), _lib.ptr(self.sign),
                                            n, self.ch, self.h // 4, self.w // 4, _lib.ptr(self.fused), _lib.stream_ptr()))
        return decode_device(self.fused, self.center, self.scale, self.tt, self.tt == "gaussian", 4.0, True, want_idx=False)


def C_ptr(t, row):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + row * t.stride(0) * t.element_size())


def roofline(net, hp, steps, dtype):
    """hipEvent-instrumented pass: per-op times -> dominant kernel class."""
    acc = None
    for _ in range(steps):
        ms, desc = net.profile(hp.xin, flip_test=True)
        acc = ms if acc is None else acc + ms
    ms = acc / steps
    esz = 2 if dtype == "bf16" else 4
    b = 2 * hp.n
    classes = {}
    # ops of one launch group run as ONE merged launch (conv_ws_multi / conv_mfma_multi): they form a class of
    # their own whose launch count is the number of groups, so `avg_launch_us` is the kernel's launch duration
    # that rocprofv3 --kernel-trace --stats reports (udp_hrnet_profile splits a merged launch's time over its
    # members, summed back here)
    groups = [int(o.group) for o in net.program(hp.h, hp.w).ops_array()]
    merged_ids = {}
    for t, (name, kind, ks, stride, cin, cout, ho, wo), gid in zip(ms, desc, groups):
        key = {0: "stem", 2: "fuse_sum", 3: "stem7x7", 4: "maxpool", 5: "bilinear", 10: "basic_block_c32"}.get(
            kind, "conv%dx%d_s%d_nb%d" % (ks, ks, stride, 4 if cout % 64 == 0 else 2))
        if gid and kind == 1:
            key = "merged_conv%dx%d_s%d" % (ks, ks, stride)
            merged_ids.setdefault(key, set()).add(gid)
        flops = 2.0 * ks * ks * cin * cout * ho * wo * b if kind in (0, 1, 3) else 0.0
        if kind == 10:                                   # fused BasicBlock: two 3x3 convs
            flops = 2 * 2.0 * 9 * cin * cout * ho * wo * b
        hin, win = ho * stride, wo * stride
        byts = (hin * win * cin * (4 if kind in (0, 3) else esz) + ho * wo * cout * esz) * b
        c = classes.setdefault(key, [0.0, 0.0, 0.0, 0])
        c[0] += float(t)
        c[1] += flops
        c[2] += byts
        c[3] += 1
    for key, ids in merged_ids.items():
        classes[key][3] = len(ids)
    dom = max(classes, key=lambda k: classes[k][0])
    t_ms, flops, byts, cnt = classes[dom]
    handle = net._compiled[(hp.h, hp.w)][0]
    lanes = int(_lib.lib().udp_hrnet_lanes(handle, hp.n)) if net.use_graph else 1
    achieved = flops / (t_ms * 1e-3) / 1e12
    table = {k: {"ms": round(v[0], 4), "launches": v[3], "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else 0,
                 "gbs": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0} for k, v in classes.items()}
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/pmc_by_op.py writes
    # profiles/r02_traffic_<dtype>.json together with the SHA-256 of the library it profiled): reported only
    # when that library is the one loaded now, null otherwise -- never a number from another build
    traffic = None
    tfile = next((f for f in (os.path.join(ROOT, "profiles", "r%02d_traffic_%s.json" % (r, dtype)) for r in (3, 2))
                  if os.path.exists(f)), os.path.join(ROOT, "profiles", "r03_traffic_%s.json" % dtype))
    tj = {}
    if os.path.exists(tfile) and hp.n == 64 and hp.h == 256:
        try:
            with open(tfile) as f:
                tj = json.load(f)
        except (OSError, ValueError):
            tj = {}                       # an unreadable profile file must never take the bench line down
    if tj.get("classes") and tj.get("lib_sha256") == lib_sha256():
        try:
            if dom == "merged_conv3x3_s1":       # the per-op PMC run has the members as 3x3 stride-1 classes
                tot = sum(tc["fetch_bytes"] + tc["write_bytes"] for k, tc in tj["classes"].items() if k.startswith("conv3x3_s1"))
                ops = sum(tc["launches"] for k, tc in tj["classes"].items() if k.startswith("conv3x3_s1"))
                n_dom = sum(1 for (_, kind, ks, st, *_), g in zip(desc, groups) if g and kind == 1 and ks == 3 and st == 1)
                traffic = round(tot * n_dom / ops / cnt, 0)
            else:
                tc = tj["classes"].get(dom)
                if tc:
                    traffic = round(tc["hbm_bytes_per_launch"], 0)
        except (KeyError, TypeError, ZeroDivisionError):
            traffic = None
    if dom.startswith("merged_"):
        kname = ("conv_ws_multi<%s> (merged weight-stationary split-fp16 convs of one HRNet depth)" % dom[11]) if dtype == "f16x2" \
            else "conv_mfma_multi<%s>" % dtype
    else:
        kname = "conv_ws_h2_kernel" if dtype == "f16x2" else "conv_mfma_kernel<%s>" % dtype
    return {"bound": "mfma", "kernel": "%s: class %s" % (kname, dom), "achieved": round(achieved, 2),
            "peak": round(PEAK_TFLOPS[dtype], 1), "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS[dtype], 4),
            "peak_note": {"f16x2": "algorithmic FLOP/s against the dense fp16 MFMA peak (2.5 PF) / 3 MFMAs per product",
                          "bf16": "dense bf16 MFMA peak", "f32": "fp32 MFMA peak"}[dtype],
            "lanes_in_timed_region": lanes,
            "measured_as": ("instrumented eager pass of the whole batch in ONE lane (the kernel alone on the chip; rocprofv3 of "
                            "`UDP_POSE_LANES=1 bench.py` agrees); the timed region runs the batch as two half-batch lanes whose "
                            "launches overlap -- see whole_step for what that adds") if lanes > 1 else
                           "instrumented eager pass of the launches of the timed region",
            "traffic": traffic, "traffic_source": os.path.basename(tfile) if traffic else None,
            "launches_per_step": cnt, "avg_launch_us": round(t_ms / cnt * 1e3, 2),
            "algorithmic_gflop_per_launch": round(flops / cnt / 1e9, 3),
            "algorithmic_gbs_same_kernel": round(byts / (t_ms * 1e-3) / 1e9, 1),
            "sum_kernel_ms_per_step": round(float(ms.sum()), 3), "classes": table}


def lib_sha256():
    import hashlib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def usable_cpus():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU
    box hands each job a share of the host; running torch's default one-thread-per-hardware-thread pool
    against a smaller quota oversubscribes and slows the baseline down several times)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(sd, sample=8):
    """The CPU oracle on this box's host cores: forward + mirrored forward + flip fuse + DARK decode."""
    from oracle import decode as odec, flip as oflip, hrnet as ohrnet
    cores = min(usable_cpus(), int(os.environ.get("UDP_POSE_CPU_THREADS", "64")))
    torch.set_num_threads(cores)
    x = torch.from_numpy(synth.synth_crops(sample, 256, 192, seed=1))
    c, s = synth.synth_center_scale(sample, seed=1)

    def run():
        y = ohrnet.hrnet_forward(sd, synth.W32_EXTRA, torch.cat([x, torch.flip(x, dims=[3])])).numpy()
        hm = oflip.flip_fuse(y[:sample], y[sample:], oflip.COCO_FLIP_PAIRS, False)
        return odec.get_final_preds("gaussian", True, 4.0, hm.copy(), c, s), hm   # (post() mutates its input)

    run()
    t0 = time.time()
    reps = 0
    while reps < 2 or time.time() - t0 < 12.0:
        (ref, hm) = run()
        reps += 1
    dt = (time.time() - t0) / reps
    return {"value": round(sample / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d crops x %d reps, fp32 torch conv graph x2 (flip) + NumPy flip fuse + DARK decode" % (sample, reps)}, \
        x, c, s, ref, hm


def parity(dtypes, sd, x, c, s, ref, ref_hm, device):
    """Keypoint / heat-map agreement of every timed mode (and fp32) with the CPU oracle on the baseline sample."""
    out = {}
    for dt in dtypes:
        _, net = build_net(dt)
        hp = HotPath(net, x.shape[0], device, seed=0)
        hp.xin.copy_(x.to(device))
        hp.center = torch.from_numpy(c.astype(np.float64)).to(device)
        hp.scale = torch.from_numpy(s.astype(np.float64)).to(device)
        preds, maxvals, _, _ = hp.step()
        torch.cuda.synchronize()
        err = np.abs(preds.cpu().numpy() - ref[0]).max(axis=2)
        hm = hp.fused.cpu().numpy()
        n, j = hm.shape[:2]
        same = hm.reshape(n, j, -1).argmax(2) == ref_hm.reshape(n, j, -1).argmax(2)
        out[dt] = {"heatmap_max_abs_err": float(np.abs(hm - ref_hm).max()), "argmax_equal_rate": float(same.mean()),
                   "keypoint_err_px_median": float(np.median(err)), "keypoint_err_px_p90": float(np.percentile(err, 90))}
    return out


def self_launch(args):
    """bench.py --gpus N from a plain shell: start N ranks (one per GPU) with torch.distributed.run and relay
    their output.  Runs before any GPU call, so the parent never initialises the device (the reference
    launches its multi-GPU test the same way, RSN/exps/RSN18.coco/test.py:158)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def time_mode(dtype, args, device, rank, in_h, in_w, tt, barrier):
    """W warm-up steps, then exactly K timed steps of the hot path in one storage mode."""
    sd, net = build_net(dtype, target_type=tt, model=args.model)
    net.use_graph = not args.no_graph
    hp = HotPath(net, args.batch, device, seed=100 + rank, h=in_h, w=in_w, target_type=tt)
    for _ in range(args.warmup):
        hp.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = hp.step()
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out[1]).all()
    return sd, net, hp, dt


def golden_parity(model, dtype, device):
    """Heat-maps of the timed mode on the crop the REFERENCE module was run on when tests/golden was made
    (oracle/gen_golden.py): max abs error and arg-max agreement.  Data files only -- nothing of the reference is read."""
    name = {"w48": "hrnet_w48_gaussian.npz", "rsn18": "rsn18_51.npz"}[model]
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    _, h, w, _ = MODELS_CFG[model]
    seed = {"w48": 6, "rsn18": 8}[model]
    _, net = build_net(dtype, target_type="offset" if model == "rsn18" else "gaussian", model=model)
    x = torch.from_numpy(synth.synth_crops(1, h, w, seed=seed)).to(device)
    got = net.to(device)(x).clone().cpu().numpy()
    ref = g["out"]
    ch = ref.shape[1]
    hm_ch = slice(0, ch, 3) if model == "rsn18" else slice(None)        # offset head: arg-max on the heat-map channels
    return {"fixture": "tests/golden/" + name + " (reference module output, 1 crop)",
            "heatmap_max_abs_err": float(np.abs(got - ref).max()),
            "argmax_equal_rate": float((got[:, hm_ch].reshape(-1, ref.shape[2] * ref.shape[3]).argmax(1) ==
                                        ref[:, hm_ch].reshape(-1, ref.shape[2] * ref.shape[3]).argmax(1)).mean()),
            "tolerance": 3e-3 if model == "rsn18" else 1e-3}


def time_other_inference(model, args, device):
    """BASELINE.json configs[3] / configs[4] in the parity-grade mode: same step definition as the headline."""
    import copy
    a = copy.copy(args)
    a.model, a.batch = model, (32 if model == "w48" else 64)
    _, h, w, _ = MODELS_CFG[model]
    tt = "offset" if model == "rsn18" else "gaussian"
    sd, net, hp, dt = time_mode("f16x2", a, device, 0, h, w, tt, torch.cuda.synchronize)
    ms = dt / a.steps * 1e3
    out = {"workload": "%s %dx%d f16x2 inference, batch=%d, flip-test on, %s decode" % (model, h, w, a.batch, "DARK" if tt == "gaussian" else "offset"),
           "value": round(a.batch * a.steps / dt, 1), "unit": "images/s", "ms_per_step": round(ms, 4), "dtype": "f16x2"}
    try:
        r = roofline(net, hp, 3, "f16x2")
        out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_us")}
        flops_per_img = 2 * 2 * net.program(h, w).macs_per_image()
        out["whole_step_tflops"] = round(flops_per_img * a.batch / (ms * 1e-3) / 1e12, 2)
    except Exception as e:                                          # noqa: BLE001
        out["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    del hp, net
    out["parity"] = golden_parity(model, "f16x2", device)
    return out


def time_train_step(dtype, args, device, sd):
    """BASELINE.json configs[2] on one GPU (its per-GPU share: 32 images): train-mode forward + JointsMSELoss +
    backward + Adam of pose_hrnet_w32 256x192, replayed as a hipGraph (function.train's path).  Parity: the step-0
    loss of 8 of the images against the CPU oracle's train-mode forward + criterion (oracle/train.py)."""
    from udp_pose_amd.train import HRNetTrainer
    cfg = {"MODEL": {"EXTRA": synth.W32_EXTRA, "NUM_JOINTS": 17, "TARGET_TYPE": "gaussian"}}
    n = 32
    x = torch.from_numpy(synth.synth_crops(n, 256, 192, seed=1)).to(device)
    tg = torch.from_numpy(synth.synth_heatmaps(n, 17, 64, 48, seed=2)).to(device)
    tw = torch.ones(n, 17, 1, device=device)
    tr = HRNetTrainer(cfg, sd, device=device, dtype=dtype)
    heat = tr.forward(x[:8].contiguous())
    loss8 = float(tr.loss_and_grad(heat, tg[:8].contiguous(), tw[:8].contiguous())[0].cpu()[0])
    tr._tape = []
    for _ in range(3):                                   # eager, capture, first replay
        tr.train_step_graphed(x, tg, tw)
    torch.cuda.synchronize()
    steps = max(5, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step_graphed(x, tg, tw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / steps * 1e3
    from udp_pose_amd.hrnet_plan import HRNetProgram
    flops = 3 * 2.0 * HRNetProgram(sd, synth.W32_EXTRA, 256, 192, "f32").macs_per_image() * n
    peak = {"f32": 157.3, "bf16": 2500.0}[dtype]
    tf = flops / (ms * 1e-3) / 1e12
    out = {"workload": "w32 256x192 training step, batch=32 (config 3's per-GPU share), JointsMSELoss, Adam, hipGraph replay",
           "value": round(n / ms * 1e3, 1), "unit": "images/s", "ms_per_step": round(ms, 3),
           "dtype": dtype + (" activations (fp32 statistics, master weights, gradients, Adam): NO reference counterpart -- the "
                             "reference trains in fp32" if dtype == "bf16" else " (the reference's precision)"),
           "loss_after_steps": [float(v) for v in loss.cpu().numpy()], "steps_timed": steps,
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                        "note": "algorithmic FLOPs of the three conv passes (forward, input gradient, weight gradient) / wall time of the whole step"}}
    del tr
    return out, loss8, (x[:8].cpu(), tg[:8].cpu(), tw[:8].cpu())


def other_configs(args, device, sd_w32):
    """Configs 4, 5 and 3 of BASELINE.json, timed after the headline on rank 0 of a one-GPU run; every entry is
    wrapped so that a failure shows up as an error string and never costs the headline line."""
    out = {}
    for model in ("w48", "rsn18"):
        try:
            out[model] = time_other_inference(model, args, device)
        except Exception as e:                                      # noqa: BLE001
            out[model] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    oracle_loss = None
    for dtype in ("f32", "bf16"):
        key = "train_w32_b32_" + dtype
        try:
            out[key], loss8, sample = time_train_step(dtype, args, device, sd_w32)
            if oracle_loss is None and not args.no_cpu_baseline:
                from oracle import train as o_train
                torch.set_num_threads(min(usable_cpus(), int(os.environ.get("UDP_POSE_CPU_THREADS", "64"))))
                t0 = time.time()
                parts, _, _ = o_train.loss_and_grads({k: v.clone() for k, v in sd_w32.items()}, synth.W32_EXTRA, sample[0],
                                                     sample[1], sample[2], "gaussian")
                oracle_loss = (float(parts[0]), time.time() - t0)
            if oracle_loss is not None:
                out[key]["parity"] = {"step0_loss_8img": loss8, "cpu_oracle_loss_8img": oracle_loss[0],
                                      "rel_err": abs(loss8 - oracle_loss[0]) / abs(oracle_loss[0]),
                                      "note": "train-mode forward (batch statistics) + JointsMSELoss on 8 of the images vs oracle/train.py "
                                              "(fwd + autograd bwd took %.1f s on the host)" % oracle_loss[1]}
        except Exception as e:                                      # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="crops per GPU per step (default 64; 32 for w48)")
    ap.add_argument("--model", default="w32", choices=sorted(MODELS_CFG), help="w32 256x192 (headline), w48 384x288, rsn18 256x192 + offset head/decode")
    ap.add_argument("--dtype", default="f16x2", choices=["f16x2", "f32", "bf16"],
                    help="storage mode that is timed as `value` (default: the parity-grade split-fp16 mode)")
    ap.add_argument("--no-other-modes", action="store_true", help="do not time the bf16 mode after the headline one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="do not time W48 / RSN-18 / the training step after the headline")
    ap.add_argument("--no-graph", action="store_true", help="eager launches on lane streams instead of hipGraph replay")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d runs under WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    extra, in_h, in_w, _ = MODELS_CFG[args.model]
    if args.batch <= 0:
        args.batch = 32 if args.model == "w48" else 64
    tt = "offset" if args.model == "rsn18" else "gaussian"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    sd, net, hp, dt = time_mode(args.dtype, args, device, rank, in_h, in_w, tt, barrier)
    dt = max_over_ranks(dt)
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    other = {}
    if not args.no_other_modes and args.dtype != "bf16":
        _, _, hp_b, dt_b = time_mode("bf16", args, device, rank, in_h, in_w, tt, barrier)
        dt_b = max_over_ranks(dt_b)
        other["bf16"] = {"value": round(world * args.batch * args.steps / dt_b, 1), "unit": "images/s",
                         "ms_per_step": round(dt_b / args.steps * 1e3, 4),
                         "note": "bf16 storage, fp32 accumulate: REDUCED PRECISION, outside the 1e-3 / arg-max parity "
                                 "contract (see parity_vs_cpu_oracle.bf16); reported for BASELINE.json configs[1] only"}
        del hp_b

    if not args.no_other_modes and args.dtype == "f16x2" and int(_lib.lib().udp_hrnet_lanes(net._compiled[(in_h, in_w)][0], args.batch)) > 1:
        # the same mode with the sub-batch lanes off: what the two concurrent chains add (and the configuration the
        # roofline pass below measures)
        saved = os.environ.get("UDP_POSE_LANES")
        os.environ["UDP_POSE_LANES"] = "1"
        try:
            _, _, hp_1, dt_1 = time_mode(args.dtype, args, device, rank, in_h, in_w, tt, barrier)
            dt_1 = max_over_ranks(dt_1)
            other["f16x2_one_lane"] = {"value": round(world * args.batch * args.steps / dt_1, 1), "unit": "images/s",
                                       "ms_per_step": round(dt_1 / args.steps * 1e3, 4),
                                       "note": "UDP_POSE_LANES=1: the whole batch as one launch sequence (round 2's configuration)"}
            del hp_1
        finally:
            if saved is None:
                os.environ.pop("UDP_POSE_LANES", None)
            else:
                os.environ["UDP_POSE_LANES"] = saved

    metric = {"w32": "images/sec HRNet-W32 256x192 (infer+decode)", "w48": "images/sec HRNet-W48 384x288 (infer+decode)",
              "rsn18": "images/sec RSN-18 256x192 + UDP offset decode (infer+decode)"}[args.model]
    line = {"metric": metric, "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x2": "f16x2 (split fp16 hi+lo operands = 22 bits, 3 fp16 MFMAs per product, fp32 accumulate)",
                      "f32": "f32", "bf16": "bf16 (reduced precision)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": "%s %dx%d %s inference, batch=%d per GPU, flip-test on, %s decode "
                                   "(forward on 2N images + flip fuse + udp_decode_%s)" %
                                   (args.model, in_h, in_w, args.dtype, args.batch, "DARK" if tt == "gaussian" else "offset", tt),
                       "global_batch": world * args.batch, "parallelism": "replicas x%d (no data-path collective)" % world,
                       "weights": "seeded synthetic, BN stats calibrated (tests/golden/bn_calib_%s_gaussian.npz)" % args.model,
                       "inputs": "8 distinct seeded synthetic crops tiled to the batch + N(0, 0.01^2) noise per image, resident in HBM"}}
    if rank == 0:
        # the headline numbers are complete at this point: nothing below may keep the line from being printed
        try:
            line["roofline"] = roofline(net, hp, max(1, min(args.steps, 5)), args.dtype)
        except Exception as e:                                      # noqa: BLE001
            line["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        flops_per_img = 2 * 2 * net.program(in_h, in_w).macs_per_image()
        line["whole_step"] = {"tflops": round(flops_per_img * args.batch / (ms_per_step * 1e-3) / 1e12, 2),
                              "algorithmic_act_gbs": round(2 * net.program(in_h, in_w).activation_elems_per_image() *
                                                           (2 if args.dtype == "bf16" else 4) * args.batch /
                                                           (ms_per_step * 1e-3) / 1e9, 1),
                              "hbm_peak_gbs": PEAK_HBM_GBS}
        if other:
            line["other_modes"] = other
        if world == 1 and not args.no_cpu_baseline and args.model == "w32":
            try:
                cb, x, c, s, ref, ref_hm = cpu_baseline(sd)
                line["cpu_baseline"] = cb
                modes = sorted({args.dtype, "f32"} | {m for m in other if m in ("f32", "bf16", "f16x2")})
                line["parity_vs_cpu_oracle"] = parity(modes, sd, x, c, s, ref, ref_hm, device)
            except Exception as e:                                  # noqa: BLE001
                line.setdefault("cpu_baseline", {"error": "%s: %s" % (type(e).__name__, e)})
                line.setdefault("parity_vs_cpu_oracle", {"error": "%s: %s" % (type(e).__name__, e)})
        if world == 1 and args.model == "w32" and not args.no_other_configs:
            del hp, net
            torch.cuda.empty_cache()
            try:
                line["other_configs"] = other_configs(args, device, sd)
            except Exception as e:                                  # noqa: BLE001
                line["other_configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
