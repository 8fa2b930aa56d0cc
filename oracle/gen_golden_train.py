#!/usr/bin/env python3
"""Golden for the training step (function.py:38-77): REFERENCE module in train mode + REFERENCE
criterion + torch.optim.Adam(lr=1e-3) for two steps on a width-16 mini HRNet (build container only).

    python oracle/gen_golden_train.py      # writes tests/golden/train_mini_{gaussian,offset}.npz

Adam's first update is lr*g/(|g|+eps): elements whose gradient is at rounding-noise level move by
+-lr with a sign no two fp32 implementations agree on, so the second step is only comparable in
aggregate (loss, norms); step 0 (loss, output, gradients, running statistics) is exact.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as gg                                   # noqa: E402
from oracle import data as o_data, train as o_train        # noqa: E402
from udp_pose_amd import synth                             # noqa: E402

EXTRA = synth.scaled_extra(16, modules=(1, 2, 2), blocks=2)
FULL = ("conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight",
        "transition1.1.0.0.weight", "stage2.0.branches.0.0.conv1.weight", "stage3.1.fuse_layers.0.2.0.weight",
        "stage3.0.fuse_layers.2.0.1.0.weight", "stage4.1.fuse_layers.0.0.0.weight", "stage4.0.branches.3.1.bn2.weight",
        "final_layer.weight", "final_layer.bias")
STATS = ("bn1.running_mean", "bn1.running_var", "stage3.1.branches.2.1.bn1.running_var",
         "stage4.1.fuse_layers.0.3.1.running_mean")


def batch(tt, n=4, h=96, w=64, nj=5, seed=11):
    x = torch.from_numpy(synth.synth_crops(n, h, w, seed=seed))
    rng = np.random.default_rng(seed + 1)
    tg, tw = [], []
    for _ in range(n):
        joints = np.zeros((nj, 3), np.float32)
        joints[:, 0] = rng.uniform(-4, w + 4, nj)
        joints[:, 1] = rng.uniform(-4, h + 4, nj)
        vis = np.ones((nj, 3), np.float32)
        vis[rng.random(nj) < 0.2] = 0
        t, wgt = o_data.generate_target(joints, vis, tt, (w, h), (w // 4, h // 4), sigma=2, kpd=4.0)
        tg.append(t)
        tw.append(wgt)
    return x, torch.from_numpy(np.stack(tg)), torch.from_numpy(np.stack(tw))


def main():
    pose_hrnet, _, ref_loss, _, _ = gg.load_reference()
    for tt in ("gaussian", "offset"):
        nj = 5
        sd0 = synth.synth_state_dict(EXTRA, nj, tt, seed=1)
        net = pose_hrnet.get_pose_net(gg.model_cfg(EXTRA, nj, tt), is_train=False)
        net.load_state_dict(sd0, strict=True)
        net.train()                                                      # function.py:38
        crit = (ref_loss.JointsMSELoss_offset if tt == "offset" else ref_loss.JointsMSELoss)(use_target_weight=True)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)                # utils.py:70-74
        sd = {k: v.clone() for k, v in sd0.items()}
        oopt = o_train.Adam(lr=1e-3)
        out = {}
        for step in range(2):
            x, tg, tw = batch(tt, seed=11 + 7 * step)
            y = net(x)
            if tt == "offset":
                l_hm, l_os = crit(y, tg, tw)
                loss, parts = l_hm + l_os, (float(l_hm), float(l_os))
            else:
                loss = crit(y, tg, tw)
                parts = (float(loss),)
            opt.zero_grad()
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
            opt.step()
            out["loss%d" % step] = np.array(parts, np.float64)
            out["y%d" % step] = y.detach().numpy()
            out["gnorm%d" % step] = np.array([float(g.double().norm()) for g in grads.values()])
            if step == 0:
                out["gkeys"] = np.array(list(grads.keys()))
                for k in FULL:
                    out["grad0_" + k] = grads[k].numpy()
            if step == 0:
                ref_sd = net.state_dict()
                for k in FULL + STATS:
                    out["after1_" + k] = ref_sd[k].numpy().copy()
            oparts, oy, ograds = o_train.train_step(sd, EXTRA, oopt, x, tg, tw, tt)
            gd = max(float((ograds[k] - grads[k]).abs().max() / (grads[k].abs().max() + 1e-12)) for k in grads)
            print(tt, "step", step, "loss", parts, "oracle", oparts, "y diff", float((oy - y.detach()).abs().max()),
                  "max rel grad diff", gd)
        ref_sd = net.state_dict()
        for k in FULL + STATS:
            out["after2_" + k] = ref_sd[k].numpy()
        out["pnorm2"] = np.array([float(v.double().norm()) for k, v in ref_sd.items() if o_train.is_param(k)])
        pd = max(float((sd[k] - ref_sd[k]).abs().max()) for k in ref_sd if not k.endswith("num_batches_tracked"))
        print(tt, "params after 2 steps: oracle vs reference max abs diff", pd)
        np.savez_compressed(os.path.join(gg.OUT, "train_mini_%s.npz" % tt), **out)


if __name__ == "__main__":
    main()
