#!/usr/bin/env python3
"""Per-parameter gradient error of HRNetTrainer vs the fp64 oracle, next to the fp32 oracle's own error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import train as o_train
from udp_pose_amd import synth
from udp_pose_amd.train import HRNetTrainer
from test_train_oracle_cpu import EXTRA, make_batch
tt = sys.argv[1] if len(sys.argv) > 1 else "gaussian"
sd0 = synth.synth_state_dict(EXTRA, 5, tt, seed=1)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
x, tg, tw = make_batch(tt, seed=seed)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()}
_, y64, g64 = o_train.loss_and_grads(sd64, EXTRA, x.double(), tg.double(), tw.double(), tt)
sd32 = {k: v.clone() for k, v in sd0.items()}
_, y32, g32 = o_train.loss_and_grads(sd32, EXTRA, x, tg, tw, tt)
tr = HRNetTrainer({"MODEL": {"EXTRA": EXTRA, "NUM_JOINTS": 5, "TARGET_TYPE": tt}}, sd0, device="cuda")
heat = tr.forward(x.cuda())
print("y err hip", float((heat.cpu().double() - y64).abs().max()), "o32", float((y32.double() - y64).abs().max()))
loss, d = tr.loss_and_grad(heat, tg.cuda(), tw.cuda())
tr.backward(d)
rows = []
for k in tr._keys:
    ex = g64[k].numpy(); mx = np.abs(ex).max() + 1e-30
    rows.append((np.abs(tr.grad_of(k).cpu().numpy() - ex).max() / mx, np.abs(g32[k].numpy() - ex).max() / mx, k, tuple(ex.shape)))
for e_hip, e_32, k, s in rows:
    flag = "  <<<" if e_hip > 3 * e_32 + 2e-4 else ""
    print("%-50s %-18s hip %.2e  o32 %.2e%s" % (k, s, e_hip, e_32, flag))
