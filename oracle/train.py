"""Oracle: one training step on CPU (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates deep_hrnet/lib/core/function.py:38-77 (``model.train()``; forward; criterion;
``zero_grad / backward / step``) with the optimizer of lib/utils/utils.py:70-74
(``optim.Adam(model.parameters(), lr=cfg.TRAIN.LR)``: betas (0.9, 0.999), eps 1e-8, no weight decay).
The forward is oracle/hrnet.py in train mode, gradients come from torch autograd over those stock
fp32 ops, the losses restate lib/core/loss.py:15-76 in torch ops, and Adam is written out
(torch.optim is the reference's third-party dependency: torch>=1.x ``Adam`` single-tensor rule).
Pinned against the reference module + reference criterion + torch.optim.Adam by
tests/golden/train_mini.npz (oracle/gen_golden_train.py).
"""
import math

import torch

from . import hrnet as o_hrnet


def joints_mse_loss_t(output, target, target_weight, use_target_weight=True):
    """loss.py:15-39."""
    b, j = output.shape[:2]
    p = output.reshape(b, j, -1)
    g = target.reshape(b, j, -1)
    loss = 0.0
    for k in range(j):
        pk, gk = p[:, k], g[:, k]
        if use_target_weight:
            w = target_weight[:, k].reshape(b, 1)
            pk, gk = pk * w, gk * w
        loss = loss + 0.5 * torch.mean((pk - gk) ** 2)
    return loss / j


def joints_mse_loss_offset_t(output, target, target_weight):
    """loss.py:41-76 (use_target_weight=True): returns (loss_hm, loss_os)."""
    b, c = output.shape[:2]
    j = c // 3
    p = output.reshape(b, c, -1)
    g = target.reshape(b, c, -1)
    l_hm, l_os = 0.0, 0.0
    for k in range(j):
        w = target_weight[:, k].reshape(b, 1)
        l_hm = l_hm + 0.5 * torch.mean((p[:, 3 * k] * w - g[:, 3 * k] * w) ** 2)
        m = g[:, 3 * k]
        l_os = l_os + 0.5 * torch.mean((m * p[:, 3 * k + 1] - m * g[:, 3 * k + 1]) ** 2)
        l_os = l_os + 0.5 * torch.mean((m * p[:, 3 * k + 2] - m * g[:, 3 * k + 2]) ** 2)
    return l_hm / j, l_os / j


def is_param(key):
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))


class Adam:
    """torch.optim.Adam defaults (lr given, betas (0.9,0.999), eps 1e-8, weight_decay 0, amsgrad False)."""

    def __init__(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0
        self.m, self.v = {}, {}

    @torch.no_grad()
    def step(self, sd, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        for k, g in grads.items():
            m = self.m.setdefault(k, torch.zeros_like(g))
            v = self.v.setdefault(k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            sd[k] = sd[k] - (self.lr / bc1) * (m / denom)


def loss_and_grads(sd, extra, x, target, target_weight, target_type="gaussian", taps=None):
    """forward(train) + criterion + backward.  ``sd`` running stats are updated in place.
    Returns (loss scalars tuple, output tensor, {param key: grad})."""
    keys = [k for k in sd if is_param(k)]
    for k in keys:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    y = o_hrnet.hrnet_forward_train(sd, extra, x, taps)
    if target_type == "offset":
        l_hm, l_os = joints_mse_loss_offset_t(y, target, target_weight)
        loss, parts = l_hm + l_os, (float(l_hm.detach()), float(l_os.detach()))
    else:
        loss = joints_mse_loss_t(y, target, target_weight)
        parts = (float(loss.detach()),)
    gs = torch.autograd.grad(loss, [sd[k] for k in keys])
    grads = {}
    for k, g in zip(keys, gs):
        grads[k] = g
        sd[k] = sd[k].detach()
    return parts, y.detach(), grads


def train_step(sd, extra, opt, x, target, target_weight, target_type="gaussian"):
    parts, y, grads = loss_and_grads(sd, extra, x, target, target_weight, target_type)
    opt.step(sd, grads)
    return parts, y, grads
