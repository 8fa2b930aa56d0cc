"""Oracle: heat-map losses (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Closed-form restatement of deep_hrnet/lib/core/loss.py:15-39 (JointsMSELoss)
and :41-76 (JointsMSELoss_offset), value and gradient w.r.t. the prediction.
fp64 accumulation so the GPU reduction order does not matter to the check.
"""
import numpy as np


def joints_mse_loss(output, target, target_weight, use_target_weight=True):
    """L = 1/J sum_j 0.5*mean_{b,p}((w_bj*pred - w_bj*gt)^2).

    Returns (loss float64, grad f32 like output)."""
    b, j = output.shape[:2]
    p = output.reshape(b, j, -1).astype(np.float64)
    g = target.reshape(b, j, -1).astype(np.float64)
    w = target_weight.reshape(b, j, 1).astype(np.float64) if use_target_weight else 1.0
    d = (p - g) * w
    hw = p.shape[2]
    loss = 0.5 * np.sum(d * d) / (j * b * hw)
    grad = d * w / (j * b * hw)
    return loss, grad.reshape(output.shape).astype(np.float32)


def joints_mse_loss_offset(output, target, target_weight):
    """loss.py:41-76 (use_target_weight=True): (L_hm, L_os) and d(L_hm+L_os)/d output.

    L_hm as above on channels 3j with target_weight; L_os weights the x/y offset
    residuals by the ground-truth disk mask (target channel 3j), not by
    target_weight."""
    b, c = output.shape[:2]
    j = c // 3
    p = output.reshape(b, j, 3, -1).astype(np.float64)
    g = target.reshape(b, j, 3, -1).astype(np.float64)
    w = target_weight.reshape(b, j, 1).astype(np.float64)
    hw = p.shape[3]
    norm = j * b * hw
    dh = (p[:, :, 0] - g[:, :, 0]) * w
    m = g[:, :, 0]
    dx = m * (p[:, :, 1] - g[:, :, 1])
    dy = m * (p[:, :, 2] - g[:, :, 2])
    l_hm = 0.5 * np.sum(dh * dh) / norm
    l_os = 0.5 * (np.sum(dx * dx) + np.sum(dy * dy)) / norm
    grad = np.stack([dh * w, dx * m, dy * m], axis=2) / norm
    return l_hm, l_os, grad.reshape(output.shape).astype(np.float32)
