"""Oracle: PCK-style accuracy on heat-maps (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates deep_hrnet/lib/core/evaluate.py:16-73 (calc_dists, dist_acc, accuracy) on top of
oracle/decode.get_max_preds.  Pinned by tests/golden/accuracy.npz (oracle/gen_golden_accuracy.py runs
the reference's own function).
"""
import numpy as np

from . import decode as odec


def accuracy(output, target, thr=0.5):
    """evaluate.py:40-73 with hm_type='gaussian'.  Returns (acc [J+1], avg_acc, cnt, pred [N,J,2])."""
    pred, _, _ = odec.get_max_preds(output)
    tgt, _, _ = odec.get_max_preds(target)
    n, j = pred.shape[:2]
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((n, 2)) * np.array([h, w]) / 10
    p32, t32 = pred.astype(np.float32), tgt.astype(np.float32)
    dists = np.full((j, n), -1.0)
    ok = (t32[..., 0] > 1) & (t32[..., 1] > 1)
    d = np.linalg.norm(p32 / norm[:, None, :] - t32 / norm[:, None, :], axis=2)
    dists[ok.T] = d.T[ok.T]
    acc = np.zeros(j + 1)
    avg, cnt = 0.0, 0
    for i in range(j):
        valid = dists[i] != -1
        acc[i + 1] = (dists[i][valid] < thr).sum() * 1.0 / valid.sum() if valid.sum() > 0 else -1
        if acc[i + 1] >= 0:
            avg += acc[i + 1]
            cnt += 1
    avg = avg / cnt if cnt else 0
    if cnt:
        acc[0] = avg
    return acc, avg, cnt, pred
