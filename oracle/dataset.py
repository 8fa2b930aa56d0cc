"""Oracle: JointsDataset.__getitem__ on CPU (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

NumPy restatement of deep_hrnet/lib/dataset/JointsDataset.py:176-256 with the AID classes of
lib/utils/transforms.py:144-224; consumes ``np.random`` / ``random`` in the reference's order.
``cv2.warpAffine`` is oracle/cv2_standin.py (parity unpinned vs real OpenCV).  Pinned by
tests/golden/dataset_getitem.npz (oracle/gen_golden_dataset.py runs the reference's own __getitem__).
"""
import copy
import random

import numpy as np

from . import cv2_standin as cv2
from . import data as o_data


def _half_body(joints, joints_vis, cfg):
    """:126-171."""
    upper, lower = [], []
    for j in range(cfg["num_joints"]):
        if joints_vis[j][0] > 0:
            (upper if j in cfg["upper_body_ids"] else lower).append(joints[j])
    sel = upper if (np.random.randn() < 0.5 and len(upper) > 2) else (lower if len(lower) > 2 else upper)
    if len(sel) < 2:
        return None, None
    sel = np.array(sel, dtype=np.float32)
    center = sel.mean(axis=0)[:2]
    lt, rb = np.amin(sel, axis=0), np.amax(sel, axis=0)
    w, h = rb[0] - lt[0], rb[1] - lt[1]
    ar = cfg["aspect_ratio"]
    if w > ar * h:
        h = w * 1.0 / ar
    elif w < ar * h:
        w = h * ar
    return center, np.array([w * 1.0 / 200, h * 1.0 / 200], dtype=np.float32) * 1.5


def _cutout(img, prob, radius_factor, num_patch):
    """transforms.py:201-224."""
    if not (np.random.rand() < prob):
        return img
    h, w, _ = img.shape
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    for _ in range(num_patch):
        center = [np.random.rand() * w, np.random.rand() * h]
        radius = radius_factor * (1 + np.random.rand(2)) * w
        dis = ((center[0] - xs) / radius[0]) ** 2 + ((center[1] - ys) / radius[1]) ** 2
        img[dis <= 1] = 0
    return img


def _hide_and_seek(img, prob, prob_hiding, grid_sizes):
    """transforms.py:160-181 (x/y swapped in the slice, as written there)."""
    if not (np.random.rand() < prob):
        return img
    h, w, _ = img.shape
    g = grid_sizes[np.random.randint(0, len(grid_sizes) - 1)]
    if g != 0:
        for x in range(0, w, g):
            for y in range(0, h, g):
                if np.random.rand() <= prob_hiding:
                    img[x:min(w, x + g), y:min(h, y + g), :] = 0
    return img


def getitem(cfg, rec, frame_bgr):
    """Returns (uint8 crop HxWx3 after AID, target, target_weight, meta)."""
    rec = copy.deepcopy(rec)
    img = frame_bgr
    if cfg["color_rgb"]:
        img = np.ascontiguousarray(img[:, :, ::-1])
    joints, vis = rec["joints_3d"], rec["joints_3d_vis"]
    c, s, r = rec["center"], rec["scale"], 0
    size = np.array(cfg["image_size"])
    if cfg["is_train"]:
        if np.sum(vis[:, 0]) > cfg["num_joints_half_body"] and np.random.rand() < cfg["prob_half_body"]:
            c2, s2 = _half_body(joints, vis, cfg)
            if c2 is not None and s2 is not None:
                c, s = c2, s2
        sf, rf = cfg["scale_factor"], cfg["rotation_factor"]
        s = s * np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf)
        r = np.clip(np.random.randn() * rf, -rf * 2, rf * 2) if random.random() <= 0.6 else 0
        if cfg["flip"] and random.random() <= 0.5:
            img = img[:, ::-1, :]
            joints[:, 0] = img.shape[1] - joints[:, 0] - 1
            for a, b in cfg["flip_pairs"]:
                joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
                vis[a, :], vis[b, :] = vis[b, :], vis[a, :].copy()
            joints = joints * vis
            c[0] = img.shape[1] - c[0] - 1
    trans = o_data.get_warpmatrix(r, c * 2.0, size - 1.0, s)
    crop = cv2.warpAffine(np.ascontiguousarray(img), trans, (int(size[0]), int(size[1])),
                          flags=cv2.WARP_INVERSE_MAP | cv2.INTER_LINEAR)
    joints[:, 0:2] = o_data.rotate_points(joints[:, 0:2], r, c, size, s, False)
    if cfg["is_train"]:
        if cfg.get("cutout"):
            crop = _cutout(crop, *cfg["cutout"])
        if cfg.get("hide_and_seek"):
            crop = _hide_and_seek(crop, *cfg["hide_and_seek"])
    target, weight = o_data.generate_target(joints, vis, cfg["target_type"], tuple(size), tuple(cfg["heatmap_size"]),
                                            sigma=cfg["sigma"], kpd=cfg["kpd"])
    return crop, target, weight, {"joints": joints, "joints_vis": vis, "center": c, "scale": s, "rotation": r}
