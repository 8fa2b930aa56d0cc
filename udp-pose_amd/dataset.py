"""Training data path on the device: JointsDataset.__getitem__ for a whole batch.

Mirrors deep_hrnet/lib/dataset/JointsDataset.py:
  get_warpmatrix :29-49, rotate_points :51-73, half_body_transform :126-171,
  __getitem__ :176-256 (half-body / scale / rotation / flip sampling :204-222, UDP warp :226-227,
  joints transform :228, AID :231-235, ToTensor+Normalize, generate_target :291-385)
and lib/utils/transforms.py: fliplr_joints :49-63, Cutout :184-224, HideAndSeek :144-181.

Split of work: the *sampling* (a handful of scalars per sample) stays on the host and consumes
``np.random`` / ``random`` in exactly the reference's call order, so a seeded run reproduces the
reference's augmentation; every *pixel* is produced on the GPU (udp_warp_affine_ex with the flip and
BGR->RGB folded into the read, udp_aid_apply, udp_target_gaussian/offset).  Frames are uint8 HxWx3
device tensors (the decoded JPEG; decoding is outside this path).
"""
import copy
import ctypes as C
import math
import random

import numpy as np
import torch

from . import _lib
from .pose_engine import IMAGENET_MEAN, IMAGENET_STD


def get_warpmatrix(theta, size_input, size_dst, size_target):
    """dst->src 2x3 float32 matrix of the unbiased (UDP) crop; JointsDataset.py:29-49."""
    size_target = size_target * 200.0
    theta = theta / 180.0 * math.pi
    cs, sn = math.cos(theta), math.sin(theta)
    sx, sy = size_target[0] / size_dst[0], size_target[1] / size_dst[1]
    m = np.zeros((2, 3), dtype=np.float32)
    m[0, 0], m[0, 1] = cs * sx, sn * sy
    m[0, 2] = -0.5 * size_target[0] * cs - 0.5 * size_target[1] * sn + 0.5 * size_input[0]
    m[1, 0], m[1, 1] = -sn * sx, cs * sy
    m[1, 2] = 0.5 * size_target[0] * sn - 0.5 * size_target[1] * cs + 0.5 * size_input[1]
    return m


def rotate_points(src_points, angle, c, dst_img_shape, size_target, do_clip=True):
    """Image points -> crop coordinates; JointsDataset.py:51-73 (scale_x uses dst_img_shape[0]: callers
    pass image_size = [w, h])."""
    size_target = size_target * 200.0
    scale_x = (dst_img_shape[0] - 1.0) / size_target[0]
    scale_y = (dst_img_shape[1] - 1.0) / size_target[1]
    rad = angle / 180.0 * math.pi
    sn, cs = -math.sin(rad), math.cos(rad)             # radian_sin = -sin (:59)
    out = np.zeros(src_points.shape, dtype=src_points.dtype)
    x = src_points[:, 0] - c[0]
    y = src_points[:, 1] - c[1]
    out[:, 0] = cs * x + sn * y
    out[:, 1] = -sn * x + cs * y
    out[:, 0] += size_target[0] * 0.5
    out[:, 1] += size_target[1] * 0.5
    out[:, 0] *= scale_x
    out[:, 1] *= scale_y
    if do_clip:
        out[:, 0] = np.clip(out[:, 0], 0, dst_img_shape[1] - 1)
        out[:, 1] = np.clip(out[:, 1], 0, dst_img_shape[0] - 1)
    return out


def fliplr_joints(joints, joints_vis, width, matched_parts):
    """transforms.py:49-63."""
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        joints_vis[a, :], joints_vis[b, :] = joints_vis[b, :], joints_vis[a, :].copy()
    return joints * joints_vis, joints_vis


class JointsPipeline:
    """Batch form of JointsDataset.__getitem__ with the pixels on the GPU."""

    def __init__(self, image_size=(192, 256), heatmap_size=(48, 64), num_joints=17, target_type="gaussian", sigma=2,
                 kpd=4.0, is_train=True, scale_factor=0.35, rotation_factor=45, flip=True, num_joints_half_body=8,
                 prob_half_body=0.3, color_rgb=True, flip_pairs=(), upper_body_ids=(), cutout=None, hide_and_seek=None,
                 pixel_std=200, device="cuda"):
        self.image_size = np.array(image_size)
        self.heatmap_size = np.array(heatmap_size)
        self.num_joints, self.target_type, self.sigma, self.kpd = num_joints, target_type, sigma, kpd
        self.is_train, self.scale_factor, self.rotation_factor, self.flip = is_train, scale_factor, rotation_factor, flip
        self.num_joints_half_body, self.prob_half_body, self.color_rgb = num_joints_half_body, prob_half_body, color_rgb
        self.flip_pairs, self.upper_body_ids, self.pixel_std = list(flip_pairs), tuple(upper_body_ids), pixel_std
        self.aspect_ratio = image_size[0] * 1.0 / image_size[1]
        self.cutout = tuple(cutout) if cutout else None                 # (prob, radius_factor, num_patch)
        self.hide_and_seek = tuple(hide_and_seek) if hide_and_seek else None   # (prob, prob_hiding, grid_sizes)
        self.device = torch.device(device)

    # ---------------------------------------------------------------- host sampling (reference RNG order)
    def half_body_transform(self, joints, joints_vis):
        """JointsDataset.py:126-171 (``np.random.randn() < 0.5`` as written there)."""
        upper, lower = [], []
        for j in range(self.num_joints):
            if joints_vis[j][0] > 0:
                (upper if j in self.upper_body_ids else lower).append(joints[j])
        if np.random.randn() < 0.5 and len(upper) > 2:
            sel = upper
        else:
            sel = lower if len(lower) > 2 else upper
        if len(sel) < 2:
            return None, None
        sel = np.array(sel, dtype=np.float32)
        center = sel.mean(axis=0)[:2]
        lt, rb = np.amin(sel, axis=0), np.amax(sel, axis=0)
        w, h = rb[0] - lt[0], rb[1] - lt[1]
        if w > self.aspect_ratio * h:
            h = w * 1.0 / self.aspect_ratio
        elif w < self.aspect_ratio * h:
            w = h * self.aspect_ratio
        scale = np.array([w * 1.0 / self.pixel_std, h * 1.0 / self.pixel_std], dtype=np.float32)
        return center, scale * 1.5

    def _sample_cutout(self, width, height):
        """Cutout.__call__/_cutout (transforms.py:201-224): ellipse list [(cx, cy, rx, ry)]."""
        prob, radius_factor, num_patch = self.cutout
        out = []
        if np.random.rand() < prob:
            for _ in range(num_patch):
                center = [np.random.rand() * width, np.random.rand() * height]
                radius = radius_factor * (1 + np.random.rand(2)) * width
                out.append((center[0], center[1], radius[0], radius[1]))
        return out

    def _sample_hide_and_seek(self, width, height):
        """HideAndSeek.__call__/_hide_and_seek (transforms.py:160-181): (grid, {(i, j) hidden cells})."""
        prob, prob_hiding, grid_sizes = self.hide_and_seek
        if not (np.random.rand() < prob):
            return 0, set()
        g = grid_sizes[np.random.randint(0, len(grid_sizes) - 1)]
        cells = set()
        if g != 0:
            for i, _x in enumerate(range(0, width, g)):
                for j, _y in enumerate(range(0, height, g)):
                    if np.random.rand() <= prob_hiding:
                        cells.add((i, j))
        return g, cells

    def sample(self, rec, frame_width):
        """The scalar part of __getitem__ (:198-228) for one record; returns the plan for the device."""
        rec = copy.deepcopy(rec)
        joints, joints_vis = rec["joints_3d"], rec["joints_3d_vis"]
        c, s, r, flipped = rec["center"], rec["scale"], 0, False
        if self.is_train:
            if np.sum(joints_vis[:, 0]) > self.num_joints_half_body and np.random.rand() < self.prob_half_body:
                c_hb, s_hb = self.half_body_transform(joints, joints_vis)
                if c_hb is not None and s_hb is not None:
                    c, s = c_hb, s_hb
            sf, rf = self.scale_factor, self.rotation_factor
            s = s * np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf)
            r = np.clip(np.random.randn() * rf, -rf * 2, rf * 2) if random.random() <= 0.6 else 0
            if self.flip and random.random() <= 0.5:
                flipped = True
                joints, joints_vis = fliplr_joints(joints, joints_vis, frame_width, self.flip_pairs)
                c[0] = frame_width - c[0] - 1
        trans = get_warpmatrix(r, c * 2.0, self.image_size - 1.0, s)
        joints[:, 0:2] = rotate_points(joints[:, 0:2], r, c, self.image_size, s, False)
        plan = {"trans": trans, "flip": flipped, "joints": joints, "joints_vis": joints_vis, "center": c, "scale": s,
                "rotation": r, "cutout": [], "hs_grid": 0, "hs_cells": set()}
        if self.is_train:
            w, h = int(self.image_size[0]), int(self.image_size[1])
            if self.cutout:
                plan["cutout"] = self._sample_cutout(w, h)
            if self.hide_and_seek:
                plan["hs_grid"], plan["hs_cells"] = self._sample_hide_and_seek(w, h)
        return plan

    # ---------------------------------------------------------------- device
    def batch(self, recs, frames):
        """recs: db records (joints_3d, joints_3d_vis, center, scale); frames: cuda uint8 [H,W,3] tensors in the
        loader's channel order (BGR when color_rgb).  Returns (input [N,3,h,w], target, target_weight, metas)."""
        L = _lib.lib()
        n = len(recs)
        w, h = int(self.image_size[0]), int(self.image_size[1])
        plans = [self.sample(rec, int(frames[i].shape[1])) for i, rec in enumerate(recs)]
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=self.device)
        mean, std = (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD)
        mats = torch.as_tensor(np.stack([p["trans"].astype(np.float64).reshape(6) for p in plans])).to(self.device)
        for i, (p, f) in enumerate(zip(plans, frames)):
            if not f.is_cuda or f.dtype != torch.uint8 or f.dim() != 3 or f.shape[2] != 3 or not f.is_contiguous():
                raise ValueError("frames must be contiguous cuda uint8 [H,W,3] tensors")
            _lib.check(L.udp_warp_affine_ex(f.data_ptr(), f.shape[0], f.shape[1], f.shape[1] * 3,
                                            mats.data_ptr() + 48 * i, 1, h, w, mean, std, int(p["flip"]),
                                            int(self.color_rgb), out.data_ptr() + 4 * 3 * h * w * i, _lib.stream_ptr()))
        n_patch = max((len(p["cutout"]) for p in plans), default=0)
        grids = [p["hs_grid"] for p in plans]
        if n_patch or any(grids):
            cut = np.zeros((n, max(n_patch, 1), 4), np.float64)
            for i, p in enumerate(plans):
                for k, e in enumerate(p["cutout"]):
                    cut[i, k] = e
            gmin = min([g for g in grids if g] or [1])
            mx, my = (w + gmin - 1) // gmin, (h + gmin - 1) // gmin
            mask = np.zeros((n, mx, my), np.uint8)
            for i, p in enumerate(plans):
                for (a, b) in p["hs_cells"]:
                    mask[i, a, b] = 1
            cut_d = torch.from_numpy(cut).to(self.device)
            grid_d = torch.tensor(grids, dtype=torch.int32, device=self.device)
            mask_d = torch.from_numpy(mask).to(self.device)
            _lib.check(L.udp_aid_apply(out.data_ptr(), n, h, w, cut_d.data_ptr(), max(n_patch, 1), grid_d.data_ptr(),
                                       mask_d.data_ptr(), mx, my, mean, std, _lib.stream_ptr()))
        joints = torch.from_numpy(np.stack([p["joints"][:, :2] for p in plans]).astype(np.float32)).to(self.device)
        vis = torch.from_numpy(np.stack([p["joints_vis"][:, 0] for p in plans]).astype(np.float32)).to(self.device)
        k = 3 if self.target_type == "offset" else 1
        hw, hh = int(self.heatmap_size[0]), int(self.heatmap_size[1])
        target = torch.empty(n, self.num_joints * k, hh, hw, dtype=torch.float32, device=self.device)
        weight = torch.empty(n, self.num_joints, 1, dtype=torch.float32, device=self.device)
        if self.target_type == "offset":
            _lib.check(L.udp_target_offset(joints.data_ptr(), vis.data_ptr(), n, self.num_joints, w, h, hw, hh,
                                           float(self.kpd), target.data_ptr(), weight.data_ptr(), _lib.stream_ptr()))
        else:
            _lib.check(L.udp_target_gaussian(joints.data_ptr(), vis.data_ptr(), n, self.num_joints, w, h, hw, hh,
                                             float(self.sigma), target.data_ptr(), weight.data_ptr(), _lib.stream_ptr()))
        metas = [{"joints": p["joints"], "joints_vis": p["joints_vis"], "center": p["center"], "scale": p["scale"],
                  "rotation": p["rotation"]} for p in plans]
        return out, target, weight, metas
