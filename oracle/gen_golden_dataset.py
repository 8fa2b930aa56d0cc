#!/usr/bin/env python3
"""Golden for the training data path: the REFERENCE's JointsDataset.__getitem__
(deep_hrnet/lib/dataset/JointsDataset.py:170-256: half-body / scale / rotation / flip sampling,
get_warpmatrix + cv2.warpAffine, rotate_points, AID Cutout / HideAndSeek (lib/utils/transforms.py:144-224),
generate_target) run on a bare instance in the build container.

cv2 is the stand-in of oracle/cv2_standin.py (imread is stubbed with synthetic frames, cvtColor
restated): the warp's pixel arithmetic is "parity unpinned" against real OpenCV as everywhere else;
RNG call order, augmentation parameters, flips, AID masks, joints and targets are the reference's own.

    python oracle/gen_golden_dataset.py      # writes tests/golden/dataset_getitem.npz
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as gg                                   # noqa: E402
from udp_pose_amd import synth                             # noqa: E402

FLIP_PAIRS, UPPER, make_db = synth.COCO_FLIP_PAIRS, synth.COCO_UPPER_BODY, synth.synth_db


def main():
    _, _, _, tfm, jd = gg.load_reference()
    cv2 = jd.cv2
    frames = {}
    cv2.IMREAD_COLOR, cv2.IMREAD_IGNORE_ORIENTATION, cv2.COLOR_BGR2RGB = 1, 128, 4
    cv2.imread = lambda path, flags=None: frames[path].copy()
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    db = make_db()
    for rec in db:
        frames[rec["image"]] = synth.synth_frame_u8(rec["frame_hw"][0], rec["frame_hw"][1], seed=rec["frame_seed"])
    out = {}
    for tag, is_train, tt, aid in (("train_gaussian", True, "gaussian", True), ("train_offset", True, "offset", False),
                                   ("val_gaussian", False, "gaussian", False)):
        ds = jd.JointsDataset.__new__(jd.JointsDataset)
        ds.num_joints, ds.pixel_std, ds.flip_pairs, ds.upper_body_ids = 17, 200, FLIP_PAIRS, UPPER
        ds.is_train, ds.data_format = is_train, "jpg"
        ds.scale_factor, ds.rotation_factor, ds.flip = 0.35, 45, True                       # w32 yaml :17-22
        ds.num_joints_half_body, ds.prob_half_body, ds.color_rgb = 8, 0.3, True
        ds.aspect_ratio = 192 / 256
        ds.cutout = tfm.Cutout(1.0, 0.2, 2) if aid else None
        ds.hide_and_seek = tfm.HideAndSeek(1.0, 0.5, (0, 16, 32, 44, 56)) if aid else None
        ds.target_type, ds.image_size, ds.heatmap_size = tt, np.array([192, 256]), np.array([48, 64])
        ds.sigma, ds.use_different_joints_weight, ds.joints_weight, ds.kpd = 2, False, 1, 4.0
        ds.db = db
        captured = []
        ds.transform = lambda img: captured.append(img.copy()) or img
        tg, tw, joints, jvis, cs, rot = [], [], [], [], [], []
        for idx in range(len(db)):
            np.random.seed(1000 + idx)
            random.seed(2000 + idx)
            _, target, target_weight, meta = ds[idx]
            tg.append(target.numpy())
            tw.append(target_weight.numpy())
            joints.append(meta["joints"])
            jvis.append(meta["joints_vis"])
            cs.append(np.concatenate([meta["center"], meta["scale"]]))
            rot.append(float(meta["rotation"]))
        out[tag + "_u8"] = np.stack(captured)
        out[tag + "_target"] = np.stack(tg)
        out[tag + "_weight"] = np.stack(tw)
        out[tag + "_joints"] = np.stack(joints)
        out[tag + "_vis"] = np.stack(jvis)
        out[tag + "_cs"] = np.stack(cs)
        out[tag + "_rot"] = np.array(rot)
        print(tag, "rot", np.round(rot, 1), "zero frac", float((out[tag + "_u8"] == 0).mean()))
    np.savez_compressed(os.path.join(gg.OUT, "dataset_getitem.npz"), **out)


if __name__ == "__main__":
    main()
