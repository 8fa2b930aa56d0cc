"""Training data path, CPU side: the oracle's __getitem__ and the product's host sampling against the
reference fixture tests/golden/dataset_getitem.npz (the reference's own JointsDataset.__getitem__)."""
import os
import random

import numpy as np
import pytest

from oracle import dataset as o_dataset
from udp_pose_amd import synth
from udp_pose_amd.dataset import JointsPipeline

CASES = (("train_gaussian", True, "gaussian", True), ("train_offset", True, "offset", False),
         ("val_gaussian", False, "gaussian", False))
CUTOUT, HIDE = (1.0, 0.2, 2), (1.0, 0.5, (0, 16, 32, 44, 56))


def oracle_cfg(is_train, tt, aid):
    return {"num_joints": 17, "upper_body_ids": synth.COCO_UPPER_BODY, "flip_pairs": synth.COCO_FLIP_PAIRS,
            "aspect_ratio": 192 / 256, "image_size": (192, 256), "heatmap_size": (48, 64), "is_train": is_train,
            "color_rgb": True, "num_joints_half_body": 8, "prob_half_body": 0.3, "scale_factor": 0.35,
            "rotation_factor": 45, "flip": True, "target_type": tt, "sigma": 2, "kpd": 4.0,
            "cutout": CUTOUT if aid else None, "hide_and_seek": HIDE if aid else None}


def pipeline(is_train, tt, aid, device):
    return JointsPipeline(target_type=tt, is_train=is_train, flip_pairs=synth.COCO_FLIP_PAIRS,
                          upper_body_ids=synth.COCO_UPPER_BODY, cutout=CUTOUT if aid else None,
                          hide_and_seek=HIDE if aid else None, device=device)


@pytest.mark.parametrize("tag,is_train,tt,aid", CASES)
def test_oracle_getitem_matches_reference(golden_dir, tag, is_train, tt, aid):
    g = np.load(os.path.join(golden_dir, "dataset_getitem.npz"))
    for i, rec in enumerate(synth.synth_db()):
        frame = synth.synth_frame_u8(rec["frame_hw"][0], rec["frame_hw"][1], seed=rec["frame_seed"])
        np.random.seed(1000 + i)
        random.seed(2000 + i)
        crop, target, weight, meta = o_dataset.getitem(oracle_cfg(is_train, tt, aid), rec, frame)
        np.testing.assert_array_equal(crop, g[tag + "_u8"][i])
        np.testing.assert_array_equal(target, g[tag + "_target"][i])
        np.testing.assert_array_equal(weight, g[tag + "_weight"][i])
        np.testing.assert_array_equal(meta["joints"], g[tag + "_joints"][i])
        np.testing.assert_array_equal(np.concatenate([meta["center"], meta["scale"]]), g[tag + "_cs"][i])


@pytest.mark.parametrize("tag,is_train,tt,aid", CASES)
def test_host_sampling_matches_reference(golden_dir, tag, is_train, tt, aid):
    """JointsPipeline.sample consumes the RNGs in the reference's order: joints, visibility, center,
    scale and rotation after half-body / jitter / flip are bit-identical."""
    g = np.load(os.path.join(golden_dir, "dataset_getitem.npz"))
    pipe = pipeline(is_train, tt, aid, "cpu")
    flips = 0
    for i, rec in enumerate(synth.synth_db()):
        np.random.seed(1000 + i)
        random.seed(2000 + i)
        p = pipe.sample(rec, rec["frame_hw"][1])
        np.testing.assert_array_equal(p["joints"], g[tag + "_joints"][i])
        np.testing.assert_array_equal(p["joints_vis"], g[tag + "_vis"][i])
        np.testing.assert_array_equal(np.concatenate([p["center"], p["scale"]]), g[tag + "_cs"][i])
        assert float(p["rotation"]) == g[tag + "_rot"][i]
        flips += p["flip"]
    assert (flips > 0) == is_train
