"""Training data path on the GPU (udp_warp_affine_ex, udp_aid_apply, udp_target_*) against the reference's
JointsDataset.__getitem__ fixture: crops bit-exact (uint8 warp through the OpenCV fixed-point rule of the
stand-in + ToTensor/Normalize in fp32), AID masks exact, targets 1e-7."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from udp_pose_amd import synth                              # noqa: E402
from udp_pose_amd.pose_engine import IMAGENET_MEAN, IMAGENET_STD   # noqa: E402
from test_dataset_cpu import CASES, pipeline                # noqa: E402


def _normalize(u8):
    x = u8.astype(np.float32) / np.float32(255.0)
    mean = np.array(IMAGENET_MEAN, np.float32)
    std = np.array(IMAGENET_STD, np.float32)
    return ((x - mean) / std).transpose(0, 3, 1, 2)


@pytest.mark.parametrize("tag,is_train,tt,aid", CASES)
def test_device_batch_matches_reference_getitem(golden_dir, tag, is_train, tt, aid):
    g = np.load(os.path.join(golden_dir, "dataset_getitem.npz"))
    db = synth.synth_db()
    frames = [torch.from_numpy(synth.synth_frame_u8(r["frame_hw"][0], r["frame_hw"][1], seed=r["frame_seed"])).cuda()
              for r in db]
    pipe = pipeline(is_train, tt, aid, "cuda")
    inputs, targets, weights = [], [], []
    for i, rec in enumerate(db):                 # one record per call: the fixture re-seeds per item
        np.random.seed(1000 + i)
        random.seed(2000 + i)
        x, t, w, metas = pipe.batch([rec], [frames[i]])
        inputs.append(x.cpu().numpy())
        targets.append(t.cpu().numpy())
        weights.append(w.cpu().numpy())
        np.testing.assert_array_equal(metas[0]["joints"], g[tag + "_joints"][i])
    x = np.concatenate(inputs)
    np.testing.assert_array_equal(x, _normalize(g[tag + "_u8"]))
    np.testing.assert_allclose(np.concatenate(targets), g[tag + "_target"], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(np.concatenate(weights), g[tag + "_weight"])
    if aid:
        assert (g[tag + "_u8"] == 0).mean() > 0.2           # the AID masks really are in the fixture


def test_device_batch_of_many_equals_singles():
    """A whole batch in one call == the same records one by one (same RNG stream)."""
    db = synth.synth_db()
    frames = [torch.from_numpy(synth.synth_frame_u8(r["frame_hw"][0], r["frame_hw"][1], seed=r["frame_seed"])).cuda()
              for r in db]
    pipe = pipeline(True, "gaussian", True, "cuda")
    np.random.seed(5)
    random.seed(6)
    xb, tb, wb, _ = pipe.batch(db, frames)
    np.random.seed(5)
    random.seed(6)
    for i, rec in enumerate(db):
        x, t, w, _ = pipe.batch([rec], [frames[i]])
        torch.testing.assert_close(xb[i:i + 1], x, rtol=0, atol=0)
        torch.testing.assert_close(tb[i:i + 1], t, rtol=0, atol=0)
