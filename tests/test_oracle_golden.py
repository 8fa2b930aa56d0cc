"""Oracle vs the reference's own outputs (tests/golden, oracle/gen_golden.py).

These pin the CPU oracle: every fixture was produced by importing the
reference's Python files in the build container.  CPU only.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import data as odata
from oracle import decode as odec
from oracle import flip as oflip
from oracle import hrnet as ohrnet
from oracle import loss as oloss
from udp_pose_amd import synth


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ---------------------------------------------------------------- decode
@pytest.mark.parametrize("tt,post,k", [("gaussian", False, 1), ("gaussian", True, 1), ("offset", False, 3)])
def test_get_final_preds_matches_reference(golden_dir, tt, post, k):
    g = _g(golden_dir, "decode.npz")
    hm = synth.synth_heatmaps(4, 17, 64, 48, seed=21, channels_per_joint=k)
    hm[0, 0] = -np.abs(hm[0, 0]) - 0.1
    hm[0, 1 * k, 10, 7] = hm[0, 1 * k].max() + 0.5
    hm[0, 1 * k, 30, 40] = hm[0, 1 * k, 10, 7]
    hm[1, 2 * k, 0, 0] = 2.0
    hm[1, 3 * k, 63, 47] = 2.0
    hm[2, 4 * k] = 0.25
    with np.errstate(all="ignore"):
        preds, maxvals, pin, idx = odec.get_final_preds(tt, post, 4.0, hm.copy(), g["center"], g["scale"])
    tag = "%s%s" % (tt, "_post" if post else "")
    assert preds.dtype == g["preds_" + tag].dtype
    np.testing.assert_array_equal(maxvals, g["maxvals_" + tag])
    # nan (flat map -> 0/0 in the min/max rescale) must be nan in both
    np.testing.assert_allclose(preds, g["preds_" + tag], rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(pin, g["pin_" + tag], rtol=0, atol=1e-9, equal_nan=True)
    if tt == "gaussian" and not post:
        # tie -> first flat index (10*48+7), all-negative map -> coords zeroed
        assert idx[0, 1] == 10 * 48 + 7
        assert np.all(preds[0, 0] == odec.transform_preds(np.zeros((1, 2), np.float32), g["center"][0],
                                                          g["scale"][0], [48, 64]).astype(np.float32))


def test_get_max_preds_matches_reference(golden_dir):
    g = _g(golden_dir, "decode.npz")
    hm = synth.synth_heatmaps(4, 17, 64, 48, seed=21)
    p, m, idx = odec.get_max_preds(hm)
    np.testing.assert_array_equal(p, g["maxpreds"])
    np.testing.assert_array_equal(m, g["maxpreds_vals"])
    np.testing.assert_array_equal(idx, (p[..., 1] * 48 + p[..., 0]).astype(np.int64))


def test_dark_recovers_subpixel_mean():
    """KAT: a noiseless Gaussian at a sub-pixel mean decodes to that mean."""
    ys = np.arange(64, dtype=np.float64)[:, None]
    xs = np.arange(48, dtype=np.float64)[None, :]
    mus = [(20.3, 30.7), (10.5, 12.25), (40.1, 50.9)]
    hm = np.stack([np.exp(-((xs - mx) ** 2 + (ys - my) ** 2) / 8.0) for mx, my in mus])[None].astype(np.float32)
    coords, _, _ = odec.get_max_preds(hm)
    res = odec.post(coords, hm.copy())
    # blur widens sigma but keeps the mean; min/max rescale + clip bias it slightly
    np.testing.assert_allclose(res[0], np.asarray(mus), atol=0.05)


# ---------------------------------------------------------------- flip
def test_flip_back_matches_reference(golden_dir):
    g = _g(golden_dir, "flip.npz")
    np.testing.assert_array_equal(oflip.flip_back(g["a"], oflip.COCO_FLIP_PAIRS), g["fa"])
    np.testing.assert_array_equal(oflip.flip_back_offset(g["b"], oflip.COCO_FLIP_PAIRS), g["fb"])
    # involution
    np.testing.assert_array_equal(oflip.flip_back(g["fa"], oflip.COCO_FLIP_PAIRS), g["a"])
    np.testing.assert_array_equal(oflip.flip_back_offset(g["fb"], oflip.COCO_FLIP_PAIRS), g["b"])


# ---------------------------------------------------------------- data path
def test_warpmatrix_rotate_points_match_reference(golden_dir):
    g = _g(golden_dir, "data.npz")
    rng = np.random.Generator(np.random.PCG64(41))
    img = np.array([192, 256])
    for i, case in enumerate(g["warp_cases"]):
        theta = float(rng.uniform(-60, 60)) if i else 0.0
        c = rng.uniform(50, 400, 2).astype(np.float32)
        s = rng.uniform(0.5, 2.5, 2).astype(np.float32)
        assert theta == case[0]
        m = odata.get_warpmatrix(theta, c * 2.0, img - 1.0, s)
        np.testing.assert_array_equal(m, g["warp_mats"][i])
        pts = rng.uniform(0, 500, (17, 2)).astype(np.float32)
        q = odata.rotate_points(pts, theta, c, img, s, False)
        np.testing.assert_array_equal(q, g["rot_pts_out"][i])
        # round trip: the dst->src matrix applied to rotate_points output gives the joints back
        back = q.astype(np.float64) @ m[:, :2].T.astype(np.float64) + m[:, 2].astype(np.float64)
        np.testing.assert_allclose(back, pts, atol=2e-3)


@pytest.mark.parametrize("tt", ["gaussian", "offset"])
def test_generate_target_matches_reference(golden_dir, tt):
    g = _g(golden_dir, "data.npz")
    for k in range(3):
        t, w = odata.generate_target(g["tgt_joints"][k], g["tgt_vis"][k], tt, [192, 256], [48, 64])
        np.testing.assert_array_equal(w, g["target_weight_" + tt][k])
        np.testing.assert_array_equal(t, g["target_" + tt][k])
    if tt == "gaussian":
        assert g["target_weight_gaussian"][0, 2, 0] == 0          # joint fully outside the map


def test_box_to_center_scale_closed_form():
    b = np.array([[100, 50, 200, 350], [10, 20, 410, 120]], np.float32)
    cs = odata.box_to_center_scale(b, [192, 256])
    np.testing.assert_allclose(cs[0], [150, 200, 225 / 200 * 1.25, 300 / 200 * 1.25], rtol=1e-6)
    np.testing.assert_allclose(cs[1], [210, 70, 400 / 200 * 1.25, (400 / 0.75) / 200 * 1.25], rtol=1e-6)


def test_engine_affine_is_uniform_biased_scale():
    m = odata.engine_affine(np.array([320.0, 240.0]), np.array([1.5, 2.0]), [192, 256])
    k = 192 / 300.0                                            # dst_w / src_w, uniform, no W-1
    np.testing.assert_allclose(m, [[k, 0, 96 - 320 * k], [0, k, 128 - 240 * k]], atol=1e-4)


# ---------------------------------------------------------------- loss
def test_losses_match_reference(golden_dir):
    g = _g(golden_dir, "loss.npz")
    l, gr = oloss.joints_mse_loss(g["mse_pred"], g["mse_gt"], g["mse_w"])
    np.testing.assert_allclose(l, g["mse_loss"], rtol=1e-6)
    np.testing.assert_allclose(gr, g["mse_grad"], rtol=1e-5, atol=1e-9)
    lh, lo, gr = oloss.joints_mse_loss_offset(g["off_pred"], g["off_gt"], g["off_w"])
    np.testing.assert_allclose(lh, g["off_loss_hm"], rtol=1e-6)
    np.testing.assert_allclose(lo, g["off_loss_os"], rtol=1e-6)
    np.testing.assert_allclose(gr, g["off_grad"], rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------- network
@pytest.mark.parametrize("tag,extra,tt", [("w32_offset", synth.W32_EXTRA, "offset"),
                                          ("w32_gaussian", synth.W32_EXTRA, "gaussian"),
                                          ("w48_gaussian", synth.scaled_extra(48), "gaussian")])
def test_state_dict_contract(golden_dir, tag, extra, tt):
    """Key names and shapes equal the reference module's state_dict()."""
    with open(os.path.join(golden_dir, "hrnet_keys_%s.json" % tag)) as f:
        ref = json.load(f)
    ours = synth.hrnet_param_shapes(extra, 17, tt)
    assert list(ours.keys()) == list(ref.keys())
    for k, s in ours.items():
        assert list(s) == ref[k], k


def test_mini_hrnet_matches_reference(golden_dir):
    g = _g(golden_dir, "hrnet_mini.npz")
    extra = synth.scaled_extra(16, modules=(1, 2, 2), blocks=2)
    calib = {k[len("calib_"):]: g[k] for k in g.files if k.startswith("calib_")}
    sd = synth.synth_state_dict(extra, 5, "gaussian", seed=1, bn_calib=calib)
    x = torch.from_numpy(synth.synth_crops(2, 96, 64, seed=3))
    taps = {}
    y = ohrnet.hrnet_forward(sd, extra, x, taps=taps).numpy()
    np.testing.assert_allclose(y, g["out"], rtol=0, atol=1e-5)
    assert y.shape == (2, 5, 24, 16)
    np.testing.assert_allclose(taps["layer1"].numpy(), g["tap_layer1"], atol=1e-5)
    for k in ("stage2.0", "stage2.1", "stage3.0", "stage3.1", "stage3.2", "stage4.0"):
        np.testing.assert_allclose(taps[k].numpy(), g["tap_" + k], atol=1e-4)
    assert g["tap_stage4.0"].shape[1] == 4 * 16                  # last fuse widens to 4C


def _psa_mini(golden_dir):
    g = _g(golden_dir, "hrnet_psa_mini.npz")
    extra = synth.scaled_extra(32, modules=(1, 2, 1), blocks=2)
    calib = {k[len("calib_"):]: g[k] for k in g.files if k.startswith("calib_")}
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=6, bn_calib=calib, psa=True)
    return g, extra, sd


def test_psa_mini_matches_reference(golden_dir):
    """pose_hrnet_psa (PSA_s after conv1 of every BasicBlock): reference module heat-maps + key contract."""
    g, extra, sd = _psa_mini(golden_dir)
    x = torch.from_numpy(synth.synth_crops(2, 128, 96, seed=24))
    y = ohrnet.hrnet_forward(sd, extra, x).numpy()
    np.testing.assert_allclose(y, g["out"], rtol=0, atol=1e-5)
    ours = sorted("%s:%s" % (k, "x".join(map(str, v))) for k, v in synth.hrnet_param_shapes(extra, 17, "gaussian", psa=True).items())
    assert ours == [str(k) for k in g["keys"]]
    # the attention is not a no-op on these weights
    plain = {k: v for k, v in sd.items() if ".deattn." not in k}
    assert np.abs(ohrnet.hrnet_forward(plain, extra, x).numpy() - y).max() > 1e-2


def test_w32_matches_reference(golden_dir):
    g = _g(golden_dir, "hrnet_w32_gaussian.npz")
    calib = dict(_g(golden_dir, "bn_calib_w32_gaussian.npz"))
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0, bn_calib=calib)
    x = torch.from_numpy(synth.synth_crops(1, 256, 192, seed=5))
    y = ohrnet.hrnet_forward(sd, synth.W32_EXTRA, x).numpy()
    np.testing.assert_allclose(y, g["out"][:1], rtol=0, atol=1e-4)


def test_w48_384x288_matches_reference(golden_dir):
    """Config 4 network (widths 48/96/192/384, 384x288 input): oracle == reference module."""
    g = _g(golden_dir, "hrnet_w48_gaussian.npz")
    calib = dict(_g(golden_dir, "bn_calib_w48_gaussian.npz"))
    extra = synth.scaled_extra(48)
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=2, bn_calib=calib)
    x = torch.from_numpy(synth.synth_crops(1, 384, 288, seed=6))
    y = ohrnet.hrnet_forward(sd, extra, x).numpy()
    assert y.shape == (1, 17, 96, 72)
    np.testing.assert_allclose(y, g["out"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("och", [17, 51])
def test_rsn18_matches_reference(golden_dir, och):
    """Config 5 backbone: oracle RSN-18 == reference module; state_dict contract."""
    from oracle import rsn as orsn
    with open(os.path.join(golden_dir, "rsn18_keys_%d.json" % och)) as f:
        ref = json.load(f)
    ours = synth.rsn18_param_shapes(och)
    assert list(ours.keys()) == list(ref.keys()) and all(list(ours[k]) == ref[k] for k in ours)
    calib = dict(_g(golden_dir, "bn_calib_rsn18_%d.npz" % och))
    sd = synth.synth_rsn18_state_dict(och, seed=4, bn_calib=calib)
    x = torch.from_numpy(synth.synth_crops(1, 256, 192, seed=8))
    y = orsn.rsn_forward(sd, x).numpy()
    np.testing.assert_allclose(y, _g(golden_dir, "rsn18_%d.npz" % och)["out"], rtol=0, atol=1e-5)


def test_accuracy_matches_reference(golden_dir):
    """oracle/evaluate.accuracy == lib/core/evaluate.py:40-73 (fixture from the reference's function)."""
    from oracle import evaluate as oev
    g = _g(golden_dir, "accuracy.npz")
    for k, (seed, shift) in enumerate(((3, 0), (4, 3), (5, 6))):
        pred, tgt = synth.synth_accuracy_case(seed, shift)
        acc, avg, cnt, p = oev.accuracy(pred, tgt)
        np.testing.assert_allclose(acc, g["acc%d" % k], rtol=0, atol=1e-12)
        assert abs(avg - float(g["avg%d" % k])) < 1e-12 and cnt == int(g["cnt%d" % k])
        np.testing.assert_array_equal(p, g["p%d" % k])
    assert 0.1 < float(g["avg1"]) < 0.9                       # a case that is neither all right nor all wrong


def test_oracle_reproduces_the_trained_peaked_fixture(golden_dir):
    """tests/golden/hrnet_peaked.npz (oracle/gen_golden_peaked.py): a mini HRNet TRAINED by the reference module +
    reference JointsMSELoss + torch.optim.Adam until its heat-maps have clear peaks.  The oracle's forward, flip
    fusion and get_final_preds reproduce the reference's outputs on the held-out crops."""
    import torch
    from oracle import decode as odec, flip as oflip, hrnet as ohrnet
    from udp_pose_amd import synth
    g = np.load(os.path.join(golden_dir, "hrnet_peaked.npz"))
    extra = synth.scaled_extra(16, modules=(1, 1, 1), blocks=1)
    sd = {k[3:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k]) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(synth.normalize_u8(g["crops_u8"]))
    out = ohrnet.hrnet_forward(sd, extra, x).numpy()
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-5)
    assert float(g["argmax_hit_rate"]) > 0.9 and g["out"].max() > 1.0            # peaked, not noise-like
    out_flip = ohrnet.hrnet_forward(sd, extra, torch.flip(x, dims=[3])).numpy()
    fused = oflip.flip_fuse(out, out_flip, oflip.COCO_FLIP_PAIRS, False)
    np.testing.assert_allclose(fused, g["fused"], rtol=0, atol=1e-5)
    p, m, pin, _ = odec.get_final_preds("gaussian", True, 4.0, g["fused"].copy(), g["center"], g["scale"])
    np.testing.assert_allclose(p, g["preds"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(m, g["maxvals"])
    np.testing.assert_allclose(pin, g["preds_in_input_space"], rtol=0, atol=1e-9)
