"""``train`` / ``validate`` / criteria with the reference's call shapes (udp-pose_amd/function.py) against the
CPU oracle: validate() fills all_preds / all_boxes as function.py:212-221 does from forward + flip test +
get_final_preds; the criteria equal oracle/loss.py; train() drives HRNetTrainer over a loader."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import decode as odec, flip as oflip, hrnet as ohrnet, loss as oloss   # noqa: E402
from udp_pose_amd import function as ufn, synth                                      # noqa: E402
from udp_pose_amd.model import MODELS                                                 # noqa: E402
from udp_pose_amd.train import HRNetTrainer                                           # noqa: E402
from udp_pose_amd.transforms import COCO_FLIP_PAIRS                                   # noqa: E402

EXTRA = synth.scaled_extra(32, modules=(1, 1, 1), blocks=1)


class _DS:
    flip_pairs = COCO_FLIP_PAIRS

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def _loader(n_batches, bs, tt, seed):
    out = []
    for b in range(n_batches):
        x = torch.from_numpy(synth.synth_crops(bs, 128, 96, seed=seed + b))
        k = 3 if tt == "offset" else 1
        tg = torch.from_numpy(synth.synth_heatmaps(bs, 17, 32, 24, seed=seed + 50 + b, channels_per_joint=k))
        tw = torch.from_numpy((np.random.default_rng(seed + b).random((bs, 17, 1)) > 0.2).astype(np.float32))
        c, s = synth.synth_center_scale(bs, seed=seed + b)
        out.append((x, tg, tw, {"center": c, "scale": s, "score": np.full(bs, 0.5 + 0.1 * b),
                                "image": ["img_%d_%d" % (b, i) for i in range(bs)]}))
    return out


@pytest.mark.parametrize("tt,post", [("gaussian", True), ("gaussian", False), ("offset", False)])
def test_validate_matches_oracle_pipeline(tt, post):
    cfg = {"MODEL": {"EXTRA": EXTRA, "NUM_JOINTS": 17, "TARGET_TYPE": tt}, "TEST": {"FLIP_TEST": True, "POST_PROCESS": post},
           "LOSS": {"KPD": 4.0}}
    sd = synth.synth_state_dict(EXTRA, 17, tt, seed=31)
    loader = _loader(2, 3, tt, 60)
    ohrnet.hrnet_forward(sd, EXTRA, loader[0][0], calibrate=True)
    net = MODELS["pose_hrnet"](cfg, is_train=False).load_state_dict(sd).to("cuda")
    crit = ufn.JointsMSELoss_offset() if tt == "offset" else ufn.JointsMSELoss()
    all_preds, all_boxes, paths, mean_loss = ufn.validate(cfg, loader, _DS(6), net, crit)
    ref_p, ref_l, idx = [], 0.0, 0
    for x, tg, tw, meta in loader:
        y = ohrnet.hrnet_forward(sd, EXTRA, torch.cat([x, torch.flip(x, dims=[3])])).numpy()
        hm = oflip.flip_fuse(y[:3], y[3:], oflip.COCO_FLIP_PAIRS, tt == "offset")
        if tt == "offset":
            l_hm, l_os, _ = oloss.joints_mse_loss_offset(hm, tg.numpy(), tw.numpy())
            ref_l += (l_hm + l_os) * 3
        else:
            ref_l += oloss.joints_mse_loss(hm, tg.numpy(), tw.numpy())[0] * 3
        p, m, _, _ = odec.get_final_preds(tt, post, 4.0, hm.copy(), meta["center"], meta["scale"])
        ref_p.append(np.concatenate([p, m], axis=2))
        np.testing.assert_allclose(all_boxes[idx:idx + 3, 4], np.prod(meta["scale"] * 200, 1))
        np.testing.assert_allclose(all_boxes[idx:idx + 3, 5], meta["score"])
        idx += 3
    ref_p = np.concatenate(ref_p)
    assert paths == ["img_%d_%d" % (b, i) for b in range(2) for i in range(3)]
    np.testing.assert_allclose(mean_loss, ref_l / 6, rtol=1e-4)
    np.testing.assert_allclose(all_preds[..., 2], ref_p[..., 2], atol=1e-3)
    err = np.abs(all_preds[..., :2] - ref_p[..., :2]).max(axis=2)
    assert np.median(err) < 1e-2 and (err < 0.5).mean() > 0.9, (np.median(err), (err < 0.5).mean())


def test_train_epoch_drives_the_trainer():
    cfg = {"MODEL": {"EXTRA": EXTRA, "NUM_JOINTS": 17, "TARGET_TYPE": "gaussian"}}
    sd = synth.synth_state_dict(EXTRA, 17, "gaussian", seed=32)
    loader = _loader(1, 4, "gaussian", 70) * 6                  # the same batch six times: the loss must fall
    tr = HRNetTrainer(cfg, sd, device="cuda", lr=1e-3)
    opt = ufn.Adam([], lr=1e-3)
    first = ufn.train(cfg, loader[:1], tr, ufn.JointsMSELoss(), opt, 0)
    last = None
    for _ in range(3):
        last = ufn.train(cfg, loader, tr, ufn.JointsMSELoss(), opt, 1)
    assert last < 0.8 * first and tr.step_count == 1 + 18


def test_integration_md_training_snippet_runs_verbatim():
    """The training sequence of INTEGRATION.md (= tools/train.py:91,116-125,164,183-196 with the import swaps):
    MODELS[name](cfg, is_train=True) -> criterion -> get_optimizer -> train() -> validate() on the same object;
    the weights validate() runs on are the trained ones; a scheduler's lr change reaches the step; unsupported
    criteria / optimizers are refused, not ignored."""
    from udp_pose_amd.config import default_config
    from udp_pose_amd.model import MODELS
    from udp_pose_amd.function import JointsMSELoss, get_optimizer, train, validate
    cfg = default_config()
    cfg.MODEL.NAME = "pose_hrnet"
    cfg.MODEL.EXTRA = EXTRA
    cfg.MODEL.NUM_JOINTS = 17
    cfg.MODEL.TARGET_TYPE = "gaussian"
    cfg.MODEL.INIT_WEIGHTS = True
    cfg.MODEL.PRETRAINED = ""
    cfg.TRAIN.LR = 1e-3
    cfg.TRAIN.OPTIMIZER = "adam"
    cfg.TEST.FLIP_TEST = True
    cfg.TEST.POST_PROCESS = True
    torch.manual_seed(5)
    train_loader = _loader(1, 4, "gaussian", 70) * 4
    valid_loader = _loader(2, 3, "gaussian", 71)
    valid_dataset = _DS(6)

    model = MODELS[cfg.MODEL.NAME](cfg, is_train=True)
    model = model.cuda()
    criterion = JointsMSELoss(use_target_weight=True)
    optimizer = get_optimizer(cfg, model)
    w0 = model.state_dict()["conv1.weight"].clone()
    assert abs(float(w0.std()) - 1e-3) < 2e-4 and float(model.state_dict()["bn1.weight"].mean()) == 1.0   # init_weights
    losses = []
    for epoch in range(3):
        losses.append(train(cfg, train_loader, model, criterion, optimizer, epoch, None, None, None))
        all_preds, all_boxes, paths, val_loss = validate(cfg, valid_loader, valid_dataset, model, criterion, None, None, None)
        assert np.isfinite(all_preds).all() and np.isfinite(val_loss)
    assert losses[-1] < losses[0]
    sd = model.state_dict()
    assert not torch.equal(sd["conv1.weight"], w0)                      # validate() saw trained weights
    assert set(sd) == set(synth.hrnet_param_shapes(EXTRA, 17, "gaussian"))
    # the eval-mode forward runs the trained weights: same heat-maps as a fresh inference model loaded from them
    x = valid_loader[0][0].cuda()
    fresh = MODELS["pose_hrnet"](cfg, is_train=False).load_state_dict(sd).cuda().eval()
    torch.testing.assert_close(model.eval()(x).clone(), fresh(x).clone(), rtol=0, atol=0)
    # a scheduler's lr reaches the kernel: with lr = 0 the parameters stop moving
    optimizer.param_groups[0]["lr"] = 0.0
    before = model.parameters()[0].clone()
    train(cfg, train_loader[:1], model, criterion, optimizer, 3, None, None, None)
    assert torch.equal(before, model.parameters()[0])
    # a real torch.optim.Adam over model.parameters() is accepted; SGD / weight decay / other criteria are refused
    train(cfg, train_loader[:1], model, criterion, torch.optim.Adam(model.parameters(), lr=1e-4), 4, None, None, None)
    for bad in (torch.optim.SGD(model.parameters(), lr=0.1), torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)):
        with pytest.raises(NotImplementedError):
            train(cfg, train_loader[:1], model, criterion, bad, 5, None, None, None)

    class JointsOHKMMSELoss:
        use_target_weight = True
    with pytest.raises(NotImplementedError):
        train(cfg, train_loader[:1], model, JointsOHKMMSELoss(), optimizer, 5, None, None, None)
    cfg.TRAIN.OPTIMIZER = "sgd"
    with pytest.raises(NotImplementedError):
        get_optimizer(cfg, model)


def test_accuracy_matches_reference_fixture(golden_dir):
    """evaluate.accuracy (PCK on heat-maps): device arg-max + host bookkeeping vs the reference's own function."""
    import os
    g = np.load(os.path.join(golden_dir, "accuracy.npz"))
    for k, (seed, shift) in enumerate(((3, 0), (4, 3), (5, 6))):
        pred, tgt = synth.synth_accuracy_case(seed, shift)
        acc, avg, cnt, p = ufn.accuracy(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda())
        np.testing.assert_allclose(acc, g["acc%d" % k], rtol=0, atol=1e-12)
        assert abs(avg - float(g["avg%d" % k])) < 1e-12 and cnt == int(g["cnt%d" % k])
        np.testing.assert_array_equal(p, g["p%d" % k])
