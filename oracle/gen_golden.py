#!/usr/bin/env python3
"""Write tests/golden/*.npz from the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference; the GPU box never
sees it).  The reference's torch/NumPy files are imported from where they lie
-- nothing is copied -- with two accommodations recorded in SURVEY.md 8c:
  * ``cv2`` is not installed: ``sys.modules['cv2']`` is pointed at
    oracle/cv2_standin.py (our restatement of OpenCV's documented rules), so
    the blur inside the reference's ``post``/offset decode is that stand-in;
  * ``np.float`` (removed from NumPy >= 1.24, used at inference.py:136) is
    aliased to ``float``.
``lib/models/__init__.py`` is not executed (it imports torchvision): the model
file is loaded under a synthetic parent package.

Each fixture holds inputs (or the seed that regenerates them) and the
reference's outputs.  tests/ compare the oracle against these, and the HIP
path against the oracle and against these.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/deep_hrnet"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import cv2_standin, hrnet as o_hrnet  # noqa: E402
from udp_pose_amd import synth  # noqa: E402


class AttrDict(dict):
    """yacs-like read access over the YAML dict (cfg.MODEL.EXTRA and cfg['MODEL'])."""
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) else v

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    sys.modules["cv2"] = cv2_standin
    if not hasattr(np, "float"):
        np.float = float                      # inference.py:136
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [os.path.join(REF, "lib", "models")]
    sys.modules["refmodels"] = pkg
    _load("refmodels.PSA", os.path.join(REF, "lib/models/PSA.py"), "refmodels")
    pose_hrnet = _load("refmodels.pose_hrnet", os.path.join(REF, "lib/models/pose_hrnet.py"), "refmodels")
    inference = _load("ref_inference", os.path.join(REF, "lib/core/inference.py"))
    loss = _load("ref_loss", os.path.join(REF, "lib/core/loss.py"))
    sys.path.insert(0, os.path.join(REF, "lib"))
    transforms = _load("utils.transforms", os.path.join(REF, "lib/utils/transforms.py"))
    upkg = types.ModuleType("utils")
    upkg.__path__ = [os.path.join(REF, "lib", "utils")]
    upkg.transforms = transforms
    sys.modules["utils"] = upkg
    jd = _load("ref_joints_dataset", os.path.join(REF, "lib/dataset/JointsDataset.py"))
    return pose_hrnet, inference, loss, transforms, jd


def model_cfg(extra, num_joints, target_type):
    return AttrDict({"MODEL": {"EXTRA": extra, "TARGET_TYPE": target_type, "NUM_JOINTS": num_joints,
                               "INIT_WEIGHTS": False, "PRETRAINED": ""}})


def sd_sha256(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.numpy()).tobytes())
    return h.hexdigest()


def ref_hrnet(pose_hrnet, extra, nj, tt, sd):
    net = pose_hrnet.get_pose_net(model_cfg(extra, nj, tt), is_train=False)
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net.eval()
    return net


def gen_hrnet(pose_hrnet):
    # (c) weight-file contract: key names + shapes of the reference module
    for tag, extra, nj, tt in (("w32_offset", synth.W32_EXTRA, 17, "offset"),
                               ("w32_gaussian", synth.W32_EXTRA, 17, "gaussian"),
                               ("w48_gaussian", synth.scaled_extra(48), 17, "gaussian")):
        net = pose_hrnet.get_pose_net(model_cfg(extra, nj, tt), is_train=False)
        keys = {k: list(v.shape) for k, v in net.state_dict().items()}
        with open(os.path.join(OUT, "hrnet_keys_%s.json" % tag), "w") as f:
            json.dump(keys, f)
        print("keys", tag, len(keys), sum(int(np.prod(s)) for k, s in keys.items()
                                         if not k.endswith("num_batches_tracked") and "running" not in k))

    # (a) mini HRNet, all taps: pins the graph incl. the 4xC last-fuse quirk
    extra = synth.scaled_extra(16, modules=(1, 2, 2), blocks=2)
    sd = synth.synth_state_dict(extra, 5, "gaussian", seed=1)
    x = torch.from_numpy(synth.synth_crops(2, 96, 64, seed=3))
    yc = o_hrnet.hrnet_forward(sd, extra, x, calibrate=True)
    calib = {k: v.numpy() for k, v in sd.items() if "running_" in k}
    calib["final_layer.scale"] = np.float32(0.25 / float(yc.std()))
    sd = synth.synth_state_dict(extra, 5, "gaussian", seed=1, bn_calib=calib)
    net = ref_hrnet(pose_hrnet, extra, 5, "gaussian", sd)
    taps = {}
    hooks = []
    for name in ("layer1", "stage2", "stage3", "stage4"):
        def mk(nm):
            def hook(_m, _i, o):
                if isinstance(o, (list, tuple)):
                    for b, t in enumerate(o):
                        taps["%s.%d" % (nm, b)] = t.detach().numpy().copy()
                else:
                    taps[nm] = o.detach().numpy().copy()
            return hook
        hooks.append(getattr(net, name).register_forward_hook(mk(name)))
    with torch.no_grad():
        y = net(x).numpy()
    np.savez_compressed(os.path.join(OUT, "hrnet_mini.npz"), out=y,
                        **{"tap_" + k: v for k, v in taps.items()},
                        **{"calib_" + k: v for k, v in calib.items()})
    print("mini out", y.shape, float(np.abs(y).max()), "taps", sorted(taps))

    # (b) full W32, gaussian + offset heads: calibrated BN stats + heat-maps of 2 crops
    for tt, nimg in (("gaussian", 2), ("offset", 1)):
        sd = synth.synth_state_dict(synth.W32_EXTRA, 17, tt, seed=0)
        xc = torch.from_numpy(synth.synth_crops(4, 256, 192, seed=11))
        yc = o_hrnet.hrnet_forward(sd, synth.W32_EXTRA, xc, calibrate=True)
        calib = {k: v.numpy() for k, v in sd.items() if "running_" in k}
        calib["final_layer.scale"] = np.float32(0.25 / float(yc.std()))
        sd = synth.synth_state_dict(synth.W32_EXTRA, 17, tt, seed=0, bn_calib=calib)
        np.savez_compressed(os.path.join(OUT, "bn_calib_w32_%s.npz" % tt), **calib)
        net = ref_hrnet(pose_hrnet, synth.W32_EXTRA, 17, tt, sd)
        x = torch.from_numpy(synth.synth_crops(nimg, 256, 192, seed=5))
        with torch.no_grad():
            y = net(x).numpy()
        np.savez_compressed(os.path.join(OUT, "hrnet_w32_%s.npz" % tt), out=y,
                            sha256=np.frombuffer(sd_sha256(sd).encode(), dtype=np.uint8))
        print("w32", tt, y.shape, "absmax", float(np.abs(y).max()), "std", float(y.std()))


def gen_hrnet_w48(pose_hrnet):
    """Config 4: pose_hrnet_w48 384x288 (widths 48/96/192/384), one crop."""
    extra = synth.scaled_extra(48)
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=2)
    xc = torch.from_numpy(synth.synth_crops(2, 384, 288, seed=13))
    yc = o_hrnet.hrnet_forward(sd, extra, xc, calibrate=True)
    calib = {k: v.numpy().astype(np.float16 if "running_var" not in k else np.float32)
             for k, v in sd.items() if "running_" in k}
    calib = {k: v.astype(np.float32) for k, v in calib.items()}      # fp16-rounded means: smaller fixture
    calib["final_layer.scale"] = np.float32(0.25 / float(yc.std()))
    np.savez_compressed(os.path.join(OUT, "bn_calib_w48_gaussian.npz"),
                        **{k: (v.astype(np.float16) if "running_mean" in k else v) for k, v in calib.items()})
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=2, bn_calib=calib)
    net = ref_hrnet(pose_hrnet, extra, 17, "gaussian", sd)
    x = torch.from_numpy(synth.synth_crops(1, 384, 288, seed=6))
    with torch.no_grad():
        y = net(x).numpy()
    np.savez_compressed(os.path.join(OUT, "hrnet_w48_gaussian.npz"), out=y)
    print("w48", y.shape, "absmax", float(np.abs(y).max()), "std", float(y.std()))


def gen_rsn18():
    """Config 5 backbone: RSN-18 (RSN/exps/RSN18.coco/network.py), 17-channel and 51-channel heads."""
    from oracle import rsn as o_rsn
    sys.path.insert(0, "/root/reference/RSN")
    net_mod = _load("ref_rsn_network", "/root/reference/RSN/exps/RSN18.coco/network.py")
    for och in (17, 51):
        cfg = AttrDict({"MODEL": {"STAGE_NUM": 1, "UPSAMPLE_CHANNEL_NUM": 256}, "DATASET": {"KEYPOINT": {"NUM": och}},
                        "OUTPUT_SHAPE": (64, 48), "LOSS": {"OHKM": True, "TOPK": 8, "COARSE_TO_FINE": True}})
        net = net_mod.RSN(cfg)
        net.eval()
        keys = {k: list(v.shape) for k, v in net.state_dict().items()}
        with open(os.path.join(OUT, "rsn18_keys_%d.json" % och), "w") as f:
            json.dump(keys, f)
        sd = synth.synth_rsn18_state_dict(och, seed=4)
        xc = torch.from_numpy(synth.synth_crops(24, 256, 192, seed=15))    # 8x6 maps at the deepest level: many crops
        yc = o_rsn.rsn_forward(sd, xc, calibrate=True)
        calib = {k: v.numpy().astype(np.float16).astype(np.float32) if "running_mean" in k else v.numpy()
                 for k, v in sd.items() if "running_" in k}
        calib["final.scale"] = np.float32(0.25 / float(yc.std()))
        np.savez_compressed(os.path.join(OUT, "bn_calib_rsn18_%d.npz" % och),
                            **{k: (v.astype(np.float16) if "running_mean" in k else v) for k, v in calib.items()})
        sd = synth.synth_rsn18_state_dict(och, seed=4, bn_calib=calib)
        net.load_state_dict(sd, strict=True)
        x = torch.from_numpy(synth.synth_crops(1, 256, 192, seed=8))
        with torch.no_grad():
            y = net(x).numpy()
        np.savez_compressed(os.path.join(OUT, "rsn18_%d.npz" % och), out=y)
        print("rsn18", och, y.shape, "absmax", float(np.abs(y).max()), "std", float(y.std()))


def gen_decode(inference):
    c, s = synth.synth_center_scale(4, seed=2)
    out = {"center": c, "scale": s}
    for tt, post, k in (("gaussian", False, 1), ("gaussian", True, 1), ("offset", False, 3)):
        hm = synth.synth_heatmaps(4, 17, 64, 48, seed=21, channels_per_joint=k)
        # adversarial rows: all-negative map, exact ties, border peaks, flat map
        hm[0, 0] = -np.abs(hm[0, 0]) - 0.1
        hm[0, 1 * k, 10, 7] = hm[0, 1 * k].max() + 0.5
        hm[0, 1 * k, 30, 40] = hm[0, 1 * k, 10, 7]
        hm[1, 2 * k, 0, 0] = 2.0
        hm[1, 3 * k, 63, 47] = 2.0
        hm[2, 4 * k] = 0.25
        cfg = AttrDict({"MODEL": {"TARGET_TYPE": tt}, "TEST": {"POST_PROCESS": post},
                        "LOSS": {"KPD": 4.0}})
        with np.errstate(all="ignore"):
            preds, maxvals, pin = inference.get_final_preds(cfg, hm.copy(), c, s)
        tag = "%s%s" % (tt, "_post" if post else "")
        out["preds_" + tag] = preds
        out["maxvals_" + tag] = maxvals
        out["pin_" + tag] = pin
        print("decode", tag, preds.dtype, preds.shape)
    # get_max_preds alone (indices are implied by coords)
    hm = synth.synth_heatmaps(4, 17, 64, 48, seed=21)
    p, m = inference.get_max_preds(hm)
    out["maxpreds"] = p
    out["maxpreds_vals"] = m
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **out)


def gen_flip(transforms):
    rng = np.random.Generator(np.random.PCG64(31))
    from oracle.flip import COCO_FLIP_PAIRS
    a = rng.standard_normal((2, 17, 8, 6)).astype(np.float32)
    b = rng.standard_normal((2, 51, 8, 6)).astype(np.float32)
    fa = np.ascontiguousarray(transforms.flip_back(a.copy(), COCO_FLIP_PAIRS))
    fb = np.ascontiguousarray(transforms.flip_back_offset(b.copy(), COCO_FLIP_PAIRS))
    np.savez_compressed(os.path.join(OUT, "flip.npz"), a=a, b=b, fa=fa, fb=fb)


def gen_data(jd):
    rng = np.random.Generator(np.random.PCG64(41))
    mats, pts_out, cases = [], [], []
    for i in range(8):
        theta = float(rng.uniform(-60, 60)) if i else 0.0
        c = rng.uniform(50, 400, 2).astype(np.float32)
        s = rng.uniform(0.5, 2.5, 2).astype(np.float32)
        img = np.array([192, 256])
        m = jd.get_warpmatrix(theta, c * 2.0, img - 1.0, s)
        pts = rng.uniform(0, 500, (17, 2)).astype(np.float32)
        q = jd.rotate_points(pts, theta, c, img, s, False)
        mats.append(m)
        pts_out.append(q)
        cases.append(np.concatenate([[theta], c, s]).astype(np.float64))
    data = {"warp_cases": np.stack(cases), "warp_mats": np.stack(mats),
            "rot_pts_out": np.stack(pts_out)}
    # generate_target on a bare instance (JointsDataset.py:291-385)
    joints = np.zeros((3, 17, 3), np.float32)
    vis = np.ones((3, 17, 3), np.float32)
    joints[..., 0] = rng.uniform(-30, 220, (3, 17))
    joints[..., 1] = rng.uniform(-30, 290, (3, 17))
    joints[0, 0, :2] = (0.0, 0.0)
    joints[0, 1, :2] = (191.0, 255.0)
    joints[0, 2, :2] = (-40.0, 100.0)           # fully outside -> weight 0 (gaussian)
    joints[0, 3, :2] = (95.5, 127.5)
    vis[1, 5] = 0
    for tt in ("gaussian", "offset"):
        ds = jd.JointsDataset.__new__(jd.JointsDataset)
        ds.num_joints = 17
        ds.target_type = tt
        ds.image_size = np.array([192, 256])
        ds.heatmap_size = np.array([48, 64])
        ds.sigma = 2
        ds.kpd = 4.0
        ds.use_different_joints_weight = False
        ds.joints_weight = 1
        tg, tw = [], []
        for k in range(3):
            t, w = ds.generate_target(joints[k], vis[k])
            tg.append(t)
            tw.append(w)
        data["tgt_joints"] = joints
        data["tgt_vis"] = vis
        data["target_" + tt] = np.stack(tg)
        data["target_weight_" + tt] = np.stack(tw)
    np.savez_compressed(os.path.join(OUT, "data.npz"), **data)


def gen_loss(loss):
    rng = np.random.Generator(np.random.PCG64(51))
    out = {}
    p = torch.from_numpy(rng.standard_normal((4, 17, 16, 12)).astype(np.float32)).requires_grad_()
    g = torch.from_numpy(rng.uniform(0, 1, (4, 17, 16, 12)).astype(np.float32))
    w = torch.from_numpy((rng.uniform(0, 1, (4, 17, 1)) > 0.3).astype(np.float32))
    l = loss.JointsMSELoss(True)(p, g, w)
    l.backward()
    out.update(mse_pred=p.detach().numpy(), mse_gt=g.numpy(), mse_w=w.numpy(),
               mse_loss=np.float64(l.item()), mse_grad=p.grad.numpy())
    p = torch.from_numpy(rng.standard_normal((4, 51, 16, 12)).astype(np.float32)).requires_grad_()
    g = torch.from_numpy(rng.uniform(0, 1, (4, 51, 16, 12)).astype(np.float32))
    g[:, 0::3] = (g[:, 0::3] > 0.7).float()
    lh, lo = loss.JointsMSELoss_offset(True)(p, g, w)
    (lh + lo).backward()
    out.update(off_pred=p.detach().numpy(), off_gt=g.numpy(), off_w=w.numpy(),
               off_loss_hm=np.float64(lh.item()), off_loss_os=np.float64(lo.item()),
               off_grad=p.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    pose_hrnet, inference, loss, transforms, jd = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "w48":
        gen_hrnet_w48(pose_hrnet)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rsn":
        gen_rsn18()
        return
    gen_hrnet_w48(pose_hrnet)
    gen_rsn18()
    gen_flip(transforms)
    gen_data(jd)
    gen_loss(loss)
    gen_decode(inference)
    gen_hrnet(pose_hrnet)
    for f in sorted(os.listdir(OUT)):
        print("%9d  %s" % (os.path.getsize(os.path.join(OUT, f)), f))


if __name__ == "__main__":
    main()
