"""Synthetic HRNet weights in the reference's state_dict format.

No pretrained weights ship with the reference, so parity tests
and the benchmark use seeded random weights.  ``hrnet_param_shapes`` enumerates
the weight-file contract of deep_hrnet/lib/models/pose_hrnet.py (key names and
shapes of ``PoseHighResolutionNet.state_dict()``, :282-342 / :189-255 /
:344-383); ``synth_state_dict`` fills it from a NumPy PCG64 stream, so the same
(seed, config) gives the same weights in the build container and on the GPU
box.  BatchNorm running statistics may be overlaid from a calibration fixture
(tests/golden/bn_calib_*.npz, written in the build container by
oracle/gen_golden.py) so that activations keep a realistic O(1) scale through
all ~60 sequential layers.
"""
from collections import OrderedDict

import numpy as np
import torch


def _bn(shapes, name, c):
    shapes[name + ".weight"] = (c,)
    shapes[name + ".bias"] = (c,)
    shapes[name + ".running_mean"] = (c,)
    shapes[name + ".running_var"] = (c,)
    shapes[name + ".num_batches_tracked"] = ()


def _psa_s(shapes, name, c):
    """PSA_s(planes, planes) parameters (deep_hrnet/lib/models/PSA.py:146-180) in registration order."""
    shapes[name + ".conv_q_right.weight"] = (1, c, 1, 1)
    shapes[name + ".conv_v_right.weight"] = (c // 2, c, 1, 1)
    shapes[name + ".conv_up.0.weight"] = (c // 8, c // 2, 1, 1)
    shapes[name + ".conv_up.0.bias"] = (c // 8,)
    shapes[name + ".conv_up.1.weight"] = (c // 8, 1, 1)       # LayerNorm([C/8,1,1])
    shapes[name + ".conv_up.1.bias"] = (c // 8, 1, 1)
    shapes[name + ".conv_up.3.weight"] = (c, c // 8, 1, 1)
    shapes[name + ".conv_up.3.bias"] = (c,)
    shapes[name + ".conv_q_left.weight"] = (c // 2, c, 1, 1)
    shapes[name + ".conv_v_left.weight"] = (c // 2, c, 1, 1)


def hrnet_param_shapes(extra, num_joints=17, target_type="gaussian", psa=False):
    """Ordered {key: shape} of the reference PoseHighResolutionNet state_dict (``psa``: the
    pose_hrnet_psa variant, a PSA_s after bn1 of every BasicBlock, pose_hrnet_psa.py:37,49)."""
    s = OrderedDict()
    s["conv1.weight"] = (64, 3, 3, 3)
    _bn(s, "bn1", 64)
    s["conv2.weight"] = (64, 64, 3, 3)
    _bn(s, "bn2", 64)
    inpl = 64
    for k in range(4):                                   # layer1: 4 Bottlenecks, planes 64
        p = "layer1.%d" % k
        s[p + ".conv1.weight"] = (64, inpl, 1, 1)
        _bn(s, p + ".bn1", 64)
        s[p + ".conv2.weight"] = (64, 64, 3, 3)
        _bn(s, p + ".bn2", 64)
        s[p + ".conv3.weight"] = (256, 64, 1, 1)
        _bn(s, p + ".bn3", 256)
        if k == 0:
            s[p + ".downsample.0.weight"] = (256, inpl, 1, 1)
            _bn(s, p + ".downsample.1", 256)
        inpl = 256
    pre = [256]
    for st in (2, 3, 4):
        cfg = extra["STAGE%d" % st]
        if cfg["BLOCK"] != "BASIC":
            raise ValueError("only BASIC blocks are supported in stages 2-4")
        chans = list(cfg["NUM_CHANNELS"])
        nb = cfg["NUM_BRANCHES"]
        if nb != len(chans) or nb != len(cfg["NUM_BLOCKS"]):
            raise ValueError("NUM_BRANCHES(%d) <> NUM_CHANNELS/NUM_BLOCKS" % nb)
        t = "transition%d" % (st - 1)
        for i in range(nb):
            if i < len(pre):
                if chans[i] != pre[i]:
                    s["%s.%d.0.weight" % (t, i)] = (chans[i], pre[i], 3, 3)
                    _bn(s, "%s.%d.1" % (t, i), chans[i])
            else:
                for k in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = chans[i] if k == i - len(pre) else cin
                    s["%s.%d.%d.0.weight" % (t, i, k)] = (cout, cin, 3, 3)
                    _bn(s, "%s.%d.%d.1" % (t, i, k), cout)
        nmod = cfg["NUM_MODULES"]
        for m in range(nmod):
            last = (st == 4 and m == nmod - 1)
            p = "stage%d.%d" % (st, m)
            for b in range(nb):
                for k in range(cfg["NUM_BLOCKS"][b]):
                    q = "%s.branches.%d.%d" % (p, b, k)
                    s[q + ".conv1.weight"] = (chans[b], chans[b], 3, 3)
                    _bn(s, q + ".bn1", chans[b])
                    if psa:
                        _psa_s(s, q + ".deattn", chans[b])
                    s[q + ".conv2.weight"] = (chans[b], chans[b], 3, 3)
                    _bn(s, q + ".bn2", chans[b])
            inch = list(chans)
            if last:
                inch[0] *= 4
            for i in range(1 if last else nb):
                for j in range(nb):
                    q = "%s.fuse_layers.%d.%d" % (p, i, j)
                    if j > i:
                        s[q + ".0.weight"] = (inch[i], inch[j], 1, 1)
                        _bn(s, q + ".1", inch[i])
                    elif j == i:
                        if last:
                            s[q + ".0.weight"] = (inch[j], inch[j] // 4, 1, 1)
                    else:
                        for k in range(i - j):
                            cout = inch[i] if k == i - j - 1 else inch[j]
                            s["%s.%d.0.weight" % (q, k)] = (cout, inch[j], 3, 3)
                            _bn(s, "%s.%d.1" % (q, k), cout)
            if last:
                chans = inch
        pre = chans
    factor = 3 if target_type == "offset" else 1
    fk = int(extra.get("FINAL_CONV_KERNEL", 1))
    s["final_layer.weight"] = (num_joints * factor, pre[0], fk, fk)
    s["final_layer.bias"] = (num_joints * factor,)
    return s


def synth_state_dict(extra, num_joints=17, target_type="gaussian", seed=0, bn_calib=None, psa=False):
    """Seeded synthetic state_dict (fp32 torch tensors, CPU).

    conv ~ N(0, 2/fan_in); BN gamma ~ U(0.6,1.2) (0.25..0.5 for the closing BN of
    a residual branch and for fuse BNs, so the residual stream does not blow
    up), beta ~ N(0,0.05); running stats (0,1) unless ``bn_calib`` (dict of
    name -> array) overlays calibrated ones.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for name, shape in hrnet_param_shapes(extra, num_joints, target_type, psa=psa).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        if name.endswith("running_mean"):
            a = np.zeros(shape, np.float32)
        elif name.endswith("running_var"):
            a = np.ones(shape, np.float32)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            a = rng.standard_normal(shape).astype(np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        elif name == "final_layer.bias":
            a = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        elif name.endswith(".weight"):
            closing = (".bn2." in name + "." and "branches" in name) or ".bn3" in name \
                or "fuse_layers" in name or "downsample" in name
            lo, hi = (0.25, 0.5) if closing else (0.6, 1.2)
            a = rng.uniform(lo, hi, shape).astype(np.float32)
        else:
            a = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        sd[name] = torch.from_numpy(a)
    if bn_calib is not None:
        for k, v in bn_calib.items():
            if k in sd:
                sd[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).copy())
        if "final_layer.scale" in bn_calib:          # brings heat-maps to O(1) like trained ones
            f = float(np.asarray(bn_calib["final_layer.scale"]))
            sd["final_layer.weight"] = sd["final_layer.weight"] * f
            sd["final_layer.bias"] = sd["final_layer.bias"] * f
    return sd


W32_EXTRA = {
    "FINAL_CONV_KERNEL": 1,
    "PRETRAINED_LAYERS": ["*"],
    "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "BLOCK": "BASIC", "NUM_BLOCKS": [4, 4],
               "NUM_CHANNELS": [32, 64], "FUSE_METHOD": "SUM"},
    "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "BLOCK": "BASIC", "NUM_BLOCKS": [4, 4, 4],
               "NUM_CHANNELS": [32, 64, 128], "FUSE_METHOD": "SUM"},
    "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "BLOCK": "BASIC", "NUM_BLOCKS": [4, 4, 4, 4],
               "NUM_CHANNELS": [32, 64, 128, 256], "FUSE_METHOD": "SUM"},
}


def scaled_extra(width, modules=(1, 4, 3), blocks=4):
    """HRNet EXTRA dict with branch widths width*(1,2,4,8) (W32: 32, W48: 48;
    small widths/modules give the mini nets used by the golden fixtures)."""
    e = {"FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"]}
    for i, st in enumerate((2, 3, 4)):
        nb = st
        e["STAGE%d" % st] = {
            "NUM_MODULES": modules[i], "NUM_BRANCHES": nb, "BLOCK": "BASIC",
            "NUM_BLOCKS": [blocks] * nb, "NUM_CHANNELS": [width * 2 ** b for b in range(nb)],
            "FUSE_METHOD": "SUM"}
    return e


# ----------------------------------------------------------------------------
# Synthetic inputs: person crops, boxes, heat-maps.
# ----------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def synth_frame_u8(h, w, seed=0, blobs=8):
    """uint8 HxWx3 frame: sum of random Gaussian blobs + uniform noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ys = np.arange(h, dtype=np.float32)[:, None]
    xs = np.arange(w, dtype=np.float32)[None, :]
    img = np.zeros((h, w, 3), np.float32)
    for _ in range(blobs):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        s = rng.uniform(0.05, 0.25) * min(h, w)
        col = rng.uniform(40, 255, 3).astype(np.float32)
        g = np.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * s * s)).astype(np.float32)
        img += g[:, :, None] * col[None, None, :]
    img += rng.uniform(0, 30, (h, w, 3)).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_crops(n, h=256, w=192, seed=0):
    """Normalised fp32 crops [N,3,H,W] (ToTensor + ImageNet Normalize of
    synthetic uint8 patches), as pose_engine.py:40-43 would hand the network."""
    out = np.empty((n, 3, h, w), np.float32)
    mean = np.asarray(IMAGENET_MEAN, np.float32)[:, None, None]
    std = np.asarray(IMAGENET_STD, np.float32)[:, None, None]
    for i in range(n):
        u8 = synth_frame_u8(h, w, seed=seed * 100003 + i)
        x = np.transpose(u8.astype(np.float32) / np.float32(255), (2, 0, 1))
        out[i] = (x - mean) / std
    return out


def synth_boxes(n, frame_w=640, frame_h=480, seed=0):
    """xyxy person boxes: centre U([100,540]x[100,380]), w~U(40,200),
    h = w/0.75*U(0.8,1.2)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    cx = rng.uniform(100, frame_w - 100, n)
    cy = rng.uniform(100, frame_h - 100, n)
    w = rng.uniform(40, 200, n)
    h = w / 0.75 * rng.uniform(0.8, 1.2, n)
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)


def synth_center_scale(n, aspect=0.75, seed=0):
    """center [N,2] / scale [N,2] (float32) from synth_boxes through the
    engine's box->center/scale rule (pose_engine.py:55-63)."""
    b = synth_boxes(n, seed=seed)
    c = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2], 1)
    w = b[:, 2] - b[:, 0]
    h = b[:, 3] - b[:, 1]
    m = w > h * aspect
    h = np.where(m, w / aspect, h)
    w = np.where(m, w, h * aspect)
    s = np.stack([w, h], 1) / 200.0 * 1.25
    return c.astype(np.float32), s.astype(np.float32)


def synth_heatmaps(n, j, h=64, w=48, seed=0, channels_per_joint=1):
    """Heat-maps [N,J*k,H,W] f32: a sigma=2 Gaussian at a sub-pixel mean,
    amplitude U(0.3,1), + N(0,0.01^2) noise.  For k=3 (offset head) channels
    3j+1 / 3j+2 carry the UDP offset field (mu - grid)/4 inside the unit disk
    plus noise, like JointsDataset.generate_target (offset) would teach."""
    rng = np.random.Generator(np.random.PCG64(seed + 13))
    ys = np.arange(h, dtype=np.float32)[:, None]
    xs = np.arange(w, dtype=np.float32)[None, :]
    out = np.empty((n, j * channels_per_joint, h, w), np.float32)
    for a in range(n):
        for b in range(j):
            mx, my = rng.uniform(2, w - 3), rng.uniform(2, h - 3)
            amp = rng.uniform(0.3, 1.0)
            g = amp * np.exp(-((xs - mx) ** 2 + (ys - my) ** 2) / 8.0)
            noise = rng.standard_normal((channels_per_joint, h, w)).astype(np.float32) * 0.01
            if channels_per_joint == 1:
                out[a, b] = g.astype(np.float32) + noise[0]
            else:
                dx = (mx - xs) / 4.0 + 0 * ys
                dy = (my - ys) / 4.0 + 0 * xs
                disk = ((dx * dx + dy * dy) <= 1.0).astype(np.float32)
                out[a, 3 * b] = disk * np.float32(amp) + noise[0]
                out[a, 3 * b + 1] = (dx * disk).astype(np.float32) + noise[1]
                out[a, 3 * b + 2] = (dy * disk).astype(np.float32) + noise[2]
    return out


# ----------------------------------------------------------------------------
# RSN-18 (RSN/exps/RSN18.coco/network.py): weight-file contract + synthetic weights
# ----------------------------------------------------------------------------
def _cbr(shapes, name, cin, cout, k):
    shapes[name + ".conv.weight"] = (cout, cin, k, k)
    shapes[name + ".conv.bias"] = (cout,)
    shapes[name + ".bn.weight"] = (cout,)
    shapes[name + ".bn.bias"] = (cout,)
    shapes[name + ".bn.running_mean"] = (cout,)
    shapes[name + ".bn.running_var"] = (cout,)
    shapes[name + ".bn.num_batches_tracked"] = ()


def rsn18_param_shapes(out_channels=17, chl_num=256):
    """Ordered {key: shape} of RSN(cfg).state_dict() for STAGE_NUM=1, layers [2,2,2,2]
    (network.py:343-396: top, stage0.downsample.layer1-4, stage0.upsample.up1-4)."""
    s = OrderedDict()
    _cbr(s, "top.conv", 3, 64, 7)
    in_planes = 64
    for layer, planes in zip(range(1, 5), (64, 128, 256, 512)):
        for b in range(2):
            p = "stage0.downsample.layer%d.%d" % (layer, b)
            bch = in_planes * 26 // 64
            _cbr(s, p + ".conv_bn_relu1", in_planes, 4 * bch, 1)
            for nm in ("2_1_1", "2_2_1", "2_2_2", "2_3_1", "2_3_2", "2_3_3", "2_4_1", "2_4_2", "2_4_3", "2_4_4"):
                _cbr(s, p + ".conv_bn_relu" + nm, bch, bch, 3)
            _cbr(s, p + ".conv_bn_relu3", 4 * bch, planes, 1)
            if b == 0 and (layer > 1 or in_planes != planes):
                _cbr(s, p + ".downsample", in_planes, planes, 1)
            in_planes = planes
    for ind, cin in enumerate((512, 256, 128, 64)):
        p = "stage0.upsample.up%d" % (ind + 1)
        _cbr(s, p + ".u_skip", cin, chl_num, 1)
        if ind > 0:
            _cbr(s, p + ".up_conv", chl_num, chl_num, 1)
        _cbr(s, p + ".res_conv1", chl_num, chl_num, 1)
        _cbr(s, p + ".res_conv2", chl_num, out_channels, 3)
    return s


def synth_rsn18_state_dict(out_channels=17, seed=0, bn_calib=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for name, shape in rsn18_param_shapes(out_channels).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        if name.endswith("running_mean"):
            a = np.zeros(shape, np.float32)
        elif name.endswith("running_var"):
            a = np.ones(shape, np.float32)
        elif len(shape) == 4:
            a = rng.standard_normal(shape).astype(np.float32) * np.float32(np.sqrt(2.0 / (shape[1] * shape[2] * shape[3])))
        elif name.endswith("conv.bias"):
            a = (rng.standard_normal(shape) * 0.02).astype(np.float32)
        elif name.endswith("bn.weight"):
            closing = "conv_bn_relu3" in name or "downsample" in name or "up_conv" in name or "u_skip" in name
            a = rng.uniform(0.2, 0.4, shape).astype(np.float32) if closing else rng.uniform(0.4, 0.8, shape).astype(np.float32)
        else:
            a = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        sd[name] = torch.from_numpy(a)
    if bn_calib is not None:
        for k, v in bn_calib.items():
            if k in sd:
                sd[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).copy())
        if "final.scale" in bn_calib:
            f = float(np.asarray(bn_calib["final.scale"]))
            for k in ("stage0.upsample.up4.res_conv2.bn.weight", "stage0.upsample.up4.res_conv2.bn.bias"):
                sd[k] = sd[k] * f
    return sd


COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]     # lib/dataset/coco.py:91-92
COCO_UPPER_BODY = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)                                              # coco.py:94


def synth_db(n=8, seed=77):
    """Synthetic dataset records shaped like COCODataset's db entries (coco.py:150-200): joints_3d,
    joints_3d_vis, center, scale (+ the size / seed of the synthetic frame that stands for the image)."""
    rng = np.random.default_rng(seed)
    db = []
    for i in range(n):
        h, w = (480, 640) if i % 2 == 0 else (427, 640)
        joints = np.zeros((17, 3), np.float32)
        vis = np.zeros((17, 3), np.float32)
        cx, cy = rng.uniform(150, w - 150), rng.uniform(120, h - 120)
        joints[:, 0] = cx + rng.normal(0, 45, 17)
        joints[:, 1] = cy + rng.normal(0, 80, 17)
        v = (rng.random(17) < (0.9 if i % 3 else 0.45)).astype(np.float32)
        vis[:, 0] = vis[:, 1] = v
        joints[:, :2] *= v[:, None]
        bw, bh = rng.uniform(80, 220), rng.uniform(150, 330)
        if bw > 0.75 * bh:
            bh = bw / 0.75
        else:
            bw = bh * 0.75
        db.append({"image": "frame_%d" % i, "joints_3d": joints, "joints_3d_vis": vis,
                   "center": np.array([cx, cy], np.float32),
                   "scale": np.array([bw / 200 * 1.25, bh / 200 * 1.25], np.float32),
                   "frame_hw": (h, w), "frame_seed": 500 + i})
    return db


def synth_person_sets(n_images, seed=0, num_joints=17):
    """Detections for OKS-NMS: per image a few distinct poses, each with near-duplicates (jittered copies).
    Returns kpts fp32 [P, J, 3] (x, y, score), areas fp64 [P], box scores fp64 [P], image offsets int32 [I+1]."""
    rng = np.random.default_rng(seed)
    kpts, areas, scores, offs = [], [], [], [0]
    for _ in range(n_images):
        n_pose = int(rng.integers(1, 5))
        for _p in range(n_pose):
            base = np.zeros((num_joints, 3), np.float32)
            cx, cy, sc = rng.uniform(100, 500), rng.uniform(100, 380), rng.uniform(40, 120)
            base[:, 0] = cx + rng.normal(0, sc / 3, num_joints)
            base[:, 1] = cy + rng.normal(0, sc, num_joints)
            area = float((2 * sc) * (3 * sc))
            for _d in range(int(rng.integers(1, 5))):
                k = base.copy()
                jit = rng.choice([0.5, 3.0, 12.0])
                k[:, :2] += rng.normal(0, jit, (num_joints, 2)).astype(np.float32)
                k[:, 2] = rng.uniform(0.05, 0.95, num_joints).astype(np.float32)
                kpts.append(k)
                areas.append(area * rng.uniform(0.9, 1.1))
                scores.append(float(rng.uniform(0.3, 1.0)))
        offs.append(len(kpts))
    return (np.stack(kpts), np.array(areas, np.float64), np.array(scores, np.float64), np.array(offs, np.int32))


def synth_accuracy_case(seed, max_shift, n=4, j=17, h=64, w=48):
    """(prediction, target) heat-maps for the PCK accuracy check: the prediction is the target with every
    (image, joint) map rolled by its own random shift of up to ``max_shift`` pixels, plus noise; joint 3 of
    image 0 is absent (all-zero target)."""
    rng = np.random.default_rng(seed)
    tgt = synth_heatmaps(n, j, h, w, seed=seed)
    pred = np.empty_like(tgt)
    for a in range(n):
        for b in range(j):
            dy, dx = rng.integers(-max_shift, max_shift + 1, 2)
            pred[a, b] = np.roll(tgt[a, b], (int(dy), int(dx)), axis=(0, 1))
    pred += 0.01 * rng.standard_normal(pred.shape).astype(np.float32)
    tgt[0, 3] = 0.0
    return pred, tgt


def synth_keypoint_scene(n, h=128, w=96, num_joints=17, seed=0):
    """Learnable synthetic pose data: every joint j is drawn as a small blob of its own colour on a noisy
    background, so a network can be TRAINED to put a heat-map peak on it (the random-weight nets of the other
    fixtures only produce noise-like maps).  Returns (uint8 crops [N,H,W,3], joints [N,J,3] in crop pixels with
    sub-pixel positions, joints_vis [N,J,3])."""
    rng = np.random.Generator(np.random.PCG64(0x5EED0 + seed))
    hue = (np.arange(num_joints) + 0.5) / num_joints
    colour = np.stack([0.5 + 0.5 * np.cos(2 * np.pi * (hue + k / 3.0)) for k in range(3)], axis=1)   # [J,3] in [0,1]
    size = 2.0 + 1.5 * ((np.arange(num_joints) * 7) % 5) / 4.0                                        # blob sigma px
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    imgs = np.empty((n, h, w, 3), np.uint8)
    joints = np.zeros((n, num_joints, 3), np.float32)
    vis = np.ones((n, num_joints, 3), np.float32)
    for i in range(n):
        img = rng.uniform(0.0, 0.25, (h, w, 3)).astype(np.float32)
        joints[i, :, 0] = rng.uniform(6, w - 7, num_joints)
        joints[i, :, 1] = rng.uniform(6, h - 7, num_joints)
        for j in range(num_joints):
            g = np.exp(-((xx - joints[i, j, 0]) ** 2 + (yy - joints[i, j, 1]) ** 2) / (2 * size[j] ** 2))
            img += g[:, :, None] * (0.75 * colour[j][None, None, :]).astype(np.float32)
        imgs[i] = np.clip(img * 255.0, 0, 255).astype(np.uint8)
    return imgs, joints, vis


def normalize_u8(imgs_u8):
    """ToTensor + ImageNet Normalize (pose_engine.py:40-43) of uint8 [N,H,W,3] -> fp32 [N,3,H,W]."""
    x = np.transpose(imgs_u8.astype(np.float32) / np.float32(255), (0, 3, 1, 2))
    mean = np.asarray(IMAGENET_MEAN, np.float32)[None, :, None, None]
    std = np.asarray(IMAGENET_STD, np.float32)[None, :, None, None]
    return ((x - mean) / std).astype(np.float32)
