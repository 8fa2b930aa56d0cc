"""pose_resnet (SimpleBaseline) weight-file contract + seeded synthetic weights.

Key names / shapes of ``PoseResNet.state_dict()`` (deep_hrnet/lib/models/pose_resnet.py:105-208,
Bottleneck :64-100) for BASELINE.json configs[0] (pose_resnet_50 256x192).  The synthetic generator
draws in state_dict order exactly like oracle/gen_golden_resnet.py did when it produced
tests/golden/resnet50_cfg0.npz, so the fixture's heat-maps can be reproduced without the reference.
"""
from collections import OrderedDict

import numpy as np
import torch


def _bn(s, name, c):
    s[name + ".weight"] = (c,)
    s[name + ".bias"] = (c,)
    s[name + ".running_mean"] = (c,)
    s[name + ".running_var"] = (c,)
    s[name + ".num_batches_tracked"] = ()


def pose_resnet_param_shapes(layers=(3, 4, 6, 3), num_joints=17, deconv_filters=(256, 256, 256), deconv_kernel=4,
                             final_kernel=1, deconv_with_bias=False):
    s = OrderedDict()
    s["conv1.weight"] = (64, 3, 7, 7)
    _bn(s, "bn1", 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for b in range(nblk):
            p = "layer%d.%d" % (li, b)
            s[p + ".conv1.weight"] = (planes, inplanes, 1, 1)
            _bn(s, p + ".bn1", planes)
            s[p + ".conv2.weight"] = (planes, planes, 3, 3)
            _bn(s, p + ".bn2", planes)
            s[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
            _bn(s, p + ".bn3", planes * 4)
            if b == 0:                                   # stride != 1 or inplanes != planes*4 (:144-151)
                s[p + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                _bn(s, p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for d, f in enumerate(deconv_filters):
        s["deconv_layers.%d.weight" % (3 * d)] = (inplanes, f, deconv_kernel, deconv_kernel)   # ConvTranspose2d: [in,out,k,k]
        if deconv_with_bias:
            s["deconv_layers.%d.bias" % (3 * d)] = (f,)
        _bn(s, "deconv_layers.%d" % (3 * d + 1), f)
        inplanes = f
    s["final_layer.weight"] = (num_joints, inplanes, final_kernel, final_kernel)
    s["final_layer.bias"] = (num_joints,)
    return s


def synth_pose_resnet_state_dict(seed=7, calib=None, final_scale=1.0, **kw):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for k, shape in pose_resnet_param_shapes(**kw).items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0, dtype=torch.long)
        elif len(shape) == 4:
            fan = shape[1] * shape[2] * shape[3]
            sd[k] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32))
        elif k.endswith("bn3.weight") or "downsample.1.weight" in k:
            sd[k] = torch.from_numpy(rng.uniform(0.2, 0.4, shape).astype(np.float32))
        elif k.endswith(".weight"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.0, shape).astype(np.float32))
        elif k.endswith(".bias"):
            sd[k] = torch.from_numpy((rng.standard_normal(shape) * 0.05).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shape)
        else:
            sd[k] = torch.zeros(shape)
    if calib:
        for k, v in calib.items():
            sd[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).copy())
    sd["final_layer.weight"] = sd["final_layer.weight"] * float(final_scale)
    sd["final_layer.bias"] = sd["final_layer.bias"] * float(final_scale)
    return sd
