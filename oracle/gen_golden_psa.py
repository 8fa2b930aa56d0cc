#!/usr/bin/env python3
"""Golden for the PSA variant (deep_hrnet/lib/models/pose_hrnet_psa.py + PSA.py): width-32 mini
HRNet-PSA heat-maps from the REFERENCE module (build container only).

    python oracle/gen_golden_psa.py        # writes tests/golden/hrnet_psa_mini.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as gg                                   # noqa: E402
from oracle import hrnet as o_hrnet                        # noqa: E402
from udp_pose_amd import synth                             # noqa: E402


def main():
    gg.load_reference()
    m = gg._load("refmodels.pose_hrnet_psa", os.path.join(gg.REF, "lib/models/pose_hrnet_psa.py"), "refmodels")
    extra = synth.scaled_extra(32, modules=(1, 2, 1), blocks=2)
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=6, psa=True)
    xc = torch.from_numpy(synth.synth_crops(6, 128, 96, seed=23))
    yc = o_hrnet.hrnet_forward(sd, extra, xc, calibrate=True)
    calib = {k: v.numpy() for k, v in sd.items() if "running_" in k}
    calib["final_layer.scale"] = np.float32(0.25 / float(yc.std()))
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=6, bn_calib=calib, psa=True)
    net = m.get_pose_net(gg.model_cfg(extra, 17, "gaussian"), is_train=False)
    res = net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.from_numpy(synth.synth_crops(2, 128, 96, seed=24))
    with torch.no_grad():
        y = net(x).numpy()
    keys = sorted("%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in net.state_dict().items())
    np.savez_compressed(os.path.join(gg.OUT, "hrnet_psa_mini.npz"), out=y, keys=np.array(keys),
                        **{"calib_" + k: v for k, v in calib.items()})
    yo = o_hrnet.hrnet_forward(sd, extra, x).numpy()
    print("psa mini", y.shape, "absmax", float(np.abs(y).max()), "oracle-vs-ref", float(np.abs(y - yo).max()))


if __name__ == "__main__":
    main()
