#!/usr/bin/env python3
"""Golden with PEAKED heat-maps: a mini HRNet TRAINED (reference module, train mode, reference JointsMSELoss,
torch.optim.Adam lr 1e-3 -- function.py:38-77, utils.py:70-74) on synthetic coloured-blob keypoints until its
heat-maps have clear peaks, then the reference module's eval-mode heat-maps, its flip-test fusion
(function.py:151-171) and the reference get_final_preds (inference.py:149-186, blur = oracle/cv2_standin) on
held-out crops.  Build container only (imports /root/reference).

    python oracle/gen_golden_peaked.py      # writes tests/golden/hrnet_peaked.npz  (~2 min of CPU training)

The random-weight fixtures give noise-like maps, on which DARK's Hessian is ill-conditioned and bf16 arg-max
agreement says little; this one pins (a) fp32 / split-fp16 end-to-end keypoints at 1e-3 px on ALL joints and
(b) what bf16 storage really does to trained maps.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as gg                                   # noqa: E402
from oracle import data as o_data                          # noqa: E402
from udp_pose_amd import synth                             # noqa: E402

EXTRA = synth.scaled_extra(16, modules=(1, 1, 1), blocks=1)
NJ, H, W = 17, 128, 96
STEPS, BATCH = int(os.environ.get("PEAKED_STEPS", "1500")), 16


def targets(joints, vis):
    tg, tw = [], []
    for j, v in zip(joints, vis):
        t, wgt = o_data.generate_target(j, v, "gaussian", (W, H), (W // 4, H // 4), sigma=2, kpd=4.0)
        tg.append(t)
        tw.append(wgt)
    return torch.from_numpy(np.stack(tg)), torch.from_numpy(np.stack(tw))


def main():
    pose_hrnet, ref_inf, ref_loss, ref_tr, _ = gg.load_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = gg.model_cfg(EXTRA, NJ, "gaussian")
    cfg["MODEL"]["INIT_WEIGHTS"] = True
    net = pose_hrnet.get_pose_net(cfg, is_train=True)       # the reference's own init_weights (pose_hrnet.py:473-489)
    net.train()
    crit = ref_loss.JointsMSELoss(use_target_weight=True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    t0 = time.time()
    for step in range(STEPS):
        u8, joints, vis = synth.synth_keypoint_scene(BATCH, H, W, NJ, seed=1000 + step)
        x = torch.from_numpy(synth.normalize_u8(u8))
        tg, tw = targets(joints, vis)
        loss = crit(net(x), tg, tw)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 100 == 0 or step == STEPS - 1:
            print("step %4d loss %.6f (%.0f s)" % (step, float(loss.detach()), time.time() - t0), flush=True)
    net.eval()
    # the fixture's weights are the trained ones ROUNDED TO fp16 (half the file size); every output below is
    # computed by the reference module from exactly these rounded values
    sd = {k: (v.detach().to(torch.float16).to(torch.float32) if v.is_floating_point() else v.detach().clone())
          for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)

    # held-out crops; eval-mode heat-maps of the reference module, plain and W-mirrored input
    u8, joints, vis = synth.synth_keypoint_scene(4, H, W, NJ, seed=7)
    x = torch.from_numpy(synth.normalize_u8(u8))
    with torch.no_grad():
        out = net(x).numpy()
        out_flip = net(torch.flip(x, dims=[3])).numpy()
    pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]     # coco.py:91-92
    fused = (out + ref_tr.flip_back(out_flip.copy(), pairs)) * 0.5                       # function.py:160-171
    c, s = synth.synth_center_scale(4, seed=3)
    cfg = gg.AttrDict({"MODEL": {"TARGET_TYPE": "gaussian"}, "TEST": {"POST_PROCESS": True}, "LOSS": {"KPD": 4.0}})
    preds, maxvals, preds_in = ref_inf.get_final_preds(cfg, fused.copy(), c, s)
    preds1, maxvals1, _ = ref_inf.get_final_preds(cfg, out.copy(), c, s)
    # how peaked: arg-max of the reference maps against the true joints (heat-map pixels)
    am = out.reshape(4, NJ, -1).argmax(2)
    err = np.hypot(am % (W // 4) - joints[:, :, 0] / (W - 1) * (W // 4 - 1), am // (W // 4) - joints[:, :, 1] / (H - 1) * (H // 4 - 1))
    print("trained maps: max %.3f, arg-max within 1 px of the true joint for %.0f %% of the joints"
          % (out.max(), 100.0 * (err < 1.0).mean()))
    blob = {"sd/" + k: (v.to(torch.float16).numpy() if v.is_floating_point() else v.numpy()) for k, v in sd.items()}
    blob.update(crops_u8=u8, joints=joints, out=out, out_flip=out_flip, fused=fused, center=c, scale=s, preds=preds,
                maxvals=maxvals, preds_in_input_space=preds_in, preds_noflip=preds1, maxvals_noflip=maxvals1,
                steps=np.int32(STEPS), argmax_hit_rate=np.float32((err < 1.0).mean()))
    np.savez_compressed(os.path.join(gg.OUT, "hrnet_peaked.npz"), **blob)
    print("wrote hrnet_peaked.npz", os.path.getsize(os.path.join(gg.OUT, "hrnet_peaked.npz")), "bytes")


if __name__ == "__main__":
    main()
