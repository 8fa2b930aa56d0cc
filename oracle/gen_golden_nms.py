#!/usr/bin/env python3
"""Golden for OKS-NMS: the REFERENCE's lib/nms/nms.py oks_iou / oks_nms (build container only).
The module's compiled-extension imports (.cpu_nms / .gpu_nms: box NMS, out of scope) are satisfied by
empty placeholder modules; nothing of them is called.

    python oracle/gen_golden_nms.py        # writes tests/golden/oks_nms.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from udp_pose_amd import synth                             # noqa: E402

REF = "/root/reference/deep_hrnet/lib/nms/nms.py"


def load_ref_nms():
    pkg = types.ModuleType("refnms")
    pkg.__path__ = []
    sys.modules["refnms"] = pkg
    for name in ("cpu_nms", "gpu_nms"):
        m = types.ModuleType("refnms." + name)
        setattr(m, name, None)
        sys.modules["refnms." + name] = m
    spec = importlib.util.spec_from_file_location("refnms.nms", REF)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "refnms"
    sys.modules["refnms.nms"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    nms = load_ref_nms()
    out = {}
    for case, (n_img, seed) in enumerate(((6, 3), (3, 9))):
        kpts, areas, scores, offs = synth.synth_person_sets(n_img, seed)
        keeps, ious0 = [], []
        for i in range(n_img):
            a, b = offs[i], offs[i + 1]
            db = [{"keypoints": kpts[p], "area": areas[p], "score": scores[p]} for p in range(a, b)]
            for thr_tag, thr, vis in (("t9", 0.9, None), ("t5", 0.5, None), ("t5v", 0.5, 0.2)):
                keep = nms.oks_nms(db, thr, None, vis)
                out["c%d_keep_%s_%d" % (case, thr_tag, i)] = np.array(keep, np.int64)
            for thr_tag, thr in (("t9", 0.9), ("t5", 0.5)):
                out["c%d_soft_%s_%d" % (case, thr_tag, i)] = np.array(nms.soft_oks_nms(db, thr), np.int64)
            flat = kpts[a:b].reshape(b - a, -1)
            ious0.append(nms.oks_iou(flat[0], flat, areas[a], areas[a:b]))
        out["c%d_iou0" % case] = np.concatenate(ious0)
        print("case", case, "persons", offs[-1], "kept@0.9", sum(len(out["c%d_keep_t9_%d" % (case, i)]) for i in range(n_img)),
              "kept@0.5", sum(len(out["c%d_keep_t5_%d" % (case, i)]) for i in range(n_img)))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oks_nms.npz"), **out)


if __name__ == "__main__":
    main()
