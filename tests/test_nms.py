"""OKS-NMS: oracle vs the reference fixture (CPU) and the HIP kernel vs both (GPU)."""
import os

import numpy as np
import pytest

from oracle import nms as o_nms
from udp_pose_amd import synth

CASES = ((6, 3), (3, 9))
THR = (("t9", 0.9, None), ("t5", 0.5, None), ("t5v", 0.5, 0.2))


def test_oracle_oks_nms_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "oks_nms.npz"))
    for case, (n_img, seed) in enumerate(CASES):
        kpts, areas, scores, offs = synth.synth_person_sets(n_img, seed)
        iou0 = []
        for i in range(n_img):
            a, b = offs[i], offs[i + 1]
            flat = kpts[a:b].reshape(b - a, -1)
            for tag, thr, vis in THR:
                assert o_nms.oks_nms(flat, scores[a:b], areas[a:b], thr, None, vis) == list(g["c%d_keep_%s_%d" % (case, tag, i)])
            for tag, thr in (("t9", 0.9), ("t5", 0.5)):
                assert o_nms.soft_oks_nms(flat, scores[a:b], areas[a:b], thr) == list(g["c%d_soft_%s_%d" % (case, tag, i)])
            iou0.append(o_nms.oks_iou(flat[0], flat, areas[a], areas[a:b]))
        np.testing.assert_allclose(np.concatenate(iou0), g["c%d_iou0" % case], rtol=0, atol=1e-15)
    assert o_nms.oks_nms(np.zeros((0, 51)), np.zeros(0), np.zeros(0), 0.9) == []


@pytest.mark.gpu
def test_hip_oks_nms_matches_reference(golden_dir):
    from udp_pose_amd import nms as u_nms
    g = np.load(os.path.join(golden_dir, "oks_nms.npz"))
    for case, (n_img, seed) in enumerate(CASES):
        kpts, areas, scores, offs = synth.synth_person_sets(n_img, seed)
        for i in range(n_img):
            a, b = offs[i], offs[i + 1]
            db = [{"keypoints": kpts[p], "area": areas[p], "score": scores[p]} for p in range(a, b)]
            for tag, thr, vis in THR:
                assert u_nms.oks_nms(db, thr, None, vis) == list(g["c%d_keep_%s_%d" % (case, tag, i)]), (case, i, tag)
            for tag, thr in (("t9", 0.9), ("t5", 0.5)):
                assert u_nms.soft_oks_nms(db, thr) == list(g["c%d_soft_%s_%d" % (case, tag, i)]), (case, i, tag)
    assert u_nms.oks_nms([], 0.9) == []


@pytest.mark.gpu
def test_hip_oks_nms_with_nan_scores_equals_numpy_order():
    """A NaN score (e.g. from a NaN maxval) must not break the kernel's ranking: NumPy's argsort puts NaN last,
    so the reference's ``scores.argsort()[::-1]`` visits it FIRST; the kernel ranks NaN as +inf and keeps the same
    list as the oracle (which runs the reference's NumPy statements)."""
    from udp_pose_amd import nms as u_nms
    kpts, areas, scores, offs = synth.synth_person_sets(4, 5)
    for i in range(4):
        a, b = offs[i], offs[i + 1]
        sc = scores[a:b].copy()
        sc[(i + 1) % (b - a)] = np.nan
        if b - a > 3:
            sc[3] = np.nan
        db = [{"keypoints": kpts[p], "area": areas[p], "score": sc[p - a]} for p in range(a, b)]
        want = o_nms.oks_nms(kpts[a:b].reshape(b - a, -1), sc, areas[a:b], 0.5, None, None)
        assert u_nms.oks_nms(db, 0.5, None, None) == want, (i, want)


@pytest.mark.gpu
def test_hip_rescore_and_nms_whole_evaluation():
    """coco.py:306-356 in one launch vs the oracle's per-image loop (rescoring restated from the text)."""
    from udp_pose_amd import nms as u_nms
    kpts, areas, scores, offs = synth.synth_person_sets(40, 21)
    p = kpts.shape[0]
    rng = np.random.default_rng(4)
    image_ids = np.repeat(np.arange(40) * 7 + 100, np.diff(offs))
    shuffle = rng.permutation(p)                                   # persons of an image need not be adjacent
    kpts, areas, scores, image_ids = kpts[shuffle], areas[shuffle], scores[shuffle], image_ids[shuffle]
    boxes = np.zeros((p, 6))
    boxes[:, 4], boxes[:, 5] = areas, scores
    got = u_nms.rescore_and_nms(kpts, boxes, image_ids, in_vis_thre=0.2, oks_thre=0.9)
    n_kept = 0
    for im in np.unique(image_ids):
        idx = np.where(image_ids == im)[0]
        sc = np.array([o_nms.rescore(kpts[q], boxes[q, 5], 0.2) for q in idx])
        keep = o_nms.oks_nms(kpts[idx].reshape(len(idx), -1), sc, areas[idx], 0.9)
        assert [int(np.where((kpts == d["keypoints"]).all(axis=(1, 2)))[0][0]) for d in got[im]] == [int(idx[k]) for k in keep]
        np.testing.assert_allclose([d["score"] for d in got[im]], sc[keep], rtol=1e-12)
        n_kept += len(keep)
    assert 0 < n_kept < p


def test_coco_result_records(tmp_path):
    """coco.py:397-428 record format, restated from the text (evaluate() itself needs pycocotools)."""
    import json
    from udp_pose_amd import nms as u_nms
    kp = np.arange(51, dtype=np.float32).reshape(17, 3)
    kept = {7: [{"keypoints": kp, "center": np.array([1.5, 2.5]), "scale": np.array([0.5, 0.75]), "area": 3.0,
                 "score": 0.25, "image": 7}], 9: []}
    rec = u_nms.coco_keypoint_results(kept)
    assert rec == [{"image_id": 7, "category_id": 1, "keypoints": [float(v) for v in range(51)], "score": 0.25,
                    "center": [1.5, 2.5], "scale": [0.5, 0.75]}]
    u_nms.write_coco_keypoint_results(kept, str(tmp_path / "r.json"))
    assert json.load(open(tmp_path / "r.json")) == rec
