"""Oracle: keypoint rescoring + OKS-NMS on CPU (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates deep_hrnet/lib/nms/nms.py:75-124 (oks_iou, oks_nms) and the rescoring loop of
deep_hrnet/lib/dataset/coco.py:321-341.  oks_iou / oks_nms are pinned against the reference's own
functions (tests/golden/oks_nms.npz, oracle/gen_golden_nms.py); the rescoring loop lives inside
COCODataset.evaluate (imports pycocotools, absent here) and is restated from its text only.
"""
import numpy as np

COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    """nms.py:75-96.  g: [3J] flat (x, y, v), d: [n, 3J].  With in_vis_thre the reference's
    ``list(vg > t) and list(vd > t)`` evaluates to the second list: only the candidate's visibilities count."""
    if not isinstance(sigmas, np.ndarray):
        sigmas = COCO_SIGMAS
    vars_ = (sigmas * 2) ** 2
    xg, yg = g[0::3], g[1::3]
    ious = np.zeros((d.shape[0]))
    for n_d in range(d.shape[0]):
        dx = d[n_d, 0::3] - xg
        dy = d[n_d, 1::3] - yg
        e = (dx ** 2 + dy ** 2) / vars_ / ((a_g + a_d[n_d]) / 2 + np.spacing(1)) / 2
        if in_vis_thre is not None:
            e = e[d[n_d, 2::3] > in_vis_thre]
        ious[n_d] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] != 0 else 0.0
    return ious


def oks_nms(kpts, scores, areas, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:99-124 on arrays: kpts [P, 3J], scores [P], areas [P] -> keep indices in selection order."""
    if len(scores) == 0:
        return []
    order = np.asarray(scores).argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, in_vis_thre)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def rescore(keypoints, box_score, in_vis_thre):
    """coco.py:326-341 for one person: mean of the joint scores above in_vis_thre (accumulated in the
    array's dtype, as ``kpt_score + t_s`` does with NumPy scalars) times the box score."""
    kpt_score, valid = 0, 0
    for j in range(keypoints.shape[0]):
        t_s = keypoints[j][2]
        if t_s > in_vis_thre:
            kpt_score = kpt_score + t_s
            valid += 1
    if valid != 0:
        kpt_score = kpt_score / valid
    return kpt_score * box_score


def soft_oks_nms(kpts, scores, areas, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:139-175 (rescore() 'gaussian' :127-136): keep indices in selection order, at most 20."""
    if len(scores) == 0:
        return []
    scores = np.asarray(scores, dtype=np.float64)
    order = scores.argsort()[::-1]
    scores = scores[order]
    keep = []
    while order.size > 0 and len(keep) < 20:
        i = order[0]
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, in_vis_thre)
        order = order[1:]
        scores = scores[1:] * np.exp(-ovr ** 2 / thresh)
        tmp = scores.argsort()[::-1]
        order, scores = order[tmp], scores[tmp]
        keep.append(int(i))
    return keep
