"""Keypoint rescoring + OKS-NMS on the GPU with the reference's call shapes.

Mirrors deep_hrnet/lib/nms/nms.py:99-124 (``oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None)`` ->
keep indices) and the per-image loop of deep_hrnet/lib/dataset/coco.py:321-356 (``rescore_and_nms``: all
images of an evaluation in ONE launch) and ``soft_oks_nms`` (:139-175, TEST.SOFT_NMS).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def _run(kpts, areas, box_scores, offsets, in_vis_thre, rescore, oks_thre, sigmas, oks_vis_thre, device, soft=False):
    kpts = np.ascontiguousarray(kpts, dtype=np.float32)
    p, j = kpts.shape[0], kpts.shape[1]
    sig = COCO_SIGMAS if not isinstance(sigmas, np.ndarray) else sigmas
    if sig.shape[0] != j:
        raise ValueError("%d sigmas for %d joints" % (sig.shape[0], j))
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    dev = torch.device(device)
    k_d = torch.from_numpy(kpts).to(dev)
    a_d = torch.from_numpy(np.ascontiguousarray(areas, dtype=np.float64)).to(dev)
    b_d = torch.from_numpy(np.ascontiguousarray(box_scores, dtype=np.float64)).to(dev)
    o_d = torch.from_numpy(offsets).to(dev)
    v_d = torch.from_numpy(np.ascontiguousarray((sig * 2) ** 2, dtype=np.float64)).to(dev)
    scores = torch.empty(p, dtype=torch.float64, device=dev)
    rank = torch.empty(p, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().udp_oks_nms(k_d.data_ptr(), a_d.data_ptr(), b_d.data_ptr(), o_d.data_ptr(),
                                      offsets.ctypes.data_as(C.POINTER(C.c_int32)), len(offsets) - 1, j, v_d.data_ptr(),
                                      float(in_vis_thre), int(rescore), float(oks_thre),
                                      int(oks_vis_thre is not None), float(oks_vis_thre or 0.0), int(soft), scores.data_ptr(),
                                      rank.data_ptr(), _lib.stream_ptr()))
    return scores.cpu().numpy(), rank.cpu().numpy()


def oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None, device="cuda"):
    """nms.py:99-124: list of {'keypoints' [J,3], 'score', 'area'} -> indices to keep, best first."""
    if len(kpts_db) == 0:
        return []
    kpts = np.stack([np.asarray(d["keypoints"], dtype=np.float32).reshape(-1, 3) for d in kpts_db])
    scores = np.array([d["score"] for d in kpts_db], dtype=np.float64)
    areas = np.array([d["area"] for d in kpts_db], dtype=np.float64)
    _, rank = _run(kpts, areas, scores, [0, len(kpts_db)], 0.0, False, thresh, sigmas, in_vis_thre, device)
    keep = np.where(rank >= 0)[0]
    return [int(i) for i in keep[np.argsort(rank[keep])]]


def soft_oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None, device="cuda"):
    """nms.py:139-175: Gaussian score decay instead of suppression, at most 20 detections."""
    if len(kpts_db) == 0:
        return []
    kpts = np.stack([np.asarray(d["keypoints"], dtype=np.float32).reshape(-1, 3) for d in kpts_db])
    scores = np.array([d["score"] for d in kpts_db], dtype=np.float64)
    areas = np.array([d["area"] for d in kpts_db], dtype=np.float64)
    _, rank = _run(kpts, areas, scores, [0, len(kpts_db)], 0.0, False, thresh, sigmas, in_vis_thre, device, soft=True)
    keep = np.where(rank >= 0)[0]
    return [int(i) for i in keep[np.argsort(rank[keep])]]


def rescore_and_nms(preds, all_boxes, image_ids, in_vis_thre, oks_thre, device="cuda", soft_nms=False):
    """coco.py:306-356 for a whole evaluation: preds [P,J,3] (x, y, score), all_boxes [P,6] (center 0:2,
    scale 2:4, area 4, score 5), image_ids [P].  Returns per image id the kept persons
    {'keypoints', 'center', 'scale', 'area', 'score' (rescored), 'image'} in keep order."""
    preds = np.asarray(preds)
    image_ids = np.asarray(image_ids)
    if preds.shape[0] == 0:
        return {}
    first = {}
    for idx, im in enumerate(image_ids.tolist()):
        first.setdefault(im, []).append(idx)            # defaultdict(list) order of coco.py:317-319
    perm = np.array([i for im in first for i in first[im]], dtype=np.int64)
    offsets = np.cumsum([0] + [len(first[im]) for im in first]).astype(np.int32)
    scores, rank = _run(preds[perm], all_boxes[perm, 4], all_boxes[perm, 5], offsets, in_vis_thre, True, oks_thre, None,
                        None, device, soft=soft_nms)
    out = {}
    for n, im in enumerate(first):
        a, b = offsets[n], offsets[n + 1]
        keep = np.where(rank[a:b] >= 0)[0]
        keep = keep[np.argsort(rank[a:b][keep])]
        out[im] = [{"keypoints": preds[perm[a + k]], "center": all_boxes[perm[a + k]][0:2],
                    "scale": all_boxes[perm[a + k]][2:4], "area": all_boxes[perm[a + k]][4],
                    "score": scores[a + k], "image": im} for k in keep]
    return out


def coco_keypoint_results(kept_by_image, cat_id=1):
    """coco.py:397-428 (``_coco_keypoint_results_one_category_kernel``): the list ``json.dump``ed as the COCO
    keypoint result file -- one record per kept person with the 3J flat keypoints as float64, the rescored
    score, center and scale.  ``kept_by_image``: the dict returned by ``rescore_and_nms`` (or a list of
    per-image lists like the reference's ``oks_nmsed_kpts``).  ``cat_id`` 1 = 'person' (coco.py:69-77)."""
    groups = kept_by_image.values() if isinstance(kept_by_image, dict) else kept_by_image
    out = []
    for img_kpts in groups:
        if len(img_kpts) == 0:
            continue
        for k in img_kpts:
            kp = np.asarray(k["keypoints"], dtype=np.float64).reshape(-1)        # x, y, score per joint
            out.append({"image_id": k["image"], "category_id": cat_id, "keypoints": [float(v) for v in kp],
                        "score": float(k["score"]), "center": [float(v) for v in k["center"]],
                        "scale": [float(v) for v in k["scale"]]})
    return out


def write_coco_keypoint_results(kept_by_image, res_file, cat_id=1):
    """coco.py:369-395: ``json.dump(results, f, sort_keys=True, indent=4)``."""
    import json
    with open(res_file, "w") as f:
        json.dump(coco_keypoint_results(kept_by_image, cat_id), f, sort_keys=True, indent=4)
