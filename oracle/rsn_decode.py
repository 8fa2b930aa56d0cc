"""Oracle: the RSN test loop's decode (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates RSN/exps/RSN18.coco.e1.se.36x8x132000_prm/test.py:174-192: flip fuse, ``outputs/255.0``,
get_max_preds, post (DARK), transform_preds with the fixed [48, 64] map size -- the functions are
verbatim copies of deep_hrnet/lib/core/inference.py, so oracle/decode.py is reused.
"""
import numpy as np

from . import decode as odec
from . import flip as oflip


def rsn_decode(outputs, outputs_flipped, centers, scales, flip_pairs=oflip.COCO_FLIP_PAIRS):
    if outputs_flipped is not None:
        outputs = (outputs + oflip.flip_back(outputs_flipped, flip_pairs)) * 0.5       # :179-181
    outputs = (outputs / 255.0).astype(np.float32)                                      # :185
    preds, maxvals, _ = odec.get_max_preds(outputs)
    preds = odec.post(preds, outputs.copy())
    for i in range(preds.shape[0]):
        preds[i] = odec.transform_preds(preds[i], centers[i], scales[i], [outputs.shape[3], outputs.shape[2]])
    return preds, maxvals, outputs
