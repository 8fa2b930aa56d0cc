"""GPU parity of the RSN test loop's decode: flip fuse, /255, arg-max, DARK,
transform_preds -- RSN/exps/RSN18.coco.e1.se.36x8x132000_prm/test.py:174-192 -- through
udp_flip_fuse_scaled + udp_decode_gaussian, against oracle/rsn_decode.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rsn_decode as ordec                         # noqa: E402
from udp_pose_amd import synth                                  # noqa: E402
from udp_pose_amd.inference import decode_device                # noqa: E402
from udp_pose_amd.transforms import COCO_FLIP_PAIRS, flip_fuse  # noqa: E402


@pytest.mark.parametrize("n", [1, 7])
def test_rsn_flip_div255_dark_decode(n):
    out = synth.synth_heatmaps(n, 17, 64, 48, seed=3) * np.float32(255.0)          # RSN heat-maps are 0..255
    flipped = synth.synth_heatmaps(n, 17, 64, 48, seed=4) * np.float32(255.0)
    c, s = synth.synth_center_scale(n, seed=6)
    rp, rm, rhm = ordec.rsn_decode(out, flipped, c, s)
    hm = flip_fuse(torch.from_numpy(out).cuda(), torch.from_numpy(flipped).cuda(), COCO_FLIP_PAIRS, False, divisor=255.0)
    np.testing.assert_array_equal(hm.cpu().numpy(), rhm)                           # fused + scaled maps bit-exact
    preds, maxvals, _, idx = decode_device(hm, torch.from_numpy(c.astype(np.float64)),
                                           torch.from_numpy(s.astype(np.float64)), "gaussian", True, 4.0, True)
    np.testing.assert_array_equal(maxvals.cpu().numpy(), rm)
    np.testing.assert_array_equal(idx.cpu().numpy(), rhm.reshape(n, 17, -1).argmax(2))
    np.testing.assert_allclose(preds.cpu().numpy(), rp, rtol=0, atol=1e-3)
