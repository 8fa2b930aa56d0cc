"""GPU parity tests of the individual kernels, called through the C ABI (ctypes).

Every test compares the HIP result with the CPU oracle (oracle/, itself pinned
to the reference by tests/test_oracle_golden.py) and, where a fixture exists,
with the reference's own outputs in tests/golden.  Tolerances follow the
north star: fp32 heat-maps 1e-3, arg-max indices bit-exact.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import data as odata            # noqa: E402
from oracle import decode as odec           # noqa: E402
from oracle import flip as oflip            # noqa: E402
from oracle import loss as oloss            # noqa: E402
from udp_pose_amd import _lib, synth        # noqa: E402
from udp_pose_amd import f16x2              # noqa: E402
from udp_pose_amd import inference as uinf  # noqa: E402
from udp_pose_amd import transforms as utr  # noqa: E402


def _adversarial(hm, k):
    hm[0, 0] = -np.abs(hm[0, 0]) - 0.1
    hm[0, 1 * k, 10, 7] = hm[0, 1 * k].max() + 0.5
    hm[0, 1 * k, 30, 40] = hm[0, 1 * k, 10, 7]
    hm[1, 2 * k, 0, 0] = 2.0
    hm[1, 3 * k, 63, 47] = 2.0
    hm[2, 4 * k] = 0.25
    return hm


# ------------------------------------------------------------------ fused conv
def _conv_case(dtype, ks, stride, cin, cout, h, w, n, relu, res, nup, nchw_out=False, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    ws = dtype == "f16x2-ws"          # split fp16 on the weight-stationary kernel (fragment-major weights)
    dtype = "f16x2" if ws else dtype
    h2 = dtype == "f16x2"
    q = (lambda t: f16x2.decode(f16x2.encode(t))) if h2 else (lambda t: t)     # operands exactly as the device holds them
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    x = q(torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32)).to(tdt))
    wt = q(torch.from_numpy((rng.standard_normal((cout, cin, ks, ks)) * np.sqrt(2.0 / (cin * ks * ks))).astype(np.float32)).to(tdt))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    r = q(torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)).to(tdt)) if res else None
    ups = [q(torch.from_numpy(rng.standard_normal((n, cout, ho >> (u + 1), wo >> (u + 1))).astype(np.float32)).to(tdt))
           for u in range(nup)]
    # reference: stock fp32 conv on the (possibly bf16-rounded) operands
    y = F.conv2d(x.float(), wt.float(), bias, stride=stride, padding=pad)
    if r is not None:
        y = y + r.float()
    for u, t in enumerate(ups):
        y = y + F.interpolate(t.float(), scale_factor=2 ** (u + 1), mode="nearest")
    if relu:
        y = F.relu(y)
    # device operands
    cout_pad = (cout + 31) // 32 * 32
    wp = torch.zeros(ks * ks, cout_pad, cin, dtype=tdt)
    wp[:, :cout] = wt.permute(2, 3, 0, 1).reshape(ks * ks, cout, cin)
    bp = torch.zeros(cout_pad)
    bp[:cout] = bias
    nhwc = lambda t: (f16x2.encode(t.permute(0, 2, 3, 1)) if h2 else t.permute(0, 2, 3, 1).contiguous()).cuda()
    wsp, wexp = f16x2.pack_weights_ws(wp) if ws else (None, 0)
    d_x, d_w, d_b = nhwc(x), (wsp if ws else f16x2.encode(wp) if h2 else wp).cuda(), bp.cuda()
    d_r = nhwc(r) if r is not None else None
    d_u = [nhwc(t) for t in ups] + [None] * (3 - nup)
    op = _lib.ConvOp()
    op.kind, op.ks, op.stride, op.relu = _lib.UDP_OP_CONV, ks, stride, int(relu)
    op.cin, op.cout, op.cout_pad = cin, cout, cout_pad
    op.hin, op.win, op.hout, op.wout = h, w, ho, wo
    op.n_up = nup
    op.wfmt, op.wexp = int(ws), wexp
    for u in range(nup):
        op.up_shift[u] = u + 1
    if nchw_out:
        op.out_buf = _lib.UDP_BUF_OUTPUT
        out = torch.full((n, cout, ho, wo), float("nan"), dtype=torch.float32, device="cuda")
    elif h2:
        out = torch.full((n, ho, wo, 2, cout), float("nan"), dtype=torch.float16, device="cuda")
    else:
        out = torch.full((n, ho, wo, cout), float("nan"), dtype=tdt, device="cuda")
    _lib.check(_lib.lib().udp_conv2d_fused(C.byref(op), _lib.DTYPES[dtype], n,
                                           _lib.ptr(d_x), _lib.ptr(d_w), _lib.ptr(d_b), _lib.ptr(d_r),
                                           _lib.ptr(d_u[0]), _lib.ptr(d_u[1]), _lib.ptr(d_u[2]), _lib.ptr(out),
                                           _lib.stream_ptr()))
    torch.cuda.synchronize()
    got = (f16x2.decode(out) if h2 and not nchw_out else out.float()).cpu()
    if not nchw_out:
        got = got.permute(0, 3, 1, 2)
    return got.numpy(), y.numpy()


CONV_CASES = [
    # ks, stride, cin, cout, h, w, n, relu, res, nup
    (3, 1, 32, 32, 64, 48, 3, True, True, 0),      # BasicBlock conv2, branch 0 (a2)
    (3, 1, 64, 64, 32, 24, 3, True, False, 0),
    (3, 1, 128, 128, 16, 12, 5, True, True, 0),
    (3, 1, 256, 256, 8, 6, 7, True, True, 0),      # multi-image tiles (G > 1, ragged last group)
    (3, 2, 64, 64, 128, 96, 2, True, False, 0),    # stem conv2
    (3, 2, 32, 64, 64, 48, 3, True, True, 2),      # fuse chain end: + identity + 2 upsampled terms (a4)
    (3, 2, 256, 64, 64, 48, 2, True, False, 0),    # transition1 new branch (a5)
    (3, 1, 256, 32, 64, 48, 2, True, False, 0),    # transition1[0]
    (1, 1, 64, 256, 64, 48, 2, True, True, 0),     # Bottleneck conv3 (a3)
    (1, 1, 256, 64, 64, 48, 2, True, False, 0),
    (1, 1, 128, 32, 16, 12, 3, False, False, 0),   # fuse 1x1 (low-res temp)
    (1, 1, 32, 128, 64, 48, 2, True, False, 3),    # last-module f_00 + three upsampled terms
    (3, 1, 32, 32, 8, 8, 1, False, False, 0),      # single tiny image
    (3, 1, 48, 48, 96, 72, 2, True, True, 0),      # W48 branch 0 at 384x288: ragged K chunk (48 = 32 + 16), 2 column tiles
    (3, 1, 96, 96, 48, 36, 2, True, True, 0),      # W48 branch 1
    (3, 2, 48, 96, 96, 72, 2, True, True, 1),      # W48 fuse chain end
    (1, 1, 192, 48, 24, 18, 2, False, False, 0),   # W48 fuse 1x1
    (3, 1, 16, 16, 24, 16, 2, True, True, 0),      # width-16 mini net of the reference fixture
    (3, 1, 32, 32, 64, 48, 5, True, True, 1),      # 3x3 stride 1 with an upsampled addend in the epilogue
    (3, 1, 64, 64, 32, 24, 7, False, True, 0),     # no ReLU, residual, odd image count
    (3, 1, 192, 192, 24, 18, 3, True, True, 0),    # W48 branch 2 (6 K chunks, ragged row tiles)
    (3, 1, 384, 384, 12, 9, 5, True, True, 0),     # W48 branch 3 (12 K chunks, two cout blocks per tile)
]


@pytest.mark.parametrize("dtype", ["f32", "f16x2", "f16x2-ws", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "k%ds%d_%d-%d_%dx%d_n%d" % c[:7])
def test_fused_conv_matches_torch_fp32(case, dtype):
    got, ref = _conv_case(dtype, *case)
    assert not np.isnan(got).any(), "output not fully written"
    scale = max(1.0, float(np.abs(ref).max()))
    # fp32: exact-fp32 MFMA, only the summation order differs.  f16x2: operands are exact in both (22-bit
    # hi + lo pairs); three fp16 MFMAs per product drop the lo*lo term (2^-22 relative), fp32 accumulate, the
    # output is rounded to 22 bits -> same gate as fp32.  bf16: operands are bf16-exact in both, error =
    # fp32 accumulation order + one bf16 rounding of the output (2^-9 relative).
    tol = 6e-3 * scale if dtype == "bf16" else 1e-4 * scale
    np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
    if dtype.startswith("f16x2"):
        assert np.abs(got - ref).max() <= 2e-5 * scale, np.abs(got - ref).max()


@pytest.mark.parametrize("dtype", ["f32", "f16x2", "bf16"])
def test_final_layer_nchw_output(dtype):
    got, ref = _conv_case(dtype, 1, 1, 128, 17, 64, 48, 3, False, False, 0, nchw_out=True)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-2 if dtype == "bf16" else 1e-4)
    got, ref = _conv_case(dtype, 1, 1, 128, 51, 64, 48, 2, False, False, 0, nchw_out=True, seed=4)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-2 if dtype == "bf16" else 1e-4)


def test_conv_rejects_bad_shapes():
    op = _lib.ConvOp()
    op.kind, op.ks, op.stride, op.cin, op.cout, op.cout_pad = _lib.UDP_OP_CONV, 3, 1, 24, 32, 32
    op.hin = op.hout = 8
    op.win = op.wout = 8
    t = torch.zeros(8 * 8 * 32, device="cuda")
    rc = _lib.lib().udp_conv2d_fused(C.byref(op), _lib.UDP_F32, 1, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None,
                                     None, None, _lib.ptr(t), _lib.stream_ptr())
    assert rc == -3 and b"multiple of 16" in _lib.lib().udp_last_error()
    rc = _lib.lib().udp_conv2d_fused(C.byref(op), _lib.UDP_F32, 1, None, None, None, None, None, None, None, None, None)
    assert rc == -1


# ------------------------------------------------------------------ decode
@pytest.mark.parametrize("tt,post,k", [("gaussian", False, 1), ("gaussian", True, 1), ("offset", False, 3)])
@pytest.mark.parametrize("cs_dtype", [np.float32, np.float64])
def test_decode_matches_oracle_and_reference(golden_dir, tt, post, k, cs_dtype):
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    hm = _adversarial(synth.synth_heatmaps(4, 17, 64, 48, seed=21, channels_per_joint=k), k)
    c, s = g["center"].astype(cs_dtype), g["scale"].astype(cs_dtype)
    cfg = {"MODEL": {"TARGET_TYPE": tt}, "TEST": {"POST_PROCESS": post}, "LOSS": {"KPD": 4.0}}
    before = hm.copy()
    preds, maxvals, pin, idx = uinf.get_final_preds(cfg, hm, c, s, return_idx=True)
    np.testing.assert_array_equal(hm, before)
    with np.errstate(all="ignore"):
        rp, rm, rpin, ridx = odec.get_final_preds(tt, post, 4.0, hm.copy(), c, s)
    np.testing.assert_array_equal(idx, ridx)                       # arg-max indices bit-exact
    np.testing.assert_array_equal(maxvals, rm)
    assert preds.dtype == rp.dtype
    if post:
        np.testing.assert_allclose(preds, rp, rtol=0, atol=1e-3, equal_nan=True)   # north-star tolerance
        np.testing.assert_allclose(pin, rpin, rtol=0, atol=1e-3, equal_nan=True)
        assert np.nanmax(np.abs(preds - rp)) < 1e-4                # what we actually reach
    else:
        np.testing.assert_array_equal(preds, rp)                   # fp32 path reproduces bit for bit
        np.testing.assert_array_equal(pin, rpin)
    if cs_dtype == np.float32:                                     # the reference's own outputs
        tag = "%s%s" % (tt, "_post" if post else "")
        np.testing.assert_array_equal(maxvals, g["maxvals_" + tag])
        np.testing.assert_allclose(preds, g["preds_" + tag], rtol=0, atol=1e-3 if post else 0, equal_nan=True)


def test_decode_384x288_maps_and_ragged_batch():
    hm = synth.synth_heatmaps(3, 16, 96, 72, seed=5)
    c, s = synth.synth_center_scale(3, seed=9)
    cfg = {"MODEL": {"TARGET_TYPE": "gaussian"}, "TEST": {"POST_PROCESS": True}, "LOSS": {"KPD": 4.0}}
    preds, maxvals, pin, idx = uinf.get_final_preds(cfg, hm, c, s, return_idx=True)
    rp, rm, rpin, ridx = odec.get_final_preds("gaussian", True, 4.0, hm.copy(), c, s)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_allclose(preds, rp, rtol=0, atol=1e-4)


def test_decode_full_batch_properties():
    """Config-2 size (N=64): decode is per-map independent and translation-consistent."""
    hm = synth.synth_heatmaps(64, 17, 64, 48, seed=77)
    c, s = synth.synth_center_scale(64, seed=3)
    cfg = {"MODEL": {"TARGET_TYPE": "gaussian"}, "TEST": {"POST_PROCESS": True}, "LOSS": {"KPD": 4.0}}
    p_all, m_all, _ = uinf.get_final_preds(cfg, hm, c, s)
    p_one, m_one, _ = uinf.get_final_preds(cfg, hm[17:18], c[17:18], s[17:18])
    np.testing.assert_array_equal(p_all[17:18], p_one)
    np.testing.assert_array_equal(m_all[17:18], m_one)
    # shifting the centre by d shifts every keypoint by d (transform_preds is affine in center)
    p_shift, _, _ = uinf.get_final_preds(cfg, hm, c + np.float32(8.0), s)
    np.testing.assert_allclose(p_shift - p_all, 8.0, atol=1e-3)


def test_gaussian_taps_equal_oracle():
    from oracle import cv2_standin
    for k in (3, 5, 7, 9, 11, 15):
        np.testing.assert_array_equal(uinf.gaussian_taps(k), cv2_standin.getGaussianKernel(k))


# ------------------------------------------------------------------ flip
@pytest.mark.parametrize("is_offset", [False, True])
def test_flip_fuse_matches_oracle(golden_dir, is_offset):
    g = np.load(os.path.join(golden_dir, "flip.npz"))
    a = g["b"] if is_offset else g["a"]
    want = g["fb"] if is_offset else g["fa"]
    fb = (utr.flip_back_offset if is_offset else utr.flip_back)(a, oflip.COCO_FLIP_PAIRS)
    np.testing.assert_array_equal(fb, want)                        # reference's own output
    rng = np.random.Generator(np.random.PCG64(5))
    c = 51 if is_offset else 17
    x = rng.standard_normal((64, c, 64, 48)).astype(np.float32)
    y = rng.standard_normal((64, c, 64, 48)).astype(np.float32)
    got = utr.flip_fuse(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), oflip.COCO_FLIP_PAIRS, is_offset)
    np.testing.assert_array_equal(got.cpu().numpy(), oflip.flip_fuse(x, y, oflip.COCO_FLIP_PAIRS, is_offset))


# ------------------------------------------------------------------ data path
def test_warp_affine_matches_oracle():
    from udp_pose_amd import pose_engine as pe
    frame = synth.synth_frame_u8(480, 640, seed=2)
    boxes = synth.synth_boxes(5, seed=4)
    boxes[0] = [-40, -30, 120, 300]                                # crop hangs over the frame border
    ref, cs = odata.engine_preprocess(frame, boxes, [192, 256])
    cs2 = pe.box_to_center_scale(boxes, [192, 256])
    np.testing.assert_array_equal(cs, cs2)
    mats = np.stack([pe.engine_affine_dst2src(c[:2], c[2:], [192, 256]) for c in cs2])
    got = pe.warp_affine_device(torch.from_numpy(frame).cuda(), mats, (256, 192)).cpu().numpy()
    assert got.shape == (5, 3, 256, 192)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_warp_affine_udp_matrix():
    """Training-side unbiased warp: get_warpmatrix (dst->src) with WARP_INVERSE_MAP."""
    from oracle import cv2_standin as cv2s
    from udp_pose_amd import pose_engine as pe
    frame = synth.synth_frame_u8(300, 400, seed=8)
    c = np.array([210.0, 140.0], np.float32)
    s = np.array([1.2, 1.6], np.float32)
    m = odata.get_warpmatrix(25.0, c * 2.0, np.array([192, 256]) - 1.0, s)
    ref = cv2s.warpAffine(frame, m, (192, 256), flags=cv2s.WARP_INVERSE_MAP | cv2s.INTER_LINEAR)
    got = pe.warp_affine_device(torch.from_numpy(frame).cuda(), m.astype(np.float64)[None], (256, 192)).cpu().numpy()
    np.testing.assert_allclose(got[0], odata.normalize_crop(ref), rtol=0, atol=2e-6)


@pytest.mark.parametrize("tt", ["gaussian", "offset"])
def test_generate_target_matches_reference(golden_dir, tt):
    g = np.load(os.path.join(golden_dir, "data.npz"))
    joints = torch.from_numpy(np.ascontiguousarray(g["tgt_joints"][..., :2])).cuda()
    vis = torch.from_numpy(np.ascontiguousarray(g["tgt_vis"][..., 0])).cuda()
    n, j = joints.shape[:2]
    k = 3 if tt == "offset" else 1
    target = torch.full((n, j * k, 64, 48), float("nan"), device="cuda")
    weight = torch.full((n, j), float("nan"), device="cuda")
    fn = _lib.lib().udp_target_offset if tt == "offset" else _lib.lib().udp_target_gaussian
    _lib.check(fn(_lib.ptr(joints), _lib.ptr(vis), n, j, 192, 256, 48, 64, 4.0 if tt == "offset" else 2.0,
                  _lib.ptr(target), _lib.ptr(weight), _lib.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(weight.cpu().numpy()[..., None], g["target_weight_" + tt])
    t = target.cpu().numpy()
    assert np.array_equal(t != 0, g["target_" + tt] != 0)          # support (disk / patch mask) exact
    np.testing.assert_allclose(t, g["target_" + tt], rtol=0, atol=1e-7)


@pytest.mark.parametrize("is_offset", [False, True])
def test_mse_loss_matches_reference(golden_dir, is_offset):
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    pre = "off_" if is_offset else "mse_"
    p, t, w = (torch.from_numpy(g[pre + k]).cuda() for k in ("pred", "gt", "w"))
    b, c = p.shape[:2]
    j = c // 3 if is_offset else c
    loss = torch.zeros(2, dtype=torch.float64, device="cuda")
    grad = torch.empty_like(p)
    _lib.check(_lib.lib().udp_mse_loss(_lib.ptr(p), _lib.ptr(t), _lib.ptr(w.contiguous()), b, j, p.shape[2] * p.shape[3],
                                       int(is_offset), _lib.ptr(loss), _lib.ptr(grad), _lib.stream_ptr()))
    l = loss.cpu().numpy()
    if is_offset:
        np.testing.assert_allclose(l[0], g["off_loss_hm"], rtol=1e-6)
        np.testing.assert_allclose(l[1], g["off_loss_os"], rtol=1e-6)
        rl = oloss.joints_mse_loss_offset(g["off_pred"], g["off_gt"], g["off_w"])
        np.testing.assert_allclose(l, rl[:2], rtol=1e-12)
    else:
        np.testing.assert_allclose(l[0], g["mse_loss"], rtol=1e-6)
        assert l[1] == 0
    np.testing.assert_allclose(grad.cpu().numpy(), g[pre + "grad"], rtol=1e-5, atol=1e-9)


def test_dark_decode_recovers_subpixel_mean_on_device():
    """KAT: noiseless Gaussians at sub-pixel means decode (POST_PROCESS on) to those
    means.  preds_in_input_space = coord/(W-1)*(4W-1) (inference.py:177-179) is inverted to heat-map pixels."""
    ys = np.arange(64, dtype=np.float64)[:, None]
    xs = np.arange(48, dtype=np.float64)[None, :]
    mus = [(20.3, 30.7), (10.5, 12.25), (40.1, 50.9), (1.2, 1.4), (46.6, 62.3)]
    hm = np.stack([np.exp(-((xs - mx) ** 2 + (ys - my) ** 2) / 8.0) for mx, my in mus])[None].astype(np.float32)
    c = torch.tensor([[96.0, 128.0]], dtype=torch.float64, device="cuda")
    s = torch.tensor([[192.0 / 200, 256.0 / 200]], dtype=torch.float64, device="cuda")
    preds, maxvals, pin, idx = uinf.decode_device(torch.from_numpy(hm).cuda(), c, s, "gaussian", True, 4.0, False)
    got = pin.cpu().numpy()[0] / np.array([4 * 48 - 1, 4 * 64 - 1]) * np.array([47, 63])
    np.testing.assert_allclose(got[:3], np.asarray(mus[:3]), atol=0.05)     # interior peaks; the two border peaks are
    # biased by the reference's own border handling (reflect-101 blur, replicate-padded log map)
    # and the same through the oracle, borders included: identical within 1e-3 px
    rp, _, rpin, _ = odec.get_final_preds("gaussian", True, 4.0, hm.copy(), c.cpu().numpy(), s.cpu().numpy())
    np.testing.assert_allclose(pin.cpu().numpy(), rpin, atol=1e-3)
    np.testing.assert_allclose(preds.cpu().numpy(), rp, atol=1e-3)
