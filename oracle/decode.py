"""Oracle: UDP heatmap decode (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Vectorised NumPy restatement of deep_hrnet/lib/core/inference.py:
  transform_preds  :20-27     get_max_preds   :30-58
  post (DARK)      :60-145    get_final_preds :149-186
Pinned against the reference functions themselves (tests/golden/decode_*.npz,
written by oracle/gen_golden.py); the Gaussian blur inside is the stand-in of
oracle/cv2_standin.py (parity unpinned against real OpenCV).
"""
import numpy as np

from . import cv2_standin as cv2s


def get_max_preds(batch_heatmaps):
    """inference.py:30-58 -- first-max flat arg-max, coords zeroed where max<=0.

    Returns preds f32 [N,J,2] (x,y), maxvals f32 [N,J,1], idx int64 [N,J].
    """
    assert isinstance(batch_heatmaps, np.ndarray) and batch_heatmaps.ndim == 4
    n, j, h, w = batch_heatmaps.shape
    flat = batch_heatmaps.reshape(n, j, h * w)
    idx = np.argmax(flat, axis=2)
    maxvals = np.take_along_axis(flat, idx[..., None], axis=2)
    preds = np.empty((n, j, 2), dtype=np.float32)
    preds[..., 0] = (idx % w).astype(np.float32)
    preds[..., 1] = (idx // w).astype(np.float32)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals.astype(batch_heatmaps.dtype), idx


def dark_prepare_map(m):
    """inference.py:74-82 for one map, fp32 throughout: blur 7x7, min/max
    rescale to the raw max (the value written back at :80), clip [1e-3, 50],
    log.  Returns (log map, rescaled map)."""
    m = np.asarray(m, dtype=np.float32)
    maxori = np.max(m)
    b = cv2s.GaussianBlur(m, (7, 7), 0)
    mx = np.max(b)
    mn = np.min(b)
    with np.errstate(divide="ignore", invalid="ignore"):
        resc = (b - mn) / (mx - mn) * maxori
    lg = np.log(np.clip(resc, np.float32(0.001), np.float32(50)))
    return lg, resc


def taylor_shift(pad, px, py):
    """inference.py:93-144: 7 samples of the replicate-padded log map around the
    integer peak -> gradient D and Hessian H -> shift = H^-1 D (fp64).  A
    singular H (np.linalg.inv raising, :129-132) gives a zero shift."""
    n, j = px.shape
    ai = np.arange(n)[:, None]
    bi = np.arange(j)[None, :]

    def at(dy, dx):
        return pad[ai, bi, py + dy, px + dx]

    I, Ix1, Ix1_ = at(1, 1), at(1, 2), at(1, 0)
    Iy1, Iy1_ = at(2, 1), at(0, 1)
    Ix1y1, Ix1_y1_ = at(2, 2), at(0, 0)
    dx = 0.5 * (Ix1 - Ix1_)
    dy = 0.5 * (Iy1 - Iy1_)
    dxx = Ix1 - 2 * I + Ix1_
    dyy = Iy1 - 2 * I + Iy1_
    dxy = 0.5 * (Ix1y1 - Ix1 - Iy1 + I + I - Ix1_ - Iy1_ + Ix1_y1_)
    det = dxx * dyy - dxy * dxy
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(det != 0.0, 1.0 / det, 0.0)
    return inv * (dyy * dx - dxy * dy), inv * (dxx * dy - dxy * dx)


def post(coords, batch_heatmaps):
    """inference.py:60-145 (DARK / Taylor refinement).  float64 result.

    MUTATES ``batch_heatmaps`` like the reference does (:80): each map is
    replaced by its blurred, rescaled version.
    """
    n, j, h, w = batch_heatmaps.shape
    logmaps = np.empty((n, j, h, w), dtype=np.float32)
    for a in range(n):
        for b in range(j):
            logmaps[a, b], batch_heatmaps[a, b] = dark_prepare_map(batch_heatmaps[a, b])
    pad = np.pad(logmaps.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
    ci = coords.astype(np.int32)
    sx, sy = taylor_shift(pad, ci[..., 0].astype(np.int64), ci[..., 1].astype(np.int64))
    res = ci.astype(np.float64)
    res[..., 0] -= sx
    res[..., 1] -= sy
    return res


def transform_preds(coords, center, scale, output_size):
    """inference.py:20-27 -- UDP unbiased heat-map -> image coordinates."""
    scale = scale * 200.0
    scale_x = scale[0] / (output_size[0] - 1.0)
    scale_y = scale[1] / (output_size[1] - 1.0)
    out = np.zeros(coords.shape)
    out[:, 0] = coords[:, 0] * scale_x + center[0] - scale[0] * 0.5
    out[:, 1] = coords[:, 1] * scale_y + center[1] - scale[1] * 0.5
    return out


def offset_blur(net_output, kpd):
    """inference.py:157-167: split (hm, ox, oy) triplets, scale offsets by KPD,
    blur hm 15x15 and offsets 7x7.  Returns three f32 [N,J,H,W] arrays."""
    out = net_output.copy()
    hm = out[:, 0::3]
    ox = out[:, 1::3] * kpd
    oy = out[:, 2::3] * kpd
    hb = np.empty_like(hm)
    for a in range(hm.shape[0]):
        for b in range(hm.shape[1]):
            hb[a, b] = cv2s.GaussianBlur(hm[a, b], (15, 15), 0)
            ox[a, b] = cv2s.GaussianBlur(ox[a, b], (7, 7), 0)
            oy[a, b] = cv2s.GaussianBlur(oy[a, b], (7, 7), 0)
    return hb, ox, oy


def get_final_preds(target_type, post_process, kpd, batch_heatmaps, center, scale):
    """inference.py:149-186.  ``target_type`` in {'gaussian','offset'}.

    Returns (preds f64 [N,J,2], maxvals f32 [N,J,1], preds_in_input_space,
    idx int64 [N,J]) -- idx is the flat arg-max the coordinates came from
    (extra, for the bit-exact index check).  Mutates batch_heatmaps when
    post_process is set, like the reference.
    """
    h, w = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    if target_type == "gaussian":
        coords, maxvals, idx = get_max_preds(batch_heatmaps)
        if post_process:
            coords = post(coords, batch_heatmaps)
    elif target_type == "offset":
        hb, ox, oy = offset_blur(batch_heatmaps, kpd)
        coords, maxvals, idx = get_max_preds(hb)
        n, j = idx.shape
        ci = coords.astype(np.int64)
        ai = np.arange(n)[:, None]
        bi = np.arange(j)[None, :]
        coords[..., 0] += ox[ai, bi, ci[..., 1], ci[..., 0]]
        coords[..., 1] += oy[ai, bi, ci[..., 1], ci[..., 0]]
    else:
        raise ValueError(target_type)
    preds = coords.copy()
    pin = preds.copy()
    pin[:, :, 0] = pin[:, :, 0] / (w - 1.0) * (4 * w - 1.0)
    pin[:, :, 1] = pin[:, :, 1] / (h - 1.0) * (4 * h - 1.0)
    for i in range(coords.shape[0]):
        preds[i] = transform_preds(coords[i], center[i], scale[i], [w, h])
    return preds, maxvals, pin, idx
