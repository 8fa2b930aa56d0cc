"""Stand-in for the three OpenCV calls on the UDP-Pose hot path (oracle only).

OpenCV (opencv-python==4.5.3.56, deep_hrnet/requirements.txt:4) is third-party
code that is neither vendored under /root/reference nor installed in this
image.  The reference calls it at
  * cv2.GaussianBlur        deep_hrnet/lib/core/inference.py:76,165-167
  * cv2.getAffineTransform  deep_hrnet/tools/infer_utils/utils.py:177
  * cv2.warpAffine          deep_hrnet/pose_engine.py:76-80,
                            deep_hrnet/lib/dataset/JointsDataset.py:227
This module restates OpenCV's *published* algorithm for exactly those uses.
PARITY UNPINNED against real OpenCV: nothing in the reference's tree pins
results at this boundary, so these functions are pinned only by their own
known-answer tests (tests/test_oracle_cv2.py).  The summation order below is
OUR definition (symmetric pair sum, rows then columns, one fp32 rounding per
operation); the HIP kernels follow the same order so that arg-max on blurred
maps is bit-reproducible between oracle and device.

The names mirror cv2 so that ``sys.modules['cv2'] = oracle.cv2_standin`` lets
the reference's own files import in the build container (gen_golden.py).
"""
import numpy as np

INTER_LINEAR = 1
WARP_INVERSE_MAP = 16
BORDER_CONSTANT = 0
BORDER_REFLECT_101 = 4

_SMALL_TAB = {
    1: [1.0],
    3: [0.25, 0.5, 0.25],
    5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
    7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125],
}


def getGaussianKernel(ksize, sigma=0.0):
    """1-D Gaussian taps as float32 (OpenCV getGaussianKernel, ktype CV_32F).

    sigma<=0: fixed table for ksize<=7, else sigma=0.3*((ksize-1)*0.5-1)+0.8.
    """
    ksize = int(ksize)
    if sigma <= 0 and ksize in _SMALL_TAB:
        t = np.asarray(_SMALL_TAB[ksize], dtype=np.float64)
    else:
        sig = sigma if sigma > 0 else 0.3 * ((ksize - 1) * 0.5 - 1.0) + 0.8
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
        t = np.exp(-0.5 / (sig * sig) * x * x)
    cf = t.astype(np.float32)
    s = 1.0 / float(np.sum(cf.astype(np.float64)))
    return (cf.astype(np.float64) * s).astype(np.float32)


def _reflect101_idx(n, r):
    idx = np.arange(-r, n + r)
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def _filter_axis(a, k, axis):
    """Symmetric 1-D filter along ``axis`` in fp32, BORDER_REFLECT_101.

    out = k[r]*x[c] + sum_{d=1..r} k[r+d]*(x[c-d] + x[c+d]), d ascending; each
    product and each sum is rounded to fp32 once (no fused multiply-add).
    """
    r = (len(k) - 1) // 2
    n = a.shape[axis]
    p = np.take(a, _reflect101_idx(n, r), axis=axis)

    def sl(o):
        s = [slice(None)] * a.ndim
        s[axis] = slice(o, o + n)
        return p[tuple(s)]

    acc = (k[r] * sl(r)).astype(np.float32)
    for d in range(1, r + 1):
        pair = (sl(r - d) + sl(r + d)).astype(np.float32)
        acc = (acc + (k[r + d] * pair).astype(np.float32)).astype(np.float32)
    return acc


def GaussianBlur(src, ksize, sigmaX, dst=None, sigmaY=0, borderType=BORDER_REFLECT_101):
    """cv2.GaussianBlur for a 2-D float32 map, square odd ksize, sigma 0."""
    a = np.ascontiguousarray(src, dtype=np.float32)
    assert a.ndim == 2 and ksize[0] == ksize[1] and ksize[0] % 2 == 1
    k = getGaussianKernel(ksize[0], float(sigmaX))
    out = _filter_axis(a, k, axis=1)       # row pass (along x)
    out = _filter_axis(out, k, axis=0)     # column pass (along y)
    return out


def getAffineTransform(src, dst):
    """2x3 float64 matrix M with M @ [x,y,1]^T = [u,v]^T for three point pairs."""
    src = np.asarray(src, dtype=np.float64).reshape(3, 2)
    dst = np.asarray(dst, dtype=np.float64).reshape(3, 2)
    a = np.concatenate([src, np.ones((3, 1))], axis=1)
    m = np.linalg.solve(a, dst)            # (3,2): columns are the two rows of M
    return np.ascontiguousarray(m.T)


def invertAffineTransform(m):
    m = np.asarray(m, dtype=np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]], dtype=np.float64)


def warp_affine_coords(m_dst2src, dsize):
    """Fixed-point source coordinates OpenCV's warpAffine uses (INTER_LINEAR).

    Returns integer source x,y (floor) and the 5-bit fractions fx,fy in [0,32)
    for every destination pixel: AB_BITS=10, INTER_BITS=5,
    X = (round((M01*y+M02)*1024) + 16 + round(M00*x*1024)) >> 5.
    """
    w, h = int(dsize[0]), int(dsize[1])
    m = np.asarray(m_dst2src, dtype=np.float64)
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = np.rint(m[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(m[1, 0] * xs * 1024.0).astype(np.int64)
    x0 = np.rint((m[0, 1] * ys + m[0, 2]) * 1024.0).astype(np.int64) + 16
    y0 = np.rint((m[1, 1] * ys + m[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    return X >> 5, Y >> 5, X & 31, Y & 31


def warpAffine(src, M, dsize, dst=None, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT,
               borderValue=0):
    """cv2.warpAffine for HxWxC (or HxW) uint8, INTER_LINEAR, constant-0 border.

    8-bit path: weights (32-fx)*(32-fy)*32 ... (15-bit, sum 32768),
    value = (sum w*p + 16384) >> 15; each out-of-image tap reads 0.
    """
    src = np.asarray(src)
    assert src.dtype == np.uint8
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    m = np.asarray(M, dtype=np.float64)
    if not (flags & WARP_INVERSE_MAP):
        m = invertAffineTransform(m)
    sx, sy, fx, fy = warp_affine_coords(m, dsize)
    H, W = src.shape[:2]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return v * ok[..., None]

    w00 = ((32 - fx) * (32 - fy) * 32)[..., None]
    w01 = (fx * (32 - fy) * 32)[..., None]
    w10 = ((32 - fx) * fy * 32)[..., None]
    w11 = (fx * fy * 32)[..., None]
    acc = (w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) +
           w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1) + 16384) >> 15
    out = np.clip(acc, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out
