"""Oracle: UDP data path (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates
  get_warpmatrix        deep_hrnet/lib/dataset/JointsDataset.py:29-49
  rotate_points         JointsDataset.py:51-73
  generate_target       JointsDataset.py:291-385 (gaussian :301-348, offset :349-381)
  box -> center/scale   deep_hrnet/pose_engine.py:45-63
  engine affine (biased) deep_hrnet/tools/infer_utils/utils.py:157-177
  engine crop+normalize deep_hrnet/pose_engine.py:69-85,40-43
"""
import math

import numpy as np

from . import cv2_standin as cv2s

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_warpmatrix(theta, size_input, size_dst, size_target):
    """JointsDataset.py:29-49: UDP unbiased dst->src 2x3 float32 matrix."""
    size_target = size_target * 200.0
    t = theta / 180.0 * math.pi
    m = np.zeros((2, 3), dtype=np.float32)
    sx = size_target[0] / size_dst[0]
    sy = size_target[1] / size_dst[1]
    c, s = math.cos(t), math.sin(t)
    m[0, 0] = c * sx
    m[0, 1] = s * sy
    m[0, 2] = -0.5 * size_target[0] * c - 0.5 * size_target[1] * s + 0.5 * size_input[0]
    m[1, 0] = -s * sx
    m[1, 1] = c * sy
    m[1, 2] = 0.5 * size_target[0] * s - 0.5 * size_target[1] * c + 0.5 * size_input[1]
    return m


def rotate_points(src_points, angle, c, dst_img_shape, size_target, do_clip=True):
    """JointsDataset.py:51-73: joints -> crop coordinates (same quirk: scale_x
    comes from dst_img_shape[0], clip uses [1]/[0] the other way round)."""
    size_target = size_target * 200.0
    sx = (dst_img_shape[0] - 1.0) / size_target[0]
    sy = (dst_img_shape[1] - 1.0) / size_target[1]
    rad = angle / 180.0 * math.pi
    rs, rc = -math.sin(rad), math.cos(rad)
    out = np.zeros(src_points.shape, dtype=src_points.dtype)
    x = src_points[:, 0] - c[0]
    y = src_points[:, 1] - c[1]
    out[:, 0] = rc * x + rs * y
    out[:, 1] = -rs * x + rc * y
    out[:, 0] += size_target[0] * 0.5
    out[:, 1] += size_target[1] * 0.5
    out[:, 0] *= sx
    out[:, 1] *= sy
    if do_clip:
        out[:, 0] = np.clip(out[:, 0], 0, dst_img_shape[1] - 1)
        out[:, 1] = np.clip(out[:, 1], 0, dst_img_shape[0] - 1)
    return out


def generate_target(joints, joints_vis, target_type, image_size, heatmap_size, sigma=2, kpd=4.0):
    """JointsDataset.py:291-385 for one sample.

    joints [J,>=2] crop coordinates, joints_vis [J,>=1]; image_size/heatmap_size
    are [w,h].  Returns target f32 ([J,h,w] gaussian / [3J,h,w] offset) and
    target_weight f32 [J,1].
    """
    image_size = np.asarray(image_size)
    heatmap_size = np.asarray(heatmap_size)
    nj = joints.shape[0]
    tw = np.ones((nj, 1), dtype=np.float32)
    tw[:, 0] = joints_vis[:, 0]
    W, H = int(heatmap_size[0]), int(heatmap_size[1])
    stride = (image_size - 1.0) / (heatmap_size - 1.0)
    if target_type == "gaussian":
        target = np.zeros((nj, H, W), dtype=np.float32)
        tmp = sigma * 3
        size = 2 * tmp + 1
        xs = np.arange(0, size, 1, np.float32)
        ys = xs[:, None]
        for j in range(nj):
            mu_x = int(joints[j][0] / stride[0] + 0.5)
            mu_y = int(joints[j][1] / stride[1] + 0.5)
            ul = [int(mu_x - tmp), int(mu_y - tmp)]
            br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
            if ul[0] >= W or ul[1] >= H or br[0] < 0 or br[1] < 0:
                tw[j] = 0
                continue
            x0 = size // 2 + (joints[j][0] / stride[0] - mu_x)
            y0 = size // 2 + (joints[j][1] / stride[1] - mu_y)
            g = np.exp(-((xs - x0) ** 2 + (ys - y0) ** 2) / (2 * sigma ** 2))
            gx = max(0, -ul[0]), min(br[0], W) - ul[0]
            gy = max(0, -ul[1]), min(br[1], H) - ul[1]
            ix = max(0, ul[0]), min(br[0], W)
            iy = max(0, ul[1]), min(br[1], H)
            if tw[j] > 0.5:
                target[j][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    elif target_type == "offset":
        target = np.zeros((nj, 3, H * W), dtype=np.float32)
        fx, fy = np.meshgrid(np.arange(0, W), np.arange(0, H))
        fx = fx.reshape(-1)
        fy = fy.reshape(-1)
        for j in range(nj):
            xo = (joints[j][0] / stride[0] - fx) / kpd
            yo = (joints[j][1] / stride[1] - fy) / kpd
            dis = xo ** 2 + yo ** 2
            keep = np.where((dis <= 1) & (dis >= 0))[0]
            if tw[j] > 0.5:
                target[j, 0, keep] = 1
                target[j, 1, keep] = xo[keep]
                target[j, 2, keep] = yo[keep]
        target = target.reshape(nj * 3, H, W)
    else:
        raise ValueError(target_type)
    return target, tw


def box_to_center_scale(boxes_xyxy, input_shape, pixel_std=200):
    """pose_engine.py:45-63: xyxy -> (cx,cy,w,h) with aspect fix, /200, *1.25.
    float32 arithmetic (the reference works on a float32 torch tensor)."""
    b = np.asarray(boxes_xyxy, dtype=np.float32)
    out = np.empty_like(b)
    out[:, 0] = (b[:, 0] + b[:, 2]) / np.float32(2)
    out[:, 1] = (b[:, 1] + b[:, 3]) / np.float32(2)
    out[:, 2] = b[:, 2] - b[:, 0]
    out[:, 3] = b[:, 3] - b[:, 1]
    r = np.float32(input_shape[0] / input_shape[1])
    mask = out[:, 2] > out[:, 3] * r
    out[mask, 3] = out[mask, 2] / r
    out[~mask, 2] = out[~mask, 3] * r
    out[:, 2:] /= np.float32(pixel_std)
    out[:, 2:] *= np.float32(1.25)
    return out


def engine_affine_points(center, scale, patch_size):
    """tools/infer_utils/utils.py:157-177 with rot=0: the three src / dst points
    (float32) fed to cv2.getAffineTransform (biased: dst uses W,H not W-1,H-1)."""
    scale_tmp = np.asarray(scale, dtype=np.float64) * 200
    src_w = scale_tmp[0]
    dst_w, dst_h = patch_size[0], patch_size[1]
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center) + np.array([0, src_w * -0.5])
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], dtype=np.float32)
    d = src[0] - src[1]
    src[2] = src[1] + np.array([-d[1], d[0]], dtype=np.float32)
    d = dst[0] - dst[1]
    dst[2] = dst[1] + np.array([-d[1], d[0]], dtype=np.float32)
    return src, dst


def engine_affine(center, scale, patch_size):
    """src->dst 2x3 float64 matrix of the engine crop (utils.py:177)."""
    src, dst = engine_affine_points(center, scale, patch_size)
    return cv2s.getAffineTransform(src, dst)


def normalize_crop(patch_u8):
    """pose_engine.py:40-43: ToTensor (/255, HWC->CHW) + Normalize(mean,std)."""
    x = patch_u8.astype(np.float32) / np.float32(255)
    x = np.transpose(x, (2, 0, 1))
    mean = np.asarray(IMAGENET_MEAN, dtype=np.float32)[:, None, None]
    std = np.asarray(IMAGENET_STD, dtype=np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32)


def engine_preprocess(img, boxes_xyxy, input_shape):
    """pose_engine.py:69-85: per-box biased affine + warpAffine + normalize."""
    cs = box_to_center_scale(boxes_xyxy, input_shape)
    patches = []
    for cx, cy, w, h in cs:
        m = engine_affine(np.array([cx, cy]), np.array([w, h]), input_shape)
        patch = cv2s.warpAffine(img, m, (int(input_shape[0]), int(input_shape[1])),
                                flags=cv2s.INTER_LINEAR)
        patches.append(normalize_crop(patch))
    return np.stack(patches), cs
