"""Drop-in engine: the ``UdpPsaPoseTorch`` contract of deep_hrnet/pose_engine.py:99-127.

    engine = UdpPsaPoseHip(model_path, config_path, device)
    keypoints, maxvals = engine.infer_pose(img, boxes)   # [N,J,2] image px, [N,J,1]

Same stages as the reference (``_preprocess`` :69-85, model forward :125,
``_postprocess`` :87-92) but device-resident: the frame is uploaded once, the
per-box biased affine crop + ToTensor + Normalize, the HRNet forward and the UDP
decode all run in libudp_pose_hip.so, and only ``[N,J,3]`` numbers come back.
``flip_test=True`` adds validate()'s second mirrored forward + fuse
(lib/core/function.py:151-171), which the reference engine itself does not do.
"""
import numpy as np
import torch

from . import _lib
from .config import load_config
from .inference import decode_device
from .model import MODELS
from .transforms import COCO_FLIP_PAIRS, MPII_FLIP_PAIRS, flip_fuse

SKELETONS = {   # pose_engine.py:17-26
    "coco": [[16, 14], [14, 12], [17, 15], [15, 13], [12, 13], [6, 12], [7, 13], [6, 7], [6, 8], [7, 9],
             [8, 10], [9, 11], [2, 3], [1, 2], [1, 3], [2, 4], [3, 5], [4, 6], [5, 7]],
    "mpii": [[9, 10], [12, 13], [12, 11], [3, 2], [2, 1], [14, 15], [15, 16], [4, 5], [5, 6], [9, 8], [8, 7],
             [7, 3], [7, 4], [9, 13], [9, 14]],
}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def box_to_center_scale(boxes, input_shape, pixel_std=200):
    """pose_engine.py:45-63 (xyxy2xywh + aspect fix + /200*1.25), float32 like the torch original.
    Returns a new [N,4] float32 array (cx, cy, w, h); the caller's boxes are not modified."""
    if isinstance(boxes, torch.Tensor):
        boxes = boxes.detach().cpu().numpy()
    b = np.array(boxes, dtype=np.float32).reshape(-1, 4)
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


def engine_affine_dst2src(center, scale, patch_size):
    """The dst->src 2x3 matrix cv2.warpAffine ends up using for the engine crop:
    tools/infer_utils/utils.py:157-177 builds three float32 point pairs (rot=0, biased: dst uses
    W,H), cv2.getAffineTransform solves src->dst in fp64, warpAffine inverts it."""
    st = np.asarray(scale, dtype=np.float64) * 200
    dst_w, dst_h = patch_size[0], patch_size[1]
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = np.asarray(center) + np.array([0, st[0] * -0.5])
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], np.float32)
    d = src[0] - src[1]
    src[2] = src[1] + np.array([-d[1], d[0]], np.float32)
    d = dst[0] - dst[1]
    dst[2] = dst[1] + np.array([-d[1], d[0]], np.float32)
    a = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    m = np.linalg.solve(a, dst.astype(np.float64)).T            # src -> dst
    det = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    det = 1.0 / det if det != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * det, m[0, 0] * det, -m[0, 1] * det, -m[1, 0] * det
    return np.array([[a11, a12, -a11 * m[0, 2] - a12 * m[1, 2]],
                     [a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]]], np.float64)


def warp_affine_device(frame_u8, mats, out_hw, out=None):
    """cv2.warpAffine(INTER_LINEAR) + ToTensor + Normalize for N crops of one frame, on device.
    frame_u8: cuda uint8 [H,W,3]; mats: [N,2,3] fp64 dst->src; out: fp32 [N,3,oh,ow]."""
    import ctypes as C
    if not frame_u8.is_cuda or frame_u8.dtype != torch.uint8 or frame_u8.dim() != 3 or frame_u8.shape[2] != 3:
        raise ValueError("frame must be a cuda uint8 [H,W,3] tensor")
    frame_u8 = frame_u8.contiguous()
    mats_t = torch.as_tensor(np.ascontiguousarray(mats, dtype=np.float64)).reshape(-1, 6).to(frame_u8.device)
    n = mats_t.shape[0]
    oh, ow = out_hw
    if out is None:
        out = torch.empty(n, 3, oh, ow, dtype=torch.float32, device=frame_u8.device)
    mean = (C.c_float * 3)(*IMAGENET_MEAN)
    std = (C.c_float * 3)(*IMAGENET_STD)
    _lib.check(_lib.lib().udp_warp_affine(_lib.ptr(frame_u8), frame_u8.shape[0], frame_u8.shape[1],
                                          frame_u8.shape[1] * 3, _lib.ptr(mats_t), n, oh, ow, mean, std,
                                          _lib.ptr(out), _lib.stream_ptr()))
    return out


class UdpPsaPoseHip:
    """MI355X engine with the UdpPsaPoseTorch interface (pose_engine.py:99-127)."""

    SKELETONS = SKELETONS

    def __init__(self, model_path, config_path, device="cuda", dtype=None, state_dict=None, config=None):
        self.config = config if config is not None else load_config(config_path)
        self.input_shape = list(self.config.MODEL.IMAGE_SIZE)          # [w, h]
        ds = str(self.config.DATASET.DATASET).lower()
        self.skeleton = SKELETONS.get(ds)
        self.flip_pairs = MPII_FLIP_PAIRS if ds == "mpii" else COCO_FLIP_PAIRS
        self.config.TEST.MODEL_FILE = model_path
        self._device = torch.device(device)
        if dtype is None:
            dtype = "f16x2"           # the parity-grade fast mode (fp32-level results on the fp16 matrix pipe)
        self.model = MODELS[self.config.MODEL.NAME](self.config, is_train=False, dtype=dtype)
        if state_dict is None:
            state_dict = torch.load(model_path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(state_dict, strict=False)
        self.model.to(self._device)
        self.model.eval()
        self._fallback, self.fp32_retries = None, 0     # fp32 twin for batches that leave fp16's range (infer_pose)

    def _box_to_center_scale(self, boxes, pixel_std=200):
        return box_to_center_scale(boxes, self.input_shape, pixel_std)

    @staticmethod
    def _bucket(n):
        """Batch size the network is run at for n boxes: 1, 2, 4, 8, then multiples of 8.  A frame's box count
        changes almost every call; per exact n the model would keep one set of I/O buffers and one hipGraph
        (instantiated from ~170 launch descriptions) -- bucketing bounds both.  The padding rows repeat the last
        box and are dropped from the result (an image's heat-maps do not depend on its batch, bit for bit:
        tests/test_gpu_e2e.py::test_each_image_is_independent_of_its_batch)."""
        if n <= 8:
            b = 1
            while b < n:
                b *= 2
            return b
        return (n + 7) // 8 * 8

    def _preprocess(self, img, boxes):
        cs = self._box_to_center_scale(boxes)
        if cs.shape[0] == 0:
            raise RuntimeError("infer_pose needs at least one box (the reference's torch.stack([]) raises)")
        n = cs.shape[0]
        nb = self._bucket(n)
        cs_run = np.concatenate([cs, np.repeat(cs[-1:], nb - n, axis=0)]) if nb > n else cs
        mats = np.stack([engine_affine_dst2src(c[:2], c[2:], self.input_shape) for c in cs_run])
        frame = torch.as_tensor(np.ascontiguousarray(img, dtype=np.uint8)).to(self._device, non_blocking=True)
        w, h = int(self.input_shape[0]), int(self.input_shape[1])
        xin, _ = self.model.io_buffers(nb, h, w, False)
        return warp_affine_device(frame, mats, (h, w), out=xin), cs

    def _heatmaps(self, model, pose_input, n, flip_test, offset):
        nb = pose_input.shape[0]
        if flip_test:
            xin, _ = model.io_buffers(nb, pose_input.shape[2], pose_input.shape[3], True)
            xin.copy_(pose_input)
            raw = model.raw_forward(xin, flip_test=True)
            return flip_fuse(raw[:n], raw[nb:nb + n], self.flip_pairs, offset)
        return model.raw_forward(pose_input)[:n]

    def _f32_model(self):
        """The same weights in the reference's precision (built on first need: an f16x2 forward left fp16's range)."""
        if self._fallback is None:
            m = MODELS[self.config.MODEL.NAME](self.config, is_train=False, dtype="f32")
            m.load_state_dict(self.model.state_dict(), strict=False)
            m.to(self._device)
            m.eval()
            self._fallback = m
        return self._fallback

    @torch.no_grad()
    def infer_pose(self, img, boxes, flip_test=False):
        pose_input, cs = self._preprocess(img, boxes)
        n = cs.shape[0]
        offset = self.config.MODEL.TARGET_TYPE == "offset"
        hm = self._heatmaps(self.model, pose_input, n, flip_test, offset)
        if self.model.dtype == "f16x2" and _lib.f16x2_overflow(reset=True):
            # split-fp16 storage has fp16's range (|x| < 65520).  The kernels flag a value that leaves it where it is
            # produced (udp_f16x2_overflow: the NaN it turns into downstream would not survive the next ReLU, so a
            # finiteness test of the heat-maps can miss it) and the batch is re-run in fp32 -- the precision and
            # range of the reference engine -- instead of returning garbage.
            self.fp32_retries += 1
            fb = self._f32_model()
            xin, _ = fb.io_buffers(pose_input.shape[0], pose_input.shape[2], pose_input.shape[3], False)
            xin.copy_(pose_input)
            hm = self._heatmaps(fb, xin, n, flip_test, offset)
        center = torch.from_numpy(cs[:, :2].astype(np.float64))
        scale = torch.from_numpy(cs[:, 2:].astype(np.float64))
        post = bool(self.config.TEST.POST_PROCESS) and not offset
        preds, maxvals, _, _ = decode_device(hm, center, scale, self.config.MODEL.TARGET_TYPE, post,
                                             float(self.config.LOSS.KPD), cs_is_f32=True, want_idx=False)
        kp, mv = preds.cpu().numpy(), maxvals.cpu().numpy()
        if not np.isfinite(mv).all():
            raise FloatingPointError("non-finite heat-maps in fp32 too: the weights or the frame hold non-finite values")
        return (kp if post else kp.astype(np.float32)), mv

    def draw_keypoints(self, image, keypoints, radius=1):
        """pose_engine.py:65-67: marks keypoints (filled squares) and skeleton end points in place."""
        h, w = image.shape[:2]
        for person in np.asarray(keypoints):
            for x, y in person[:, :2]:
                if np.isfinite(x) and np.isfinite(y):
                    xi, yi = int(round(float(x))), int(round(float(y)))
                    image[max(0, yi - radius):min(h, yi + radius + 1), max(0, xi - radius):min(w, xi + radius + 1)] = (0, 255, 0)
        return image


UdpPsaPoseTorch = UdpPsaPoseHip          # name the reference's callers import (inference_engine.py:236)
