"""udp-pose_amd: MI355X-native UDP-Pose hot path (HRNet forward + UDP decode).

Host side (Python) mirrors the reference's call contracts -- ``get_final_preds``
(deep_hrnet/lib/core/inference.py:149), ``MODELS[name](cfg, is_train)``
(deep_hrnet/lib/models/__init__.py:28-41), ``UdpPsaPoseTorch.infer_pose``
(deep_hrnet/pose_engine.py:99-127) -- over the C ABI of
``include/udp_pose_hip.h`` (hand-written HIP kernels for gfx950).  There is no
CPU fallback: importing a compute entry point without the built library raises.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: E402
from .config import load_config  # noqa: E402,F401


def __getattr__(name):
    """Lazy exports so that ``import udp_pose_amd`` works without torch being imported yet."""
    if name in ("get_final_preds", "decode_device", "gaussian_taps"):
        from . import inference
        return getattr(inference, name)
    if name in ("MODELS", "get_pose_net", "PoseHighResolutionNetHip"):
        from . import model
        return getattr(model, name)
    if name in ("UdpPsaPoseHip", "UdpPsaPoseTorch", "box_to_center_scale", "warp_affine_device"):
        from . import pose_engine
        return getattr(pose_engine, name)
    if name in ("flip_back", "flip_back_offset", "flip_fuse"):
        from . import transforms
        return getattr(transforms, name)
    raise AttributeError(name)
