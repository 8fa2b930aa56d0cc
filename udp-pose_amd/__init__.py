"""udp-pose_amd: MI355X-native UDP-Pose hot path (HRNet forward + UDP decode).

Host side (Python) mirrors the reference's call contracts -- ``get_final_preds``
(deep_hrnet/lib/core/inference.py:149), ``MODELS[name](cfg, is_train)``
(deep_hrnet/lib/models/__init__.py:28-41), ``UdpPsaPoseTorch.infer_pose``
(deep_hrnet/pose_engine.py:99-127) -- over the C ABI of
``include/udp_pose_hip.h`` (hand-written HIP kernels for gfx950).  There is no
CPU fallback: importing a compute entry point without the built library raises.
"""
__version__ = "0.1.0"
