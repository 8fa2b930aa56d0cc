"""CPU oracle for the UDP-Pose hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy / stock torch.nn.functional, fp32+fp64)
of the reference algorithms named in SURVEY.md section 8(a).  It exists only to
*check* the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under
``udp-pose_amd/`` imports it, and the product path raises if the HIP library is
missing instead of falling back to this code.

Pinning: every function here is compared against the reference's own Python
code, imported in the build container from /root/reference by
``oracle/gen_golden.py`` (the script that wrote ``tests/golden/*.npz``).  The
OpenCV calls the reference makes (``cv2.GaussianBlur``, ``cv2.warpAffine``,
``cv2.getAffineTransform``) are third-party code that is absent from
/root/reference and not installed here (opencv-python==4.5.3.56,
deep_hrnet/requirements.txt:4): ``oracle/cv2_standin.py`` restates their
documented rules, and results that depend on them are **parity unpinned**
against real OpenCV (pinned only against this stand-in).
"""
