"""Import shim: ``import udp_pose_amd`` resolves to the sources in ``udp-pose_amd/``.

The package directory carries the repository's name (``udp-pose_amd``), which
is not a valid Python identifier; this one-file package points ``__path__`` at
it so sub-modules import as ``udp_pose_amd.<module>``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "udp-pose_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
