"""xivo_amd - MI355X-native EKF measurement-update path for XIVO.

The product is the C-ABI shared library ``libxivo_hip.so`` (include/xivo_hip.h,
hand-written HIP for gfx950). This package only holds the ctypes binding used by
tests/ and bench.py, and the build recipe. There is no CPU fallback: importing
``xivo_amd.lib`` without the built library raises.
"""
from .lib import Context, XivoHipError, load_library, lib_path  # noqa: F401
