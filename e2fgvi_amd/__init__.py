"""e2fgvi_amd: MI355X-native (gfx950) E2FGVI inference forward.

Python host code + hand-written HIP kernels behind a C ABI (include/e2fgvi_hip.h).
"""
__version__ = "0.1.0"
