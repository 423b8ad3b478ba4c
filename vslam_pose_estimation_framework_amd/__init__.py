"""MI355X-native stereo visual-odometry front end (ProSLAM hot path) — host-side Python glue.

The product is ``csrc/libvslam_hip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/vslam_hip.h``); this package only binds it (ctypes), generates synthetic data and
evaluates trajectories.  There is no CPU fallback: ``hip.load()`` raises when the library or a
GPU is missing.
"""
from .capi import CApi, Config, FrameInfo, VslamError  # noqa: F401
