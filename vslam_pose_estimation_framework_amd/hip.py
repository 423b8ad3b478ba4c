"""Loader of the product library (csrc/libvslam_hip.so).  Fails loudly: no CPU fallback exists."""
import os

from .capi import CApi

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.environ.get("VSLAM_HIP_LIB") or os.path.join(CSRC, "libvslam_hip.so")   # override: build variants of the same HIP library


def lib_path():
    if not os.path.exists(LIB):
        raise RuntimeError("libvslam_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C %s`; the HIP path has no CPU fallback." % (LIB, CSRC))
    return LIB


def load():
    """Bind libvslam_hip.so (prefix vslam_).  Creating a context additionally needs an MI355X."""
    return CApi(lib_path(), "vslam_")
