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


_torch_checked = False


def _initialise_torch_first():
    """A process that uses both PyTorch-ROCm and this library holds TWO ROCm runtimes: torch's wheel bundles its own libamdhip64 /
    libhsa-runtime64, libvslam_hip.so links the system's (/opt/rocm).  Measured on this image (tests/validation/torch_after_hip.py):
    when the system runtime creates a context first, torch's later fails with "No HIP GPUs are available"; the other way round
    both work.  So when torch is installed its GPU context is initialised before the library's first HIP call."""
    global _torch_checked
    if _torch_checked:
        return
    _torch_checked = True
    try:
        import torch
    except ImportError:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:      # no usable GPU for torch: the library reports its own error at vslam_create
        pass


def load():
    """Bind libvslam_hip.so (prefix vslam_).  Creating a context additionally needs an MI355X."""
    _initialise_torch_first()
    return CApi(lib_path(), "vslam_")
