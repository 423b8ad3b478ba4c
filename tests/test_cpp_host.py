"""The C++ host mirror of the reference's plug-in interface (host/proslam_hip.hpp) against the oracle: compiled
with g++ against the C ABI only (no hipcc, no torch), run as its own process on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    subprocess.run(["make", "-C", CPP], check=True, capture_output=True)
    return os.path.join(CPP, "test_host_tracker")


def test_cpp_host_mirror_compiles_against_c_abi_only():
    exe = _build()
    assert os.access(exe, os.X_OK)
    with open(os.path.join(ROOT, "host", "proslam_hip.hpp")) as f:
        text = f.read()
    assert "torch" not in text and "hip_runtime" not in text   # plain C ABI consumer


@pytest.mark.gpu
def test_cpp_host_tracker_matches_oracle():
    exe = _build()
    out = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical to the oracle" in out.stdout
