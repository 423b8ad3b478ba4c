"""Identity of the build a measurement belongs to.

The GPU box receives the working tree without `.git/`, so a profile cannot record a commit; it records `source_sha16`, a hash of
every source file the native library is built from.  bench.py reports a counter file (profiles/*pmc_traffic*.json) only when the
hash recorded in it equals the hash of the sources the running library was built from."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    csrc = os.path.join(ROOT, "vslam_pose_estimation_framework_amd", "csrc")
    files = glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")) + [os.path.join(csrc, "Makefile")]
    files += glob.glob(os.path.join(ROOT, "include", "*.h"))
    return sorted(files)


def source_sha16():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_sha16():
    lib = os.path.join(ROOT, "vslam_pose_estimation_framework_amd", "csrc", "libvslam_hip.so")
    with open(lib, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def git_head():
    """Commit of the working tree when `.git` is there (the build container), else None (the GPU box)."""
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip() or None
    except Exception:
        return None


if __name__ == "__main__":
    import json
    print(json.dumps({"source_sha16": source_sha16(), "library_sha16": library_sha16(), "git_head": git_head()}))
