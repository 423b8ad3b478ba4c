"""Synthetic KITTI-shaped stereo sequences rendered on the GPU (csrc/libvslam_synth.so).

Data generator for benchmarks: the world of tools/synth/synth_scene.h is rendered straight into HBM
(torch uint8 tensors) so inputs are resident before the timed region.  Tests render small images on the
CPU through the oracle library instead."""
import ctypes as C
import os

import numpy as np

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libvslam_synth.so")


class SynthScene(C.Structure):
    """struct synth_scene (tools/synth/synth_scene.h)."""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("baseline_m", C.c_double),
                ("cam_height_m", C.c_double), ("wall_half_m", C.c_double), ("max_depth_m", C.c_double),
                ("cell_m", C.c_double), ("speed_m", C.c_double), ("sway_m", C.c_double),
                ("sway_rate", C.c_double), ("seed", C.c_uint64), ("bob_m", C.c_double), ("roll_amp", C.c_double),
                ("pitch_amp", C.c_double), ("roll_rate", C.c_double), ("pitch_rate", C.c_double), ("contrast", C.c_double), ("noise_seed", C.c_uint64)]


class Synth(object):
    def __init__(self):
        if not os.path.exists(LIB):
            raise RuntimeError("libvslam_synth.so is not built: run __graft_entry__.build()")
        self.lib = C.CDLL(LIB)
        self.lib.synth_render_device.restype = C.c_int

    def scene_kitti(self, seed=7):
        s = SynthScene()
        self.lib.synth_scene_default_kitti(C.byref(s))
        s.seed = seed
        return s

    def scene_euroc(self, seed=7):
        s = SynthScene()
        self.lib.synth_scene_default_euroc(C.byref(s))
        s.seed = seed
        return s

    def render_device(self, scene, frame0, n_frames, left_ptr, right_ptr, row_stride, frame_stride, stream_ptr=0):
        rc = self.lib.synth_render_device(C.byref(scene), C.c_int(frame0), C.c_int(n_frames), C.c_void_p(left_ptr),
                                          C.c_void_p(right_ptr), C.c_int(row_stride), C.c_size_t(frame_stride),
                                          C.c_void_p(stream_ptr))
        if rc != 0:
            raise RuntimeError("synth_render_device failed: %d" % rc)

    def gt_pose(self, scene, frame):
        out = np.zeros(12, np.float64)
        self.lib.synth_pose_host(C.byref(scene), C.c_int(frame), out.ctypes.data_as(C.c_void_p))
        return out.reshape(3, 4)


def config_for_scene(api, scene, which="kitti"):
    cfg = api.default_config(which)
    cfg.rows, cfg.cols = scene.rows, scene.cols
    K = [scene.fx, 0, scene.cx, 0, scene.fy, scene.cy, 0, 0, 1]
    for i in range(9):
        cfg.K[i] = K[i]
    cfg.baseline_h[0] = -scene.fx * scene.baseline_m
    cfg.baseline_h[1] = 0.0
    cfg.baseline_h[2] = 0.0
    return cfg
