// synth_render.hip — libvslam_synth.so: renders the seeded synthetic stereo world (tools/synth/synth_scene.h)
// straight into HBM so that benchmark inputs are resident before the timed region.  Data generator only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../tools/synth/synth_scene.h"

__global__ __launch_bounds__(256) void k_synth(const synth_scene sc, int frame0, uint8_t* left, uint8_t* right, int stride,
                                               size_t frame_stride) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int k = blockIdx.z;
  if (x >= sc.cols || y >= sc.rows) return;
  double R[9], t[3];
  synth_pose(&sc, frame0 + k, R, t);
  left[(size_t)k * frame_stride + (size_t)y * stride + x] = synth_pixel(&sc, R, t, frame0 + k, 0, x, y);
  right[(size_t)k * frame_stride + (size_t)y * stride + x] = synth_pixel(&sc, R, t, frame0 + k, 1, x, y);
}

extern "C" __attribute__((visibility("default"))) int synth_render_device(const synth_scene* sc, int frame0, int n_frames,
                                                                          uint8_t* left, uint8_t* right, int stride,
                                                                          size_t frame_stride, void* stream) {
  if (!sc || !left || !right || n_frames < 1 || stride < sc->cols) return -1;
  dim3 grid((sc->cols + 63) / 64, (sc->rows + 3) / 4, n_frames);
  hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, (hipStream_t)stream, *sc, frame0, left, right, stride, frame_stride);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) void synth_scene_default_kitti(synth_scene* s) { synth_default_kitti(s); }
extern "C" __attribute__((visibility("default"))) void synth_scene_default_euroc(synth_scene* s) { synth_default_euroc(s); }
extern "C" __attribute__((visibility("default"))) void synth_pose_host(const synth_scene* s, int k, double* c2w) {
  double R[9], t[3];
  synth_pose(s, k, R, t);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c2w[4 * i + j] = R[3 * i + j]; c2w[4 * i + 3] = t[i]; }
}
