// test_host_tracker.cpp — the C++ host mirror of the reference's plug-in interface (tests/cpp/proslam_hip_mirror.hpp) driving
// libvslam_hip.so call by call, checked frame by frame against the CPU oracle (test infrastructure) on a rendered
// synthetic sequence.  Reads like the reference's own harness (executables/test_stereo_frontend.cpp: initialize ->
// track -> compute per frame), with assertions instead of a display.  Exit code 0 = pass.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "proslam_hip_mirror.hpp"
#include "../../tools/synth/synth_scene.h"

extern "C" {
struct orc_ctx;
void orc_default_config_kitti(vslam_config*);
int orc_create(const vslam_config*, int, int, orc_ctx**);
void orc_destroy(orc_ctx*);
int orc_process_host(orc_ctx*, const uint8_t*, const uint8_t*, int32_t, size_t);
int orc_get_frame_info(orc_ctx*, int, vslam_frame_info*);
void orc_synth_default_kitti(synth_scene*);
void orc_synth_render(const synth_scene*, int, uint8_t*, uint8_t*, int32_t);
}

#define REQUIRE(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "FAILED frame %d: %s | ", k, #cond); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n_frames = argc > 1 ? std::atoi(argv[1]) : 10;
  synth_scene scene;
  orc_synth_default_kitti(&scene);
  const double scale = 0.5;
  scene.rows = (int)std::lround(scene.rows * scale); scene.cols = (int)std::lround(scene.cols * scale);
  scene.fx *= scale; scene.fy *= scale; scene.cx *= scale; scene.cy *= scale; scene.seed = 33;
  vslam_config cfg;
  orc_default_config_kitti(&cfg);
  cfg.rows = scene.rows; cfg.cols = scene.cols;
  const double K[9] = {scene.fx, 0, scene.cx, 0, scene.fy, scene.cy, 0, 0, 1};
  std::memcpy(cfg.K, K, sizeof K);
  cfg.baseline_h[0] = -scene.fx * scene.baseline_m; cfg.baseline_h[1] = 0; cfg.baseline_h[2] = 0;

  using namespace proslam_hip;
  int k = -1;
  try {
    HipContext hip(cfg, 0);
    StereoFramePointGenerator* generator = new StereoFramePointGenerator(&hip);
    generator->configure();
    StereoUVAligner* aligner = new StereoUVAligner(&hip);
    aligner->configure();
    PoseTracker3D tracker(generator, aligner);   // owns and deletes both plug-ins, as the reference
    tracker.configure();
    orc_ctx* oracle = nullptr;
    if (orc_create(&cfg, 0, 1, &oracle) != 0) { std::fprintf(stderr, "oracle create failed\n"); return 2; }
    std::vector<uint8_t> L((size_t)scene.rows * scene.cols), R(L.size());
    for (k = 0; k < n_frames; ++k) {
      orc_synth_render(&scene, k, L.data(), R.data(), scene.cols);
      tracker.setIntensityImageLeft(L.data(), scene.cols);
      tracker.setImageSecondary(R.data());
      tracker.compute();
      orc_process_host(oracle, L.data(), R.data(), scene.cols, 0);
      vslam_frame_info fo;
      orc_get_frame_info(oracle, 0, &fo);
      const vslam_frame_info& fg = tracker.currentFrame().info;
      REQUIRE(fg.status == fo.status, "%d vs %d", fg.status, fo.status);
      REQUIRE(fg.n_keypoints_left == fo.n_keypoints_left && fg.n_keypoints_right == fo.n_keypoints_right, "keypoints");
      REQUIRE(fg.n_tracked == fo.n_tracked && fg.n_lost == fo.n_lost, "%d/%d vs %d/%d", fg.n_tracked, fg.n_lost, fo.n_tracked, fo.n_lost);
      REQUIRE(fg.n_inliers == fo.n_inliers && fg.n_outliers == fo.n_outliers, "inliers %d vs %d", fg.n_inliers, fo.n_inliers);
      REQUIRE(fg.n_after_prune == fo.n_after_prune && fg.n_recovered == fo.n_recovered, "prune/recover");
      REQUIRE(fg.n_active_landmarks == fo.n_active_landmarks && fg.n_new_stereo == fo.n_new_stereo && fg.n_points == fo.n_points, "points %d vs %d", fg.n_points, fo.n_points);
      REQUIRE(fg.window_pixels == fo.window_pixels && fg.tau_track == fo.tau_track, "tracker state");
      double num = 0, den = 0;
      for (int i = 0; i < 12; ++i) { const double d = fg.camera_left_to_world[i] - fo.camera_left_to_world[i]; num += d * d; den += fo.camera_left_to_world[i] * fo.camera_left_to_world[i]; }
      REQUIRE(std::sqrt(num / den) <= 1e-4, "pose differs: %g", std::sqrt(num / den));
    }
    // error behaviour of the interface: null frames throw std::runtime_error like the reference
    bool thrown = false;
    try { generator->initialize(nullptr); } catch (const std::runtime_error&) { thrown = true; }
    REQUIRE(thrown, "initialize(nullptr) must throw");
    orc_destroy(oracle);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception at frame %d: %s\n", k, e.what());
    return 3;
  }
  std::printf("host tracker ok: %d frames identical to the oracle\n", n_frames);
  return 0;
}
