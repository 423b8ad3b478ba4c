// test_shim.cpp — shim/proslam_hip_plugin.h compiled against the declaration stubs of tests/shim_stubs/ and driven through
// libvslam_hip.so: both plug-in classes are instantiated through the C ABI, a small harness makes the calls
// PoseTracker3D::compute makes on its plug-ins (initialize -> track -> aligner initialize + converge -> prune ->
// recoverPoints -> landmark bookkeeping -> compute; pose_tracker_3d.cpp:32-222) and every frame is compared with a second
// context that runs the fused device path (vslam_process_host): counters, poses, and the host objects the shim
// materialised (Frame::points(): keypoints, coordinates, previous links, epipolar offsets, descriptors).
//   test_shim            : CPU check — the shim builds, links, and reports "no HIP device" through std::runtime_error
//   test_shim <n_frames> : GPU run
#include <cstdio>
#include <cstdlib>

#include "shim_harness.hpp"
#include "../../tools/synth/synth_scene.h"

#define REQUIRE(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "FAILED frame %d: %s | ", k, #cond); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n_frames = argc > 1 ? std::atoi(argv[1]) : 0;
  const bool recovery = !(argc > 2 && std::atoi(argv[2]) == 0);
  synth_scene scene;
  synth_default_kitti(&scene);
  const double scale = 0.5;
  scene.rows = (int)std::lround(scene.rows * scale); scene.cols = (int)std::lround(scene.cols * scale);
  scene.fx *= scale; scene.fy *= scale; scene.cx *= scale; scene.cy *= scale; scene.seed = 41;
  CameraMatrix K; K(0, 0) = scene.fx; K(0, 2) = scene.cx; K(1, 1) = scene.fy; K(1, 2) = scene.cy; K(2, 2) = 1;
  Camera camera_left(scene.rows, scene.cols, K), camera_right(scene.rows, scene.cols, K);
  camera_right.setBaselineHomogeneous(Vector3(-scene.fx * scene.baseline_m, 0, 0));
  // configuration_kitti.yaml values (the structs default to parameters.h)
  StereoFramePointGeneratorParameters generator_parameters;
  generator_parameters.descriptor_type = (argc > 3 && std::atoi(argv[3]) == 1) ? "ORB-256" : "BRIEF";   // configuration_kitti.yaml:60 / _euroc.yaml:52
  AlignerParameters aligner_parameters; aligner_parameters.error_delta_for_convergence = 1e-3; aligner_parameters.maximum_error_kernel = 4; aligner_parameters.damping = 5;
  PoseTracker3DParameters tracker_parameters; tracker_parameters.aligner = &aligner_parameters;
  tracker_parameters.minimum_track_length_for_landmark_creation = 1; tracker_parameters.minimum_number_of_landmarks_to_track = 5;
  tracker_parameters.tunnel_vision_ratio = 0.5; tracker_parameters.good_tracking_ratio = 0.2; tracker_parameters.enable_landmark_recovery = recovery;
  LandmarkParameters landmark_parameters;

  int k = -1;
  HipContext hip;
  hip.tracker_parameters = &tracker_parameters; hip.landmark_parameters = &landmark_parameters;
  hip.config.max_keypoints = 8192; hip.config.max_points = 4096; hip.config.max_history_frames = 64;
  HipStereoFramePointGenerator generator(&generator_parameters, &hip);
  generator.setCameraLeft(&camera_left); generator.setCameraRight(&camera_right);
  generator.configure();
  HipStereoUVAligner aligner(&aligner_parameters, &hip);
  aligner.setMaximumReliableDepthMeters(generator_parameters.maximum_reliable_depth_meters);
  aligner.setMinimumReliableDepthMeters(generator_parameters.minimum_depth_meters);
  try {
    aligner.configure();    // vslam_create
  } catch (const std::runtime_error& e) {
    if (n_frames == 0 && std::string(e.what()).find("no HIP device") != std::string::npos) {
      std::printf("shim ok (CPU): both plug-in classes instantiated, the device context fails loudly without a GPU: %s\n", e.what());
      return 0;
    }
    std::fprintf(stderr, "configure failed: %s\n", e.what());
    return 2;
  }
  if (n_frames == 0) { std::printf("shim ok: device context created\n"); return 0; }

  try {
    vslam_ctx* fused = nullptr;
    hipCheck(nullptr, vslam_create(&hip.config, 0, 1, &fused), "fused context");
    Harness h{&generator, &aligner, &tracker_parameters, &generator_parameters, &camera_left, &camera_right};
    h.window = generator_parameters.maximum_projection_tracking_distance_pixels; h.tau = generator_parameters.minimum_descriptor_distance_tracking;
    std::vector<uint8_t> L((size_t)scene.rows * scene.cols), R(L.size());
    std::vector<int16_t> kp((size_t)hip.config.max_points * 4); std::vector<int32_t> meta((size_t)hip.config.max_points * 6);
    std::vector<double> cam((size_t)hip.config.max_points * 3); std::vector<uint8_t> desc((size_t)hip.config.max_points * 64);
    int max_tracked = 0, total_recovered = 0;
    double chrono_previous[3] = {0, 0, 0};
    for (k = 0; k < n_frames; ++k) {
      double Rm[9], t[3];
      synth_pose(&scene, k, Rm, t);
      for (int y = 0; y < scene.rows; ++y) for (int x = 0; x < scene.cols; ++x) {
        L[(size_t)y * scene.cols + x] = synth_pixel(&scene, Rm, t, k, 0, x, y); R[(size_t)y * scene.cols + x] = synth_pixel(&scene, Rm, t, k, 1, x, y); }
      Frame* frame = h.step(L.data(), R.data(), scene.rows, scene.cols);
      hipCheck(fused, vslam_process_host(fused, L.data(), R.data(), scene.cols, 0), "fused");
      vslam_frame_info ff, fs = generator.frameInfo();
      { const vslam_frame_info& fl = generator.lastFrameInfo();      // what compute() received with its stage view: the same report
        REQUIRE(std::memcmp(&fl, &fs, sizeof fs) == 0, "lastFrameInfo differs from vslam_get_frame_info"); }
      hipCheck(fused, vslam_get_frame_info(fused, 0, &ff), "fused info");
      REQUIRE((int)h.status == ff.status, "status %d vs %d", (int)h.status, ff.status);
      REQUIRE(fs.n_keypoints_left == ff.n_keypoints_left && (int)frame->keypointsLeft().size() == ff.n_keypoints_left && (int)frame->keypointsRight().size() == ff.n_keypoints_right, "keypoints");
      REQUIRE(fs.n_tracked == ff.n_tracked && fs.n_lost == ff.n_lost && (int)h.lost.size() == (k ? ff.n_lost : 0), "track %d/%d vs %d/%d", fs.n_tracked, fs.n_lost, ff.n_tracked, ff.n_lost);
      REQUIRE(fs.n_tracked_landmarks == ff.n_tracked_landmarks && (k == 0 || (int)generator.numberOfTrackedLandmarks() == ff.n_tracked_landmarks), "tracked landmarks host %u device %d fused %d", generator.numberOfTrackedLandmarks(), fs.n_tracked_landmarks, ff.n_tracked_landmarks);
      REQUIRE(fs.n_inliers == ff.n_inliers && (ff.aligner_ran == 0 || (int)aligner.numberOfInliers() == ff.n_inliers), "inliers %d vs %d", fs.n_inliers, ff.n_inliers);
      REQUIRE(fs.n_after_prune == ff.n_after_prune && fs.n_recovered == ff.n_recovered, "prune %d/%d vs %d/%d", fs.n_after_prune, fs.n_recovered, ff.n_after_prune, ff.n_recovered);
      REQUIRE(fs.n_active_landmarks == ff.n_active_landmarks && (int)h.active_landmarks == ff.n_active_landmarks, "active landmarks host %u device %d fused %d", h.active_landmarks, fs.n_active_landmarks, ff.n_active_landmarks);
      REQUIRE(fs.n_new_stereo == ff.n_new_stereo && fs.n_points == ff.n_points && (int)frame->points().size() == ff.n_points, "points host %zu device %d fused %d", frame->points().size(), fs.n_points, ff.n_points);
      REQUIRE(h.window == ff.window_pixels && h.tau == ff.tau_track, "tracker state %d/%g vs %d/%g", h.window, h.tau, ff.window_pixels, ff.tau_track);
      for (int i = 0; i < 12; ++i) REQUIRE(frame->cameraLeftToWorld().m(i / 4, i % 4) == ff.camera_left_to_world[i], "pose element %d: %.17g vs %.17g", i, frame->cameraLeftToWorld().m(i / 4, i % 4), ff.camera_left_to_world[i]);
      // the host objects against the fused context's frame
      int32_t n = 0;
      hipCheck(fused, vslam_get_frame_points(fused, 0, 0, hip.config.max_points, &n, kp.data(), meta.data(), cam.data(), nullptr, desc.data()), "fused points");
      REQUIRE(n == (int)frame->points().size(), "point count");
      Frame* previous = frame->previous();
      for (int i = 0; i < n; ++i) {
        FramePoint* q = frame->points()[i];
        REQUIRE(q->keypointLeft().pt.x == kp[4 * i] && q->keypointLeft().pt.y == kp[4 * i + 1] && q->keypointRight().pt.x == kp[4 * i + 2] && q->keypointRight().pt.y == kp[4 * i + 3], "keypoint of point %d", i);
        REQUIRE(q->descriptorDistanceTriangulation() == meta[6 * i] && q->epipolarOffset() == meta[6 * i + 1] && (int)q->trackLength() == meta[6 * i + 3], "meta of point %d: %g/%d/%u vs %d/%d/%d", i,
                q->descriptorDistanceTriangulation(), q->epipolarOffset(), q->trackLength(), meta[6 * i], meta[6 * i + 1], meta[6 * i + 3]);
        REQUIRE((q->previous() == nullptr) == (meta[6 * i + 2] < 0) && (meta[6 * i + 2] < 0 || q->previous() == previous->points()[meta[6 * i + 2]]), "previous link of point %d", i);
        REQUIRE((q->landmark() != nullptr) == (meta[6 * i + 4] > 0), "landmark of point %d: host %d device updates %d", i, q->landmark() != nullptr, meta[6 * i + 4]);
        for (int c = 0; c < 3; ++c) REQUIRE(q->cameraCoordinatesLeft()(c) == cam[3 * i + c], "coordinates of point %d", i);
        REQUIRE(std::memcmp(q->descriptorLeft().ptr<uint8_t>(0), &desc[(size_t)64 * i], 32) == 0 && std::memcmp(q->descriptorRight().ptr<uint8_t>(0), &desc[(size_t)64 * i + 32], 32) == 0, "descriptors of point %d", i);
      }
      max_tracked = std::max(max_tracked, ff.n_tracked); total_recovered += ff.n_recovered;
      // the chronometers SLAMAssembly::printReport reads from the generator (slam_assembly.cpp:709-719) accumulate frame by frame
      const double chrono[3] = {generator.getTimeConsumptionSeconds_keypoint_detection(), generator.getTimeConsumptionSeconds_descriptor_extraction(),
                                generator.getTimeConsumptionSeconds_point_triangulation()};
      for (int q = 0; q < 3; ++q) {
        REQUIRE(chrono[q] > chrono_previous[q] && chrono[q] < 10.0, "chronometer %d does not grow: %.9f after %.9f", q, chrono[q], chrono_previous[q]);
        chrono_previous[q] = chrono[q];
      }
    }
    REQUIRE((int)h.status == VSLAM_TRACKING && max_tracked > 50, "the tracker must lock on: status %d, tracked %d", (int)h.status, max_tracked);
    REQUIRE(!recovery || total_recovered > 0, "recovery never produced a point");
    bool thrown = false;
    try { generator.initialize(nullptr); } catch (const std::runtime_error&) { thrown = true; }
    REQUIRE(thrown, "initialize(nullptr) must throw");
    vslam_destroy(fused);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception at frame %d: %s\n", k, e.what());
    return 3;
  }
  std::printf("shim ok: %d frames, host objects and counters identical to the fused device path (recovery %s, descriptor %s)\n", n_frames,
              recovery ? "on" : "off", hip.config.descriptor_type == VSLAM_DESCRIPTOR_ORB ? "ORB" : "BRIEF");
  return 0;
}
