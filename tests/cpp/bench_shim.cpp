// bench_shim.cpp — frames/s of the REAL drop-in path: the shim's plug-in classes (shim/proslam_hip_plugin.h) driven as
// PoseTracker3D::compute drives them (slam_assembly.cpp:61-76 wiring, the process() loop timed at :428-447), host objects
// materialised (Frame::keypoints / descriptors / points(), FramePoint links) — against the fused one-stream device path
// (vslam_process_host) on the same host images.  Built against the declaration stubs (tests/shim_stubs/): the reference's own
// headers need OpenCV / Eigen / srrg, absent here.
//   bench_shim <images.bin> <rows> <cols> <row_stride> <n_frames> <bin_size> [warmup] [pinned]
// pinned = 1: each frame's two images are first placed in pinned buffers from vslam_host_alloc (outside the timed region — what a
// loader that decodes into such buffers gives the tracker), so the upload is a direct asynchronous copy.
// images.bin: n_frames x (left image, right image), each rows x row_stride bytes.  Prints one JSON line.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "shim_harness.hpp"
#include "../../tools/synth/synth_scene.h"

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: bench_shim <images.bin> <rows> <cols> <row_stride> <n_frames> <bin_size> [warmup]\n"); return 2; }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), stride = std::atoi(argv[4]), n_frames = std::atoi(argv[5]), bin = std::atoi(argv[6]);
  const int warmup = argc > 7 ? std::atoi(argv[7]) : 20;
  const bool pinned = argc > 8 && std::atoi(argv[8]) != 0;
  const size_t img = (size_t)rows * stride;
  std::vector<uint8_t> data((size_t)n_frames * 2 * img);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(data.data(), 1, data.size(), f) != data.size()) { std::fprintf(stderr, "cannot read %zu bytes from %s\n", data.size(), argv[1]); return 2; }
  std::fclose(f);
  synth_scene scene;
  synth_default_kitti(&scene);                       // intrinsics / baseline of the images (bench.py renders them from the same scene)
  CameraMatrix K; K(0, 0) = scene.fx; K(0, 2) = scene.cx; K(1, 1) = scene.fy; K(1, 2) = scene.cy; K(2, 2) = 1;
  Camera camera_left(rows, cols, K), camera_right(rows, cols, K);
  camera_right.setBaselineHomogeneous(Vector3(-scene.fx * scene.baseline_m, 0, 0));
  StereoFramePointGeneratorParameters generator_parameters;    // configuration_kitti.yaml values
  generator_parameters.descriptor_type = "BRIEF";
  generator_parameters.bin_size_pixels = bin;
  AlignerParameters aligner_parameters; aligner_parameters.error_delta_for_convergence = 1e-3; aligner_parameters.maximum_error_kernel = 4; aligner_parameters.damping = 5;
  PoseTracker3DParameters tracker_parameters; tracker_parameters.aligner = &aligner_parameters;
  tracker_parameters.minimum_track_length_for_landmark_creation = 1; tracker_parameters.minimum_number_of_landmarks_to_track = 5;
  tracker_parameters.tunnel_vision_ratio = 0.5; tracker_parameters.good_tracking_ratio = 0.2; tracker_parameters.enable_landmark_recovery = true;
  LandmarkParameters landmark_parameters;
  try {
    HipContext hip;
    hip.tracker_parameters = &tracker_parameters; hip.landmark_parameters = &landmark_parameters;
    hip.config.max_keypoints = 8192; hip.config.max_points = 4096; hip.config.max_history_frames = 64;
    HipStereoFramePointGenerator generator(&generator_parameters, &hip);
    generator.setCameraLeft(&camera_left); generator.setCameraRight(&camera_right);
    generator.configure();
    HipStereoUVAligner aligner(&aligner_parameters, &hip);
    aligner.setMaximumReliableDepthMeters(generator_parameters.maximum_reliable_depth_meters);
    aligner.setMinimumReliableDepthMeters(generator_parameters.minimum_depth_meters);
    aligner.configure();
    Harness h{&generator, &aligner, &tracker_parameters, &generator_parameters, &camera_left, &camera_right};
    h.window = generator_parameters.maximum_projection_tracking_distance_pixels; h.tau = generator_parameters.minimum_descriptor_distance_tracking;
    std::vector<double> frame_s;
    double total = 0, points = 0, tracked = 0, keypoints = 0;
    uint8_t* pin[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [frame parity][side]
    if (pinned) for (int q = 0; q < 4; ++q) hipCheck(nullptr, vslam_host_alloc((void**)&pin[q / 2][q % 2], img), "pinned image buffer");
    for (int k = 0; k < n_frames; ++k) {
      if (k == warmup) for (double& s : h.seconds) s = 0;
      uint8_t* L = &data[(size_t)(2 * k) * img]; uint8_t* R = &data[(size_t)(2 * k + 1) * img];
      if (pinned) { std::memcpy(pin[k & 1][0], L, img); std::memcpy(pin[k & 1][1], R, img); L = pin[k & 1][0]; R = pin[k & 1][1]; }
      const auto t0 = std::chrono::steady_clock::now();
      Frame* frame = h.step(L, R, rows, cols, (size_t)stride);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (k >= warmup) { frame_s.push_back(dt); total += dt; points += frame->points().size(); tracked += h.tracked_points; keypoints += frame->keypointsLeft().size(); }
    }
    const vslam_frame_info shim_info = generator.frameInfo();
    for (int q = 0; q < 4; ++q) vslam_host_free(pin[q / 2][q % 2]);
    // the fused one-stream path on the same host images (what exact_mode.single_sequence times with device-resident images)
    vslam_ctx* fused = nullptr;
    hipCheck(nullptr, vslam_create(&hip.config, 0, 1, &fused), "fused context");
    double fused_total = 0;
    for (int k = 0; k < n_frames; ++k) {
      const auto t0 = std::chrono::steady_clock::now();
      hipCheck(fused, vslam_process_host(fused, &data[(size_t)(2 * k) * img], &data[(size_t)(2 * k + 1) * img], stride, 0), "fused");
      if (k == n_frames - 1 || k == warmup - 1) hipCheck(fused, vslam_synchronize(fused), "fused");
      if (k >= warmup) fused_total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    vslam_frame_info fused_info;
    hipCheck(fused, vslam_get_frame_info(fused, 0, &fused_info), "fused info");
    vslam_destroy(fused);
    bool same = shim_info.n_points == fused_info.n_points && shim_info.n_tracked == fused_info.n_tracked && shim_info.n_inliers == fused_info.n_inliers;
    for (int i = 0; i < 12; ++i) same = same && shim_info.camera_left_to_world[i] == fused_info.camera_left_to_world[i];
    const int n = (int)frame_s.size();
    std::sort(frame_s.begin(), frame_s.end());
    static const char* names[Harness::T_STAGES] = {"initialize", "track", "aligner_initialize_converge", "host_prune", "recoverPoints", "host_landmark_bookkeeping", "compute"};
    std::printf("{\"frames\": %d, \"warmup\": %d, \"ms_per_frame\": %.4f, \"frames_per_s\": %.1f, \"median_ms\": %.4f, \"min_ms\": %.4f, \"max_ms\": %.4f, \"stage_ms\": {", n, warmup,
                total / n * 1e3, n / total, frame_s[n / 2] * 1e3, frame_s[0] * 1e3, frame_s[n - 1] * 1e3);
    for (int q = 0; q < Harness::T_STAGES; ++q) std::printf("%s\"%s\": %.4f", q ? ", " : "", names[q], h.seconds[q] / n * 1e3);
#ifdef PROSLAM_HIP_PROFILE
    std::printf("}, \"host_breakdown_ms\": {");      // accumulated over ALL frames (warm-up included)
    for (int q = 0; q < HipProfile::N; ++q) std::printf("%s\"%s\": %.4f", q ? ", " : "", HipProfile::name(q), HipProfile::acc()[q] / n_frames * 1e3);
#endif
    std::printf("}, \"mean_keypoints_left\": %.1f, \"mean_points\": %.1f, \"mean_tracked\": %.1f, \"fused_host_images_ms_per_frame\": %.4f, \"shim_over_fused\": %.3f, "
                "\"last_frame_identical_to_fused\": %s, \"host_images\": \"%s\"}\n", keypoints / n, points / n, tracked / n, fused_total / n * 1e3, (total / n) / (fused_total / n), same ? "true" : "false", pinned ? "pinned (vslam_host_alloc)" : "pageable");
    return same ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "bench_shim: %s\n", e.what());
    return 3;
  }
}
