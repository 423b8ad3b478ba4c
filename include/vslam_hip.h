/* vslam_hip.h — C ABI of the MI355X-native ProSLAM front end (libvslam_hip.so).
 *
 * Drop-in boundary for the per-frame hot path of Ssellu/vslam-pose-estimation-framework
 * (a ProSLAM fork).  The reference has no FFI: its plug points are C++ virtuals wired in
 * SLAMAssembly::_createStereoTracker (src/system/slam_assembly.cpp:61-76).  Each entry
 * point below names the reference interface it replaces (paths relative to the reference
 * repository root).  The header-only C++ shim that subclasses the reference classes and
 * forwards to these calls is shim/proslam_hip_plugin.h; INTEGRATION.md shows the two lines
 * a maintainer changes.
 *
 * Conventions
 *   - plain C, no exceptions, no callbacks; every call returns 0 (VSLAM_OK) or a negative
 *     vslam_status; vslam_last_error() gives the text (reference: std::runtime_error thrown
 *     at stereo_framepoint_generator.cpp:30-33,75-78,139-142,468-471, caught in app.cpp:128).
 *   - a context owns `n_streams` independent sequences ("streams": whole KITTI sequences
 *     or chunks of one).  Every per-frame call processes ONE stereo pair per stream, all
 *     streams in the same kernels (grid dimension = stream).  n_streams = 1 is the literal
 *     drop-in for the reference's single tracker.
 *   - images: 8-bit grayscale, row-major, `stride` bytes per row
 *     (reference: Frame::intensityImageLeft/Right, CV_8UC1, src/types/frame.h:114-119).
 *   - transforms: double[12], row-major 3x4 [R|t] (reference: TransformMatrix3D,
 *     src/types/definitions.h:62).
 *   - descriptors: 32 bytes (256 bit), Hamming norm (definitions.h:45-49).
 *   - a context is single-caller (reference: one caller thread, app.cpp:96).
 */
#ifndef VSLAM_HIP_H
#define VSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSLAM_MAX_REGIONS 16      /* detector grid cells (det_rows*det_cols)          */
#define VSLAM_DESC_BYTES 32       /* SRRG_PROSLAM_DESCRIPTOR_SIZE_BITS=256            */
#define VSLAM_MAX_EPI 8           /* maximum_epipolar_search_offset_pixels upper bound */

typedef enum {
  VSLAM_OK = 0,
  VSLAM_ERR_INVALID = -1,   /* bad argument / null frame (reference: runtime_error)          */
  VSLAM_ERR_NO_DEVICE = -2, /* HIP device or kernel image missing: the product path never   */
                            /* falls back to a CPU implementation                           */
  VSLAM_ERR_HIP = -3,       /* a HIP runtime call failed                                    */
  VSLAM_ERR_CAPACITY = -4,  /* an output array of the caller, or the scratch buffers of a stand-alone entry, are too
                               small.  Overflow of a context's device-resident buffers inside vslam_process_* / the
                               stage calls is NOT a return code: processing continues on the truncated lists and the
                               frame's vslam_frame_info.error_flags reports it per stream (bit 0 keypoints, bit 1 points,
                               bit 2 history) */
  VSLAM_ERR_STATE = -5      /* call sequence violated (e.g. track before frame_begin)       */
} vslam_status;

typedef enum { VSLAM_LOCALIZING = 0, VSLAM_TRACKING = 1 } vslam_tracker_status; /* Frame::Status */

/* All parameters the hot path reads.  Names follow the reference's YAML keys
 * (configurations/configuration_kitti.yaml:49-134, src/types/parameters.h:64-330). */
typedef struct vslam_config {
  /* camera (src/types/camera.h): left == right intrinsics for a rectified pair */
  int32_t rows, cols;
  double K[9];          /* cameraMatrix(), row-major                                        */
  double baseline_h[3]; /* Camera::baselineHomogeneous() of the right camera: (-fx*B, 0, 0) */

  /* base_framepoint_generation */
  int32_t det_rows, det_cols;            /* number_of_detectors_vertical / _horizontal      */
  int32_t detector_threshold_minimum;    /* FAST thresholds (parameters.h:176-177)           */
  int32_t detector_threshold_maximum;
  double detector_threshold_maximum_change;
  double target_number_of_keypoints_tolerance;
  int32_t bin_size_pixels;
  int32_t enable_keypoint_binning;
  int32_t minimum_projection_tracking_distance_pixels;
  int32_t maximum_projection_tracking_distance_pixels;
  double minimum_descriptor_distance_tracking;
  double maximum_descriptor_distance_tracking;
  double maximum_reliable_depth_meters;
  double maximum_depth_meters;
  double minimum_depth_meters;

  /* stereo_framepoint_generation */
  double maximum_matching_distance_triangulation;
  double minimum_disparity_pixels;
  int32_t maximum_epipolar_search_offset_pixels;

  /* tracking (PoseTracker3DParameters) */
  int32_t minimum_track_length_for_landmark_creation;
  int32_t minimum_number_of_landmarks_to_track;
  double tunnel_vision_ratio;
  double good_tracking_ratio;
  int32_t enable_landmark_recovery;
  double minimum_delta_angular_for_movement;
  double minimum_delta_translational_for_movement;

  /* tracking.aligner (AlignerParameters) */
  double aligner_error_delta_for_convergence;
  double aligner_maximum_error_kernel;
  double aligner_damping;
  int32_t aligner_maximum_number_of_iterations;
  int32_t aligner_minimum_number_of_inliers;

  /* landmark (LandmarkParameters, parameters.h:97-112) */
  double landmark_maximum_error_squared_meters;
  int32_t landmark_maximum_number_of_iterations;

  /* capacities of the device-resident buffers (no reference counterpart: std::vector grows) */
  int32_t max_keypoints;       /* per image, 64..65535                                        */
  int32_t max_points;          /* framepoints per frame                                       */
  int32_t max_history_frames;  /* frames of per-point history kept for landmark refinement     */

  /* descriptor extractor (base_framepoint_generator.cpp:184-224): 0 = BRIEF-32 (descriptor_type "BRIEF",
   * configuration_kitti.yaml:60; 28 px border), 1 = ORB (cv::ORB::create() used as extractor on the detector's FAST
   * keypoints: descriptor_type "ORB" and every unknown string, e.g. configuration_euroc.yaml:52 "ORB-256"; rBRIEF at
   * level 0 of the 7x7 Gaussian-blurred image, steered by KeyPoint::angle = -1 as FAST leaves it, 31 px border) */
  int32_t descriptor_type;
} vslam_config;
#define VSLAM_DESCRIPTOR_BRIEF 0
#define VSLAM_DESCRIPTOR_ORB 1

/* Per-frame, per-stream report: the scalars PoseTracker3D / SLAMAssembly read back from the
 * plug-ins (pose_tracker_3d.cpp:111,132,242-248,361; slam_assembly.cpp:644-742). */
typedef struct vslam_frame_info {
  int32_t frame_index;         /* frames processed by this stream so far (this one included)  */
  int32_t status;              /* tracker status AFTER the frame (vslam_tracker_status)       */
  int32_t status_at_start;     /* status the frame was created with                           */
  int32_t n_keypoints_left, n_keypoints_right;   /* after the descriptor border filter        */
  int32_t n_detected_left, n_detected_right;     /* raw FAST detections (controller input)    */
  int32_t thresholds[VSLAM_MAX_REGIONS];         /* detector thresholds AFTER adjust          */
  int32_t track_attempts;      /* 1 + recursive re-registrations (pose_tracker_3d.cpp:300)    */
  int32_t n_tracked;           /* points linked by the final track() call                     */
  int32_t n_lost;              /* lost list of the final track() call                         */
  int32_t n_tracked_landmarks; /* numberOfTrackedLandmarks()                                  */
  int32_t aligner_ran;         /* StereoUVAligner::converge() was called on the final points  */
  int32_t aligner_iterations;  /* oneRound() calls of that converge()                         */
  int32_t aligner_converged;   /* hasSystemConverged()                                        */
  int32_t n_inliers, n_outliers;
  double total_error;          /* totalError()                                                */
  int32_t n_after_prune;       /* points surviving _prunePoints                               */
  int32_t n_recovered;         /* points added by recoverPoints                               */
  int32_t n_active_landmarks;  /* _number_of_active_landmarks                                 */
  int32_t n_new_stereo;        /* points appended by compute()                                */
  int32_t n_points;            /* frame->points().size() at the end of the frame              */
  int32_t track_broken;        /* breakTrack() happened                                       */
  int32_t fallback;            /* _fallbackEstimate() was used                                */
  int32_t window_pixels;       /* _projection_tracking_distance_pixels after the frame        */
  int32_t error_flags;         /* bit0 keypoint capacity, bit1 point capacity, bit2 history   */
  double tau_track;            /* _current_descriptor_distance_tracking after the frame       */
  double tau_triangulation;    /* _current_maximum_descriptor_distance_triangulation          */
  double camera_left_to_world[12];  /* frame pose (robotToWorld with identity robot offset)   */
  double previous_to_current[12];   /* motion prior kept for the next frame                   */
} vslam_frame_info;

typedef struct vslam_ctx vslam_ctx;

/* ---- lifetime --------------------------------------------------------------------------
 * Replaces: new StereoFramePointGenerator(params) + setCameraLeft/Right + configure(),
 *           new StereoUVAligner(params) + setMaximum/MinimumReliableDepthMeters + configure(),
 *           PoseTracker3D::configure()   (slam_assembly.cpp:61-76, pose_tracker_3d.cpp:11-21).
 * `device` is the HIP device ordinal.  Fails with VSLAM_ERR_NO_DEVICE when no GPU is usable. */
int vslam_create(const vslam_config* cfg, int device, int n_streams, vslam_ctx** out);
void vslam_destroy(vslam_ctx* ctx);
const char* vslam_last_error(const vslam_ctx* ctx); /* ctx may be NULL: last create() error */
void vslam_default_config_kitti(vslam_config* cfg); /* configuration_kitti.yaml values, KITTI-00 calib */
void vslam_default_config_euroc(vslam_config* cfg); /* configuration_euroc.yaml values                */
/* Restart every stream as a fresh sequence (PoseTracker3D::configure, WorldMap::clear). */
int vslam_reset(vslam_ctx* ctx);
/* Exact mode with whole sequences of different lengths per stream: a stream whose sequence has ended is
 * switched off — every kernel skips it, its last frame's state, report and pose log stay readable — while the other
 * streams of the context carry on; vslam_reset_stream restarts ONE stream as a fresh sequence (a new PoseTracker3D /
 * generator / aligner / WorldMap: pose_tracker_3d.cpp:11-21) so that a stream can work through a queue of sequences.
 * vslam_set_stream_active synchronises the context; vslam_reset_stream is asynchronous (queued between two frames on the
 * context's HIP streams).  Neither may be called between vslam_frame_begin and the end of that frame.
 * The images passed for an inactive stream are ignored (the pointer arithmetic still reserves its slot). */
int vslam_set_stream_active(vslam_ctx* ctx, int stream, int active);
int vslam_reset_stream(vslam_ctx* ctx, int stream);
int vslam_reset_streams(vslam_ctx* ctx, int32_t n, const int32_t* streams);   /* several streams, one launch per state half */
/* Use the caller's HIP stream (hipStream_t passed as void*) for all work of this context. */
int vslam_set_hip_stream(vslam_ctx* ctx, void* hip_stream);

/* ---- whole-frame entry point -------------------------------------------------------------
 * Replaces PoseTracker3D::compute() (pose_tracker_3d.cpp:32-222) for every stream of the
 * context: initialize -> track -> StereoUVAligner -> prune -> recoverPoints -> landmark
 * update -> compute, with the tracker's control logic evaluated on the device.  Asynchronous
 * on the context stream; no host synchronisation.
 *   left/right: n_streams images each; image s starts at base + s*image_stride_bytes.
 *   *_device variant: pointers are device memory (inputs already resident in HBM).
 *   host variant: pointers are host memory, copied to the device first (hipMemcpyAsync on the image stream: one
 *     copy per side when the images are back to back, else one per image; pinned memory makes it truly asynchronous).
 *     The buffers must stay valid and unchanged until the copy has run (vslam_synchronize, or any read-back call). */
int vslam_process_device(vslam_ctx* ctx, const uint8_t* left, const uint8_t* right,
                         int32_t row_stride_bytes, size_t image_stride_bytes);
int vslam_process_host(vslam_ctx* ctx, const uint8_t* left, const uint8_t* right,
                       int32_t row_stride_bytes, size_t image_stride_bytes);
/* Block until all queued work of the context is done; returns the sticky HIP error state (VSLAM_ERR_HIP once a runtime
 * call has failed, else VSLAM_OK; capacity overflows are reported in vslam_frame_info.error_flags, see VSLAM_ERR_CAPACITY). */
int vslam_synchronize(vslam_ctx* ctx);

/* ---- stage entry points (the reference's plug-in virtuals, one call each) ------------------
 * These drive the same device code as vslam_process_* but leave the control flow to the
 * caller (the shim's PoseTracker3D keeps the reference's own logic). All act on every stream. */

/* StereoFramePointGenerator::initialize(frame, extract_features=true)
 * (stereo_framepoint_generator.cpp:73-133): FAST per detector region + threshold controller
 * (base_framepoint_generator.cpp:355-459), BRIEF-32, triangulation-distance rule, feature
 * stores.  Frame::status() of the new frame is the tracker status held in the context. */
int vslam_frame_begin(vslam_ctx* ctx, const uint8_t* left, const uint8_t* right,
                      int32_t row_stride_bytes, size_t image_stride_bytes, int on_device);
/* Second half of the two-call form of vslam_process_*: everything PoseTracker3D::compute does after
 * initialize() (track .. compute), evaluated on the device.  vslam_frame_begin + vslam_frame_finish ==
 * vslam_process_*. */
int vslam_frame_finish(vslam_ctx* ctx);
/* initialize(frame, extract_features=false): rebuild both feature stores (…:128-132). */
int vslam_frame_restore(vslam_ctx* ctx);
/* StereoFramePointGenerator::track (…:464-681) using the tracker state held in the context
 * (prior, window, descriptor distance; setters below). */
int vslam_track(vslam_ctx* ctx, int by_appearance);
/* StereoUVAligner::initialize + converge (stereouv_aligner.cpp:10-69,210-264). */
int vslam_align(vslam_ctx* ctx, int enable_inverse_depth_as_information);
/* PoseTracker3D::_prunePoints + recoverPoints (pose_tracker_3d.cpp:437-472,
 * stereo_framepoint_generator.cpp:683-869). */
int vslam_prune_recover(vslam_ctx* ctx);
/* PoseTracker3D::_updatePoints incl. Landmark create/update (pose_tracker_3d.cpp:475-520,
 * landmark.cpp:8-33,66-167). */
int vslam_update_points(vslam_ctx* ctx);
/* StereoFramePointGenerator::compute (…:135-462). */
int vslam_stereo_new(vslam_ctx* ctx);
int vslam_compute(vslam_ctx* ctx);          /* vslam_update_points + vslam_stereo_new in ONE launch (the shim's compute())  */
/* Tracker-owned state the reference pokes through setters
 * (setProjectionTrackingDistancePixels, setMaximumDescriptorDistanceTracking,
 * base_framepoint_generator.h:166-167; Frame::setRobotToWorld; Frame::setStatus). */
int vslam_set_tracker_state(vslam_ctx* ctx, int stream, int status, const double prior[12],
                            int window_pixels, double tau_track);
int vslam_set_pose(vslam_ctx* ctx, int stream, const double camera_left_to_world[12]);

/* ---- readback (synchronises the context stream) -------------------------------------------- */
int vslam_get_frame_info(vslam_ctx* ctx, int stream, vslam_frame_info* out);
/* Frame::keypointsLeft/Right + descriptorsLeft/Right (frame.h:64-67).  xy = (x,y) int16 pairs
 * (FAST keypoints are integer pixels), score = KeyPoint::response, desc = n*32 bytes.
 * Order: image row-major.  Any output pointer may be NULL.  cap = capacity in keypoints. */
int vslam_get_keypoints(vslam_ctx* ctx, int stream, int side, int32_t cap, int32_t* n,
                        int16_t* xy, int32_t* score, uint8_t* desc);
/* Frame::points() of the current frame (frame.h:88-89) as SoA.
 *   kp   : n*4 int16 (xL,yL,xR,yR)              FramePoint::keypointLeft/Right().pt
 *   meta : n*6 int32 (hamming_LR, epipolar_offset, previous_index, track_length,
 *                      landmark_updates, disparity)
 *   cam  : n*3 double  cameraCoordinatesLeft()
 *   lm   : n*3 double  landmark()->coordinates() (world), valid where landmark_updates>0
 *   chi  : n double    StereoUVAligner::errors() of the point (-1 if none), inl: inliers() */
int vslam_get_points(vslam_ctx* ctx, int stream, int32_t cap, int32_t* n, int16_t* kp,
                     int32_t* meta, double* cam, double* lm);
int vslam_get_aligner_result(vslam_ctx* ctx, int stream, int32_t cap, int32_t* n, double* chi,
                             uint8_t* inlier, double T[12], double H[36]);
/* Stage-granular read-backs for a host that keeps the reference's object model (shim/proslam_hip_plugin.h).
 * vslam_get_track_result: what the last vslam_track left — out4 = (index in frame_previous->points(), left feature,
 *   right feature, Hamming L-R) per tracked point in the order of the previous points (feature = index in the row-major
 *   keypoint list of vslam_get_keypoints), lost = indices of the previous points on the lost list
 *   (stereo_framepoint_generator.cpp:659-665).  cap bounds both lists.
 * vslam_get_frame_points: as vslam_get_points, plus the 64 descriptor bytes (left | right) of every point; in_progress = 1
 *   reads the frame the stage calls are assembling (after vslam_prune_recover: survivors of _prunePoints followed by the
 *   recovered points), 0 the finished frame. */
int vslam_get_track_result(vslam_ctx* ctx, int stream, int32_t cap, int32_t* n_tracked, int32_t* out4, int32_t* n_lost,
                           int32_t* lost);
int vslam_get_frame_points(vslam_ctx* ctx, int stream, int in_progress, int32_t cap, int32_t* n, int16_t* kp, int32_t* meta,
                           double* cam, double* lm, uint8_t* desc);
/* Pinned (page-locked) host memory.  Host images handed to vslam_process_host / vslam_frame_begin from ordinary (pageable) memory
 * are staged through a pinned buffer of the context first (one memcpy per image, ~30 us per 467 KB); images that already live in
 * memory from vslam_host_alloc (e.g. the cv::Mat a loader decodes into, constructed on such a buffer) are copied to the device
 * directly and asynchronously. */
int vslam_host_alloc(void** out, size_t bytes);
void vslam_host_free(void* p);

/* ---- stage views: zero-copy read-back for a host that keeps the reference's object model --------------------------------------
 * The reference's tracker reads its plug-ins' results after every virtual call (pose_tracker_3d.cpp:32-222: keypoints after
 * initialize(), the tracked list after track(), errors()/inliers() after converge(), Frame::points() after recoverPoints() and
 * compute()).  A vslam_view_* call packs exactly the live elements of that stage's results into a pinned host buffer owned by the
 * context (one small kernel, the GPU writes host memory directly) and synchronises the stream's frame queue ONCE; the pointers it
 * returns stay valid until the next vslam_view_* call on the context (one report buffer per context); the keypoint arrays (coordinates,
 * scores, descriptors) have a region of their own that only the next frame's keypoint report rewrites, so they may still be copied after
 * vslam_track has been LAUNCHED (the shim copies descriptor rows while the device tracks).  Same data, same order and same meaning as the
 * vslam_get_* calls above, which copy array by array. */
typedef struct vslam_keypoints_view {
  int32_t n[2];                /* left, right                                           */
  const int16_t* xy[2];        /* n * (x, y), image row-major                           */
  const uint8_t* score[2];     /* n FAST scores (KeyPoint::response)                    */
  const uint8_t* desc[2];      /* n * 32 descriptor bytes                               */
} vslam_keypoints_view;
typedef struct vslam_track_view {
  int32_t n_tracked, n_lost, n_tracked_landmarks;
  const int32_t* tracked4;     /* as vslam_get_track_result out4                        */
  const int32_t* lost;
} vslam_track_view;
typedef struct vslam_aligner_view {
  int32_t n, n_inliers, n_outliers, iterations, converged;
  double total_error;
  const double* chi;           /* StereoUVAligner::errors()                             */
  const uint8_t* inlier;       /* StereoUVAligner::inliers()                            */
  double T[12], H[36];
} vslam_aligner_view;
typedef struct vslam_points_view {
  int32_t n;
  int32_t first_full;          /* meta / cam / desc are filled for points first_full .. n-1: 0 for the finished frame; for the frame in
                                  assembly (in_progress = 1) the number of survivors of the prune — those are objects the caller already
                                  holds, only the recovered points behind them are new                                                  */
  const int16_t* kp;           /* n * (xL, yL, xR, yR), every point                     */
  const int32_t* meta;         /* n * 6, the layout of vslam_get_points                 */
  const double* cam;           /* n * 3                                                 */
  const uint8_t* desc;         /* n * 64 (left | right); in_progress = 1 only, else NULL (the finished frame's new points carry the
                                  descriptors of the caller's own features)             */
  vslam_frame_info info;       /* the stream's report as of this stage                  */
  /* in-kernel chronometers of the stream, accumulated seconds (the tracker-side five of vslam_get_timers) */
  double seconds_tracking, seconds_pose_optimization, seconds_point_recovery, seconds_landmark_optimization, seconds_point_triangulation;
} vslam_points_view;
int vslam_view_keypoints(vslam_ctx* ctx, int stream, vslam_keypoints_view* out);            /* after vslam_frame_begin    */
/* The same with desc[] = NULL, as soon as the detector has written coordinates and scores (the descriptors are still being computed): a
 * caller builds its cv::KeyPoint lists meanwhile and fetches the descriptors with vslam_view_keypoints afterwards.  One-stream contexts
 * inside a frame; anywhere else it is vslam_view_keypoints. */
int vslam_view_keypoints_xy(vslam_ctx* ctx, int stream, vslam_keypoints_view* out);
int vslam_view_track(vslam_ctx* ctx, int stream, vslam_track_view* out);                    /* after vslam_track          */
int vslam_view_aligner(vslam_ctx* ctx, int stream, vslam_aligner_view* out);                /* after vslam_align          */
int vslam_view_points(vslam_ctx* ctx, int stream, int in_progress, vslam_points_view* out); /* after vslam_prune_recover (1) / vslam_stereo_new (0) */
/* StereoUVAligner::_weights_translation as the stream's last initialize() left it (stereouv_aligner.cpp:22,57-61): the
 * vector is a MEMBER of the aligner, `resize(n, 1)` keeps the elements it already holds, and they are rewritten only while
 * enable_inverse_depth_as_information is set — so Localizing frames (flag off, pose_tracker_3d.cpp:124) reuse the weights the
 * last Tracking frame left at the same indices.  n = the vector's size. */
int vslam_get_aligner_weights(vslam_ctx* ctx, int stream, int32_t cap, int32_t* n, double* weight);
/* The 8 chronometers SLAMAssembly::printReport prints (slam_assembly.cpp:703-742), seconds of
 * device time accumulated per stage over all streams (HIP events): keypoint_detection,
 * descriptor_extraction, point_triangulation, tracking, track_creation, pose_optimization,
 * landmark_optimization, point_recovery. */
int vslam_get_timers(vslam_ctx* ctx, double seconds[8]);
int vslam_enable_timers(vslam_ctx* ctx, int on);
/* Per-kernel device time (HIP events on the context stream, recorded while timers are enabled):
 * accumulated milliseconds and launch counts of k_fast_box, k_emit, k_brief, k_track_candidates, k_frame (its
 * three phase launches together, counted once), k_recover_brief, k_update_landmarks, k_stereo_dist.
 * Used by bench.py for the roofline of the dominant kernel.  Synchronises; vslam_enable_timers(ctx,1)
 * clears the accumulators.  Stage path of a one-stream context (vslam_frame_begin): the image pipeline is timed by three events
 * instead of two per kernel, so k_fast_box's figure covers k_emit as well (ms[0] + ms[1] = keypoint detection either way) and
 * k_stereo_dist is not timed. */
int vslam_get_kernel_times(vslam_ctx* ctx, double ms[8], int32_t launches[8]);

/* ---- stand-alone kernels (unit parity, and the reference's optional knnMatch block) -------- */
/* cv::FastFeatureDetector::detect on one ROI (base_framepoint_generator.cpp:12-25,367):
 * FAST-9/16, non-max suppression, output row-major, coordinates relative to the ROI. */
int vslam_fast_detect(vslam_ctx* ctx, const uint8_t* image_host, int32_t rows, int32_t cols,
                      int32_t stride, int32_t roi_x, int32_t roi_y, int32_t roi_w, int32_t roi_h,
                      int32_t threshold, int32_t cap, int32_t* n, int16_t* xy, int32_t* score);
/* _descriptor_extractor->compute (base_framepoint_generator.cpp:431-438): BRIEF-32 at given
 * integer keypoints; keypoints closer than 28 px to the border are removed (keep[i]=0). */
int vslam_brief_describe(vslam_ctx* ctx, const uint8_t* image_host, int32_t rows, int32_t cols,
                         int32_t stride, int32_t n, const int16_t* xy, uint8_t* keep, uint8_t* desc);
/* matcher->knnMatch(query, train, k=2) of the use_matches block (stereo_framepoint_generator.cpp:168-206); the matcher type
 * selects the norm (:175-197).  norm: 0 = NORM_HAMMING on the bits (BRUTEFORCE_HAMMING), 1 = NORM_L2 on the bytes converted
 * to float (convertTo(CV_32F) + BRUTEFORCE), 2 = NORM_L1 (BRUTEFORCE_L1), 3 = squared L2 (BRUTEFORCE_SL2).
 * idx: nq*2 int32 (-1 if fewer than 2 train rows), dist: nq*2 float. Ties -> lowest index. */
int vslam_knn2(vslam_ctx* ctx, int norm, int32_t nq, const uint8_t* query, int32_t nt,
               const uint8_t* train, int32_t* idx, float* dist);
/* StereoUVAligner on caller-provided correspondences (stereouv_aligner.cpp:72-264):
 * moving n*3, fixed n*4, omega n, weight n. */
int vslam_align_points(vslam_ctx* ctx, int32_t n, const double* moving, const double* fixed,
                       const double* omega, const double* weight, const double T_init[12],
                       double T_out[12], double* chi, uint8_t* inlier, int32_t* n_inliers,
                       double* total_error, int32_t* iterations, double H_out[36]);

/* The translation weights over a sequence of StereoUVAligner::initialize calls on ONE aligner object
 * (stereouv_aligner.cpp:22,57-61), for known-answer tests: call k has n[k] measurements with depths depth[off_k ..]
 * (off_k = n[0] + .. + n[k-1]) and the flag inverse_depth[k]; out receives the n[k] weights in effect after call k.
 * Runs the weight rule of the fused tracker's aligner (maximum_reliable_depth_meters of ctx's config). */
int vslam_aligner_weights(vslam_ctx* ctx, int32_t n_calls, const int32_t* n, const int32_t* inverse_depth,
                          const double* depth, double* out);

/* UVDAligner (RGB-D mode, uvd_aligner.cpp:72-232) on caller-provided correspondences: moving n*3 (previous camera
 * coordinates), fixed n*3 (u, v, depth), omega_uv n and omega_depth n (the diagonal of the information matrix:
 * uvd_aligner.cpp:28-61 sets (1 + landmark updates) and 10x that, or 0 for unreliable depth), weight n (translation
 * weight).  Same solver, damping, convergence rule and outputs as vslam_align_points; inlier-only rounds need more
 * than 100 inliers (uvd_aligner.cpp:211).  SURVEY.md 8f row 4, first half. */
int vslam_align_points_uvd(vslam_ctx* ctx, int32_t n, const double* moving, const double* fixed_uvd,
                           const double* omega_uv, const double* omega_depth, const double* weight,
                           const double T_init[12], double T_out[12], double* chi, uint8_t* inlier,
                           int32_t* n_inliers, double* total_error, int32_t* iterations, double H_out[36]);
/* StereoFramePointGenerator::track (stereo_framepoint_generator.cpp:464-681, with
 * IntensityFeatureMatcher::getMatchingFeatureInRectangularRegion, intensity_feature_matcher.cpp:81-148) on
 * caller-provided data, for known-answer tests: nP previous points (left-camera coordinates n*3, left / right
 * descriptors n*32, epipolar offsets), motion prior T, window d, the feature sets of the current images as
 * (row, col) pairs + descriptors (one feature per pixel).  Geometry (rows, cols, K, baseline) from ctx's config.
 * out4 = (previous index, left feature, right feature, L-R distance) per tracked point in the order of the
 * previous points, cap nP; lost = indices of the previous points that go to the lost list, cap nP. */
int vslam_track_match(vslam_ctx* ctx, const double T[12], int32_t d, double tau_track, double tau_tri,
                      int32_t by_appearance, int32_t nP, const double* cam, const uint8_t* prev_desc_left,
                      const uint8_t* prev_desc_right, const int32_t* epipolar_offset, int32_t nL,
                      const int32_t* rc_left, const uint8_t* desc_left, int32_t nR, const int32_t* rc_right,
                      const uint8_t* desc_right, int32_t* n_tracked, int32_t* out4, int32_t* n_lost, int32_t* lost);

/* StereoFramePointGenerator::compute (stereo_framepoint_generator.cpp:135-462: epipolar sweep over the configured
 * offsets, ordering constraint, minimum disparity, bin competition if enabled in ctx's config) on caller-provided
 * feature sets ((row, col) pairs + descriptors, one feature per pixel), for known-answer tests.
 * out4 = (left feature, right feature, L-R distance, epipolar offset) per new framepoint in emission order. */
int vslam_stereo_match(vslam_ctx* ctx, double tau_tri, int32_t nL, const int32_t* rc_left, const uint8_t* desc_left,
                       int32_t nR, const int32_t* rc_right, const uint8_t* desc_right, int32_t cap, int32_t* n_out,
                       int32_t* out4);

/* StereoFramePointGenerator::recoverPoints (stereo_framepoint_generator.cpp:683-869) on caller-provided data, for known-answer
 * tests: n lost points (landmark flag, landmark world coordinates n*3, last left / right descriptors n*32), the frame's
 * world_to_camera_left and both images (ctx's image size, geometry, depth and disparity limits, descriptor type).  Per lost point
 * in list order: projection into both cameras (:704-745), depth and 5 * keypoint.size border gates (:746-764), descriptors at the
 * rounded projections, the three descriptor gates and the minimum disparity (:773-842).  Outputs (cap n) in list order:
 * rec_index = position in the lost list, rec_xy4 = (xL, yL, xR, yR), rec_dist = left-right distance, rec_desc = left | right
 * descriptor (64 B), rec_xyz = triangulated left-camera coordinates (:871-895). */
int vslam_stereo_recover(vslam_ctx* ctx, const uint8_t* image_left, const uint8_t* image_right, int32_t row_stride,
                         const double world_to_camera[12], int32_t n, const uint8_t* has_landmark, const double* landmark_world,
                         const uint8_t* prev_desc_left, const uint8_t* prev_desc_right, double tau_track, double tau_tri,
                         int32_t* n_recovered, int32_t* rec_index, int32_t* rec_xy4, int32_t* rec_dist, uint8_t* rec_desc,
                         double* rec_xyz);

/* Landmark::update (types/landmark.cpp:66-167) for n landmarks, stand-alone (the fused tracker runs the same refinement inside
 * its frame kernel): landmark i owns the measurements offsets[i] .. offsets[i+1]-1 in the caller's order, the LAST one being
 * the new observation; measurement m was taken in frame frame_of[m] (pose tables world_to_camera / camera_to_world, 3x4 each)
 * at left-camera coordinates cam[m] with information 1 / cam[m].z (landmark.h:22-36).  Gauss-Newton on the world position
 * with the saturated kernel (landmark_maximum_error_squared_meters, landmark_maximum_number_of_iterations of ctx's config),
 * 3x3 full-pivot LU; on convergence the estimate is taken if it has more inliers than the landmark had updates, reset to the
 * mean of the measurements if inliers < outliers, kept otherwise.  world (n*3) and updates (n) are read and written. */
int vslam_landmark_update(vslam_ctx* ctx, int32_t n, const int32_t* offsets, const int32_t* frame_of, int32_t n_frames,
                          const double* world_to_camera, const double* camera_to_world, const double* cam, double* world,
                          int32_t* updates);

/* ---- RGB-D components: the pieces of DepthFramePointGenerator, stand-alone -----------------
 * Not wired into the fused stereo tracker; same role as vslam_align_points_uvd (the RGB-D aligner): the kernels a
 * depth-mode shim calls, each checked against the oracle and an independent fixture. */
typedef struct vslam_depth_params {
  int32_t rows, cols;
  double K_left[9];             /* _camera_left->cameraMatrix(), row-major                                        */
  double K_left_inverse[9];     /* its inverse as the caller's linear algebra gives it (Eigen .inverse() upstream) */
  double K_right_inverse[9];    /* inverse of the depth camera's matrix (depth_framepoint_generator.cpp:443)      */
  double right_to_left[12];     /* _camera_left->robotToCamera()*_camera_right->cameraToRobot() (:446), 3x4       */
  double depth_scale_factor_intensity_to_meters;       /* parameters.h:251 (1e-3)                               */
  double minimum_depth_meters, maximum_depth_meters;    /* parameters.h:197-198                                   */
  int32_t enable_point_triangulation;                   /* parameters.h:256                                       */
  int32_t enable_keypoint_binning, bin_size_pixels;     /* base generator parameters                              */
  int32_t descriptor_type;                              /* extractor of recoverPoints: VSLAM_DESCRIPTOR_BRIEF / _ORB (the RGB-D
                                                           configurations say "ORB-256", i.e. cv::ORB::create())   */
  int32_t detector_type;                                /* vslam_rgbd_*: VSLAM_DETECTOR_FAST (every shipped configuration) or
                                                           VSLAM_DETECTOR_ORB = OrbDetector, cv::ORB::create(5000, 1.2, 8, 31, 0, 2,
                                                           HARRIS_SCORE, 31, threshold) per detector region
                                                           (base_framepoint_generator.cpp:52-70, :242-247; parameters.cpp:341)  */
} vslam_depth_params;
#define VSLAM_DETECTOR_FAST 0
#define VSLAM_DETECTOR_ORB 1

/* DepthFramePointGenerator::_computeDepthMap (depth_framepoint_generator.cpp:410-485) without the optional bilateral
 * filter: every non-zero u16 depth pixel is back-projected through K_right_inverse, moved into the left camera,
 * projected with K_left and z-buffered (strict "stored float > new double", scan order) into the rows x cols x 3 float
 * space map, initialised to (0, 0, maximum_depth); row_map / col_map receive the winning source pixel (-1: none).
 * depth: host image, row stride in ELEMENTS.  Outputs may be NULL; the map also stays resident in the context for
 * vslam_depth_compute. */
int vslam_depth_space_map(vslam_ctx* ctx, const vslam_depth_params* p, const uint16_t* depth, int32_t row_stride,
                          float* space_map, int16_t* row_map, int16_t* col_map);

/* DepthFramePointGenerator::compute (:45-164) on caller-provided features: rc_features = nF (row, col) pairs in the
 * feature vector's order (row-major sorted), rc_tracked = nT (row, col) of the framepoints the frame already holds
 * (tracked / recovered: they own their bin).  space_map: host map (rows*cols*3 floats) or NULL = the one the last
 * vslam_depth_space_map call left in the context.  New points (measured depth) come back in emission order — bin grid
 * row-major when binning is on (lower depth wins an untracked bin, first wins ties), feature order otherwise — as
 * feature index + left-camera coordinates; temporary points (depth >= maximum and triangulation enabled, :84-96) in
 * feature order with the coordinates K_left_inverse * (col*max, row*max, max).  cap bounds both lists. */
int vslam_depth_compute(vslam_ctx* ctx, const vslam_depth_params* p, const float* space_map, int32_t nF,
                        const int32_t* rc_features, int32_t nT, const int32_t* rc_tracked, int32_t cap, int32_t* n_new,
                        int32_t* new_feature, double* new_xyz, int32_t* n_temporary, int32_t* temporary_feature,
                        double* temporary_xyz);

/* DepthFramePointGenerator::track (:166-287) on caller-provided data: the previous frame's points followed by its
 * temporary points (:181-184) as left-camera coordinates cam (nP*3), left descriptors (nP*32) and flags (bit 0: has a
 * landmark, bit 1: hasUnreliableDepth); the motion prior T (previous -> current camera), the search window d, the
 * descriptor threshold tau (minimum_descriptor_distance_tracking in the reference) and the search mode; the current left
 * features as (row, col) pairs + descriptors (one feature per pixel); the space map as in vslam_depth_compute.
 * Outputs, all in the order of the previous points (the greedy, order-dependent outcome of the serial loop, reproduced
 * exactly): out2 = (previous index, left feature) of the tracked points with their measured coordinates xyz; temp2 = the
 * same pairs for matches on pixels without depth (temporary points, triangulation enabled); lost = previous points that go
 * to the lost list (:281-284).  Every list has room for nP entries. */
int vslam_depth_track(vslam_ctx* ctx, const vslam_depth_params* p, const float* space_map, const double T[12], int32_t d,
                      double tau, int32_t by_appearance, int32_t nP, const double* cam, const uint8_t* previous_desc,
                      const uint8_t* previous_flags, int32_t nL, const int32_t* rc_left, const uint8_t* desc_left,
                      int32_t* n_tracked, int32_t* out2, double* xyz, int32_t* n_temporary, int32_t* temp2, int32_t* n_lost,
                      int32_t* lost, int32_t* n_tracked_landmarks);

/* DepthFramePointGenerator::recoverPoints (:289-407) on caller-provided data: the lost points' landmarks in world
 * coordinates (n*3; has_landmark[i] = 0 skips the point as :305 does), their last left descriptors (n*32), the frame's
 * world_to_camera_left (3x4), the LEFT intensity image (host, u8) and the keypoint size (7 for FAST; the search border is
 * 5*size + 1 px).  A landmark is recovered when it projects into the image, its pixel (rint of the projection) has a
 * measured depth in [minimum, maximum), the projection keeps the border and BRIEF at the projection — the ROI origin is
 * the rounded corner, cv::Rect_<float> -> cv::Rect — is within tau of the previous descriptor.  Outputs in the order of
 * the lost list, room for n entries each: rec_index (position in the lost list), rec_xy (the new keypoint: the sub-pixel
 * projection as the reference's float arithmetic leaves it), rec_desc, rec_xyz (the space-map entry).  A projection whose
 * rounded pixel falls outside the map (x == cols or y == rows: an out-of-bounds read upstream) is skipped. */
int vslam_depth_recover(vslam_ctx* ctx, const vslam_depth_params* p, const float* space_map, const uint8_t* image_left,
                        int32_t row_stride, const double world_to_camera_left[12], int32_t n, const uint8_t* has_landmark,
                        const double* landmark_world, const uint8_t* previous_desc, float keypoint_size, double tau,
                        int32_t* n_recovered, int32_t* rec_index, float* rec_xy, uint8_t* rec_desc, double* rec_xyz);

/* BaseFramePointGenerator::getPointInCamera (base_framepoint_generator.cpp:461-494) for n point pairs: midpoint
 * triangulation of a previous / current image point pair under the motion T (previous -> current camera); the 3x2
 * least-squares problem is solved through its singular value decomposition (minimum-norm for a rank-deficient pair).
 * xy_*: n*2 floats; out: n*3 doubles.  Floating point: agrees with the reference's JacobiSVD to rounding (tests: 1e-9). */
int vslam_point_in_camera(vslam_ctx* ctx, int32_t n, const float* xy_previous, const float* xy_current,
                          const double T[12], const double K[9], double* out);

/* ---- ORB as descriptor extractor (second half; base_framepoint_generator.cpp:190-196,219-224) -------
 * cv::ORB::create()->compute(image, keypoints, descriptors) on provided keypoints, restated [recalled, OpenCV 3.x orb.cpp
 * detectAndCompute with useProvidedKeypoints]: keypoints closer than 31 px to the border are removed, the image (level 0:
 * FAST keypoints carry octave 0) is blurred with GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) — 8-bit fixed-point
 * separable filter, kernel round(256 k) = {18, 34, 49, 55, 49, 34, 18}, result (sum + 2^15) >> 16 — and every keypoint gets
 * 256 steered tests I(c + R p1) < I(c + R p2), R = rotation by KeyPoint::angle (degrees; float arithmetic, cvRound), bit k
 * of byte i = test 8i + k.  The 256 pairs are REPO-DEFINED (include/vslam_orb_pattern.h; OpenCV's bit_pattern_31_ is not
 * in the reference tree).  vslam_gaussian_blur7_u8: host image in, host image out (dense, cols bytes per row); rows, cols >= 4
 * (one reflection per border).
 * vslam_orb_describe: n integer keypoints xy with one angle for all (FAST: -1), keep[i] = 0 for removed keypoints. */
int vslam_gaussian_blur7_u8(vslam_ctx* ctx, const uint8_t* image, int32_t rows, int32_t cols, int32_t row_stride, uint8_t* blurred);

/* ---- the descriptor test pairs are DATA ------------------------------------------------------------------------------------
 * BRIEF-32 and ORB each compare 256 fixed pixel pairs.  OpenCV's tables (xfeatures2d generated_32.i; features2d orb.cpp
 * bit_pattern_31_) are not in the reference tree, so this library ships tables of its own (include/vslam_brief_pattern.h,
 * vslam_orb_pattern.h): same construction, different numbers — descriptors are NOT bit-compatible with an OpenCV build until
 * the caller passes OpenCV's tables in.  An integration that has them (shim/proslam_hip_plugin.h: HipContext::brief_pattern /
 * orb_pattern) calls these once before the first frame; maps, relocalization data and Hamming thresholds tuned on OpenCV
 * descriptors then keep their meaning.
 *   BRIEF pair i = {y1, x1, y2, x2}: bit i (byte i / 8, MSB first) = box9x9(p + (x1, y1)) < box9x9(p + (x2, y2)); |.| <= 24.
 *   ORB   pair i = {x1, y1, x2, y2}: bit i (byte i / 8, LSB first) = I(c + R (x1, y1)) < I(c + R (x2, y2)); |x|, |y| <= 15, the
 *   31 x 31 patch (OpenCV's bit_pattern_31_, whose points reach radius 17.7, fits).
 * The tables live in the device's constant memory: the setting holds for every context on `device` in this process.  Call
 * them while no frame is in flight.  vslam_get_*_pattern returns the table in effect (256 * 4 bytes). */
int vslam_set_brief_pattern(int device, const int8_t* pairs_y1x1y2x2 /* 256 * 4 */);
int vslam_set_orb_pattern(int device, const int8_t* pairs_x1y1x2y2 /* 256 * 4 */);
int vslam_get_brief_pattern(int device, int8_t* pairs_out /* 256 * 4 */);
int vslam_get_orb_pattern(int device, int8_t* pairs_out /* 256 * 4 */);
int vslam_orb_describe(vslam_ctx* ctx, const uint8_t* image_host, int32_t rows, int32_t cols, int32_t stride, int32_t n,
                       const int16_t* xy, float angle_degrees, uint8_t* keep, uint8_t* desc);

/* ---- RGB-D mode end to end: PoseTracker3D with DepthFramePointGenerator + UVDAligner ---------------
 * (slam_assembly.cpp _createDepthTracker; configuration_{icl,tum,xtion}.yaml).  Two implementations inside the library, same results
 * frame by frame: the device-resident loop (csrc/kernels_rgbd.h + csrc/rgbd_device.h, the default: framepoints, temporary points,
 * landmarks, history and the tracker's scalars live in HBM; a frame is two copies in, one launch sequence on two HIP streams and one
 * 1 KB state block out) and the host-driven loop over the stand-alone entry points above (csrc/rgbd_tracker.h: VSLAM_RGBD_HOST=1 in the
 * environment of vslam_rgbd_create; it also serves p->detector_type = VSLAM_DETECTOR_ORB).  One sequence per object, any detector
 * grid (configuration_icl.yaml: 2 x 2).  cfg carries the tracker / aligner / landmark / detector values, p the depth camera
 * (p->descriptor_type selects the extractor: the RGB-D configurations say "ORB-256").  A frame that fails (capacity, HIP error) leaves
 * the tracker refusing further frames until vslam_rgbd_reset (the reference throws out of compute() and the run ends, app.cpp:128).
 * vslam_rgbd_process_host = PoseTracker3D::compute for one frame: left = 8-bit image, depth = 16-bit depth image (row
 * strides in bytes / in elements).  vslam_rgbd_get_frame_info fills the counters that exist in this mode (n_keypoints_left,
 * n_detected_left, thresholds[region], track_attempts, n_tracked, n_lost, n_tracked_landmarks, aligner_*, n_inliers, n_after_prune,
 * n_recovered, n_active_landmarks, n_new_stereo = new points with measured depth, n_points, window_pixels, tau_track, status,
 * poses) and the number of temporary points of the frame.  vslam_rgbd_get_points: Frame::points() as xy (float: recovered
 * points sit at sub-pixel projections), cam, meta = (index of the predecessor in the previous frame's points followed by its
 * temporary points or -1, track length, landmark updates or 0, hasUnreliableDepth), 32 descriptor bytes. */
typedef struct vslam_rgbd vslam_rgbd;
int vslam_rgbd_create(const vslam_config* cfg, const vslam_depth_params* p, int device, vslam_rgbd** out);
void vslam_rgbd_destroy(vslam_rgbd* t);
int vslam_rgbd_reset(vslam_rgbd* t);
const char* vslam_rgbd_last_error(const vslam_rgbd* t);
int vslam_rgbd_process_host(vslam_rgbd* t, const uint8_t* left, int32_t left_row_stride, const uint16_t* depth, int32_t depth_row_stride);
/* vslam_rgbd_process_host in two halves: submit copies the frame in and enqueues its kernels, wait returns its status (after it the frame's
 * info and points can be read; between submit and wait the getters return VSLAM_ERR_STATE: the frame in flight is rewriting the lists they
 * would copy).  One frame in flight per tracker.  Several trackers — one sequence each, HIP streams of their own — overlap
 * on the GPU when their frames are submitted before any of them is waited for (tests/validation/rgbd_bench.py --trackers). */
int vslam_rgbd_submit_host(vslam_rgbd* t, const uint8_t* left, int32_t left_row_stride, const uint16_t* depth, int32_t depth_row_stride);
int vslam_rgbd_wait(vslam_rgbd* t);
/* Several sequences in ONE context (device-resident loop only): n_streams independent sequences of the same camera and configuration advance
 * together, one frame each per call — the launch sequence of a frame serves all of them (one workgroup per sequence in its single-workgroup
 * kernels, a grid dimension in the wide ones), so a step costs the time of its slowest sequence (tests/validation/rgbd_batch.py).  Images: n_streams
 * images `left_stream_stride` bytes apart, depth images `depth_stream_stride` ELEMENTS apart.  A sequence that needs another registration attempt
 * gets it without disturbing the others.  Results per sequence through the *_stream getters. */
int vslam_rgbd_create_batch(const vslam_config* cfg, const vslam_depth_params* p, int device, int32_t n_streams, vslam_rgbd** out);
int vslam_rgbd_process_batch_host(vslam_rgbd* t, const uint8_t* left, int32_t left_row_stride, size_t left_stream_stride, const uint16_t* depth,
                                  int32_t depth_row_stride, size_t depth_stream_stride);
int vslam_rgbd_submit_batch_host(vslam_rgbd* t, const uint8_t* left, int32_t left_row_stride, size_t left_stream_stride, const uint16_t* depth,
                                 int32_t depth_row_stride, size_t depth_stream_stride);   /* then vslam_rgbd_wait */
/* the same with images that are already in HBM (device pointers; the caller orders their producer before this call on the device, e.g.
 * by a device synchronisation): nothing is copied.  Depth images dense per sequence (depth_stream_stride == rows * depth_row_stride). */
int vslam_rgbd_submit_batch_device(vslam_rgbd* t, const uint8_t* left_device, int32_t left_row_stride, size_t left_stream_stride,
                                   const uint16_t* depth_device, int32_t depth_row_stride, size_t depth_stream_stride);
int vslam_rgbd_get_frame_info_stream(vslam_rgbd* t, int32_t stream, vslam_frame_info* out, int32_t* n_temporary);
int vslam_rgbd_get_points_stream(vslam_rgbd* t, int32_t stream, int32_t cap, int32_t* n, float* xy, double* cam, int32_t* meta4, uint8_t* desc);
int vslam_rgbd_get_frame_info(vslam_rgbd* t, vslam_frame_info* out, int32_t* n_temporary);
int vslam_rgbd_get_points(vslam_rgbd* t, int32_t cap, int32_t* n, float* xy, double* cam, int32_t* meta4, uint8_t* desc);

/* ---- OrbDetector components (first half; base_framepoint_generator.cpp:52-70) ----------------------
 * The reference's OrbDetector is cv::ORB::create(5000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, threshold) used as a DETECTOR
 * (descriptors still come from the configured extractor).  OpenCV is not in the reference tree: the published algorithm
 * (features2d/src/orb.cpp computeKeyPoints, imgproc resize) is restated [recalled]; parity is pinned against the
 * repo's independent numpy restatement only. */

/* cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8-bit single channel: 11-bit fixed-point bilinear weights,
 * horizontal pass in 32-bit, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  Host images. */
int vslam_resize_linear_u8(vslam_ctx* ctx, const uint8_t* src, int32_t rows, int32_t cols, int32_t row_stride,
                           uint8_t* dst, int32_t dst_rows, int32_t dst_cols);

/* HarrisResponses (block 7, k = 0.04) and ICAngles (half patch 15, fastAtan2) of orb.cpp at n integer pixel positions
 * (xy: n*2 int16, at least 16 px from the border): response n floats, angle n floats (degrees). */
int vslam_harris_angle(vslam_ctx* ctx, const uint8_t* image, int32_t rows, int32_t cols, int32_t row_stride, int32_t n,
                       const int16_t* xy, float* response, float* angle);

/* ORB::detect with HARRIS_SCORE, firstLevel 0, no mask: pyramid by successive INTER_LINEAR resizes, FAST-9/16 + NMS per
 * level, 31 px border filter, retainBest(2 n_level) on the FAST score, Harris response, retainBest(n_level), intensity
 * centroid angle, coordinates scaled back to level 0.  retainBest keeps every keypoint whose response ties the n-th
 * (as KeyPointsFilter does); within a level the keypoints stay in row-major order (std::nth_element leaves the order
 * unspecified upstream).  out: 6 floats per keypoint (x, y, size, angle, response, octave), levels in ascending order. */
int vslam_orb_detect(vslam_ctx* ctx, const uint8_t* image, int32_t rows, int32_t cols, int32_t row_stride,
                     int32_t nfeatures, float scale_factor, int32_t nlevels, int32_t edge_threshold, int32_t patch_size,
                     int32_t fast_threshold, int32_t cap, int32_t* n, float* keypoints);

/* cv::ORB::create()->compute() on keypoints that carry an octave and an angle — what BaseFramePointGenerator::computeDescriptors
 * (base_framepoint_generator.cpp:431-438) does with an OrbDetector's keypoints when the extractor is ORB [recalled: orb.cpp
 * detectAndCompute with useProvidedKeypoints]: runByImageBorder(31) on the level-0 coordinates (Rect::contains rounds the float point), a
 * pyramid up to the highest octave present (level l from level l-1, INTER_LINEAR, cvRound(cols / scale) x cvRound(rows / scale), scale =
 * (float)pow(scale_factor, l)), GaussianBlur 7x7 sigma 2 per level, 256 tests steered by the keypoint's own angle around
 * (cvRound(x / scale), cvRound(y / scale)) of its level.  keypoints: n x 6 floats as vslam_orb_detect writes them.  keep[i] = 0: removed
 * (border filter; also a keypoint whose pattern would leave its level, which an OrbDetector never produces). */
int vslam_orb_describe_keypoints(vslam_ctx* ctx, const uint8_t* image, int32_t rows, int32_t cols, int32_t row_stride, int32_t n,
                                 const float* keypoints, float scale_factor, uint8_t* keep, uint8_t* descriptors);

/* ---- multi-GPU: trajectory assembly ---------------------------------------------------------
 * No reference counterpart (single process).  The pose all-gather is issued by the host
 * launcher through RCCL (torch.distributed backend "nccl"); these helpers pack/unpack. */
int vslam_get_poses(vslam_ctx* ctx, int stream, int32_t first_frame, int32_t n_frames,
                    double* camera_left_to_world /* n_frames*12 */);
/* Same, for ALL streams, into DEVICE memory (dst[stream][frame][12], asynchronous on the context
 * stream): the send buffer of the RCCL all-gather. */
int vslam_copy_poses_device(vslam_ctx* ctx, int32_t first_frame, int32_t n_frames, double* dst_device);

/* ---- multi-GPU: the pose all-gather itself, on RCCL, for callers that are not Python ---------------
 * One process per GPU.  Rank 0 asks for a unique id (128 bytes, ncclUniqueId) and hands it to the other ranks by whatever
 * channel launched them (MPI, a file, torch.distributed's store); every rank then joins with vslam_comm_init.  RCCL
 * (librccl.so) is loaded on first use: a single-GPU user of the library never needs it.
 * vslam_allgather_poses: send_device holds this rank's count doubles (e.g. [streams][frames][12] as vslam_copy_poses_device
 * leaves them), recv_device receives nranks * count doubles ordered by rank; one ncclAllGather (xGMI on one node), queued
 * on hip_stream (NULL: the default stream), asynchronous — synchronise the stream before reading. */
typedef struct vslam_comm vslam_comm;
#define VSLAM_COMM_ID_BYTES 128
/* Local precondition of vslam_comm_init (librccl.so loadable with the expected symbols, the device selectable): every rank
 * calls it and the ranks agree on the result over the launcher's channel BEFORE anyone enters the collective
 * ncclCommInitRank — a rank that cannot take part must not leave its peers waiting in it. */
int vslam_comm_available(int device);
int vslam_comm_unique_id(uint8_t id[VSLAM_COMM_ID_BYTES]);
int vslam_comm_init(int rank, int nranks, const uint8_t id[VSLAM_COMM_ID_BYTES], int device, vslam_comm** out);
int vslam_allgather_poses(vslam_comm* comm, const double* send_device, double* recv_device, size_t count, void* hip_stream);
void vslam_comm_destroy(vslam_comm* comm);
const char* vslam_comm_last_error(void);

/* The pose (camera_left_to_world) of the frame every stream processed last, into DEVICE memory dst[stream][12],
 * asynchronous on the context stream: one row block of the all-gather's send buffer per step. */
int vslam_copy_current_poses_device(vslam_ctx* ctx, double* dst_device);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_H */
