// dev_types.h — device-side data model of libvslam_hip (gfx950).
//
// Everything a stream (one sequence / chunk) owns lives in HBM as flat SoA arrays sized by the
// capacities of vslam_config; kernels address them through `DevBuf` (passed by value) using the
// stream index (blockIdx) — no per-frame allocation, no host round trip.
//
// Layout per stream s, image side d (0 = left, 1 = right):
//   box      u16 [rows][bstride]     9x9 box sums of the image (BRIEF's smoothed image; <= 20655)
//   score8   u8  [rows][bstride]     FAST score, written only at surviving corners
//   mask     u64 [rows][TX]          1 bit / pixel: FAST corner that survived 3x3 NMS
//   kp_*, desc   [NMAX]              keypoints inside the 28 px descriptor border, image row-major
//   rowcell  i32 [rows][CW+1]        CSR: index of the first keypoint of row r with x >= 16*c
//                                    (replaces the reference's rows x cols pointer lattice,
//                                    intensity_feature_matcher.cpp:36-46)
//   used     u8  [NMAX]              feature removed from the "feature_vector" (matched / parallax)
//   kill     i32 [NMAX]              scratch of the order-exact track resolution
#pragma once
#include <stdint.h>
#include "../../include/vslam_hip.h"

#define VS_TILE_W 64
#ifndef VS_TILE_H
#define VS_TILE_H 64          // rows of a k_fast_box tile (multiple of 8); 32 .. 96 measured with the 20 KB LDS footprint:
                              // 64 is the fastest (48 was, while the horizontal sums had their own 7 KB)
#endif
#define VS_CELL 16
#define VS_MAXCAND 16        // candidate list length per previous point (overflow -> exact rescan)
#ifndef VS_WG
#define VS_WG 512            // threads of the per-stream frame kernel
#endif
#ifdef VS_FRAME_WAVES_PER_EU       // optional register budget of the frame kernel: 512 / this VGPRs per lane
#define VS_FRAME_BOUNDS __launch_bounds__(VS_WG, VS_FRAME_WAVES_PER_EU)
#else
#define VS_FRAME_BOUNDS __launch_bounds__(VS_WG)
#endif
#ifndef VS_TRAIL
#define VS_TRAIL 32           // predecessors a framepoint knows by index (64 B per point): the landmark refinement reads its measurements without chasing the per-frame `prev` links
#endif
#ifndef VS_ARENA
#define VS_ARENA (128 * 1024) // bytes of LDS scratch the frame kernel stages hot index arrays in (one frame workgroup per CU)
#endif
#define VS_POSE_LOG 32768    // frames of trajectory kept per stream

struct DevRegion { int32_t x, y, w, h; };

struct DevCfg {
  vslam_config c;
  int32_t TX, CW, bstride;           // mask words per row, 16-px cells per row, box/score row stride
  int32_t n_regions;
  DevRegion regions[VSLAM_MAX_REGIONS];
  int32_t rows_bin, cols_bin, target_kp, target_per_detector;
  int32_t n_offsets;
  int32_t offsets[2 * VSLAM_MAX_EPI + 1];
  int32_t NMAX, MAXP, HCAP;
  int32_t trail;           // 1: framepoints carry the indices of their track's last VS_TRAIL predecessors (MAXP <= 65535)
  int32_t n_streams;
  int32_t mono;            // 1: one image per stream (RGB-D mode): the tile kernels' grid z is the stream, not 2 * stream + side
  // ORB extractor: the rotation of the pattern by the FAST keypoints' angle (-1 degree), evaluated on the host exactly as
  // OpenCV does — angle *= (float)(CV_PI/180.f); (float)cos(angle), (float)sin(angle) — and the fixed-point Gaussian taps
  float orb_cos, orb_sin;
  int32_t gauss7[4];
};

// scalars of the frame in flight, handed from one phase kernel of the frame to the next
struct FrameCarry {
  int32_t status, status0, win, attempts, broken, fallback, n_after_prune, aligner_valid, n_tracked_landmarks;
  int32_t n_cur, n_lost, n_recovered, n_active;
  int32_t lm_pb, lm_f;    // point buffer and frame index of the frame whose landmarks k_update_landmarks refines (it may run beside the frame's last phase, which advances StreamState::cur / frame_count)
  double tau_track, tau_gen, tau_tri;
  double prior[12];
  unsigned long long t0;
};

// tracker + generator state carried from frame to frame (PoseTracker3D / BaseFramePointGenerator members)
struct StreamState {
  int32_t thr[VSLAM_MAX_REGIONS];        // FastDetector thresholds in effect for the next detect
  int32_t status;                        // _status
  int32_t win;                           // _projection_tracking_distance_pixels
  int32_t frame_count;                   // frames processed
  int32_t has_prev;
  int32_t n_tracked_landmarks_prev;
  int32_t cur;                           // which point buffer holds the CURRENT frame (0/1)
  int32_t aligner_valid;
  int32_t error_flags;                   // sticky
  double tau_track;                      // _current_descriptor_distance_tracking
  double tau_tri;                        // _current_maximum_descriptor_distance_triangulation
  double prior[12];                      // _previous_to_current_camera
  double pose[12];                       // WorldMap::robot_to_world (camera_left_to_world)
  // scratch scalars of the frame in flight
  int32_t by_appearance;                 // mode the candidate kernel must use for attempt 0
  int32_t n_trk, n_lost, n_tracked_landmarks;
  int32_t al_n, al_inliers, al_outliers, al_iterations, al_converged;
  int32_t al_wsize;                      // size of StereoUVAligner::_weights_translation after the last initialize()
  double al_total_error;
  double al_T[12];
  double al_H[36];
  // in-kernel chronometers (wall_clock64 ticks, 100 MHz): tracking, pose_optimization, point_recovery,
  // landmark_optimization, track_creation (== point_triangulation)
  unsigned long long ticks[5];
  unsigned long long dbg[12];              // fine-grained phase ticks (profiling builds of the bench)
  // stage-granular API (the shim's host-driven PoseTracker3D): values handed from one stage call to the next
  double tau_gen;                        // generator's _maximum_descriptor_distance_tracking (last track())
  int32_t n_cur, n_active, n_after_prune, n_recovered, n_new, track_calls;
  FrameCarry fc;
};

// detector bookkeeping of one frame's image pipeline (double-buffered with the image products so that frame
// t+1 can be detected/described while frame t is still being tracked)
struct ImgInfo {
  int32_t thr_after[VSLAM_MAX_REGIONS];   // thresholds after adjustDetectorThresholds of this frame
  int32_t raw_count[2][VSLAM_MAX_REGIONS];
  int32_t ticket;                         // k_emit: arrival counter of the stream's two per-image workgroups
};

#define VS_MAX_STREAMS 4096   // streams per context (one bit each in DevBuf::active)
struct DevBuf {
  int32_t s0;          // first stream of the launch (streams are processed in independent groups)
  // Workgroup b of a launch runs on XCD (q0 + b) % 8, where q0 belongs to the hardware queue (HIP stream) — measured: constant from
  // launch to launch, idle or busy, different for different HIP streams (tools/probe/xcd_map.hip).  xcd_rot = (q0 - s0) & 7 of the
  // queue THIS launch goes to (calibrated when the context is created), so that the relabelling below puts stream s on the SAME
  // physical XCD s % 8 in every kernel of the step, whichever queue it is launched on.  Speed only: any value is a valid permutation.
  int32_t xcd_rot;
  // one bit per stream: a cleared bit makes every kernel skip the stream (its sequence has ended while other streams of
  // the context still run: whole sequences of different lengths per stream, SURVEY.md 8e exact mode).  By value in the
  // kernel arguments: a scalar load from the kernarg segment, no global round trip in the wide kernels.
  uint32_t active[VS_MAX_STREAMS / 32];
  // current input images (device pointers)
  const uint8_t* img[2];
  int32_t img_row_stride;
  size_t img_stream_stride;
  // image-pipeline products
  uint16_t* box;
  uint8_t* score8;
  unsigned long long* mask;
  int16_t* kp_xy;      // [B][2][NMAX][2]
  uint8_t* kp_score;   // [B][2][NMAX]
  uint8_t* desc;       // [B][2][NMAX][32]
  int32_t* n_kp;       // [B][2]
  int32_t* rowcell;    // [B][2][rows][CW+1]
  uint8_t* used;       // [B][2][NMAX]
  int32_t* kill;       // [B][2][NMAX]
  ImgInfo* iinfo;      // [B]
  // per-stream state
  StreamState* st;
  vslam_frame_info* info;
  double* pose_log;    // [B][VS_POSE_LOG][12]
  // frame points, ping-pong [B][2][MAXP]
  int16_t* p_kp;       // [..][4]
  uint8_t* p_desc;     // [..][64]
  int32_t* p_meta;     // [..][6]  dist, epi, prev, track_len, lm_updates, has_next
  double* p_cam;       // [..][3]
  double* p_camlm;     // [..][3]
  double* p_lm;        // [..][3]
  uint16_t* p_trail;   // [..][VS_TRAIL] index of the track's point in frame f-1, f-2, ... (0xFFFF: the track starts before); see wg_publish_history
  int32_t* n_points;   // [B][2]
  // track candidates / resolution  [B][MAXP]
  int32_t* proj;       // [..][8] row, col, candidate count (-1 = projection outside the image), |epipolar offset|,
                       //         right-candidate count of the first left candidate (-1 dead, 9 overflow), its x, -, -
  uint32_t* cand_rkey; // [..][8] sorted (reject << 31 | distance << 16 | right feature index)
  double* proj_q;      // [..][2] right-image projection u/w, v/w of the previous point under the prior
  uint32_t* cand_key;  // [..][VS_MAXCAND] sorted (primary << 16 | left feature index)
  int32_t* res;        // [..][8] fl, fr, dist, flag (bit0 success, bit1 lost-eligible), x of fl, row of fr, pad
  int32_t* trk;        // [..][4] prev, fl, fr, dist  (compacted, order of previous points)
  int32_t* lost;       // [..]
  // aligner SoA [B][MAXP]
  double* al_moving;   // [..][3]
  double* al_fixed;    // [..][4]
  double* al_omega;
  double* al_weight;
  double* al_chi;
  uint8_t* al_inl;
  // recovery scratch [B][MAXP]
  int32_t* rec;        // [..][6] flag, xL, yL, xR, yR, dist
  uint8_t* rec_desc;   // [..][64]
  // stereo scratch
  int32_t* st_match;   // [B][NMAX][3]  matched right index / distance per left feature; [2*NMAX..) winners of the bin grid
  int32_t* sc;         // [B][NMAX][4]  fl, fr, dist, epi  (new candidates in sweep order)
  int32_t* bin_occ;    // [B][rows_bin*cols_bin]
  uint8_t* sdist;      // [B][NMAX][16] precomputed L-R Hamming distances of the stereo sweep
  int32_t* bin_aux;    // [B][2*(nb+1) + NMAX]  per-bin counts, starts, candidate lists
  // history ring for landmark refinement [B][HCAP]
  double* h_pose;      // [..][24]  cam_to_world, world_to_cam
  double* h_cam;       // [..][MAXP][4]  camera coordinates and 1 / z
  int32_t* h_prev;     // [..][MAXP]
};
__device__ __forceinline__ bool vs_active(const DevBuf& b, int s) { return (b.active[s >> 5] >> (s & 31)) & 1u; }
// one-workgroup-per-stream grids: block i of n works on local stream xcd_local_stream(i, n, rot) — inside every aligned group of eight
// blocks (which covers the eight XCDs once) the block on physical XCD x takes the stream with (s0 + stream) % 8 == x; a bijection on 0 .. n-1
__device__ __forceinline__ int xcd_local_stream(int i, int n, int rot) { return i < ((n >> 3) << 3) ? ((i & ~7) | ((i + rot) & 7)) : i; }
