/* synth_scene.h — seeded synthetic stereo world used for tests and benchmarks.
 *
 * Data generator only (neither oracle nor product): no KITTI / EuRoC data exists in this
 * pipeline, so sequences are rendered from a static analytic world: a textured
 * "street canyon" (ground plane + two side walls) seen by a rectified pinhole stereo pair that
 * drives along it on a smooth, closed-form path.  Geometry defaults are KITTI-00-shaped.
 * The same inline code compiles for the host (g++) and the device (hipcc).
 */
#ifndef SYNTH_SCENE_H
#define SYNTH_SCENE_H
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define SYNTH_HD __host__ __device__ inline
#else
#define SYNTH_HD inline
#endif

typedef struct synth_scene {
  int32_t rows, cols;
  double fx, fy, cx, cy;
  double baseline_m;     /* stereo baseline in metres (right camera at +x)      */
  double cam_height_m;   /* ground plane at y = +cam_height (y points down)     */
  double wall_half_m;    /* walls at x = -/+ wall_half                          */
  double max_depth_m;    /* beyond this everything is uniform "sky"             */
  double cell_m;         /* finest texture cell edge                            */
  double speed_m;        /* forward motion per frame                            */
  double sway_m;         /* lateral sway amplitude                              */
  double sway_rate;      /* radians per frame of the sway                       */
  uint64_t seed;
  double bob_m;          /* vertical bob amplitude (0.03 m in the KITTI-shaped scene)            */
  double roll_amp, pitch_amp;   /* radians; 0 = planar motion (the KITTI-shaped scene)           */
  double roll_rate, pitch_rate; /* radians per frame of the two oscillations                     */
  double contrast;       /* texture amplitude factor (1 = the KITTI-shaped scene)                  */
  uint64_t noise_seed;   /* 0 = the scene's own sensor noise; other values: same world and path, another realisation of the
                          * +-2 grey levels of sensor noise (ATE spread under input perturbation, tools/eval_ate_noise.py)  */
} synth_scene;

SYNTH_HD uint32_t synth_hash(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  uint64_t z = a * 0x9E3779B97F4A7C15ull + b * 0xC2B2AE3D27D4EB4Full + c * 0x165667B19E3779F9ull +
               d * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

/* camera-left-to-world pose of frame k: R (row-major 3x3), t.  Forward along +z with a sinusoidal lateral
 * sway, heading tangent to the path, small vertical bob; with roll_amp / pitch_amp != 0 the camera also rolls
 * and pitches (6-DoF, the EuRoC-shaped scene): R = R_yaw * R_pitch * R_roll. */
SYNTH_HD void synth_pose(const synth_scene* s, int k, double R[9], double t[3]) {
  const double a = s->sway_rate * (double)k;
  const double x = s->sway_m * sin(a);
  const double dx = s->sway_m * s->sway_rate * cos(a);
  const double yaw = atan2(dx, s->speed_m);
  const double cy = cos(yaw), sy = sin(yaw);
  /* rotation about the camera y axis (down): x_world = cy*x + sy*z, z_world = -sy*x + cy*z */
  R[0] = cy; R[1] = 0; R[2] = sy;
  R[3] = 0;  R[4] = 1; R[5] = 0;
  R[6] = -sy; R[7] = 0; R[8] = cy;
  if (s->roll_amp != 0.0 || s->pitch_amp != 0.0) {
    const double pa = s->pitch_amp * sin(s->pitch_rate * (double)k), ra = s->roll_amp * sin(s->roll_rate * (double)k);
    const double cp = cos(pa), sp = sin(pa), cr = cos(ra), sr = sin(ra);
    /* pitch about x: [1 0 0; 0 cp -sp; 0 sp cp], roll about z: [cr -sr 0; sr cr 0; 0 0 1] */
    const double P[9] = {1, 0, 0, 0, cp, -sp, 0, sp, cp};
    const double Q[9] = {cr, -sr, 0, sr, cr, 0, 0, 0, 1};
    double A[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[3 * i + j] = R[3 * i] * P[j] + R[3 * i + 1] * P[3 + j] + R[3 * i + 2] * P[6 + j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = A[3 * i] * Q[j] + A[3 * i + 1] * Q[3 + j] + A[3 * i + 2] * Q[6 + j];
  }
  t[0] = x;
  t[1] = s->bob_m * sin(0.37 * (double)k);
  t[2] = s->speed_m * (double)k;
}

/* intensity of one ray; (u,v) sub-pixel image coordinates, side 0 = left, 1 = right */
SYNTH_HD double synth_sample(const synth_scene* s, const double R[9], const double t[3], int side,
                             double u, double v) {
  const double dx = (u - s->cx) / s->fx, dy = (v - s->cy) / s->fy;
  const double Dx = R[0] * dx + R[1] * dy + R[2];
  const double Dy = R[3] * dx + R[4] * dy + R[5];
  const double Dz = R[6] * dx + R[7] * dy + R[8];
  const double b = side ? s->baseline_m : 0.0;
  const double ox = t[0] + R[0] * b, oy = t[1] + R[3] * b, oz = t[2] + R[6] * b;
  double best = 1e30;
  int surf = -1;
  if (Dy > 1e-12) { const double q = (s->cam_height_m - oy) / Dy; if (q > 0 && q < best) { best = q; surf = 0; } }
  if (Dx < -1e-12) { const double q = (-s->wall_half_m - ox) / Dx; if (q > 0 && q < best) { best = q; surf = 1; } }
  if (Dx > 1e-12) { const double q = (s->wall_half_m - ox) / Dx; if (q > 0 && q < best) { best = q; surf = 2; } }
  if (surf < 0 || best > s->max_depth_m) return 96.0;
  const double px = ox + best * Dx, py = oy + best * Dy, pz = oz + best * Dz;
  const double a = (surf == 0) ? px : py;
  const double c = pz;
  /* perpendicular distance camera -> surface: sets the projected size of a texture cell along its
   * foreshortened axis, minor_px = cell * f * dist / depth^2.  Octaves whose cells would be thinner
   * than ~2 px are faded out smoothly (no depth-keyed discontinuities, which would be features
   * that move with the camera). */
  const double dist = (surf == 0) ? (s->cam_height_m - oy) : (surf == 1 ? (ox + s->wall_half_m) : (s->wall_half_m - ox));
  const double k = s->fx * dist / (best * best);
  double cell = s->cell_m;
  double acc = 127.5;
  const double amp[5] = {0.45, 0.40, 0.35, 0.30, 0.30};
  for (int o = 0; o < 5; ++o) {
    double w = (cell * k - 2.0) * 0.5;
    w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
    if (w > 0.0) {
      const int64_t ia = (int64_t)floor(a / cell), ic = (int64_t)floor(c / cell);
      const uint32_t h = synth_hash(s->seed + 17 * (uint64_t)surf + 1000 * (uint64_t)o, (uint64_t)ia, (uint64_t)ic, 7);
      acc += s->contrast * (w * amp[o] * ((double)(h & 255u) - 127.5));
    }
    cell *= 2.0;
  }
  return acc < 0.0 ? 0.0 : (acc > 255.0 ? 255.0 : acc);
}

/* depth of the LEFT camera's pixel (x, y) along the optical axis, in units of `unit_m` metres (RGB-D tests: unit 2 mm keeps
 * the 90 m scene inside 16 bits); 0 = no surface within max_depth_m.  The ray parameter of synth_sample IS the camera-frame z. */
SYNTH_HD uint16_t synth_depth(const synth_scene* s, const double R[9], const double t[3], int x, int y, double unit_m) {
  const double dx = ((double)x - s->cx) / s->fx, dy = ((double)y - s->cy) / s->fy;
  const double Dx = R[0] * dx + R[1] * dy + R[2];
  const double Dy = R[3] * dx + R[4] * dy + R[5];
  double best = 1e30;
  if (Dy > 1e-12) { const double q = (s->cam_height_m - t[1]) / Dy; if (q > 0 && q < best) best = q; }
  if (Dx < -1e-12) { const double q = (-s->wall_half_m - t[0]) / Dx; if (q > 0 && q < best) best = q; }
  if (Dx > 1e-12) { const double q = (s->wall_half_m - t[0]) / Dx; if (q > 0 && q < best) best = q; }
  if (best > s->max_depth_m) return 0;
  const double v = floor(best / unit_m + 0.5);
  return v > 65535.0 ? 0 : (uint16_t)v;
}

/* one output pixel: 2x2 supersampling + +-2 grey levels of deterministic sensor noise */
SYNTH_HD uint8_t synth_pixel(const synth_scene* s, const double R[9], const double t[3], int frame,
                             int side, int x, int y) {
  double acc = 0.0;
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 2; ++i)
      acc += synth_sample(s, R, t, side, (double)x - 0.25 + 0.5 * i,
                          (double)y - 0.25 + 0.5 * j);
  const int noise = (int)(synth_hash(s->seed ^ 0xABCDull ^ (s->noise_seed * 0x9E3779B97F4A7C15ull), (uint64_t)frame * 2 + side, (uint64_t)x,
                                     (uint64_t)y) % 5u) - 2;
  int val = (int)floor(acc * 0.25 + 0.5) + noise;
  if (val < 0) val = 0;
  if (val > 255) val = 255;
  return (uint8_t)val;
}

SYNTH_HD void synth_default_kitti(synth_scene* s) {
  s->rows = 376; s->cols = 1241;
  s->fx = 718.856; s->fy = 718.856; s->cx = 607.1928; s->cy = 185.2157;
  s->baseline_m = 0.5371657; /* 386.1448 / 718.856 */
  s->cam_height_m = 1.65; s->wall_half_m = 7.0; s->max_depth_m = 90.0; s->cell_m = 0.18;
  s->speed_m = 0.9; s->sway_m = 1.2; s->sway_rate = 0.05; s->seed = 7;
  s->bob_m = 0.03; s->roll_amp = 0.0; s->pitch_amp = 0.0; s->roll_rate = 0.0; s->pitch_rate = 0.0; s->contrast = 1.0; s->noise_seed = 0;
}

/* EuRoC-MH-shaped: 752x480, the cam0 intrinsics and 0.11 m baseline of configuration_euroc.yaml's data set, an indoor
 * hall (walls 3 m either side, floor 1.2 m below, nothing beyond 25 m) and a slow 6-DoF MAV motion: <= 5 cm and
 * <= 1 degree per frame (forward 4 cm, sway, bob, roll, pitch, heading along the path). */
SYNTH_HD void synth_default_euroc(synth_scene* s) {
  s->rows = 480; s->cols = 752;
  s->fx = 458.654; s->fy = 457.296; s->cx = 367.215; s->cy = 248.375;
  s->baseline_m = 0.11;
  s->cam_height_m = 1.2; s->wall_half_m = 3.0; s->max_depth_m = 25.0; s->cell_m = 0.045;
  s->speed_m = 0.04; s->sway_m = 0.2; s->sway_rate = 0.05; s->seed = 7;
  s->bob_m = 0.02; s->roll_amp = 0.05; s->pitch_amp = 0.03; s->roll_rate = 0.11; s->pitch_rate = 0.07; s->contrast = 0.5; s->noise_seed = 0;
}
#endif
