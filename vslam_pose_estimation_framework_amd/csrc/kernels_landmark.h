// kernels_landmark.h — Landmark::update (types/landmark.cpp:66-167) on caller-provided measurement lists, stand-alone.
// The fused tracker runs the same refinement inside its frame kernel (kernels_frame2.h landmark_point, walking the history
// ring); this entry exists so that the arithmetic is pinned by an independent fixture.  One thread per landmark: a 3x3
// Gauss-Newton over 3..100 measurements is latency-bound bookkeeping, the batch supplies the parallelism.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_math.h"

__global__ __launch_bounds__(256) void k_landmark_update(int n, const int32_t* offsets, const int32_t* frame_of, const double* w2c,
                                                         const double* c2w, const double* cam, double* world, int32_t* updates,
                                                         int max_iterations, double kernel) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = offsets[i], e = offsets[i + 1];
  if (e <= a) return;
  double wv[3] = {world[3 * (size_t)i], world[3 * (size_t)i + 1], world[3 * (size_t)i + 2]};
  double err_prev = 0;
  for (int it = 0; it < max_iterations; ++it) {
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0};
    double err = 0;
    int n_out = 0;
    for (int m = a; m < e; ++m) {
      const double* W = w2c + 12 * (size_t)frame_of[m];
      const double* mc = cam + 3 * (size_t)m;
      double sp[3];
      tf_apply(W, wv, sp);                                                          // :97
      if (sp[2] <= 0) { ++n_out; continue; }                                        // :98-101
      const double er[3] = {sp[0] - mc[0], sp[1] - mc[1], sp[2] - mc[2]};           // :104
      double om = 1 / mc[2];                                                        // :107 (Measurement::inverse_depth_meters)
      const double e2 = om * ((er[0] * er[0] + er[1] * er[1]) + er[2] * er[2]);     // :110
      err += e2;
      if (e2 > kernel) { om *= kernel / e2; ++n_out; }                              // :114-117
      for (int r = 0; r < 3; ++r) {                                                 // :120-127, J = R
        for (int cc = 0; cc < 3; ++cc) H[3 * r + cc] += om * ((W[r] * W[cc] + W[4 + r] * W[4 + cc]) + W[8 + r] * W[8 + cc]);
        bv[r] += om * ((W[r] * er[0] + W[4 + r] * er[1]) + W[8 + r] * er[2]);
      }
    }
    double nb[3] = {-bv[0], -bv[1], -bv[2]}, dx[3];
    full_piv_solve_regs<3>(H, nb, dx);                                                   // :131
    for (int q = 0; q < 3; ++q) wv[q] += dx[q];
    if (fabs(err - err_prev) < 1e-5 || it == 999) {                                 // :134
      const int n_in = (e - a) - n_out;
      if ((unsigned)n_in > (unsigned)updates[i]) {                                  // :138-142
        for (int q = 0; q < 3; ++q) world[3 * (size_t)i + q] = wv[q];
        updates[i] = n_in;
      } else if (n_in < n_out) {                                                    // :145-155
        double acc[3] = {0, 0, 0};
        for (int m = a; m < e; ++m) {
          double wp[3];
          tf_apply(c2w + 12 * (size_t)frame_of[m], cam + 3 * (size_t)m, wp);
          for (int q = 0; q < 3; ++q) acc[q] += wp[q];
        }
        for (int q = 0; q < 3; ++q) world[3 * (size_t)i + q] = acc[q] / (double)(e - a);
      }
      break;
    }
    err_prev = err;
  }
}
