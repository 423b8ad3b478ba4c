// dev_math.h — fixed-size fp64 helpers for the device (Eigen / srrg_core semantics restated;
// SURVEY.md §8c).  Operation order is written out explicitly and the file is compiled with
// -ffp-contract=off so that integer decisions derived from these values (truncated projections,
// gates) are bit-identical to the CPU oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VS_HD __host__ __device__ __forceinline__

VS_HD void tf_identity(double* T) {
  for (int i = 0; i < 12; ++i) T[i] = 0.0;
  T[0] = T[5] = T[10] = 1.0;
}
VS_HD void tf_apply(const double* T, const double* p, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = ((T[4 * i + 0] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2]) + T[4 * i + 3];
}
VS_HD void tf_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) C[4 * i + j] = (A[4 * i + 0] * B[j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j];
    C[4 * i + 3] = ((A[4 * i + 0] * B[3] + A[4 * i + 1] * B[7]) + A[4 * i + 2] * B[11]) + A[4 * i + 3];
  }
}
VS_HD void tf_inverse(const double* A, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[4 * i + j] = A[4 * j + i];
  for (int i = 0; i < 3; ++i) C[4 * i + 3] = -((C[4 * i + 0] * A[3] + C[4 * i + 1] * A[7]) + C[4 * i + 2] * A[11]);
}
VS_HD void mat3_mul_vec(const double* K, const double* p, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = (K[3 * i + 0] * p[0] + K[3 * i + 1] * p[1]) + K[3 * i + 2] * p[2];
}
// srrg_core::v2t: translation + vector part of a unit quaternion
VS_HD void v2t(const double* v, double* T) {
  double qx = v[3], qy = v[4], qz = v[5], qw;
  const double n2 = (qx * qx + qy * qy) + qz * qz;
  if (n2 < 1) {
    qw = sqrt(1 - n2);
  } else {
    const double n = sqrt(n2);
    qx /= n; qy /= n; qz /= n; qw = 0;
  }
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T[0] = 1 - (tyy + tzz); T[1] = txy - twz;       T[2] = txz + twy;
  T[4] = txy + twz;       T[5] = 1 - (txx + tzz); T[6] = tyz - twx;
  T[8] = txz - twy;       T[9] = tyz + twx;       T[10] = 1 - (txx + tyy);
  T[3] = v[0]; T[7] = v[1]; T[11] = v[2];
}
// Gaussian elimination with full pivoting (Eigen::FullPivLU::solve): pivot = first strict maximum
// in column-major order of the remaining corner.
template <int N>
VS_HD void full_piv_solve(const double* A_in, const double* b_in, double* x) {
  double A[N * N], b[N], y[N];
  int perm[N];
  for (int i = 0; i < N * N; ++i) A[i] = A_in[i];
  for (int i = 0; i < N; ++i) { b[i] = b_in[i]; perm[i] = i; y[i] = 0; }
  int rank = N;
  for (int k = 0; k < N; ++k) {
    int pr = k, pc = k;
    double best = 0;
    for (int j = k; j < N; ++j)
      for (int i = k; i < N; ++i) {
        const double a = fabs(A[i * N + j]);
        if (a > best) { best = a; pr = i; pc = j; }
      }
    if (best == 0) { rank = k; break; }
    if (pr != k) {
      for (int j = 0; j < N; ++j) { const double t = A[k * N + j]; A[k * N + j] = A[pr * N + j]; A[pr * N + j] = t; }
      const double t = b[k]; b[k] = b[pr]; b[pr] = t;
    }
    if (pc != k) {
      for (int i = 0; i < N; ++i) { const double t = A[i * N + k]; A[i * N + k] = A[i * N + pc]; A[i * N + pc] = t; }
      const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t;
    }
    for (int i = k + 1; i < N; ++i) {
      const double f = A[i * N + k] / A[k * N + k];
      A[i * N + k] = 0;
      for (int j = k + 1; j < N; ++j) A[i * N + j] -= f * A[k * N + j];
      b[i] -= f * b[k];
    }
  }
  for (int i = rank - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < rank; ++j) s -= A[i * N + j] * y[j];
    y[i] = s / A[i * N + i];
  }
  for (int i = 0; i < N; ++i) x[perm[i]] = y[i];
}
#ifdef __HIPCC__
// The same elimination with every index static: rows, columns and the permutation are exchanged through selects, so the
// matrix lives in registers (the generic form above indexes its arrays with the pivot position, which puts them into
// scratch memory on the device: a memory round trip per access).  Same operands, same operations, same order: same bits.
template <int N>
__device__ __forceinline__ void full_piv_solve_regs(const double* A_in, const double* b_in, double* x) {
  double A[N][N], b[N], y[N];
  int perm[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) A[i][j] = A_in[i * N + j];
    b[i] = b_in[i]; perm[i] = i; y[i] = 0;
  }
  int rank = N;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (k < rank) {
      int pr = k, pc = k;
      double best = 0;
#pragma unroll
      for (int j = k; j < N; ++j)
#pragma unroll
        for (int i = k; i < N; ++i) {
          const double a = fabs(A[i][j]);
          if (a > best) { best = a; pr = i; pc = j; }
        }
      if (best == 0) {
        rank = k;
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {   // rows k <-> pr
          const double t = A[k][j];
          double src = t;
#pragma unroll
          for (int r = k + 1; r < N; ++r) { if (pr == r) { src = A[r][j]; A[r][j] = t; } }
          A[k][j] = src;
        }
        {
          const double t = b[k];
          double src = t;
#pragma unroll
          for (int r = k + 1; r < N; ++r) { if (pr == r) { src = b[r]; b[r] = t; } }
          b[k] = src;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {   // columns k <-> pc
          const double t = A[i][k];
          double src = t;
#pragma unroll
          for (int cc = k + 1; cc < N; ++cc) { if (pc == cc) { src = A[i][cc]; A[i][cc] = t; } }
          A[i][k] = src;
        }
        {
          const int t = perm[k];
          int src = t;
#pragma unroll
          for (int cc = k + 1; cc < N; ++cc) { if (pc == cc) { src = perm[cc]; perm[cc] = t; } }
          perm[k] = src;
        }
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
          const double f = A[i][k] / A[k][k];
          A[i][k] = 0;
#pragma unroll
          for (int j = k + 1; j < N; ++j) A[i][j] -= f * A[k][j];
          b[i] -= f * b[k];
        }
      }
    }
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    if (i < rank) {
      double sv = b[i];
#pragma unroll
      for (int j = i + 1; j < N; ++j) if (j < rank) sv -= A[i][j] * y[j];
      y[i] = sv / A[i][i];
    }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) {
    double v = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) if (perm[i] == q) v = y[i];
    x[q] = v;
  }
}
#endif
// |cv::Rodrigues(R)| : rotation angle (WorldMap::toOrientationRodrigues(...).norm())
VS_HD double rotation_angle(const double* T) {
  const double rx = T[9] - T[6], ry = T[2] - T[8], rz = T[4] - T[1];
  const double s = sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
  double c = ((T[0] + T[5]) + T[10] - 1) * 0.5;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  if (s < 1e-5) return c > 0 ? 0.0 : 3.14159265358979323846;
  return acos(c);
}
// C++ double -> int32 conversion (truncation), guarded against the undefined out-of-range case
VS_HD bool to_int32(double v, int32_t* out) {
  if (!(v > -2147483648.0 && v < 2147483648.0)) return false;
  *out = (int32_t)v;
  return true;
}
__device__ __forceinline__ int hamming32(const uint32_t* a, const uint32_t* b) {
  int d = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) d += __popc(a[k] ^ b[k]);
  return d;
}
