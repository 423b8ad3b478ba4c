// kernels_orb.h — OrbDetector components (first half): cv::ORB used as a detector
// (base_framepoint_generator.cpp:52-70).  gfx950, wave64.  OpenCV is not part of the reference tree; the algorithms
// are its published ones [recalled]: imgproc resize (INTER_LINEAR, 8UC1), features2d orb.cpp computeKeyPoints /
// HarrisResponses / ICAngles, KeyPointsFilter::retainBest, core fastAtan2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- cv::resize INTER_LINEAR, 8-bit: 11-bit fixed-point weights, 32-bit horizontal pass, vertical pass
//      ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  One thread per destination pixel: the four
//      source bytes of neighbouring threads share cache lines (the scale is 1.2 in the pyramid).
__device__ __forceinline__ void resize_coeff(int d, double scale, int n_src, int* ofs, int* w0, int* w1, bool horizontal) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (horizontal) {
    if (s < 0) { f = 0; s = 0; }
    if (s >= n_src - 1) { *ofs = n_src - 1; *w0 = 2048; *w1 = 0; return; }   // replicated last column, full weight
  }
  const int a0 = (int)rintf((1.f - f) * 2048.f), a1 = (int)rintf(f * 2048.f);     // saturate_cast<short>: cvRound
  *ofs = s; *w0 = min(max(a0, -32768), 32767); *w1 = min(max(a1, -32768), 32767);
}
__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* src, int rows, int cols, int stride, uint8_t* dst, int drows,
                                                          int dcols, int dstride) {
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= dcols) return;
  int sx, a0, a1, sy, b0, b1;
  resize_coeff(dx, (double)cols / dcols, cols, &sx, &a0, &a1, true);
  resize_coeff(dy, (double)rows / drows, rows, &sy, &b0, &b1, false);
  const int y0 = min(max(sy, 0), rows - 1), y1 = min(max(sy + 1, 0), rows - 1);
  const int x1 = min(sx + 1, cols - 1);   // weight 0 when sx is the last column
  const uint8_t* r0 = src + (size_t)y0 * stride;
  const uint8_t* r1 = src + (size_t)y1 * stride;
  const int h0 = r0[sx] * a0 + r0[x1] * a1, h1 = r1[sx] * a0 + r1[x1] * a1;
  dst[(size_t)dy * dstride + dx] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
}

// ---- KeyPointsFilter::retainBest(n): keep every keypoint whose response is >= the n-th largest (ties included), in input
//      order.  One workgroup: 4-pass MSB radix select of the n-th largest key (256-bin LDS histograms), then an ordered
//      compaction.  Responses come as u8 FAST scores or as floats; both map to order-preserving 32-bit keys.
__device__ __forceinline__ uint32_t orb_key(float r) {
  uint32_t b = __float_as_uint(r);
  if (b == 0x80000000u) b = 0;                       // -0 == +0
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
template <typename R>
__global__ __launch_bounds__(1024) void k_orb_select(const int32_t* n_in_p, const int16_t* xy_in, const R* resp_in, int n_keep,
                                                     int32_t* n_out_p, int16_t* xy_out, float* resp_out, int cap_out) {
  __shared__ int hist[256];
  __shared__ int sh[17];
  __shared__ uint32_t s_prefix;
  __shared__ int s_rank;
  const int tid = threadIdx.x;
  const int n = *n_in_p;
  uint32_t thr_key = 0;                               // keep everything
  if (n_keep <= 0) thr_key = 0xffffffffu;             // keep nothing (keys never reach it: NaN-free responses)
  else if (n_keep < n && sizeof(R) == 1) {
    // FAST scores: one histogram over the 256 values
    for (int i = tid; i < 256; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[(int)resp_in[i] & 255], 1);
    __syncthreads();
    if (tid == 0) {
      int rank = n_keep, bin = 255;
      for (; bin > 0; --bin) { if (hist[bin] >= rank) break; rank -= hist[bin]; }
      s_prefix = orb_key((float)bin);
    }
    __syncthreads();
    thr_key = s_prefix;
  } else if (n_keep < n) {
    if (tid == 0) { s_prefix = 0; s_rank = n_keep; }  // looking for the s_rank-th largest among keys matching the prefix
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = tid; i < 256; i += 1024) hist[i] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix, mask = pass ? (0xffffffffu << (shift + 8)) : 0u;
      for (int i = tid; i < n; i += 1024) {
        const uint32_t k = orb_key((float)resp_in[i]);
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int rank = s_rank, bin = 255;
        for (; bin > 0; --bin) { if (hist[bin] >= rank) break; rank -= hist[bin]; }
        s_prefix = prefix | ((uint32_t)bin << shift);
        s_rank = rank;
      }
      __syncthreads();
    }
    thr_key = s_prefix;
  }
  int done = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    float r = 0;
    int keep = 0;
    if (i < n) { r = (float)resp_in[i]; keep = (n_keep > 0 && orb_key(r) >= thr_key) ? 1 : 0; }
    int total;
    const int at = done + block_exclusive_scan(keep, sh, &total);
    done += total;
    if (keep && at < cap_out) { xy_out[2 * at] = xy_in[2 * i]; xy_out[2 * at + 1] = xy_in[2 * i + 1]; resp_out[at] = r; }
  }
  if (tid == 0) *n_out_p = min(done, cap_out);
}

// ---- HarrisResponses (block 7, k = 0.04): one wavefront per keypoint, lane = block position, integer sums by wave
//      reduction (exact), the float formula on lane 0 exactly as written upstream.
__global__ __launch_bounds__(256) void k_orb_harris(const uint8_t* img, int stride, const int32_t* n_p, const int16_t* xy, float* resp) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int n = *n_p;
  for (int i = wave; i < n; i += nwaves) {
    int a = 0, b = 0, c = 0;
    if (lane < 49) {
      const int by = lane / 7, bx = lane - 7 * by;
      const uint8_t* q = img + (size_t)(xy[2 * i + 1] - 3 + by) * stride + (xy[2 * i] - 3 + bx);
      const int Ix = (q[1] - q[-1]) * 2 + (q[-stride + 1] - q[-stride - 1]) + (q[stride + 1] - q[stride - 1]);
      const int Iy = (q[stride] - q[-stride]) * 2 + (q[stride - 1] - q[-stride - 1]) + (q[stride + 1] - q[-stride + 1]);
      a = Ix * Ix; b = Iy * Iy; c = Ix * Iy;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
    if (lane == 0) {
      const float scale = 1.f / ((1 << 2) * 7 * 255.f);
      const float scale_sq_sq = scale * scale * scale * scale;
      resp[i] = ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }
  }
}

// core fastAtan2 (float, degrees)
__device__ __forceinline__ float fast_atan2f_deg(float y, float x) {
  const float s = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) { c = ay / (ax + (float)2.220446049250313e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  else { c = ax / (ay + (float)2.220446049250313e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- ICAngles: intensity centroid over the circular patch (rows v = -half..half, |u| <= umax[|v|]); lane = column u,
//      integer moments by wave reduction.  out == nullptr: angles only (stand-alone component); otherwise the keypoints of
//      this level are appended to the detector's output (x, y, size, angle, response, octave), scaled back to level 0.
struct OrbUmax { int v[34]; };
__global__ __launch_bounds__(256) void k_orb_angle(const uint8_t* img, int stride, const int32_t* n_p, const int16_t* xy, const float* resp,
                                                   int half, const OrbUmax um, float* angle, float* out, const int32_t* total_p, int cap,
                                                   float scale, int level, int patch) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int n = *n_p;
  const int base = total_p ? *total_p : 0;   // keypoints of the lower levels (advanced by k_orb_advance after this launch)
  for (int i = wave; i < n; i += nwaves) {
    const int x0 = xy[2 * i], y0 = xy[2 * i + 1];
    const int u = lane - half;
    int m01 = 0, m10 = 0;
    if (lane <= 2 * half) {
      const uint8_t* center = img + (size_t)y0 * stride + x0;
      m10 = u * center[u];
      for (int v = 1; v <= half; ++v) {
        if (abs(u) > um.v[v]) continue;
        const int vp = center[u + v * stride], vm = center[u - v * stride];
        m01 += v * (vp - vm);
        m10 += u * (vp + vm);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_xor(m01, o, 64); m10 += __shfl_xor(m10, o, 64); }
    if (lane == 0) {
      const float ang = fast_atan2f_deg((float)m01, (float)m10);
      if (angle) angle[i] = ang;
      if (out && base + i < cap) {
        float* o = out + 6 * (size_t)(base + i);
        o[0] = level ? (float)x0 * scale : (float)x0; o[1] = level ? (float)y0 * scale : (float)y0;
        o[2] = (float)patch * scale; o[3] = ang; o[4] = resp[i]; o[5] = (float)level;
      }
    }
  }
}
__global__ void k_orb_advance(int32_t* total_p, const int32_t* n_p) { *total_p += *n_p; }
