// kernels_frame2.h — second half of the frame kernel: prune, recovery, landmark refinement,
// stereo sweep + binning, and the PoseTracker3D control flow that strings the stages together.
#pragma once
#include "kernels_frame.h"
#include "kernels_report.h"
#include <type_traits>

// write one framepoint (Frame::createFramepoint, types/frame.cpp:61-84) from a left/right feature pair
__device__ __forceinline__ void materialize_point(const DevCfg& c, const DevBuf& b, int s, const PtView& cv, int j, int fl,
                                                  int fr, int dist, int epi, int prev, int tlen) {
  const int16_t* kxyL = kpxy_of(c, b, s, 0);
  const int16_t* kxyR = kpxy_of(c, b, s, 1);
  const int xL = kxyL[2 * fl], yL = kxyL[2 * fl + 1], xR = kxyR[2 * fr], yR = kxyR[2 * fr + 1];
  cv.kp[4 * (size_t)j] = (int16_t)xL; cv.kp[4 * (size_t)j + 1] = (int16_t)yL;
  cv.kp[4 * (size_t)j + 2] = (int16_t)xR; cv.kp[4 * (size_t)j + 3] = (int16_t)yR;
  const uint32_t* dl = reinterpret_cast<const uint32_t*>(desc_of(c, b, s, 0) + (size_t)32 * fl);
  const uint32_t* dr = reinterpret_cast<const uint32_t*>(desc_of(c, b, s, 1) + (size_t)32 * fr);
  uint32_t* o = reinterpret_cast<uint32_t*>(cv.desc + (size_t)64 * j);
  for (int k = 0; k < 8; ++k) { o[k] = dl[k]; o[8 + k] = dr[k]; }
  int32_t* m = cv.meta + (size_t)j * META;
  m[M_DIST] = dist; m[M_EPI] = epi; m[M_PREV] = prev; m[M_TLEN] = tlen; m[M_LMUP] = 0; m[M_NEXT] = 0;
  triangulate(c, xL, yL, xR, yR, cv.cam + 3 * (size_t)j);
  for (int k = 0; k < 3; ++k) { cv.camlm[3 * (size_t)j + k] = 0; cv.lm[3 * (size_t)j + k] = 0; }
}

// _prunePoints (pose_tracker_3d.cpp:437-472) fused with the materialisation of the surviving tracked
// points into the current frame's point arrays.  Quirk B.3: aligner not run on these points -> drop all.
__device__ __forceinline__ void wg_prune(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_prev, int pb_cur, bool aligner_valid) {
  const int tid = threadIdx.x;
  const int n = sh.n_trk;
  const PtView pv = pts_of(c, b, s, pb_prev);
  const PtView cv = pts_of(c, b, s, pb_cur);
  const int32_t* trk = b.trk + (size_t)s * c.MAXP * 4;
  const double* chi = b.al_chi + (size_t)s * c.MAXP;
  const uint8_t* inl = b.al_inl + (size_t)s * c.MAXP;
  const int16_t* kxyL = kpxy_of(c, b, s, 0);
  const int16_t* kxyR = kpxy_of(c, b, s, 1);
  const bool by_inlier = aligner_valid && (sh.E / (double)n < c.c.aligner_maximum_error_kernel);
  const int per = (n + VS_WG - 1) / VS_WG;
  const int u0 = tid * per, u1 = min(u0 + per, n);
  int cnt = 0;
  for (int u = u0; u < u1; ++u) {
    bool keep = false;
    if (aligner_valid) keep = by_inlier ? (inl[u] != 0) : (chi[u] != -1 && chi[u] < 100 * c.c.aligner_maximum_error_kernel);
    if (keep) ++cnt;
  }
  int total;
  int off = block_exclusive_scan(cnt, sh.scan, &total);
  for (int u = u0; u < u1; ++u) {
    bool keep = false;
    if (aligner_valid) keep = by_inlier ? (inl[u] != 0) : (chi[u] != -1 && chi[u] < 100 * c.c.aligner_maximum_error_kernel);
    const int ip = trk[4 * u];
    if (keep) {
      const int fl = trk[4 * u + 1], fr = trk[4 * u + 2];
      materialize_point(c, b, s, cv, off, fl, fr, trk[4 * u + 3], kxyR[2 * fr + 1] - kxyL[2 * fl + 1], ip,
                        pv.meta[(size_t)ip * META + M_TLEN] + 1);
      // the landmark travels with the track (origin()->landmark())
      cv.meta[(size_t)off * META + M_LMUP] = pv.meta[(size_t)ip * META + M_LMUP];
      for (int k = 0; k < 3; ++k) cv.lm[3 * (size_t)off + k] = pv.lm[3 * (size_t)ip + k];
      ++off;
    } else {
      pv.meta[(size_t)ip * META + M_NEXT] = 0;  // FramePoint::clear unlinks previous->next
    }
  }
  if (tid == 0) sh.n_cur = min(total, c.MAXP);
  __syncthreads();
}

// recoverPoints (stereo_framepoint_generator.cpp:683-869) in three steps:
//   project : one thread per lost point: landmark -> both image planes, depth and border gates (:704-764)
//   brief   : one wavefront per surviving point: BRIEF at both projections from the box images, the three
//             descriptor gates and the disparity gate (:773-842).  Runs inside the workgroup (stage path) or as
//             the wide kernel k_recover_brief over all streams (fused path).
//   append  : survivors are appended in lost-list order (:844-864)
// rec[6q] : 0 = rejected, 2 = projected (needs BRIEF), 1 = recovered; then xL, yL, xR, yR, Hamming L-R
// `list` (LDS, optional): compact work list of the projected points for the in-workgroup BRIEF step — 6 ints per entry
// (q, previous point, xL, yL, xR, yR), count in *n_list — so that step does not chase rec[] through HBM point by point.
__device__ __forceinline__ void wg_recover_project(const DevCfg& c, const DevBuf& b, int s, int n_lost, int pb_prev, const double* w2c,
                                                   int32_t* list = nullptr, int list_cap = 0, int* n_list = nullptr) {
  const PtView pv = pts_of(c, b, s, pb_prev);
  const int32_t* lost = b.lost + (size_t)s * c.MAXP;
  int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
  for (int q = threadIdx.x; q < n_lost; q += blockDim.x) {
    const int ip = lost[q];
    int ok = pv.meta[(size_t)ip * META + M_LMUP] > 0 ? 2 : 0;
    int xL = 0, yL = 0, xR = 0, yR = 0;
    if (ok) {
      double pc[3], uL[3], uR[3];
      tf_apply(w2c, pv.lm + 3 * (size_t)ip, pc);
      mat3_mul_vec(c.c.K, pc, uL);
      for (int k = 0; k < 3; ++k) uR[k] = uL[k] + c.c.baseline_h[k];
      if (uL[2] < c.c.minimum_depth_meters || uL[2] > c.c.maximum_depth_meters || uR[2] < c.c.minimum_depth_meters ||
          uR[2] > c.c.maximum_depth_meters) ok = 0;
      if (ok) {
        const float pLx = (float)rint(uL[0] / uL[2]), pLy = (float)rint(uL[1] / uL[2]);
        const float pRx = (float)rint(uR[0] / uR[2]), pRy = (float)rint(uR[1] / uR[2]);
        const float border = 35.f;  // 5 * keypoint.size (FAST: 7)
        if (pLx < border + 1 || pLx > c.c.cols - border - 1 || pRx < border + 1 || pRx > c.c.cols - border - 1 ||
            pLy < border + 1 || pLy > c.c.rows - border - 1 || pRy < border + 1 || pRy > c.c.rows - border - 1) ok = 0;
        xL = (int)pLx; yL = (int)pLy; xR = (int)pRx; yR = (int)pRy;
      }
    }
    rec[6 * q] = ok; rec[6 * q + 1] = xL; rec[6 * q + 2] = yL; rec[6 * q + 3] = xR; rec[6 * q + 4] = yR; rec[6 * q + 5] = 0;
    if (ok && list) {
      const int k = atomicAdd(n_list, 1);
      if (k < list_cap) { int32_t* e = list + 6 * k; e[0] = q; e[1] = ip; e[2] = xL; e[3] = yL; e[4] = xR; e[5] = yR; }
    }
  }
}

__device__ __forceinline__ void recover_brief_wave(const DevCfg& c, const DevBuf& b, int s, int pb_prev, int q, int lane,
                                                   double tau_track, double tau_tri) {
  int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
  if (rec[6 * q] != 2) return;   // wave-uniform
  const PtView pv = pts_of(c, b, s, pb_prev);
  const int ip = (b.lost + (size_t)s * c.MAXP)[q];
  const int xL = rec[6 * q + 1], yL = rec[6 * q + 2], xR = rec[6 * q + 3], yR = rec[6 * q + 4];
  const uint16_t* boxL = box_of(c, b, s, 0);
  const uint16_t* boxR = box_of(c, b, s, 1);
  // both descriptors in (uniform) registers: the 16 box gathers of a lane are issued together, then 8 ballots
  int aL[4], bL[4], aR[4], bR[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = j * 64 + lane;
    aL[j] = boxL[(size_t)(yL + c_brief[i][0]) * c.bstride + (xL + c_brief[i][1])];
    bL[j] = boxL[(size_t)(yL + c_brief[i][2]) * c.bstride + (xL + c_brief[i][3])];
    aR[j] = boxR[(size_t)(yR + c_brief[i][0]) * c.bstride + (xR + c_brief[i][1])];
    bR[j] = boxR[(size_t)(yR + c_brief[i][2]) * c.bstride + (xR + c_brief[i][3])];
  }
  unsigned long long dL[4], dR[4];
  const unsigned long long* pd = reinterpret_cast<const unsigned long long*>(pv.desc + (size_t)64 * ip);
  int hL = 0, hR = 0, dist = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dL[j] = __builtin_bswap64(__brevll(__ballot(aL[j] < bL[j])));
    dR[j] = __builtin_bswap64(__brevll(__ballot(aR[j] < bR[j])));
    hL += __popcll(dL[j] ^ pd[j]);
    hR += __popcll(dR[j] ^ pd[4 + j]);
    dist += __popcll(dL[j] ^ dR[j]);
  }
  int ok = 1;
  if ((double)hL > tau_track) ok = 0;
  if (ok && (double)((float)xL - (float)xR) < c.c.minimum_disparity_pixels) ok = 0;
  if (ok && (double)hR > tau_track) ok = 0;
  if (ok && (double)dist > tau_tri) ok = 0;
  if (lane == 0) {
    rec[6 * q] = ok; rec[6 * q + 5] = dist;
    if (ok) {
      unsigned long long* dl = reinterpret_cast<unsigned long long*>(b.rec_desc + ((size_t)s * c.MAXP + q) * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dl[j] = dL[j]; dl[4 + j] = dR[j]; }
    }
  }
}

// recoverPoints with the ORB extractor (descriptor_type 1): the steered tests of both projections straight from the
// Gaussian-blurred images (the extractor runs on the 71 x 71 region around the projection upstream,
// stereo_framepoint_generator.cpp:773-812; the pattern stays >= 14 px inside it, so the region's own border handling never
// reaches a tap).  Gates as in recover_brief_wave.
__device__ __forceinline__ void recover_orb_wave(const DevCfg& c, const DevBuf& b, int s, int pb_prev, int q, int ip, int xL, int yL, int xR, int yR,
                                                 int lane, double tau_track, double tau_tri, const OrbTaps& taps) {
  int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
  const PtView pv = pts_of(c, b, s, pb_prev);
  unsigned long long dL[4], dR[4];
  orb_wave(blur_of(c, b, s, 0) + (size_t)yL * c.bstride + xL, taps, dL);
  orb_wave(blur_of(c, b, s, 1) + (size_t)yR * c.bstride + xR, taps, dR);
  const unsigned long long* pd = reinterpret_cast<const unsigned long long*>(pv.desc + (size_t)64 * ip);
  int hL = 0, hR = 0, dist = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { hL += __popcll(dL[j] ^ pd[j]); hR += __popcll(dR[j] ^ pd[4 + j]); dist += __popcll(dL[j] ^ dR[j]); }
  int ok = 1;
  if ((double)hL > tau_track) ok = 0;
  if (ok && (double)((float)xL - (float)xR) < c.c.minimum_disparity_pixels) ok = 0;
  if (ok && (double)hR > tau_track) ok = 0;
  if (ok && (double)dist > tau_tri) ok = 0;
  if (lane == 0) {
    rec[6 * q] = ok; rec[6 * q + 5] = dist;
    if (ok) {
      unsigned long long* dl = reinterpret_cast<unsigned long long*>(b.rec_desc + ((size_t)s * c.MAXP + q) * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dl[j] = dL[j]; dl[4 + j] = dR[j]; }
    }
  }
}

// Same computation with the two 49 x 49 box patches staged in LDS by coalesced 16-byte row loads (7 lanes per row):
// the 1024 scattered 2-byte gathers per point of recover_brief_wave keep the CU's texture-address unit busy for ~1000
// cycles; 12 wide loads take a fraction of that.  `patch` = this wavefront's LDS area, VS_RPATCH bytes.
#define VS_RP_W 56
#define VS_RP_H (2 * VSLAM_BRIEF_PATCH_HALF + 1)
#define VS_RPATCH (2 * VS_RP_H * VS_RP_W * 2)
__device__ __forceinline__ void recover_brief_patch(const DevCfg& c, const DevBuf& b, int s, int pb_prev, int q, int ip, int xL, int yL,
                                                    int xR, int yR, int lane, double tau_track, double tau_tri, uint16_t* patch) {
  int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
  const PtView pv = pts_of(c, b, s, pb_prev);
  const int xy[2][2] = {{xL, yL}, {xR, yR}};
  constexpr int NLD = (VS_RP_H * 7 + 63) / 64;   // 6
  uint4 v[2][NLD];
  int cx[2];
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    const int col0 = (xy[sd][0] - VSLAM_BRIEF_PATCH_HALF) & ~7;   // 16-byte aligned; the patch ends at col0 + 55 at most
    cx[sd] = xy[sd][0] - col0;
    const uint16_t* base = box_of(c, b, s, sd) + (size_t)(xy[sd][1] - VSLAM_BRIEF_PATCH_HALF) * c.bstride + col0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int t = lane + 64 * u, row = t / 7, seg = t - 7 * row;
      v[sd][u] = make_uint4(0u, 0u, 0u, 0u);
      if (row < VS_RP_H) v[sd][u] = *reinterpret_cast<const uint4*>(base + (size_t)row * c.bstride + 8 * seg);
    }
  }
  unsigned long long pd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) pd[j] = reinterpret_cast<const unsigned long long*>(pv.desc + (size_t)64 * ip)[j];
#pragma unroll
  for (int sd = 0; sd < 2; ++sd)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int t = lane + 64 * u, row = t / 7, seg = t - 7 * row;
      if (row < VS_RP_H) *reinterpret_cast<uint4*>(patch + (sd * VS_RP_H + row) * VS_RP_W + 8 * seg) = v[sd][u];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  unsigned long long dL[4], dR[4];
  int hL = 0, hR = 0, dist = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = j * 64 + lane;
    const int oa = (VSLAM_BRIEF_PATCH_HALF + c_brief[i][0]) * VS_RP_W + c_brief[i][1], ob = (VSLAM_BRIEF_PATCH_HALF + c_brief[i][2]) * VS_RP_W + c_brief[i][3];
    const uint16_t* pl = patch + cx[0];
    const uint16_t* pr = patch + VS_RP_H * VS_RP_W + cx[1];
    dL[j] = __builtin_bswap64(__brevll(__ballot(pl[oa] < pl[ob])));
    dR[j] = __builtin_bswap64(__brevll(__ballot(pr[oa] < pr[ob])));
    hL += __popcll(dL[j] ^ pd[j]);
    hR += __popcll(dR[j] ^ pd[4 + j]);
    dist += __popcll(dL[j] ^ dR[j]);
  }
  __builtin_amdgcn_wave_barrier();
  int ok = 1;
  if ((double)hL > tau_track) ok = 0;
  if (ok && (double)((float)xy[0][0] - (float)xy[1][0]) < c.c.minimum_disparity_pixels) ok = 0;
  if (ok && (double)hR > tau_track) ok = 0;
  if (ok && (double)dist > tau_tri) ok = 0;
  if (lane == 0) {
    rec[6 * q] = ok; rec[6 * q + 5] = dist;
    if (ok) {
      unsigned long long* dl = reinterpret_cast<unsigned long long*>(b.rec_desc + ((size_t)s * c.MAXP + q) * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dl[j] = dL[j]; dl[4 + j] = dR[j]; }
    }
  }
}

// in-workgroup BRIEF step: the LDS work list first, then (list overflow only) the remaining points through rec[]
#define VS_RLIST_OFF ((VS_WG / 64) * VS_RPATCH)
// builds whose LDS arena is too small for the patches (co-scheduling experiments: a frame workgroup that fits into the hole one
// image-kernel workgroup leaves) gather the 2 x 512 taps of a point straight from the box images instead, like k_recover_brief
#define VS_RPATCH_IN_LDS (VS_RLIST_OFF + 24 * 64 <= VS_ARENA)
#define VS_RLIST_CAP (VS_RPATCH_IN_LDS ? (VS_ARENA - VS_RLIST_OFF) / 24 : 0)
__device__ __forceinline__ void wg_recover_brief(const DevCfg& c, const DevBuf& b, int s, int pb_prev, int n_lost, int n_list,
                                                 double tau_track, double tau_tri, unsigned char* arena) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if constexpr (!VS_RPATCH_IN_LDS) {
    if (c.c.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
      const OrbTaps taps = orb_taps(lane, c.orb_cos, c.orb_sin, c.bstride);
      const int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
      const int32_t* lost = b.lost + (size_t)s * c.MAXP;
      for (int q = w; q < n_lost; q += VS_WG / 64) {
        if (rec[6 * q] != 2) continue;   // wave-uniform
        recover_orb_wave(c, b, s, pb_prev, q, lost[q], rec[6 * q + 1], rec[6 * q + 2], rec[6 * q + 3], rec[6 * q + 4], lane, tau_track, tau_tri, taps);
      }
    } else {
      for (int q = w; q < n_lost; q += VS_WG / 64) recover_brief_wave(c, b, s, pb_prev, q, lane, tau_track, tau_tri);
    }
    return;
  }
  uint16_t* patch = reinterpret_cast<uint16_t*>(arena + (size_t)w * VS_RPATCH);
  const int32_t* list = reinterpret_cast<const int32_t*>(arena + VS_RLIST_OFF);
  if (c.c.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
    const OrbTaps taps = orb_taps(lane, c.orb_cos, c.orb_sin, c.bstride);
    const int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
    const int32_t* lost = b.lost + (size_t)s * c.MAXP;
    for (int q = w; q < n_lost; q += VS_WG / 64) {
      if (rec[6 * q] != 2) continue;   // wave-uniform
      recover_orb_wave(c, b, s, pb_prev, q, lost[q], rec[6 * q + 1], rec[6 * q + 2], rec[6 * q + 3], rec[6 * q + 4], lane, tau_track, tau_tri, taps);
    }
    return;
  }
  if (n_list <= VS_RLIST_CAP) {
    for (int k = w; k < n_list; k += VS_WG / 64) {
      const int32_t* e = list + 6 * k;
      recover_brief_patch(c, b, s, pb_prev, e[0], e[1], e[2], e[3], e[4], e[5], lane, tau_track, tau_tri, patch);
    }
  } else {
    const int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
    const int32_t* lost = b.lost + (size_t)s * c.MAXP;
    for (int q = w; q < n_lost; q += VS_WG / 64) {
      if (rec[6 * q] != 2) continue;   // wave-uniform
      recover_brief_patch(c, b, s, pb_prev, q, lost[q], rec[6 * q + 1], rec[6 * q + 2], rec[6 * q + 3], rec[6 * q + 4], lane, tau_track, tau_tri, patch);
    }
  }
}

template <int NT, class SH>
__device__ __forceinline__ void wg_recover_append_t(const DevCfg& c, const DevBuf& b, int s, SH& sh, int pb_prev, int pb_cur) {
  const int tid = threadIdx.x;
  const PtView pv = pts_of(c, b, s, pb_prev);
  const PtView cv = pts_of(c, b, s, pb_cur);
  const int32_t* lost = b.lost + (size_t)s * c.MAXP;
  const int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
  const uint8_t* rdesc = b.rec_desc + (size_t)s * c.MAXP * 64;
  const int nl = sh.n_lost;
  const int per = (nl + NT - 1) / NT;
  const int q0 = tid * per, q1 = min(q0 + per, nl);
  int cnt = 0;
  for (int q = q0; q < q1; ++q) cnt += rec[6 * q] == 1 ? 1 : 0;
  int total;
  int off = sh.n_cur + block_exclusive_scan(cnt, sh.scan, &total);
  for (int q = q0; q < q1; ++q) {
    if (rec[6 * q] != 1) continue;
    if (off < c.MAXP) {
      const int ip = lost[q], j = off;
      const int xL = rec[6 * q + 1], yL = rec[6 * q + 2], xR = rec[6 * q + 3], yR = rec[6 * q + 4];
      cv.kp[4 * (size_t)j] = (int16_t)xL; cv.kp[4 * (size_t)j + 1] = (int16_t)yL; cv.kp[4 * (size_t)j + 2] = (int16_t)xR; cv.kp[4 * (size_t)j + 3] = (int16_t)yR;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(rdesc + (size_t)64 * q);
      uint32_t* dst = reinterpret_cast<uint32_t*>(cv.desc + (size_t)64 * j);
      for (int k = 0; k < 16; ++k) dst[k] = src[k];
      int32_t* m = cv.meta + (size_t)j * META;
      m[M_DIST] = rec[6 * q + 5]; m[M_EPI] = 0; m[M_PREV] = ip; m[M_TLEN] = pv.meta[(size_t)ip * META + M_TLEN] + 1;
      m[M_LMUP] = pv.meta[(size_t)ip * META + M_LMUP]; m[M_NEXT] = 0;
      triangulate(c, xL, yL, xR, yR, cv.cam + 3 * (size_t)j);
      for (int k = 0; k < 3; ++k) { cv.lm[3 * (size_t)j + k] = pv.lm[3 * (size_t)ip + k]; cv.camlm[3 * (size_t)j + k] = 0; }
      pv.meta[(size_t)ip * META + M_NEXT] = 1;
    } else {
      atomicOr(&b.st[s].error_flags, 2);
    }
    ++off;
  }
  __syncthreads();
  if (tid == 0) { sh.flag = total; sh.n_cur = min(sh.n_cur + total, c.MAXP); }
  __syncthreads();
}

__device__ __forceinline__ void wg_recover_append(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_prev, int pb_cur) {
  wg_recover_append_t<VS_WG, FrameShared>(c, b, s, sh, pb_prev, pb_cur);
}

// whole recovery inside one workgroup (stage path)
__device__ __forceinline__ void wg_recover(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_prev, int pb_cur, const double* w2c,
                           double tau_track, double tau_tri, unsigned char* arena) {
  if (threadIdx.x == 0) sh.n_proj = 0;
  __syncthreads();
  wg_recover_project(c, b, s, sh.n_lost, pb_prev, w2c, reinterpret_cast<int32_t*>(arena + VS_RLIST_OFF), VS_RLIST_CAP, &sh.n_proj);
  __syncthreads();
  wg_recover_brief(c, b, s, pb_prev, sh.n_lost, sh.n_proj, tau_track, tau_tri, arena);
  __syncthreads();
  wg_recover_append(c, b, s, sh, pb_prev, pb_cur);
}

// fused path: BRIEF of the projected lost points of ALL streams, one wavefront each
__global__ __launch_bounds__(256) void k_recover_brief(const DevCfg c, const DevBuf b) {
  int bx, sy;
  xcd_stream_block(&bx, &sy, b.xcd_rot);
  const int s = b.s0 + sy;
  if (!vs_active(b, s)) return;
  const StreamState& st = b.st[s];
  const int lane = threadIdx.x & 63;
  const int wave = bx * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int nl = st.fc.n_lost;
  if (c.c.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
    const OrbTaps taps = orb_taps(lane, c.orb_cos, c.orb_sin, c.bstride);
    const int32_t* rec = b.rec + (size_t)s * c.MAXP * 6;
    const int32_t* lost = b.lost + (size_t)s * c.MAXP;
    for (int q = wave; q < nl; q += nwaves) {
      if (rec[6 * q] != 2) continue;   // wave-uniform
      recover_orb_wave(c, b, s, st.cur, q, lost[q], rec[6 * q + 1], rec[6 * q + 2], rec[6 * q + 3], rec[6 * q + 4], lane, st.fc.tau_gen, st.fc.tau_tri, taps);
    }
    return;
  }
  for (int q = wave; q < nl; q += nwaves) recover_brief_wave(c, b, s, st.cur, q, lane, st.fc.tau_gen, st.fc.tau_tri);
}

// _updatePoints (pose_tracker_3d.cpp:475-520): one thread per framepoint; landmark creation = mean of the
// track's world coordinates (landmark.cpp:19-31), update = Gauss-Newton over all measurements of the
// track (landmark.cpp:66-167).  Measurements are reached by walking the per-frame `prev` links of the
// history ring (frame f, index i) -> (f-1, prev[i]).
// landmark of framepoint i of the current frame: creation = mean of the track's world coordinates
// (Landmark::Landmark, landmark.cpp:19-31), otherwise Gauss-Newton refinement over all measurements of the track
// (Landmark::update, :66-167).  Returns true when the point carries an active landmark afterwards.
// LDS variant (fused frame kernel): the poses of the last VS_LM_CN frames are staged once per workgroup and the first
// VS_LM_CN measurements of the point's track once per point (the chain walk is a chase of dependent HBM loads, and the
// Gauss-Newton rounds would repeat it); longer tracks continue in HBM from where the cache ends.  Same order of accumulation,
// same bits.
#ifndef VS_LM_CN
#define VS_LM_CN 6
#endif
#define VS_LM_NP 48   // world_to_camera of the last VS_LM_NP frames staged in LDS (one copy for all points of the frame)
template <int NT, int CN>
struct LmCacheT { static constexpr int kCN = CN; static constexpr int kBatch = NT >= 512 ? 4 : 2; double w2c[VS_LM_NP][12]; double rtr[VS_LM_NP][9]; double cam[NT][CN][4]; };
// R^T R of world_to_camera k (J^T J of every measurement taken in that frame: a property of the frame, symmetric to the bit) next to the staged poses
template <class LC>
__device__ __forceinline__ void lm_stage_rtr(LC* lc, int f, int hcap, int nthreads) {
  for (int t = threadIdx.x; t < VS_LM_NP * 9; t += nthreads) {
    const int k = t / 9, e = t - 9 * k, rr = e / 3, cc = e - 3 * rr;
    if (f - k >= 0 && k < hcap) { const double* W = lc->w2c[k]; lc->rtr[k][e] = (W[rr] * W[cc] + W[4 + rr] * W[4 + cc]) + W[8 + rr] * W[8 + cc]; }
  }
}
typedef LmCacheT<VS_WG, VS_LM_CN> LmCache;
template <bool LDS, class LC = LmCache>
__device__ __forceinline__ bool landmark_point_t(const DevCfg& c, const DevBuf& b, int s, const PtView& cv, int f, int i, LC* lc) {
  constexpr int VS_LM_CN_ = LC::kCN;
  const double* w2c_cur = hpose_of(c, b, s, f) + 12;
  {
    int32_t* m = cv.meta + (size_t)i * META;
    const int tlen = m[M_TLEN];
    if (tlen < c.c.minimum_track_length_for_landmark_creation) return false;
    int len = tlen + 1;
    if (len > c.HCAP) { len = c.HCAP; atomicOr(&b.st[s].error_flags, 4); }
    double wpos[3];
    if (m[M_LMUP] == 0) {
      // Landmark::Landmark: average of the world coordinates along the track
      double acc[3] = {0, 0, 0};
      int ff = f, ii = i;
      for (int k = 0; k < len; ++k) {
        double wp[3];
        tf_apply(hpose_of(c, b, s, ff), hcam_of(c, b, s, ff) + 4 * (size_t)ii, wp);
        for (int q = 0; q < 3; ++q) acc[q] += wp[q];
        ii = hprev_of(c, b, s, ff)[ii];
        --ff;
        if (ii < 0 && k + 1 < len) { len = k + 1; break; }
      }
      for (int q = 0; q < 3; ++q) wpos[q] = acc[q] / (double)len;
      m[M_LMUP] = len;
    } else {
      // Landmark::update
      double wv[3] = {cv.lm[3 * (size_t)i], cv.lm[3 * (size_t)i + 1], cv.lm[3 * (size_t)i + 2]};
      for (int q = 0; q < 3; ++q) wpos[q] = wv[q];
      double err_prev = 0;
      const double kern = c.c.landmark_maximum_error_squared_meters;
      // one measurement of the track: residual, saturated kernel, H += R^T om R, b += R^T om e
      // mc = x, y, z, 1 / z of the measurement (the history ring keeps the inverse depth: one division per measurement, not one per round);
      // RtR = the staged R^T R of the measurement's frame, or null (computed here)
      auto accumulate = [&](const double* W, const double* RtR, const double* mc, double* H, double* bv, double& err, int& n_out) {
        double sp[3];
        tf_apply(W, wv, sp);
        if (sp[2] <= 0) {
          ++n_out;
        } else {
          const double e[3] = {sp[0] - mc[0], sp[1] - mc[1], sp[2] - mc[2]};
          double om = mc[3];
          const double e2 = om * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
          err += e2;
          if (e2 > kern) { om *= kern / e2; ++n_out; }
          if (RtR) {
            const double h01 = om * RtR[1], h02 = om * RtR[2], h12 = om * RtR[5];
            H[0] += om * RtR[0]; H[4] += om * RtR[4]; H[8] += om * RtR[8];
            H[1] += h01; H[3] += h01; H[2] += h02; H[6] += h02; H[5] += h12; H[7] += h12;
          } else {
            for (int r = 0; r < 3; ++r)
              for (int cc = 0; cc < 3; ++cc) H[3 * r + cc] += om * ((W[r] * W[cc] + W[4 + r] * W[4 + cc]) + W[8 + r] * W[8 + cc]);
          }
          for (int r = 0; r < 3; ++r) bv[r] += om * ((W[r] * e[0] + W[4 + r] * e[1]) + W[8 + r] * e[2]);
        }
      };
      // Measurement k of the track (frame f - k): index i for k = 0, the trail's entry k - 1 up to k = VS_TRAIL, then the
      // `prev` links of the history ring.  n_direct = measurements addressed without a link walk; a 0xFFFF entry (the track
      // starts there) ends the list like a negative `prev` link does.
      const uint16_t* tr = cv.trail + (size_t)i * VS_TRAIL;
      int n_direct = 1;
      bool ended = false;
      if (c.trail) {
        const int want = min(len, VS_TRAIL + 1);
        for (int q0 = 0; q0 < VS_TRAIL && n_direct < want && !ended; q0 += 8) {
          const uint4 v = *reinterpret_cast<const uint4*>(tr + q0);
          const uint32_t wv4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const uint32_t ent = (wv4[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
            if (!ended && n_direct < want) { if (ent == 0xFFFFu) ended = true; else ++n_direct; }
          }
        }
      }
      auto index_at = [&](int k) -> int { return k == 0 ? i : (int)tr[k - 1]; };   // k < n_direct
      // the first VS_LM_CN_ measurements into the thread's LDS slots, once
      int ncache = 0, ffc = f, iic = i;
      if constexpr (LDS) {
        double (*slot)[4] = lc->cam[threadIdx.x];
        if (c.trail) {
          const int nc = min(min(len, VS_LM_CN_), n_direct);
          double mv[VS_LM_CN_][4];
#pragma unroll
          for (int k = 0; k < VS_LM_CN_; ++k)
            if (k < nc) { const double* mc = hcam_of(c, b, s, f - k) + 4 * (size_t)index_at(k); mv[k][0] = mc[0]; mv[k][1] = mc[1]; mv[k][2] = mc[2]; mv[k][3] = mc[3]; }
#pragma unroll
          for (int k = 0; k < VS_LM_CN_; ++k)
            if (k < nc) { slot[k][0] = mv[k][0]; slot[k][1] = mv[k][1]; slot[k][2] = mv[k][2]; slot[k][3] = mv[k][3]; }
          ncache = nc;
        } else {
          for (int k = 0; k < len && k < VS_LM_CN_; ++k) {
            const double* mc = hcam_of(c, b, s, ffc) + 4 * (size_t)iic;
            slot[k][0] = mc[0]; slot[k][1] = mc[1]; slot[k][2] = mc[2]; slot[k][3] = mc[3];
            ++ncache;
            iic = hprev_of(c, b, s, ffc)[iic];
            --ffc;
            if (iic < 0) { ended = true; break; }
          }
        }
      }
      if (c.trail) {
        // where the link walk continues after the directly addressed measurements (only tracks longer than VS_TRAIL + 1)
        if (!ended && n_direct < len) { ffc = f - (n_direct - 1); iic = hprev_of(c, b, s, ffc)[index_at(n_direct - 1)]; --ffc; if (iic < 0) ended = true; }
        else ended = true;
      }
      for (int it = 0; it < c.c.landmark_maximum_number_of_iterations; ++it) {
        double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0};
        double err = 0;
        int n_out = 0;
        if constexpr (LDS) {
          const double (*slot)[4] = lc->cam[threadIdx.x];
          for (int k = 0; k < ncache; ++k) accumulate(lc->w2c[k], lc->rtr[k], slot[k], H, bv, err, n_out);   // frame f - k
        }
        if (c.trail) {
          // directly addressed measurements, four at a time: their (independent) loads are in flight together
          constexpr int NB = LC::kBatch;     // loads in flight together (fewer in the small-register tail kernel)
          for (int k0 = ncache; k0 < n_direct; k0 += NB) {
            double mc[NB][4];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
              const int k = min(k0 + u, n_direct - 1);
              const double* src = hcam_of(c, b, s, f - k) + 4 * (size_t)index_at(k);
              mc[u][0] = src[0]; mc[u][1] = src[1]; mc[u][2] = src[2]; mc[u][3] = src[3];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
              const int k = k0 + u;
              if (k < n_direct) {
                if constexpr (LDS) {
                  if (k < VS_LM_NP) accumulate(lc->w2c[k], lc->rtr[k], mc[u], H, bv, err, n_out);
                  else accumulate(hpose_of(c, b, s, f - k) + 12, nullptr, mc[u], H, bv, err, n_out);
                } else {
                  accumulate(hpose_of(c, b, s, f - k) + 12, nullptr, mc[u], H, bv, err, n_out);
                }
              }
            }
          }
        }
        if (!ended) {
          int ff = ffc, ii = iic;
          for (int k = c.trail ? n_direct : ncache; k < len; ++k) {
            accumulate(hpose_of(c, b, s, ff) + 12, nullptr, hcam_of(c, b, s, ff) + 4 * (size_t)ii, H, bv, err, n_out);
            ii = hprev_of(c, b, s, ff)[ii];
            --ff;
            if (ii < 0) break;
          }
        }
        double nb[3] = {-bv[0], -bv[1], -bv[2]}, dx[3];
        full_piv_solve_regs<3>(H, nb, dx);
        for (int q = 0; q < 3; ++q) wv[q] += dx[q];
        if (fabs(err - err_prev) < 1e-5 || it == 999) {
          const int n_in = len - n_out;
          if (n_in > m[M_LMUP]) {
            for (int q = 0; q < 3; ++q) wpos[q] = wv[q];
            m[M_LMUP] = n_in;
          } else if (n_in < n_out) {
            double acc[3] = {0, 0, 0};
            int f2 = f, i2 = i;
            for (int k = 0; k < len; ++k) {
              double wp[3];
              tf_apply(hpose_of(c, b, s, f2), hcam_of(c, b, s, f2) + 4 * (size_t)i2, wp);
              for (int q = 0; q < 3; ++q) acc[q] += wp[q];
              i2 = hprev_of(c, b, s, f2)[i2];
              --f2;
              if (i2 < 0) break;
            }
            for (int q = 0; q < 3; ++q) wpos[q] = acc[q] / (double)len;
          }
          break;
        }
        err_prev = err;
      }
    }
    for (int q = 0; q < 3; ++q) cv.lm[3 * (size_t)i + q] = wpos[q];
    tf_apply(w2c_cur, wpos, cv.camlm + 3 * (size_t)i);
  }
  return true;
}

__device__ __forceinline__ bool landmark_point(const DevCfg& c, const DevBuf& b, int s, const PtView& cv, int f, int i) {
  return landmark_point_t<false, LmCache>(c, b, s, cv, f, i, (LmCache*)nullptr);
}

// Landmark::update of a LONG track by a team of eight lanes (fused frame kernel).  One lane per landmark walks a chain of ~65
// dependent fp64 operations per measurement and round; the longest track of the frame (dozens of measurements, three rounds)
// kept the phase waiting for one wavefront.  Only the thirteen ADDITIONS into H, b and the error have to happen in the list's
// order: the eight lanes evaluate eight consecutive measurements at once (projection, residual, kernel, om * R^T R, om * R^T e),
// park the terms in LDS, and every lane adds the eight terms in list order into its own copy of the sums — the same operations on
// the same operands in the same order as landmark_point_t's serial loop, an eighth of the multiplications on the critical path.
// Measurements beyond the trail (k >= n_direct: only reachable through the `prev` links) follow serially on every lane alike.
#ifndef VS_LM_TEAMS
#define VS_LM_TEAMS 1
#endif
#define VS_LM_TEAM_G 8          // lanes per team
#ifndef VS_LM_TEAM_WAVES
#define VS_LM_TEAM_WAVES 2      // wavefronts of the workgroup that run teams when the frame has long tracks (16 teams at a time)
#endif
#ifndef VS_LM_TEAM_MIN
#define VS_LM_TEAM_MIN 9        // measurements from which a track goes to a team
#endif
struct LmTerm { double e2, h[6], b[3]; int kind, pad; };   // kind 0: behind the camera (outlier, nothing added), 1: inlier, 2: outlier with the saturated kernel
#define VS_LM_TEAM_LDS (VS_LM_TEAM_WAVES * (64 / VS_LM_TEAM_G) * VS_LM_TEAM_G * (int)sizeof(LmTerm))
// classification used by the work lists: an update (not a creation) of a track with VS_LM_TEAM_MIN or more measurements
__device__ __forceinline__ bool landmark_is_long(const DevCfg& c, const int32_t* m) {
  return m[M_LMUP] != 0 && min(m[M_TLEN] + 1, c.HCAP) >= VS_LM_TEAM_MIN && c.trail;
}
__device__ __forceinline__ bool landmark_team(const DevCfg& c, const DevBuf& b, int s, const PtView& cv, int f, int i, LmCache* lc, LmTerm* terms, int gl) {
  const double* w2c_cur = hpose_of(c, b, s, f) + 12;
  int32_t* m = cv.meta + (size_t)i * META;
  const int tlen = m[M_TLEN];
  const int lmup0 = m[M_LMUP];
  int len = tlen + 1;
  if (len > c.HCAP) { len = c.HCAP; if (gl == 0) atomicOr(&b.st[s].error_flags, 4); }
  double wpos[3];
  double wv[3] = {cv.lm[3 * (size_t)i], cv.lm[3 * (size_t)i + 1], cv.lm[3 * (size_t)i + 2]};
  for (int q = 0; q < 3; ++q) wpos[q] = wv[q];
  const double kern = c.c.landmark_maximum_error_squared_meters;
  // directly addressed measurements (landmark_point_t's n_direct)
  const uint16_t* tr = cv.trail + (size_t)i * VS_TRAIL;
  int n_direct = 1;
  bool ended = false;
  {
    const int want = min(len, VS_TRAIL + 1);
    for (int q0 = 0; q0 < VS_TRAIL && n_direct < want && !ended; q0 += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(tr + q0);
      const uint32_t wv4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t ent = (wv4[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
        if (!ended && n_direct < want) { if (ent == 0xFFFFu) ended = true; else ++n_direct; }
      }
    }
  }
  // this lane's measurements of the directly addressed part (k = gl, gl + 8, ...) into its LDS slots, all loads in flight, once
  constexpr int NG = LmCache::kCN;                       // groups whose measurements have a slot (k < 8 * NG)
  double (*slot)[4] = lc->cam[threadIdx.x];
  {
    double mv[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = VS_LM_TEAM_G * g + gl;
      if (k < n_direct) {
        const double* mc = hcam_of(c, b, s, f - k) + 4 * (size_t)(k == 0 ? i : (int)tr[k - 1]);
        mv[g][0] = mc[0]; mv[g][1] = mc[1]; mv[g][2] = mc[2]; mv[g][3] = mc[3];
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (VS_LM_TEAM_G * g + gl < n_direct) { slot[g][0] = mv[g][0]; slot[g][1] = mv[g][1]; slot[g][2] = mv[g][2]; slot[g][3] = mv[g][3]; }
  }
  // where the link walk continues after the directly addressed measurements
  int ffc = f, iic = i;
  if (!ended && n_direct < len) { ffc = f - (n_direct - 1); iic = hprev_of(c, b, s, ffc)[n_direct == 1 ? i : (int)tr[n_direct - 2]]; --ffc; if (iic < 0) ended = true; }
  else ended = true;
  double err_prev = 0;
  for (int it = 0; it < c.c.landmark_maximum_number_of_iterations; ++it) {
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0};
    double err = 0;
    int n_out = 0;
    for (int k0 = 0; k0 < n_direct; k0 += VS_LM_TEAM_G) {
      const int k = k0 + gl;
      LmTerm t;
      t.kind = -1;
      if (k < n_direct) {
        double mc[4];
        const int g = k0 / VS_LM_TEAM_G;
        if (g < NG) { mc[0] = slot[g][0]; mc[1] = slot[g][1]; mc[2] = slot[g][2]; mc[3] = slot[g][3]; }
        else { const double* src = hcam_of(c, b, s, f - k) + 4 * (size_t)tr[k - 1]; mc[0] = src[0]; mc[1] = src[1]; mc[2] = src[2]; mc[3] = src[3]; }
        const double* W;
        double rtr_far[9];
        const double* RtR;
        if (k < VS_LM_NP) { W = lc->w2c[k]; RtR = lc->rtr[k]; }
        else {
          W = hpose_of(c, b, s, f - k) + 12;
          for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) rtr_far[3 * r + cc] = (W[r] * W[cc] + W[4 + r] * W[4 + cc]) + W[8 + r] * W[8 + cc];
          RtR = rtr_far;
        }
        double sp[3];
        tf_apply(W, wv, sp);
        if (sp[2] <= 0) { t.kind = 0; }
        else {
          const double e[3] = {sp[0] - mc[0], sp[1] - mc[1], sp[2] - mc[2]};
          double om = mc[3];
          t.e2 = om * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
          t.kind = 1;
          if (t.e2 > kern) { om *= kern / t.e2; t.kind = 2; }
          t.h[0] = om * RtR[0]; t.h[1] = om * RtR[1]; t.h[2] = om * RtR[2]; t.h[3] = om * RtR[4]; t.h[4] = om * RtR[5]; t.h[5] = om * RtR[8];
          for (int r = 0; r < 3; ++r) t.b[r] = om * ((W[r] * e[0] + W[4 + r] * e[1]) + W[8 + r] * e[2]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();     // the previous batch's terms have been read by every lane of the team
      terms[gl] = t;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int nb_ = min(VS_LM_TEAM_G, n_direct - k0);
#pragma unroll
      for (int u = 0; u < VS_LM_TEAM_G; ++u) {
        if (u < nb_) {
          const LmTerm q = terms[u];
          if (q.kind == 0) { ++n_out; }
          else {
            err += q.e2;
            if (q.kind == 2) ++n_out;
            H[0] += q.h[0]; H[4] += q.h[3]; H[8] += q.h[5];
            { const double h01 = q.h[1], h02 = q.h[2], h12 = q.h[4]; H[1] += h01; H[3] += h01; H[2] += h02; H[6] += h02; H[5] += h12; H[7] += h12; }
            bv[0] += q.b[0]; bv[1] += q.b[1]; bv[2] += q.b[2];
          }
        }
      }
    }
    if (!ended) {
      // beyond the trail: the serial loop of landmark_point_t, on every lane of the team alike
      int ff = ffc, ii = iic;
      for (int k = n_direct; k < len; ++k) {
        const double* W = hpose_of(c, b, s, ff) + 12;
        const double* mc = hcam_of(c, b, s, ff) + 4 * (size_t)ii;
        double sp[3];
        tf_apply(W, wv, sp);
        if (sp[2] <= 0) {
          ++n_out;
        } else {
          const double e[3] = {sp[0] - mc[0], sp[1] - mc[1], sp[2] - mc[2]};
          double om = mc[3];
          const double e2 = om * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
          err += e2;
          if (e2 > kern) { om *= kern / e2; ++n_out; }
          for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) H[3 * r + cc] += om * ((W[r] * W[cc] + W[4 + r] * W[4 + cc]) + W[8 + r] * W[8 + cc]);
          for (int r = 0; r < 3; ++r) bv[r] += om * ((W[r] * e[0] + W[4 + r] * e[1]) + W[8 + r] * e[2]);
        }
        ii = hprev_of(c, b, s, ff)[ii];
        --ff;
        if (ii < 0) break;
      }
    }
    double nb[3] = {-bv[0], -bv[1], -bv[2]}, dx[3];
    full_piv_solve_regs<3>(H, nb, dx);
    for (int q = 0; q < 3; ++q) wv[q] += dx[q];
    if (fabs(err - err_prev) < 1e-5 || it == 999) {
      const int n_in = len - n_out;
      if (n_in > lmup0) {
        for (int q = 0; q < 3; ++q) wpos[q] = wv[q];
        if (gl == 0) m[M_LMUP] = n_in;
      } else if (n_in < n_out) {
        double acc[3] = {0, 0, 0};
        int f2 = f, i2 = i;
        for (int k = 0; k < len; ++k) {
          double wp[3];
          tf_apply(hpose_of(c, b, s, f2), hcam_of(c, b, s, f2) + 4 * (size_t)i2, wp);
          for (int q = 0; q < 3; ++q) acc[q] += wp[q];
          i2 = hprev_of(c, b, s, f2)[i2];
          --f2;
          if (i2 < 0) break;
        }
        for (int q = 0; q < 3; ++q) wpos[q] = acc[q] / (double)len;
      }
      break;
    }
    err_prev = err;
  }
  if (gl == 0) {
    for (int q = 0; q < 3; ++q) cv.lm[3 * (size_t)i + q] = wpos[q];
    tf_apply(w2c_cur, wpos, cv.camlm + 3 * (size_t)i);
  }
  return true;
}

// Besides the history ring (camera coordinates and `prev` link of every point of frame f), every point gets its trail: the
// indices of its track's points in frames f-1 .. f-VS_TRAIL (its predecessor, then the predecessor's own trail shifted by one;
// 0xFFFF where the track starts before that).  The landmark refinement then addresses its measurements directly instead of
// walking the links, a chain of dependent HBM loads per measurement.
__device__ __forceinline__ void wg_publish_history(const DevCfg& c, const DevBuf& b, int s, int n, int pb_cur, int f) {
  const PtView cv = pts_of(c, b, s, pb_cur);
  const PtView pv = pts_of(c, b, s, pb_cur ^ 1);
  double* hc = hcam_of(c, b, s, f);
  int32_t* hp = hprev_of(c, b, s, f);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    { const double x = cv.cam[3 * (size_t)i], y = cv.cam[3 * (size_t)i + 1], z = cv.cam[3 * (size_t)i + 2];
      reinterpret_cast<double2*>(hc + 4 * (size_t)i)[0] = make_double2(x, y); reinterpret_cast<double2*>(hc + 4 * (size_t)i)[1] = make_double2(z, 1 / z); }   // Measurement::inverse_depth_meters
    const int ip = cv.meta[(size_t)i * META + M_PREV];
    hp[i] = ip;
    if (c.trail) {
      uint4* dst = reinterpret_cast<uint4*>(cv.trail + (size_t)i * VS_TRAIL);
      if (ip < 0) {
        dst[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // only entry 0 is ever reached: len = 1
      } else {
        const uint4* src = reinterpret_cast<const uint4*>(pv.trail + (size_t)ip * VS_TRAIL);
        uint32_t w[VS_TRAIL / 2];
#pragma unroll
        for (int q = 0; q < VS_TRAIL / 8; ++q) { const uint4 v = src[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
        uint32_t o[VS_TRAIL / 2];
        o[0] = (uint32_t)ip | (w[0] << 16);
#pragma unroll
        for (int q = 1; q < VS_TRAIL / 2; ++q) o[q] = (w[q - 1] >> 16) | (w[q] << 16);
#pragma unroll
        for (int q = 0; q < VS_TRAIL / 8; ++q) dst[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
      }
    }
  }
}

// Landmark creation / refinement of the frame's points inside the stream's workgroup (PoseTracker3D::_updatePoints' landmark part): the poses of the
// last VS_LM_NP frames and every lane's first measurements staged in the LDS arena, short tracks one lane each, long tracks a team of eight lanes
// (landmark_point_t<true> / landmark_team).  Returns the number of active landmarks (block-uniform).  Used by k_frame's fused launches and by the
// stage path's UPDATE / COMPUTE stages; history of frame f must have been published (wg_publish_history).
__device__ __forceinline__ int wg_landmarks_lds(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_cur, int f, unsigned char* arena) {
  const int tid = threadIdx.x;
  const PtView cvu = pts_of(c, b, s, pb_cur);
  int active = 0;
  static_assert(sizeof(LmCache) <= VS_ARENA, "landmark measurement cache must fit the LDS arena");
  LmCache* lc = reinterpret_cast<LmCache*>(arena);
  for (int t = tid; t < VS_LM_NP * 12; t += VS_WG) { const int k = t / 12; if (f - k >= 0 && k < c.HCAP) lc->w2c[k][t - 12 * k] = hpose_of(c, b, s, f - k)[12 + t - 12 * k]; }
  __syncthreads();
  lm_stage_rtr(lc, f, c.HCAP, VS_WG);
  // The points that carry a landmark (track long enough: creation or refinement) are compacted into a work list first: ~40 % of
  // the frame's points, one per thread in a single round instead of two half-empty ones (a thread's refinement is a serial chain).
#if VS_LM_TEAMS
  static_assert(sizeof(LmCache) + VS_LM_TEAM_LDS + 4096 <= VS_ARENA, "landmark cache + team terms must leave room for the work lists");
  constexpr int LIST_CAP = (VS_ARENA - (int)sizeof(LmCache) - VS_LM_TEAM_LDS) / 2;
  LmTerm* team_terms = reinterpret_cast<LmTerm*>(arena + VS_ARENA - VS_LM_TEAM_LDS);
#else
  constexpr int LIST_CAP = (VS_ARENA - (int)sizeof(LmCache)) / 2;
#endif
  uint16_t* work = reinterpret_cast<uint16_t*>(arena + sizeof(LmCache));
  const bool listed = sh.n_cur <= LIST_CAP && sh.n_cur <= 65535;
  if (tid == 0) { sh.flag = 0; sh.n_proj = 0; }       // n_proj (recovery is over): the count of long tracks
  __syncthreads();
  if (listed) {
    // short tracks (one lane each) from the front of the list, long ones (a team of eight lanes each) from its end
    for (int i0 = 0; i0 < sh.n_cur; i0 += VS_WG) {
      const int i = i0 + tid;
      const int32_t* mi = cvu.meta + (size_t)min(i, sh.n_cur - 1) * META;
      const bool need = i < sh.n_cur && mi[M_TLEN] >= c.c.minimum_track_length_for_landmark_creation;
#if VS_LM_TEAMS
      const bool lng = need && landmark_is_long(c, mi);
#else
      const bool lng = false;
#endif
      const unsigned long long m = __ballot(need && !lng), ml = __ballot(lng);
      int base = 0, basel = 0;
      if ((tid & 63) == 0 && m) base = atomicAdd(&sh.flag, __popcll(m));
      if ((tid & 63) == 0 && ml) basel = atomicAdd(&sh.n_proj, __popcll(ml));
      base = __builtin_amdgcn_readfirstlane(base); basel = __builtin_amdgcn_readfirstlane(basel);
      if (need && !lng) work[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
      if (lng) work[LIST_CAP - 1 - (basel + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u)))] = (uint16_t)i;
    }
    __syncthreads();
    const int n_work = sh.flag;
#if VS_LM_TEAMS
    const int n_long = sh.n_proj;
    const int team_waves = min(VS_LM_TEAM_WAVES, (n_long + 64 / VS_LM_TEAM_G - 1) / (64 / VS_LM_TEAM_G));
    const int wv_ = tid >> 6;
    if (wv_ < team_waves) {
      const int team = tid / VS_LM_TEAM_G, gl = tid % VS_LM_TEAM_G;
      for (int q = team; q < n_long; q += team_waves * (64 / VS_LM_TEAM_G))
        active += (landmark_team(c, b, s, cvu, f, work[LIST_CAP - 1 - q], lc, team_terms + team * VS_LM_TEAM_G, gl) && gl == 0) ? 1 : 0;
    } else {
      for (int q = tid - 64 * team_waves; q < n_work; q += VS_WG - 64 * team_waves) active += landmark_point_t<true>(c, b, s, cvu, f, work[q], lc) ? 1 : 0;
    }
#else
    for (int q = tid; q < n_work; q += VS_WG) active += landmark_point_t<true>(c, b, s, cvu, f, work[q], lc) ? 1 : 0;
#endif
  } else {
    for (int i = tid; i < sh.n_cur; i += VS_WG) active += landmark_point_t<true>(c, b, s, cvu, f, i, lc) ? 1 : 0;
  }
  int total;
  block_exclusive_scan(active, sh.scan, &total);
  __syncthreads();
  if (tid == 0) sh.flag = 0;
  __syncthreads();
  return total;
}

__device__ __forceinline__ void wg_update_points(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_cur, int f, unsigned char* arena) {
  const int tid = threadIdx.x;
  // publish the current frame's cam/prev to the history ring first (chains start here)
  wg_publish_history(c, b, s, sh.n_cur, pb_cur, f);
  __syncthreads();
  const int total = wg_landmarks_lds(c, b, s, sh, pb_cur, f, arena);     // the fused launch's refinement (LDS-cached, teams): 31 us per KITTI-sized frame where one thread per track took 80
  if (tid == 0) sh.n_lm = total;  // _number_of_active_landmarks
  __syncthreads();
}

// fused path: one thread per framepoint of every stream
// count: 1 = the kernel also counts the frame's active landmarks into FrameCarry::n_active (it runs BETWEEN phase 1 and phase 2); 0 = phase 4 has
// counted them (a point is active iff its track is long enough: the refinement's outcome does not enter) and the kernel runs BESIDE phase 2
__global__ __launch_bounds__(256) void k_update_landmarks(const DevCfg c, const DevBuf b, int count) {
  const int s = b.s0 + blockIdx.y;
  if (!vs_active(b, s)) return;
  StreamState& st = b.st[s];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = st.fc.n_cur;
  if (blockIdx.x * blockDim.x >= n) return;
  const PtView cv = pts_of(c, b, s, st.fc.lm_pb);
  const bool act = i < n && landmark_point(c, b, s, cv, st.fc.lm_f, i);
  const int cnt = __popcll(__ballot(act));
  if (count && (threadIdx.x & 63) == 0 && cnt) atomicAdd(&st.fc.n_active, cnt);
}

// The same refinement spread over `gridDim.x` workgroups per stream, each with the frame workgroup's own machinery (poses of the last VS_LM_NP frames
// and every lane's first measurements in LDS, teams of eight lanes for long tracks: landmark_team / landmark_point_t<true>, i.e. the operations of
// k_frame's landmark phase in the same order): launch sequence 4 (k_tail_lm) and the stage path of a one-stream context (k_stage_lm) run it beside the
// frame's last phase / stage, inside the same launch, where the wide one-thread-per-track kernel above (56 us for one KITTI-sized stream) would be
// longer than what it hides behind.
// share g of G of stream s; n_short / n_long_sh / scan: workgroup-shared scratch (two counters, 17 ints); tick: the share-0 workgroup adds its duration to the
// stream's landmark chronometer (the callers that are not timed by HIP events)
__device__ __forceinline__ void lm_teams_body(const DevCfg& c, const DevBuf& b, int s, int g, int G, int count, bool tick, unsigned char* arena, int& n_short, int& n_long_sh, int* scan) {
  const int tid = threadIdx.x;
  const unsigned long long t_begin = wall_clock64();
  StreamState& st = b.st[s];
  const int n_cur = st.fc.n_cur, f = st.fc.lm_f;
  const PtView cvu = pts_of(c, b, s, st.fc.lm_pb);
  LmCache* lc = reinterpret_cast<LmCache*>(arena);
  for (int t = tid; t < VS_LM_NP * 12; t += VS_WG) { const int k = t / 12; if (f - k >= 0 && k < c.HCAP) lc->w2c[k][t - 12 * k] = hpose_of(c, b, s, f - k)[12 + t - 12 * k]; }
  if (tid == 0) { n_short = 0; n_long_sh = 0; }
  __syncthreads();
  lm_stage_rtr(lc, f, c.HCAP, VS_WG);
  int active = 0;
#if VS_LM_TEAMS
  constexpr int LIST_CAP = (VS_ARENA - (int)sizeof(LmCache) - VS_LM_TEAM_LDS) / 2;
  LmTerm* team_terms = reinterpret_cast<LmTerm*>(arena + VS_ARENA - VS_LM_TEAM_LDS);
#else
  constexpr int LIST_CAP = (VS_ARENA - (int)sizeof(LmCache)) / 2;
#endif
  uint16_t* work = reinterpret_cast<uint16_t*>(arena + sizeof(LmCache));
  const bool listed = n_cur <= LIST_CAP && n_cur <= 65535;
  __syncthreads();
  if (listed) {
    // a workgroup's share: the points i with i % G == g (the lists' order depends on the order of the atomics: a share must not be defined through it);
    // its two work lists: short tracks from the front, long ones from the end
    for (int i0 = 0; i0 < n_cur; i0 += VS_WG) {
      const int i = i0 + tid;
      const int32_t* mi = cvu.meta + (size_t)min(i, n_cur - 1) * META;
      const bool need = i < n_cur && (i % G) == g && mi[M_TLEN] >= c.c.minimum_track_length_for_landmark_creation;
#if VS_LM_TEAMS
      const bool lng = need && landmark_is_long(c, mi);
#else
      const bool lng = false;
#endif
      const unsigned long long m = __ballot(need && !lng), ml = __ballot(lng);
      int base = 0, basel = 0;
      if ((tid & 63) == 0 && m) base = atomicAdd(&n_short, __popcll(m));
      if ((tid & 63) == 0 && ml) basel = atomicAdd(&n_long_sh, __popcll(ml));
      base = __builtin_amdgcn_readfirstlane(base); basel = __builtin_amdgcn_readfirstlane(basel);
      if (need && !lng) work[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
      if (lng) work[LIST_CAP - 1 - (basel + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u)))] = (uint16_t)i;
    }
    __syncthreads();
    const int n_work = n_short;
#if VS_LM_TEAMS
    const int n_long = n_long_sh;
    constexpr int TEAMS = VS_LM_TEAM_WAVES * (64 / VS_LM_TEAM_G);      // teams of a workgroup
    if ((tid >> 6) < VS_LM_TEAM_WAVES) {
      const int team = tid / VS_LM_TEAM_G, gl = tid % VS_LM_TEAM_G;
      for (int q = team; q < n_long; q += TEAMS)
        active += (landmark_team(c, b, s, cvu, f, work[LIST_CAP - 1 - q], lc, team_terms + team * VS_LM_TEAM_G, gl) && gl == 0) ? 1 : 0;
    } else {
      constexpr int SH = VS_WG - 64 * VS_LM_TEAM_WAVES;
      for (int q = tid - 64 * VS_LM_TEAM_WAVES; q < n_work; q += SH) active += landmark_point_t<true>(c, b, s, cvu, f, work[q], lc) ? 1 : 0;
    }
#else
    for (int q = tid; q < n_work; q += VS_WG) active += landmark_point_t<true>(c, b, s, cvu, f, work[q], lc) ? 1 : 0;
#endif
  } else {
    for (int i = g * VS_WG + tid; i < n_cur; i += G * VS_WG) active += landmark_point_t<true>(c, b, s, cvu, f, i, lc) ? 1 : 0;
  }
  if (count) {
    int total;
    block_exclusive_scan(active, scan, &total);
    if (tid == 0 && total) atomicAdd(&st.fc.n_active, total);
  }
  if (tick && g == 0 && tid == 0) st.ticks[3] += wall_clock64() - t_begin;
}
// sdist[i][k], k < 16: Hamming distance of left feature i to right feature g0 + w0 + k of its row [g0, g1), where the
// window [w0, m) holds the (up to 16) nearest right features at or left of the left feature: m = number of right
// features of the row with x <= xl, w0 = max(m - 16, 0).  Nothing is written for m = 0 or m >= 255.
template <class XR>
__device__ __forceinline__ void stereo_dist_row(const uint8_t* descL, const uint8_t* descR, int i, int g0, int g1, int xl, XR xr,
                                                uint8_t* sdist) {
  int m;
  {  // m = right features of the row with x <= xl (x-sorted): binary search, 6 dependent loads instead of up to 255
    int lo = g0, hi = g1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (xl - xr(mid) >= 0) lo = mid + 1; else hi = mid; }
    m = min(lo - g0, 255);
  }
  if (m == 0 || m >= 255) return;
  const int w0 = max(m - 16, 0), mw = m - w0;
  const uint4 la = reinterpret_cast<const uint4*>(descL + (size_t)32 * i)[0], lb = reinterpret_cast<const uint4*>(descL + (size_t)32 * i)[1];
  uint32_t pk[4] = {0, 0, 0, 0};
  // four right descriptors in flight per step (the window is a contiguous index range)
  for (int k0 = 0; k0 < mw; k0 += 4) {
    uint4 ra[4], rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4* rp = reinterpret_cast<const uint4*>(descR + (size_t)32 * (g0 + w0 + min(k0 + u, mw - 1)));
      ra[u] = rp[0]; rb[u] = rp[1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int h = __popc(la.x ^ ra[u].x) + __popc(la.y ^ ra[u].y) + __popc(la.z ^ ra[u].z) + __popc(la.w ^ ra[u].w) +
                    __popc(lb.x ^ rb[u].x) + __popc(lb.y ^ rb[u].y) + __popc(lb.z ^ rb[u].z) + __popc(lb.w ^ rb[u].w);
      if (k0 + u < mw) pk[k0 >> 2] |= (uint32_t)(h > 255 ? 255 : h) << (8 * u);
    }
  }
  *reinterpret_cast<uint4*>(sdist + (size_t)i * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

// compute() (stereo_framepoint_generator.cpp:135-462): stereo sweep with one thread per image row (rows are
// independent: the right cursor only moves inside a row), then the order-dependent bin competition with
// one thread per bin, then emission in bin-grid row-major order.
// FULL: the arena can hold the whole staging / the 32-bit bin tables (the 512-thread frame kernel); false compiles those paths out
template <int NT, class SH, bool FULL = true>
__device__ __forceinline__ void wg_stereo_t(const DevCfg& c, const DevBuf& b, int s, SH& sh, int pb_cur, double tau_tri, int f,
                            unsigned char* arena, int arena_bytes) {
  const int tid = threadIdx.x;
  const PtView cv = pts_of(c, b, s, pb_cur);
  const int rows = c.c.rows, CW1 = c.CW + 1;
  const int nL = b.n_kp[s * 2];
  const int32_t* rcL = rowcell_of(c, b, s, 0);
  const int32_t* rcR = rowcell_of(c, b, s, 1);
  const int16_t* kxyL = kpxy_of(c, b, s, 0);
  const int16_t* kxyR = kpxy_of(c, b, s, 1);
  const uint8_t* descL = desc_of(c, b, s, 0);
  const uint8_t* descR = desc_of(c, b, s, 1);
  uint8_t* usedL = used_of(c, b, s, 0);
  uint8_t* usedR = used_of(c, b, s, 1);
  int32_t* match = b.st_match + (size_t)s * c.NMAX * 3;
  int32_t* sc = b.sc + (size_t)s * c.NMAX * 4;
  int32_t* bin_occ = b.bin_occ + (size_t)s * c.rows_bin * c.cols_bin;
  const int n_tracked = sh.n_cur;
  const int nR = b.n_kp[s * 2 + 1];
  // The sweep (:235-360) is sequential per image row only through the right cursor (a match at right feature g forbids
  // g and everything left of it to later left features of the row).  It runs in two steps on LDS copies of what it
  // touches:
  //  (A) every left feature, in parallel: over the window of its (up to 16) nearest right features at or left of it —
  //      whose descriptor distances k_stereo_dist precomputed — the first-minimum right feature for EVERY possible
  //      cursor position (a suffix-argmin table, 16 nibbles);
  //  (B) one thread per row replays the cursor with one table lookup per left feature.  Only a cursor left of the
  //      window (more than 16 unconsumed right features behind the left feature) needs the reference's explicit scan.
  const int nLp = (nL + 7) & ~7, nRp = (nR + 7) & ~7, rowsp = (rows + 8) & ~7;
  const size_t stage_bytes = (size_t)4 * 2 * rowsp + (size_t)(8 + 4 + 4 + 2 + 1 + 1) * nLp + (size_t)(2 + 1) * nRp;
  const bool staged = FULL && stage_bytes <= (size_t)arena_bytes;
  // the distance rows of the first pass (from k_stereo_dist of the image pipeline) ride along when they fit
  const bool sd_lds = staged && ((stage_bytes + 15) & ~(size_t)15) + (size_t)16 * nL <= (size_t)arena_bytes;
  unsigned long long* ssuf = reinterpret_cast<unsigned long long*>(arena);   // step A: suffix-argmin nibbles
  int32_t* srL = reinterpret_cast<int32_t*>(ssuf + nLp);     // row starts, left / right
  int32_t* srR = srL + rowsp;
  uint32_t* sxyL = reinterpret_cast<uint32_t*>(srR + rowsp); // x | y << 16
  int32_t* smatch = reinterpret_cast<int32_t*>(sxyL + nLp);  // sweep result per left feature: -1 or distance << 16 | right index
  uint16_t* sval = reinterpret_cast<uint16_t*>(smatch + nLp);// step A: bit c = a candidate exists at window position >= c
  int16_t* sxR = reinterpret_cast<int16_t*>(sval + nLp);
  uint8_t* suL = reinterpret_cast<uint8_t*>(sxR + nRp);      // used flags
  uint8_t* suR = suL + nLp;
  uint8_t* smL = suR + nRp;                                  // right features of the row at or left of the left feature (<= 255)
  uint4* sd4 = reinterpret_cast<uint4*>(arena + ((stage_bytes + 15) & ~(size_t)15));
  const int itau = (int)ceil(tau_tri);   // integer h < tau_tri  <=>  h < ceil(tau_tri)
  int n_cand = 0;
  VS_PHASE_BEGIN(tq);
#define DBG_STAMP(k) do { __syncthreads(); VS_PHASE_STAMP(k, tq); } while (0)
  for (int oi = 0; oi < c.n_offsets; ++oi) {
    const int o = c.offsets[oi];
    uint8_t* sdist = b.sdist + (size_t)s * c.NMAX * 16;
    if (staged) {
      for (int r = tid; r <= rows; r += NT) {
        srL[r] = r < rows ? rcL[(size_t)r * CW1] : rcL[(size_t)(rows - 1) * CW1 + c.CW];
        srR[r] = r < rows ? rcR[(size_t)r * CW1] : rcR[(size_t)(rows - 1) * CW1 + c.CW];
      }
#pragma unroll 4
      for (int i = tid; i < nL; i += NT) { sxyL[i] = reinterpret_cast<const uint32_t*>(kxyL)[i]; suL[i] = usedL[i]; }
#pragma unroll 4
      for (int g = tid; g < nR; g += NT) { sxR[g] = kxyR[2 * g]; suR[g] = usedR[g]; }
      if (oi == 0 && sd_lds) {
#pragma unroll 4
        for (int i = tid; i < nL; i += NT) sd4[i] = reinterpret_cast<const uint4*>(sdist)[i];
      }
      __syncthreads();
      // distances of the first pass came from k_stereo_dist (image pipeline); later offsets recompute them here
      if (oi > 0) {
        for (int i = tid; i < nL; i += NT) {
          if (suL[i]) continue;
          const int rr = (int)(sxyL[i] >> 16) - o;
          if (rr < 0 || rr >= rows) continue;
          stereo_dist_row(descL, descR, i, srR[rr], srR[rr + 1], (int)(sxyL[i] & 0xFFFFu), [&](int g) { return (int)sxR[g]; }, sdist);
        }
        __syncthreads();
      }
      // (instantiated twice: distance rows in LDS or in HBM — a run-time pointer select would turn every access into a
      // FLAT instruction, which waits on both the LDS and the HBM counters)
      auto steps_ab = [&](auto sd_tag) {
        constexpr bool SD = decltype(sd_tag)::value;
        // ---- step A ---------------------------------------------------------------------------------------------------
        for (int i = tid; i < nL; i += NT) {
          unsigned long long suf = 0;
          unsigned val = 0;
          int m = 0;
          const uint32_t xy = sxyL[i];
          const int rr = (int)(xy >> 16) - o;
          if (!suL[i] && rr >= 0 && rr < rows) {
            const int g0 = srR[rr], g1 = srR[rr + 1];
            const int xl = (int)(xy & 0xFFFFu);
            {  // m = right features of the row with x <= xl (x-sorted): binary search
              int lo = g0, hi = g1;
              while (lo < hi) { const int mid = (lo + hi) >> 1; if (xl - sxR[mid] >= 0) lo = mid + 1; else hi = mid; }
              m = min(lo - g0, 255);
            }
            if (m > 0 && m < 255) {
              const int w0 = max(m - 16, 0), mw = m - w0;
              uint4 dq;
              if constexpr (SD) dq = sd4[i]; else dq = *reinterpret_cast<const uint4*>(sdist + (size_t)i * 16);
              const unsigned long long d01 = ((unsigned long long)dq.y << 32) | dq.x, d23 = ((unsigned long long)dq.w << 32) | dq.z;
              int bh = 0, bj = -1;
              for (int k = mw - 1; k >= 0; --k) {
                const int h = (int)(((k < 8 ? d01 : d23) >> (8 * (k & 7))) & 255ull);
                if (!suR[g0 + w0 + k] && h < itau && (bj < 0 || h <= bh)) { bh = h; bj = k; }
                if (bj >= 0) { suf |= (unsigned long long)bj << (4 * k); val |= 1u << k; }
              }
            }
          }
          ssuf[i] = suf; sval[i] = (uint16_t)val; smL[i] = (uint8_t)m;
        }
        __syncthreads();
        // ---- step B ---------------------------------------------------------------------------------------------------
        for (int r = tid; r < rows; r += NT) {
          const int rr = r - o;  // right row: L.row == R.row + o
          const bool rv = rr >= 0 && rr < rows;
          const int l0 = srL[r], l1 = srL[r + 1];
          const int g0 = rv ? srR[rr] : 0, g1 = rv ? srR[rr + 1] : 0;
          int cur = g0;
          for (int i = l0; i < l1; ++i) {
            const int m = smL[i];
            int bg = -1, best = 0;
            if (m > 0 && cur < g0 + m) {
              const int w0 = max(m - 16, 0), cpos = cur - g0 - w0;
              if (m < 255 && cpos >= 0) {
                const unsigned val = sval[i];
                if ((val >> cpos) & 1u) {
                  const int k = (int)((ssuf[i] >> (4 * cpos)) & 15ull);
                  bg = g0 + w0 + k;
                  if constexpr (SD) best = reinterpret_cast<const uint8_t*>(sd4 + i)[k]; else best = sdist[(size_t)i * 16 + k];
                }
              } else {
                // cursor left of the window: the reference's explicit scan from the cursor
                const int xl = (int)(sxyL[i] & 0xFFFFu);
                uint32_t ld[8];
                for (int q = 0; q < 8; ++q) ld[q] = reinterpret_cast<const uint32_t*>(descL + (size_t)32 * i)[q];
                best = itau;
                for (int g = cur; g < g1; ++g) {
                  if (suR[g]) continue;
                  if (xl - sxR[g] < 0) break;
                  const int h = hamming32(ld, reinterpret_cast<const uint32_t*>(descR + (size_t)32 * g));
                  if (h < best) { best = h; bg = g; }
                }
              }
            }
            int res = -1;
            if (bg >= 0 && !((double)((int)(sxyL[i] & 0xFFFFu) - sxR[bg]) < c.c.minimum_disparity_pixels)) {
              res = (best << 16) | bg;
              cur = bg + 1;
            }
            smatch[i] = res;   // LDS: a global store here would put an HBM round trip into every step of the replay
          }
        }
      };
      if (oi == 0 && sd_lds) steps_ab(std::true_type{}); else steps_ab(std::false_type{});
    } else if (arena_bytes >= 8192) {
      // The staging does not fit the arena as a whole (small-LDS builds of the frame's tail kernel, very large feature counts): the
      // same two steps band by band.  Rows are independent, a band of rows [r0, r1) owns the contiguous left features
      // [l0, l1) and the right features [g0, g1) of the rows r - o, so a band stages only its own slices (local indices = global
      // index minus l0 / g0) and appends its matches before the next band starts: same results, same order.
      const int32_t nLend = rcL[(size_t)(rows - 1) * CW1 + c.CW], nRend = rcR[(size_t)(rows - 1) * CW1 + c.CW];
      auto rsL = [&](int r) { return r < rows ? rcL[(size_t)r * CW1] : nLend; };
      auto rsR = [&](int r) { return r <= 0 ? 0 : (r < rows ? rcR[(size_t)r * CW1] : nRend); };
      int rb = max(1, (int)((size_t)rows * (size_t)arena_bytes * 3 / (4 * stage_bytes)));   // first guess: average density, 25 % slack
      int r0 = 0;
      while (r0 < rows) {
        int r1, l0, l1, g0, g1;
        size_t need;
        for (;;) {   // shrink the band until its slices fit (one row always does: <= 255 usable right features, a few hundred left ones)
          r1 = min(r0 + rb, rows);
          l0 = rsL(r0); l1 = rsL(r1);
          g0 = rsR(min(max(r0 - o, 0), rows)); g1 = rsR(min(max(r1 - o, 0), rows));
          const int nLb = ((l1 - l0) + 7) & ~7, nRb = ((g1 - g0) + 7) & ~7, rbp = (r1 - r0 + 8) & ~7;
          need = (size_t)4 * 2 * rbp + (size_t)20 * nLb + (size_t)3 * nRb;
          if (need <= (size_t)arena_bytes || rb == 1) break;
          rb = max(1, rb / 2);
        }
        const int nLb = ((l1 - l0) + 7) & ~7, nRb = ((g1 - g0) + 7) & ~7, rbn = r1 - r0, rbp = (rbn + 8) & ~7;
        if (need > (size_t)arena_bytes) {   // a single row beyond the arena (cannot happen below ~600 features in one row): reference loop on HBM
          for (int i = l0 + tid; i < l1; i += NT) match[2 * i] = -1;
          __syncthreads();
          if (tid == 0) {
            const int rr = r0 - o;
            if (rr >= 0 && rr < rows) {
              int cur = g0;
              for (int i = l0; i < l1 && cur < g1; ++i) {
                if (usedL[i]) continue;
                const int xl = kxyL[2 * i];
                uint32_t ld[8];
                for (int q = 0; q < 8; ++q) ld[q] = reinterpret_cast<const uint32_t*>(descL + (size_t)32 * i)[q];
                int best = itau, bg = -1;
                for (int g = cur; g < g1; ++g) {
                  if (usedR[g]) continue;
                  if (xl - kxyR[2 * g] < 0) break;
                  const int h = hamming32(ld, reinterpret_cast<const uint32_t*>(descR + (size_t)32 * g));
                  if (h < best) { best = h; bg = g; }
                }
                if (bg >= 0 && !((double)(xl - kxyR[2 * bg]) < c.c.minimum_disparity_pixels)) { match[2 * i] = bg; match[2 * i + 1] = best; cur = bg + 1; }
              }
            }
          }
          __syncthreads();
          const int perb = (l1 - l0 + NT - 1) / NT;
          const int i0 = l0 + tid * perb, i1 = min(i0 + perb, l1);
          int cnt = 0;
          for (int i = i0; i < i1; ++i) cnt += match[2 * i] >= 0 ? 1 : 0;
          int total;
          int off = n_cand + block_exclusive_scan(cnt, sh.scan, &total);
          for (int i = i0; i < i1; ++i) {
            const int g = match[2 * i];
            if (g < 0) continue;
            sc[4 * off] = i; sc[4 * off + 1] = g; sc[4 * off + 2] = match[2 * i + 1]; sc[4 * off + 3] = o;
            usedL[i] = 1; usedR[g] = 1;
            ++off;
          }
          n_cand += total;
          r0 = r1;
          continue;
        }
        unsigned long long* bsuf = reinterpret_cast<unsigned long long*>(arena);
        int32_t* brL = reinterpret_cast<int32_t*>(bsuf + nLb);       // left row starts of rows r0 .. r1 (global indices)
        int32_t* brR = brL + rbp;                                     // right row starts of rows r0 - o .. r1 - o
        uint32_t* bxyL = reinterpret_cast<uint32_t*>(brR + rbp);
        int32_t* bmatch = reinterpret_cast<int32_t*>(bxyL + nLb);
        uint16_t* bval = reinterpret_cast<uint16_t*>(bmatch + nLb);
        int16_t* bxR = reinterpret_cast<int16_t*>(bval + nLb);
        uint8_t* buL = reinterpret_cast<uint8_t*>(bxR + nRb);
        uint8_t* buR = buL + nLb;
        uint8_t* bmL = buR + nRb;
        __syncthreads();   // the previous band's arrays are dead
        for (int q = tid; q <= rbn; q += NT) { brL[q] = rsL(r0 + q); brR[q] = rsR(min(max(r0 + q - o, 0), rows)); }
        for (int i = l0 + tid; i < l1; i += NT) { bxyL[i - l0] = reinterpret_cast<const uint32_t*>(kxyL)[i]; buL[i - l0] = usedL[i]; }
        for (int g = g0 + tid; g < g1; g += NT) { bxR[g - g0] = kxyR[2 * g]; buR[g - g0] = usedR[g]; }
        __syncthreads();
        if (oi > 0) {   // later offsets recompute their distance rows (the first pass came from k_stereo_dist)
          for (int i = l0 + tid; i < l1; i += NT) {
            if (buL[i - l0]) continue;
            const int q = (int)(bxyL[i - l0] >> 16) - r0, rr = r0 + q - o;
            if (rr < 0 || rr >= rows) continue;
            stereo_dist_row(descL, descR, i, brR[q], brR[q + 1], (int)(bxyL[i - l0] & 0xFFFFu), [&](int g) { return (int)bxR[g - g0]; }, sdist);
          }
          __syncthreads();
        }
        // ---- step A (band) ----
        for (int i = l0 + tid; i < l1; i += NT) {
          unsigned long long suf = 0;
          unsigned val = 0;
          int m = 0;
          const uint32_t xy = bxyL[i - l0];
          const int q = (int)(xy >> 16) - r0, rr = r0 + q - o;
          if (!buL[i - l0] && rr >= 0 && rr < rows) {
            const int h0 = brR[q], h1 = brR[q + 1];
            const int xl = (int)(xy & 0xFFFFu);
            { int lo = h0, hi = h1; while (lo < hi) { const int mid = (lo + hi) >> 1; if (xl - bxR[mid - g0] >= 0) lo = mid + 1; else hi = mid; } m = min(lo - h0, 255); }
            if (m > 0 && m < 255) {
              const int w0 = max(m - 16, 0), mw = m - w0;
              const uint4 dq = *reinterpret_cast<const uint4*>(sdist + (size_t)i * 16);
              const unsigned long long d01 = ((unsigned long long)dq.y << 32) | dq.x, d23 = ((unsigned long long)dq.w << 32) | dq.z;
              int bh = 0, bj = -1;
              for (int k = mw - 1; k >= 0; --k) {
                const int h = (int)(((k < 8 ? d01 : d23) >> (8 * (k & 7))) & 255ull);
                if (!buR[h0 + w0 + k - g0] && h < itau && (bj < 0 || h <= bh)) { bh = h; bj = k; }
                if (bj >= 0) { suf |= (unsigned long long)bj << (4 * k); val |= 1u << k; }
              }
            }
          }
          bsuf[i - l0] = suf; bval[i - l0] = (uint16_t)val; bmL[i - l0] = (uint8_t)m;
        }
        __syncthreads();
        // ---- step B (band): one thread per row ----
        for (int q = tid; q < rbn; q += NT) {
          const int rr = r0 + q - o;
          const bool rv = rr >= 0 && rr < rows;
          const int a0 = brL[q], a1 = brL[q + 1];
          const int h0 = rv ? brR[q] : 0, h1 = rv ? brR[q + 1] : 0;
          int cur = h0;
          for (int i = a0; i < a1; ++i) {
            const int m = bmL[i - l0];
            int bg = -1, best = 0;
            if (m > 0 && cur < h0 + m) {
              const int w0 = max(m - 16, 0), cpos = cur - h0 - w0;
              if (m < 255 && cpos >= 0) {
                const unsigned val = bval[i - l0];
                if ((val >> cpos) & 1u) {
                  const int k = (int)((bsuf[i - l0] >> (4 * cpos)) & 15ull);
                  bg = h0 + w0 + k;
                  best = 0x100 | k;     // bit 8: "distance = sdist[i][k]", fetched by the parallel append below (no HBM round trip in this loop)
                }
              } else {
                const int xl = (int)(bxyL[i - l0] & 0xFFFFu);
                uint32_t ld[8];
                for (int u = 0; u < 8; ++u) ld[u] = reinterpret_cast<const uint32_t*>(descL + (size_t)32 * i)[u];
                best = itau;
                for (int g = cur; g < h1; ++g) {
                  if (buR[g - g0]) continue;
                  if (xl - bxR[g - g0] < 0) break;
                  const int h = hamming32(ld, reinterpret_cast<const uint32_t*>(descR + (size_t)32 * g));
                  if (h < best) { best = h; bg = g; }
                }
              }
            }
            int res = -1;
            if (bg >= 0 && !((double)((int)(bxyL[i - l0] & 0xFFFFu) - bxR[bg - g0]) < c.c.minimum_disparity_pixels)) {
              res = (best << 16) | bg;
              cur = bg + 1;
            }
            bmatch[i - l0] = res;
          }
        }
        __syncthreads();
        // append the band's matches in sorted-left order
        {
          const int perb = (l1 - l0 + NT - 1) / NT;
          const int i0 = l0 + tid * perb, i1 = min(i0 + perb, l1);
          int cnt = 0;
          for (int i = i0; i < i1; ++i) cnt += bmatch[i - l0] >= 0 ? 1 : 0;
          int total;
          int off = n_cand + block_exclusive_scan(cnt, sh.scan, &total);
          for (int i = i0; i < i1; ++i) {
            const int sm = bmatch[i - l0];
            if (sm < 0) continue;
            const int g = sm & 0xFFFF;
            int dist = sm >> 16;
            if (dist & 0x100) dist = sdist[(size_t)i * 16 + (dist & 15)];
            sc[4 * off] = i; sc[4 * off + 1] = g; sc[4 * off + 2] = dist; sc[4 * off + 3] = o;
            usedL[i] = 1; usedR[g] = 1;
            ++off;
          }
          n_cand += total;
        }
        r0 = r1;
      }
      DBG_STAMP(0);
      DBG_STAMP(1);
      continue;   // this offset's matches are appended
    } else {
      // no usable arena: the reference's loop on HBM, one thread per row
      for (int i = tid; i < nL; i += NT) match[2 * i] = -1;
      __syncthreads();
      for (int r = tid; r < rows; r += NT) {
        const int rr = r - o;  // right row: L.row == R.row + o
        if (rr < 0 || rr >= rows) continue;
        const int l0 = rcL[(size_t)r * CW1], l1 = rcL[(size_t)r * CW1 + c.CW];
        const int g0 = rcR[(size_t)rr * CW1], g1 = rcR[(size_t)rr * CW1 + c.CW];
        int cur = g0;
        for (int i = l0; i < l1; ++i) {
          if (usedL[i]) continue;
          if (cur >= g1) break;
          const int xl = kxyL[2 * i];
          uint32_t ld[8];
          for (int q = 0; q < 8; ++q) ld[q] = reinterpret_cast<const uint32_t*>(descL + (size_t)32 * i)[q];
          int best = itau, bg = -1;
          for (int g = cur; g < g1; ++g) {
            if (usedR[g]) continue;
            if (xl - kxyR[2 * g] < 0) break;
            const int h = hamming32(ld, reinterpret_cast<const uint32_t*>(descR + (size_t)32 * g));
            if (h < best) { best = h; bg = g; }
          }
          if (bg >= 0) {
            if ((double)(xl - kxyR[2 * bg]) < c.c.minimum_disparity_pixels) continue;
            match[2 * i] = bg; match[2 * i + 1] = best;
            cur = bg + 1;
          }
        }
      }
    }
    DBG_STAMP(0);
    // append the matches of this offset in sorted-left order; mark both features used (prune)
    const int per = (nL + NT - 1) / NT;
    const int i0 = tid * per, i1 = min(i0 + per, nL);
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += (staged ? smatch[i] : match[2 * i]) >= 0 ? 1 : 0;
    int total;
    int off = n_cand + block_exclusive_scan(cnt, sh.scan, &total);
    for (int i = i0; i < i1; ++i) {
      int g, dist;
      if (staged) { const int sm = smatch[i]; g = sm < 0 ? -1 : (sm & 0xFFFF); dist = sm >> 16; }
      else { g = match[2 * i]; dist = g >= 0 ? match[2 * i + 1] : 0; }
      if (g < 0) continue;
      sc[4 * off] = i; sc[4 * off + 1] = g; sc[4 * off + 2] = dist; sc[4 * off + 3] = o;
      usedL[i] = 1; usedR[g] = 1;
      ++off;
    }
    n_cand += total;
    DBG_STAMP(1);
  }
  // ---- binning (:147-155, :371-394, :435-456) ---------------------------------------------------
  const int nb = c.rows_bin * c.cols_bin;
  const double bin = (double)c.c.bin_size_pixels;
  int added = 0;
  if (c.c.enable_keypoint_binning) {
    // working arrays of the bin competition: in LDS (the sweep's staging is dead by now) when they fit, else in HBM
    const bool bl = FULL && ((size_t)3 * (nb + 1) + (size_t)4 * n_cand) * 4 <= (size_t)arena_bytes;
    // small arenas: the same competition on 16-bit tables (bin cursors two to a word, candidate lists as u16) — 4 (nb + 2) + 8 n_cand
    // + a few bytes, e.g. 15 KB for 2158 bins and 800 candidates
    const size_t cw_words = ((size_t)nb + 2) / 2, occ_words = ((size_t)nb + 2) / 2;
    const bool bc = !bl && n_cand < 32767 && (cw_words + occ_words + (size_t)n_cand) * 4 + (size_t)n_cand * 4 + 16 <= (size_t)arena_bytes;
    if (bc) {
      uint32_t* cw = reinterpret_cast<uint32_t*>(arena);                     // per bin: count -> start -> fill cursor (u16 halves)
      int16_t* occ16 = reinterpret_cast<int16_t*>(cw + cw_words);            // -1 empty, -2 tracked occupant, >= 0 winning candidate
      uint32_t* cpk = reinterpret_cast<uint32_t*>(occ16) + occ_words;        // [n_cand] disparity << 16 | distance
      uint16_t* items = reinterpret_cast<uint16_t*>(cpk + n_cand);           // [n_cand] per-bin lists, then the winners in bin order
      uint16_t* cbin16 = items + n_cand;                                     // [n_cand] bin of candidate q
      auto half = [&](int k) -> int { return (int)((__hip_atomic_load(cw + (k >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (16 * (k & 1))) & 0xFFFFu); };
      __syncthreads();
      for (int k = tid; k < (int)cw_words; k += NT) cw[k] = 0u;
      for (int k = tid; k < (int)occ_words; k += NT) reinterpret_cast<uint32_t*>(occ16)[k] = 0xFFFFFFFFu;
      __syncthreads();
      for (int j = tid; j < n_tracked; j += NT) {      // tracked points seed the grid: a tracked occupant is never replaced
        const int rb = min((int)rint((double)cv.kp[4 * (size_t)j + 1] / bin), c.rows_bin - 1);
        const int cb = min((int)rint((double)cv.kp[4 * (size_t)j] / bin), c.cols_bin - 1);
        occ16[rb * c.cols_bin + cb] = (int16_t)-2;
      }
      for (int q = tid; q < n_cand; q += NT) {
        const int4 e = *reinterpret_cast<const int4*>(sc + 4 * q);
        const int lxy = *reinterpret_cast<const int32_t*>(kxyL + 2 * e.x);
        const int xl = (int16_t)(lxy & 0xFFFF), yl = lxy >> 16;
        const int rb = min((int)rint((double)yl / bin), c.rows_bin - 1);
        const int cb = min((int)rint((double)xl / bin), c.cols_bin - 1);
        const int k = rb * c.cols_bin + cb;
        cbin16[q] = (uint16_t)k;
        cpk[q] = ((uint32_t)(xl - kxyR[2 * e.y]) << 16) | (uint32_t)(e.z & 0xFFFF);
        atomicAdd(cw + (k >> 1), 1u << (16 * (k & 1)));
      }
      __syncthreads();
      {   // counts -> exclusive starts, in place
        const int perb = (nb + NT - 1) / NT;
        const int k0 = tid * perb, k1 = min(k0 + perb, nb);
        int cnt = 0;
        for (int k = k0; k < k1; ++k) cnt += half(k);
        int total;
        int off = block_exclusive_scan(cnt, sh.scan, &total);
        uint16_t* ch = reinterpret_cast<uint16_t*>(cw);
        for (int k = k0; k < k1; ++k) { const int m = ch[k]; ch[k] = (uint16_t)off; off += m; }
      }
      __syncthreads();
      for (int q = tid; q < n_cand; q += NT) {          // fill: the cursor of bin k moves from its start to its end
        const int k = cbin16[q];
        const uint32_t old = atomicAdd(cw + (k >> 1), 1u << (16 * (k & 1)));
        items[(old >> (16 * (k & 1))) & 0xFFFFu] = (uint16_t)q;
      }
      __syncthreads();
      for (int k = tid; k < nb; k += NT) {              // one thread per bin replays its candidates in sweep order
        if (occ16[k] != -1) continue;                   // tracked occupant
        const int i0 = k ? half(k - 1) : 0, m = half(k) - i0;
        int win = -1, wdisp = 0, wdist = 0, last = -1;
        for (int t = 0; t < m; ++t) {
          int q = 0x7FFFFFFF;
          for (int u = 0; u < m; ++u) { const int v = items[i0 + u]; if (v > last && v < q) q = v; }
          last = q;
          const uint32_t pk = cpk[q];
          const int disp = (int)pk >> 16, dist = (int)(pk & 0xFFFFu);
          if (win < 0 || (disp > wdisp && dist <= wdist)) { win = q; wdisp = disp; wdist = dist; }
        }
        occ16[k] = (int16_t)win;
      }
      __syncthreads();
      const int per = (nb + NT - 1) / NT;
      const int k0 = tid * per, k1 = min(k0 + per, nb);
      int cnt = 0;
      for (int k = k0; k < k1; ++k) cnt += occ16[k] >= 0 ? 1 : 0;
      int total;
      int off = block_exclusive_scan(cnt, sh.scan, &total);      // (its barriers also retire every reader of `items`)
      for (int k = k0; k < k1; ++k) { const int q = occ16[k]; if (q >= 0) items[off++] = (uint16_t)q; }
      __syncthreads();
      if (n_tracked + total > c.MAXP && tid == 0) atomicOr(&b.st[s].error_flags, 2);
      for (int t = tid; t < total && n_tracked + t < c.MAXP; t += NT) {
        const int4 e = *reinterpret_cast<const int4*>(sc + 4 * items[t]);
        materialize_point(c, b, s, cv, n_tracked + t, e.x, e.y, e.z, e.w, -1, 0);
      }
      added = total;
    } else {
    int32_t* occ = bl ? reinterpret_cast<int32_t*>(arena) : bin_occ;
    int32_t* bcnt = bl ? occ + (nb + 1) : b.bin_aux + (size_t)s * (2 * ((size_t)nb + 1) + c.NMAX);
    int32_t* bstart = bcnt + (nb + 1);
    int32_t* bitems = bstart + (nb + 1);                       // [n_cand]
    int32_t* cbin = bl ? bitems + n_cand : match;              // [n_cand][2]: bin id, (disparity << 16 | distance)
    int32_t* emit_q = bl ? cbin + 2 * n_cand : match + 2 * (size_t)c.NMAX;   // [<= n_cand] winners in bin order
    __syncthreads();
    for (int k = tid; k < nb; k += NT) { occ[k] = -1; bcnt[k] = 0; }
    __syncthreads();
    // tracked points seed the grid; later points overwrite earlier ones -> keep the largest index
    for (int j = tid; j < n_tracked; j += NT) {
      const int rb = min((int)rint((double)cv.kp[4 * (size_t)j + 1] / bin), c.rows_bin - 1);
      const int cb = min((int)rint((double)cv.kp[4 * (size_t)j] / bin), c.cols_bin - 1);
      atomicMax(occ + rb * c.cols_bin + cb, j);
    }
    // bin id / disparity / distance of every candidate, once; per-bin counts
    for (int q = tid; q < n_cand; q += NT) {
      const int4 e = *reinterpret_cast<const int4*>(sc + 4 * q);
      const int lxy = *reinterpret_cast<const int32_t*>(kxyL + 2 * e.x);
      const int xl = (int16_t)(lxy & 0xFFFF), yl = lxy >> 16;
      const int rb = min((int)rint((double)yl / bin), c.rows_bin - 1);
      const int cb = min((int)rint((double)xl / bin), c.cols_bin - 1);
      const int k = rb * c.cols_bin + cb;
      cbin[2 * q] = k;
      cbin[2 * q + 1] = ((xl - kxyR[2 * e.y]) << 16) | (e.z & 0xFFFF);
      atomicAdd(bcnt + k, 1);
    }
    __syncthreads();
    DBG_STAMP(2);
    // per-bin candidate lists by counting sort (arrival order inside a bin is arbitrary, restored by a tiny sort)
    {
      const int perb = (nb + NT - 1) / NT;
      const int k0 = tid * perb, k1 = min(k0 + perb, nb);
      int cnt = 0;
      for (int k = k0; k < k1; ++k) cnt += ld_relaxed(bcnt + k);   // written by atomics: read past the vector L1
      int total;
      int off = block_exclusive_scan(cnt, sh.scan, &total);
      for (int k = k0; k < k1; ++k) { const int m = ld_relaxed(bcnt + k); bstart[k] = off; off += m; __hip_atomic_store(bcnt + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      if (tid == 0) bstart[nb] = total;
    }
    __syncthreads();
    for (int q = tid; q < n_cand; q += NT) {
      const int k = cbin[2 * q];
      bitems[bstart[k] + atomicAdd(bcnt + k, 1)] = q;
    }
    __syncthreads();
    // one thread per bin replays its candidates in sweep order (the rule is not an argmax)
    for (int k = tid; k < nb; k += NT) {
      const int o0 = ld_relaxed(occ + k);
      if (o0 >= 0) { occ[k] = -2 - o0; continue; }  // tracked occupant: never replaced
      const int i0 = bstart[k], m = bstart[k + 1] - i0;
      int win = -1, wdisp = 0, wdist = 0, last = -1;
      for (int t = 0; t < m; ++t) {
        // next candidate in ascending sweep order: smallest q greater than the last one taken
        int q = 0x7FFFFFFF;
        for (int u = 0; u < m; ++u) { const int v = ld_relaxed(bitems + i0 + u); if (v > last && v < q) q = v; }
        last = q;
        const int pk = cbin[2 * q + 1];
        const int disp = pk >> 16, dist = pk & 0xFFFF;
        if (win < 0 || (disp > wdisp && dist <= wdist)) { win = q; wdisp = disp; wdist = dist; }
      }
      occ[k] = win;  // -1 empty, >= 0 candidate index
    }
    __syncthreads();
    DBG_STAMP(3);
    // winners in bin-grid row-major order, then one thread per new point
    const int per = (nb + NT - 1) / NT;
    const int k0 = tid * per, k1 = min(k0 + per, nb);
    int cnt = 0;
    for (int k = k0; k < k1; ++k) cnt += occ[k] >= 0 ? 1 : 0;
    int total;
    int off = block_exclusive_scan(cnt, sh.scan, &total);
    for (int k = k0; k < k1; ++k) { const int q = occ[k]; if (q >= 0) emit_q[off++] = q; }
    __syncthreads();
    if (n_tracked + total > c.MAXP && tid == 0) atomicOr(&b.st[s].error_flags, 2);
    for (int t = tid; t < total && n_tracked + t < c.MAXP; t += NT) {
      const int4 e = *reinterpret_cast<const int4*>(sc + 4 * emit_q[t]);
      materialize_point(c, b, s, cv, n_tracked + t, e.x, e.y, e.z, e.w, -1, 0);
    }
    added = total;
    }   // !bc
  } else {
    for (int q = tid; q < n_cand; q += NT) {
      const int j = n_tracked + q;
      if (j < c.MAXP) materialize_point(c, b, s, cv, j, sc[4 * q], sc[4 * q + 1], sc[4 * q + 2], sc[4 * q + 3], -1, 0);
      else atomicOr(&b.st[s].error_flags, 2);
    }
    added = n_cand;
  }
  __syncthreads();
  DBG_STAMP(4);
  const int n_final = min(n_tracked + added, c.MAXP);
  // history of the appended points
  double* hc = hcam_of(c, b, s, f);
  int32_t* hp = hprev_of(c, b, s, f);
  for (int j = n_tracked + tid; j < n_final; j += NT) {
    { const double x = cv.cam[3 * (size_t)j], y = cv.cam[3 * (size_t)j + 1], z = cv.cam[3 * (size_t)j + 2];
      reinterpret_cast<double2*>(hc + 4 * (size_t)j)[0] = make_double2(x, y); reinterpret_cast<double2*>(hc + 4 * (size_t)j)[1] = make_double2(z, 1 / z); }
    hp[j] = -1;
    if (c.trail) *reinterpret_cast<uint4*>(cv.trail + (size_t)j * VS_TRAIL) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // a track starts here
  }
  if (tid == 0) { sh.n_cand = added; sh.n_cur = n_final; }
  __syncthreads();
}

__device__ __forceinline__ void wg_stereo(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_cur, double tau_tri, int f,
                          unsigned char* arena, int arena_bytes) {
  wg_stereo_t<VS_WG, FrameShared>(c, b, s, sh, pb_cur, tau_tri, f, arena, arena_bytes);
}

// L-R Hamming distances of the first epipolar pass for every left feature of every stream (image pipeline): the window
// wg_stereo's step A reads.  One thread per left feature.
__global__ __launch_bounds__(256) void k_stereo_dist(const DevCfg c, const DevBuf b) {
  int bx, sy;
  xcd_stream_block(&bx, &sy, b.xcd_rot);
  const int s = b.s0 + sy;
  if (!vs_active(b, s)) return;
  const int i = bx * blockDim.x + threadIdx.x;
  const int nL = b.n_kp[s * 2];
  if (i >= nL) return;
  const int16_t* kxyL = kpxy_of(c, b, s, 0);
  const int16_t* kxyR = kpxy_of(c, b, s, 1);
  const int rows = c.c.rows, CW1 = c.CW + 1, o = c.offsets[0];
  const int rr = kxyL[2 * i + 1] - o;
  if (rr < 0 || rr >= rows) return;
  const int32_t* rcR = rowcell_of(c, b, s, 1);
  stereo_dist_row(desc_of(c, b, s, 0), desc_of(c, b, s, 1), i, rcR[(size_t)rr * CW1], rcR[(size_t)rr * CW1 + c.CW], kxyL[2 * i],
                  [&](int g) { return (int)kxyR[2 * g]; }, b.sdist + (size_t)s * c.NMAX * 16);
}

// ==============================================================================================
// K5: PoseTracker3D::compute for one stream (pose_tracker_3d.cpp:32-222)
// ==============================================================================================
__device__ __forceinline__ void set_pose(const DevCfg& c, const DevBuf& b, int s, int f, const double* c2w) {
  double* hp = hpose_of(c, b, s, f);
  for (int k = 0; k < 12; ++k) hp[k] = c2w[k];
  tf_inverse(c2w, hp + 12);
}


// Phase 2 of the frame (status switch, stereo sweep + binning + emission, the frame's report and the state carried to the next frame): the end of
// k_frame, and the frame workgroups of k_tail_lm.  sh.n_cur / sh.n_cand are set by the caller.
__device__ __forceinline__ void frame_phase2(const DevCfg& c, const DevBuf& b, int s, StreamState& st, vslam_frame_info& info, FrameCarry& fc, FrameShared& sh,
                                             int pb_cur, int f, unsigned char* arena) {
  const int tid = threadIdx.x;
  const int n_active = fc.n_active;
  int status = fc.status;
  if (n_active > c.c.minimum_number_of_landmarks_to_track) status = VSLAM_TRACKING;
  const double tau_tri2 = fc.tau_tri;
  const unsigned long long ts = wall_clock64();
  wg_stereo(c, b, s, sh, pb_cur, tau_tri2, f, arena, VS_ARENA);
  if (tid == 0) {
    st.ticks[4] += wall_clock64() - ts;
    const double* c2w = hpose_of(c, b, s, f);
    *pts_of(c, b, s, pb_cur).n = sh.n_cur;
    st.status = status; st.win = fc.win; st.tau_track = fc.tau_track; st.tau_tri = tau_tri2;
    for (int k = 0; k < 12; ++k) { st.prior[k] = fc.prior[k]; st.pose[k] = c2w[k]; }
    st.n_tracked_landmarks_prev = n_active;
    st.frame_count = f + 1; st.has_prev = 1; st.cur = pb_cur; st.aligner_valid = fc.aligner_valid;
    info.frame_index = f + 1; info.status = status; info.status_at_start = fc.status0;
    info.n_keypoints_left = b.n_kp[s * 2]; info.n_keypoints_right = b.n_kp[s * 2 + 1];
    int rl = 0, rr = 0;
    for (int r = 0; r < c.n_regions; ++r) { rl += b.iinfo[s].raw_count[0][r]; rr += b.iinfo[s].raw_count[1][r]; info.thresholds[r] = b.iinfo[s].thr_after[r]; }
    for (int r = c.n_regions; r < VSLAM_MAX_REGIONS; ++r) info.thresholds[r] = 0;
    info.n_detected_left = rl; info.n_detected_right = rr;
    info.track_attempts = fc.attempts; info.n_after_prune = fc.n_after_prune; info.n_recovered = fc.n_recovered;
    info.n_active_landmarks = n_active; info.n_new_stereo = sh.n_cand; info.n_points = sh.n_cur;
    info.track_broken = fc.broken; info.fallback = fc.fallback; info.window_pixels = fc.win;
    info.error_flags = st.error_flags; info.tau_track = fc.tau_track; info.tau_triangulation = tau_tri2;
    for (int k = 0; k < 12; ++k) { info.camera_left_to_world[k] = c2w[k]; info.previous_to_current[k] = fc.prior[k]; }
    if (f < VS_POSE_LOG) { double* pl = b.pose_log + ((size_t)s * VS_POSE_LOG + f) * 12; for (int k = 0; k < 12; ++k) pl[k] = c2w[k]; }
    st.dbg[8] += wall_clock64() - fc.t0;
  }
}

// The frame is processed by three phase launches of this kernel with wide kernels in between (fused path):
//   phase 0  track resolution, registration (aligner, recursion, fallback / break), prune, recovery projection
//   [k_recover_brief]   BRIEF of the projected lost points, all streams, one wavefront each
//   phase 1  recovery append, history publication
//   [k_update_landmarks] landmark creation / refinement, one thread per framepoint; [k_stereo_dist] L-R distances
//   phase 2  status switch, stereo sweep + binning + emission, report
// phase < 0 runs everything in one launch (the wide steps inside the workgroup).
// phase 4 = phase 1 + the count of active landmarks: the refinement then runs in workgroups of its own BESIDE phase 2 (k_tail_lm, launch sequence 4, few streams:
// nothing of phase 2 reads what the refinement writes — landmark coordinates and update counts of the tracked points; the next frame's phase 0 does).
// phase 3 = phases 1 and 2 in one launch with the landmark refinement inside the workgroup: with phase 0 and the wide recovery
// kernel in front this is the two-launch sequence used for few streams (the ~140 recovery patches of a frame spread over the
// idle CUs: 67 -> 9 us; the wide landmark kernel, one thread per track, is slower than the workgroup's LDS-cached one).
// The configuration and the buffer table arrive as pointers into the CONSTANT address space (device-resident copies the
// context uploads once): passed by value, the ~60 pointers of DevBuf are all loaded in the prologue, cannot stay in the
// 100-odd SGPRs and are parked in VGPR lanes — 1900 v_readlane instructions kernel-wide, ~190 in every aligner round.
// Through the constant address space each use is a scalar load next to where it is needed.
typedef const DevCfg __attribute__((address_space(4))) ConstDevCfg;
typedef const DevBuf __attribute__((address_space(4))) ConstDevBuf;
__global__ VS_FRAME_BOUNDS void k_frame(ConstDevCfg* cp, ConstDevBuf* bp, int phase) {
  const DevCfg& c = *(const DevCfg*)cp;
  const DevBuf& b = *(const DevBuf*)bp;
  __shared__ FrameShared sh;
  __shared__ __align__(16) unsigned char arena[VS_ARENA];
#ifdef VS_FRAME_PRIO
  __builtin_amdgcn_s_setprio(VS_FRAME_PRIO);   // co-scheduled builds: the latency-bound frame wavefronts issue ahead of the image kernels' wavefronts on their SIMD
#endif
  const int s = b.s0 + xcd_local_stream(blockIdx.x, gridDim.x, b.xcd_rot), tid = threadIdx.x;
  if (!vs_active(b, s)) return;
  StreamState& st = b.st[s];
  vslam_frame_info& info = b.info[s];
  const int f = st.frame_count;             // index of the frame being processed
  const int has_prev = st.has_prev;
  const int pb_prev = st.cur, pb_cur = st.cur ^ 1;
  const int status0 = st.status;
  const int n_lm_prev = st.n_tracked_landmarks_prev;
  FrameCarry& fc = st.fc;
  if (phase <= 0) {   // ======================================= phase 0 =======================================
  if (tid == 0) {
    sh.status = status0; sh.win = st.win; sh.tau_track = st.tau_track; sh.attempts = 0; sh.broken = 0; sh.fallback = 0;
    sh.aligner_ran = 0; sh.n_trk = 0; sh.n_lost = 0; sh.n_lm = 0; sh.n_cur = 0; sh.n_cand = 0; sh.its = 0; sh.conv = 0; sh.n_proj = 0;
    sh.inl = 0; sh.outl = 0; sh.E = 0; sh.flag = 0;
    for (int k = 0; k < 12; ++k) sh.T[k] = 0;
    for (int k = 0; k < 36; ++k) sh.H[k] = 0;
    set_pose(c, b, s, f, st.pose);          // frame created at WorldMap::robot_to_world
  }
  const unsigned long long tK0 = wall_clock64();
  (void)tK0;
  // the motion prior lives in LDS (sh.prior), not in 24 VGPRs of every thread across the whole registration loop
  if (tid < 12) sh.prior[tid] = st.prior[tid];
  const double* prior = sh.prior;
  const double tau_tri = tau_tri_rule(c, status0, b.n_kp[s * 2]);
  int win = st.win;
  double tau_track = st.tau_track;          // tracker's _current_descriptor_distance_tracking
  double tau_gen = tau_track;               // generator's _maximum_descriptor_distance_tracking (last _track)
  __syncthreads();
  const double* prev_c2w = hpose_of(c, b, s, f > 0 ? f - 1 : 0);
  int n_tracked_landmarks = 0, n_after_prune = 0;
  bool aligner_valid = false;

  if (has_prev) {
    const PtView pv = pts_of(c, b, s, pb_prev);
    const int P = *pv.n;
    for (int i = tid; i < P; i += VS_WG) pv.meta[(size_t)i * META + M_NEXT] = 0;
    __syncthreads();
    // ---- up to three track/register attempts (_registerRecursive, :300-419) -----------------------
    int by_app = status0 == VSLAM_LOCALIZING;
    bool done = false;
    for (int attempt = 0; attempt < 3 && !done; ++attempt) {
      // _track (:225-298)
      if (by_app) win = c.c.maximum_projection_tracking_distance_pixels;
      tau_gen = tau_track;
      if (attempt > 0) {
        const unsigned long long tc = wall_clock64();
        // initialize(frame, false): fresh feature stores; candidates for the new prior / window / mode
        const int lane = tid % VS_CGL, w = tid / VS_CGL;
        for (int i = w; i < P; i += VS_WG / VS_CGL) candidates_wave(c, b, s, pb_prev, i, lane, reinterpret_cast<CandWave*>(arena) + w, prior, win, tau_gen, tau_tri, by_app);
        __syncthreads();
        if (tid == 0) st.ticks[0] += wall_clock64() - tc;
      }
      aligner_valid = false;
      unsigned long long t0 = wall_clock64();
      wg_track_resolve(c, b, s, sh, pb_prev, arena, win, tau_gen, tau_tri, by_app);
      if (tid == 0) st.ticks[0] += wall_clock64() - t0;
      const int n_trk = sh.n_trk;
      n_tracked_landmarks = sh.n_lm;
      {
        const double ratio = (double)n_trk / (double)P;
        const double lm_per_pt = (double)n_tracked_landmarks / (double)n_trk;
        const double succ = (double)n_trk / (double)c.target_kp;
        const int wmax = c.c.maximum_projection_tracking_distance_pixels, wmin = c.c.minimum_projection_tracking_distance_pixels;
        if (ratio < c.c.good_tracking_ratio / 2) {
          if (win < wmax) win = (int)fmin(win * 1 / c.c.tunnel_vision_ratio, (double)wmax);
        } else {
          if (win > wmin) win = (int)fmax(win * c.c.tunnel_vision_ratio, (double)wmin);
        }
        if (ratio < c.c.good_tracking_ratio || n_trk < c.c.aligner_minimum_number_of_inliers || (lm_per_pt < 0.5 && succ < 0.25)) {
          tau_track += 5;
          if (tau_track > c.c.maximum_descriptor_distance_tracking) tau_track = c.c.maximum_descriptor_distance_tracking;
        } else {
          tau_track -= 5;
          if (tau_track < c.c.minimum_descriptor_distance_tracking) tau_track = c.c.minimum_descriptor_distance_tracking;
        }
      }
      if (tid == 0) ++sh.attempts;
      // ---- registration ------------------------------------------------------------------------------
      bool accept = false, fall = false, brk = false;
      if (status0 == VSLAM_LOCALIZING) {
        // :103-161
        if (n_trk < c.c.minimum_number_of_landmarks_to_track) {
          fall = true;
        } else {
          const unsigned long long ta = wall_clock64();
          wg_align(c, b, s, sh, pb_prev, false, prior, arena, VS_ARENA);
          if (tid == 0) st.ticks[1] += wall_clock64() - ta;
          aligner_valid = true;
          if (sh.inl < c.c.minimum_number_of_landmarks_to_track) fall = true; else accept = true;
        }
        done = true;
      } else {
        const double rel = (double)n_tracked_landmarks / (double)n_lm_prev;
        if (n_tracked_landmarks == 0 || rel < 0.1) {
          if (attempt < 2) {
            if (tid == 0) tf_identity(sh.prior);
            by_app = 1;
          } else {
            brk = true; done = true;
          }
        } else {
          const unsigned long long ta = wall_clock64();
          wg_align(c, b, s, sh, pb_prev, true, prior, arena, VS_ARENA);
          if (tid == 0) st.ticks[1] += wall_clock64() - ta;
          aligner_valid = true;
          if (sh.inl > c.c.minimum_number_of_landmarks_to_track) {
            accept = true; done = true;
          } else if (attempt < 2) {
            if (win < c.c.maximum_projection_tracking_distance_pixels) ++win;
            by_app = 0;
          } else {
            brk = true; done = true;
          }
        }
      }
      if (accept) {
        // accept-or-fallback on the size of the motion (:139-159, :372-388)
        const double dang = rotation_angle(sh.T);
        const double dtr = sqrt((sh.T[3] * sh.T[3] + sh.T[7] * sh.T[7]) + sh.T[11] * sh.T[11]);
        if (dang > c.c.minimum_delta_angular_for_movement || dtr > c.c.minimum_delta_translational_for_movement) {
          if (tid == 0) {
            for (int k = 0; k < 12; ++k) sh.prior[k] = sh.T[k];
            double inv[12], c2w[12];
            tf_inverse(sh.T, inv);
            tf_mul(prev_c2w, inv, c2w);
            set_pose(c, b, s, f, c2w);
          }
        } else {
          fall = true;
        }
      }
      if (fall) {  // _fallbackEstimate (:551-566)
        if (tid == 0) { tf_identity(sh.prior); set_pose(c, b, s, f, prev_c2w); sh.fallback = 1; }
      }
      if (brk) {   // breakTrack (:422-435)
        if (tid == 0) { tf_identity(sh.prior); set_pose(c, b, s, f, prev_c2w); sh.broken = 1; sh.status = VSLAM_LOCALIZING; }
      }
      __syncthreads();
    }
    // report the final track() / converge() before their buffers are consumed
    if (tid == 0) {
      info.n_tracked = sh.n_trk; info.n_lost = sh.n_lost; info.n_tracked_landmarks = n_tracked_landmarks;
      info.aligner_ran = aligner_valid ? 1 : 0;
      info.aligner_iterations = aligner_valid ? sh.its : 0; info.aligner_converged = aligner_valid ? sh.conv : 0;
      info.n_inliers = aligner_valid ? sh.inl : 0; info.n_outliers = aligner_valid ? sh.outl : 0;
      info.total_error = aligner_valid ? sh.E : 0;
      st.al_n = aligner_valid ? sh.n_trk : 0;
      for (int k = 0; k < 12; ++k) st.al_T[k] = sh.T[k];
      for (int k = 0; k < 36; ++k) st.al_H[k] = sh.H[k];
    }
    VS_PHASE_BEGIN(tP);
    wg_prune(c, b, s, sh, pb_prev, pb_cur, aligner_valid);
    VS_PHASE_STAMP(6, tP);
    n_after_prune = sh.n_cur;
    if (c.c.enable_landmark_recovery) {
      if (phase < 0) wg_recover_project(c, b, s, sh.n_lost, pb_prev, hpose_of(c, b, s, f) + 12, reinterpret_cast<int32_t*>(arena + VS_RLIST_OFF), VS_RLIST_CAP, &sh.n_proj);
      else wg_recover_project(c, b, s, sh.n_lost, pb_prev, hpose_of(c, b, s, f) + 12);
    }
  } else if (tid == 0) {
    info.n_tracked = 0; info.n_lost = 0; info.n_tracked_landmarks = 0; info.aligner_ran = 0; info.aligner_iterations = 0;
    info.aligner_converged = 0; info.n_inliers = 0; info.n_outliers = 0; info.total_error = 0; st.al_n = 0;
  }
  __syncthreads();
  if (tid == 0) {
    fc.status = sh.status; fc.status0 = status0; fc.win = win; fc.attempts = sh.attempts; fc.broken = sh.broken; fc.fallback = sh.fallback;
    fc.n_after_prune = n_after_prune; fc.aligner_valid = aligner_valid ? 1 : 0; fc.n_tracked_landmarks = n_tracked_landmarks;
    fc.n_cur = sh.n_cur; fc.n_lost = (has_prev && c.c.enable_landmark_recovery) ? sh.n_lost : 0; fc.n_recovered = 0; fc.n_active = 0;
    fc.tau_track = tau_track; fc.tau_gen = tau_gen; fc.tau_tri = tau_tri; fc.t0 = tK0;
    for (int k = 0; k < 12; ++k) fc.prior[k] = prior[k];
  }
  __syncthreads();
  if (phase == 0) return;
  }  // phase 0

  if (phase == 1 || phase < 0 || phase == 3 || phase == 4) {   // ========================== phase 1 (4: + the count of active landmarks) ==========================
    if (tid == 0) { sh.n_cur = fc.n_cur; sh.n_lost = fc.n_lost; sh.flag = 0; }
    __syncthreads();
    if (has_prev && c.c.enable_landmark_recovery) {
      const unsigned long long tr = wall_clock64();
      if (phase < 0) {
        wg_recover_brief(c, b, s, pb_prev, sh.n_lost, sh.n_proj, fc.tau_gen, fc.tau_tri, arena);
        __syncthreads();
      }
      wg_recover_append(c, b, s, sh, pb_prev, pb_cur);
      if (tid == 0) { st.ticks[2] += wall_clock64() - tr; fc.n_recovered = sh.flag; }
    }
    wg_publish_history(c, b, s, sh.n_cur, pb_cur, f);
    if (tid == 0) { fc.n_cur = sh.n_cur; fc.n_active = 0; fc.lm_pb = pb_cur; fc.lm_f = f; }
    __syncthreads();
    if (phase == 4) {
      // the landmark kernel will run BESIDE phase 2: the number of active landmarks — what the status switch needs — is the number of points
      // whose track is long enough for a landmark (landmark_point_t returns false for nothing else)
      const PtView cvc = pts_of(c, b, s, pb_cur);
      int active = 0;
      for (int i = tid; i < sh.n_cur; i += VS_WG) active += cvc.meta[(size_t)i * META + M_TLEN] >= c.c.minimum_track_length_for_landmark_creation ? 1 : 0;
      int total;
      block_exclusive_scan(active, sh.scan, &total);
      if (tid == 0) fc.n_active = total;
      return;
    }
    if (phase == 1) return;
  }

  // ============================================== phase 2 ==============================================
  if (tid == 0) { sh.n_cur = fc.n_cur; sh.n_cand = 0; }
  __syncthreads();
  if (phase < 0 || phase == 3) {
    const unsigned long long tu = wall_clock64();
    const int total = wg_landmarks_lds(c, b, s, sh, pb_cur, f, arena);
    if (tid == 0) { fc.n_active = total; st.ticks[3] += wall_clock64() - tu; }
    __syncthreads();
  }
  frame_phase2(c, b, s, st, info, fc, sh, pb_cur, f, arena);
}

// Launch sequence 4's last launch: the n frame workgroups run phase 2, G more workgroups per stream the landmark refinement (lm_teams_body) — ONE launch
// instead of a second queue with a fork and a join around it (each costs the frame queue ~7 us).  Nothing in phase 2 reads what the refinement writes.
// (Folding the recovery descriptors and phase 1 into the same launch as well — workgroups handing over through counters — was measured: every
// workgroup of a launch carries the frame workgroup's 140 KB of LDS, so the recovery workers own whole CUs and the next frame's image kernels lose
// them: 0.226 -> 0.232 ms for one stream, 0.292 -> 0.353 for eleven.)
__global__ VS_FRAME_BOUNDS void k_tail_lm(ConstDevCfg* cp, ConstDevBuf* bp, int n, int G) {
  const DevCfg& c = *(const DevCfg*)cp;
  const DevBuf& b = *(const DevBuf*)bp;
  __shared__ FrameShared sh;
  __shared__ __align__(16) unsigned char arena[VS_ARENA];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < n) {
    const int s = b.s0 + xcd_local_stream(blockIdx.x, n, b.xcd_rot);
    if (!vs_active(b, s)) return;
    StreamState& st = b.st[s];
    FrameCarry& fc = st.fc;
    if (tid == 0) { sh.n_cur = fc.n_cur; sh.n_cand = 0; }
    __syncthreads();
    frame_phase2(c, b, s, st, b.info[s], fc, sh, st.cur ^ 1, st.frame_count, arena);
  } else {
    const int i = (int)blockIdx.x - n, sl = i / G;
    const int s = b.s0 + sl;
    if (!vs_active(b, s)) return;
    lm_teams_body(c, b, s, i - sl * G, G, 0, true, arena, sh.flag, sh.n_proj, sh.scan);
  }
}

// ==============================================================================================
// The frame's TAIL as a kernel of its own (launch sequence 3: k_track_candidates, k_frame phase 0, k_recover_brief, k_tail):
// recovery append, history, landmark creation / refinement, status switch, stereo sweep + binning + emission, report — everything
// after registration and pruning.  None of it needs the aligner's 245 registers, so this kernel is shaped to fit the HOLE ONE
// image-kernel workgroup leaves on a busy CU (tools/probe/cosched.hip: 256 threads, <= 128 VGPRs, <= ~22 KB LDS start within
// microseconds beside k_fast_box's flood; anything with 512 threads, more registers or more LDS waits until the flood's grid is
// exhausted): it runs BESIDE the image pipeline of the next frame instead of on CUs of its own.  Its wavefronts raise their
// priority: a latency-bound chain loses nothing beside VALU-busy neighbours (same probe) and should not queue behind them.
// Same device functions as k_frame's phases 1-2, instantiated for 256 threads, the landmark cache one measurement deep, the
// stereo sweep band by band and the bin competition on 16-bit tables (wg_stereo_t).
// ==============================================================================================
#define VS_TAIL_WG 256
#ifndef VS_TAIL_ARENA
#define VS_TAIL_ARENA 17408
#endif
struct TailShared { int scan[17]; int flag; int n_lost, n_cur, n_cand; };
__global__ __launch_bounds__(VS_TAIL_WG, 4) void k_tail(ConstDevCfg* cp, ConstDevBuf* bp) {
  const DevCfg& c = *(const DevCfg*)cp;
  const DevBuf& b = *(const DevBuf*)bp;
  __shared__ TailShared sh;
  __shared__ __align__(16) unsigned char arena[VS_TAIL_ARENA];
  __builtin_amdgcn_s_setprio(2);
  const int s = b.s0 + xcd_local_stream(blockIdx.x, gridDim.x, b.xcd_rot), tid = threadIdx.x;
  if (!vs_active(b, s)) return;
  StreamState& st = b.st[s];
  vslam_frame_info& info = b.info[s];
  const int f = st.frame_count;
  const int has_prev = st.has_prev;
  const int pb_prev = st.cur, pb_cur = st.cur ^ 1;
  FrameCarry& fc = st.fc;
  if (tid == 0) { sh.n_cur = fc.n_cur; sh.n_lost = fc.n_lost; sh.flag = 0; sh.n_cand = 0; }
  __syncthreads();
  if (has_prev && c.c.enable_landmark_recovery) {
    const unsigned long long tr = wall_clock64();
    wg_recover_append_t<VS_TAIL_WG, TailShared>(c, b, s, sh, pb_prev, pb_cur);
    if (tid == 0) { st.ticks[2] += wall_clock64() - tr; fc.n_recovered = sh.flag; }
  }
  wg_publish_history(c, b, s, sh.n_cur, pb_cur, f);
  __syncthreads();
  // ---- _updatePoints: landmarks --------------------------------------------------------------------------------------
  int n_active = 0;
  {
    const unsigned long long tu = wall_clock64();
    const PtView cvu = pts_of(c, b, s, pb_cur);
    typedef LmCacheT<VS_TAIL_WG, 1> LC;
    static_assert(sizeof(LC) + 1024 <= VS_TAIL_ARENA, "landmark cache + work list must fit the tail's arena");
    LC* lc = reinterpret_cast<LC*>(arena);
    for (int t = tid; t < VS_LM_NP * 12; t += VS_TAIL_WG) { const int k = t / 12; if (f - k >= 0 && k < c.HCAP) lc->w2c[k][t - 12 * k] = hpose_of(c, b, s, f - k)[12 + t - 12 * k]; }
    __syncthreads();
    lm_stage_rtr(lc, f, c.HCAP, VS_TAIL_WG);
    constexpr int LIST_CAP = (VS_TAIL_ARENA - (int)sizeof(LC)) / 2;
    uint16_t* work = reinterpret_cast<uint16_t*>(arena + sizeof(LC));
    int active = 0;
    // the points that carry a landmark, compacted into a work list (LIST_CAP points of the frame at a time): one per thread
    for (int c0 = 0; c0 < sh.n_cur; c0 += LIST_CAP) {
      const int c1 = min(c0 + LIST_CAP, sh.n_cur);
      __syncthreads();
      if (tid == 0) sh.flag = 0;
      __syncthreads();
      for (int i0 = c0; i0 < c1; i0 += VS_TAIL_WG) {
        const int i = i0 + tid;
        const bool need = i < c1 && cvu.meta[(size_t)i * META + M_TLEN] >= c.c.minimum_track_length_for_landmark_creation;
        const unsigned long long m = __ballot(need);
        int base = 0;
        if ((tid & 63) == 0 && m) base = atomicAdd(&sh.flag, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (need) work[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(i - c0);
      }
      __syncthreads();
      const int n_work = sh.flag;
      for (int q = tid; q < n_work; q += VS_TAIL_WG) active += landmark_point_t<true, LC>(c, b, s, cvu, f, c0 + work[q], lc) ? 1 : 0;
    }
    block_exclusive_scan(active, sh.scan, &n_active);
    if (tid == 0) { fc.n_active = n_active; sh.flag = 0; st.ticks[3] += wall_clock64() - tu; }
    __syncthreads();
  }
  int status = fc.status;
  if (n_active > c.c.minimum_number_of_landmarks_to_track) status = VSLAM_TRACKING;
  const double tau_tri2 = fc.tau_tri;
  const unsigned long long ts = wall_clock64();
  wg_stereo_t<VS_TAIL_WG, TailShared, false>(c, b, s, sh, pb_cur, tau_tri2, f, arena, VS_TAIL_ARENA);
  if (tid == 0) {
    st.ticks[4] += wall_clock64() - ts;
    const double* c2w = hpose_of(c, b, s, f);
    *pts_of(c, b, s, pb_cur).n = sh.n_cur;
    st.status = status; st.win = fc.win; st.tau_track = fc.tau_track; st.tau_tri = tau_tri2;
    for (int k = 0; k < 12; ++k) { st.prior[k] = fc.prior[k]; st.pose[k] = c2w[k]; }
    st.n_tracked_landmarks_prev = n_active;
    st.frame_count = f + 1; st.has_prev = 1; st.cur = pb_cur; st.aligner_valid = fc.aligner_valid;
    info.frame_index = f + 1; info.status = status; info.status_at_start = fc.status0;
    info.n_keypoints_left = b.n_kp[s * 2]; info.n_keypoints_right = b.n_kp[s * 2 + 1];
    int rl = 0, rr = 0;
    for (int r = 0; r < c.n_regions; ++r) { rl += b.iinfo[s].raw_count[0][r]; rr += b.iinfo[s].raw_count[1][r]; info.thresholds[r] = b.iinfo[s].thr_after[r]; }
    for (int r = c.n_regions; r < VSLAM_MAX_REGIONS; ++r) info.thresholds[r] = 0;
    info.n_detected_left = rl; info.n_detected_right = rr;
    info.track_attempts = fc.attempts; info.n_after_prune = fc.n_after_prune; info.n_recovered = fc.n_recovered;
    info.n_active_landmarks = n_active; info.n_new_stereo = sh.n_cand; info.n_points = sh.n_cur;
    info.track_broken = fc.broken; info.fallback = fc.fallback; info.window_pixels = fc.win;
    info.error_flags = st.error_flags; info.tau_track = fc.tau_track; info.tau_triangulation = tau_tri2;
    for (int k = 0; k < 12; ++k) { info.camera_left_to_world[k] = c2w[k]; info.previous_to_current[k] = fc.prior[k]; }
    if (f < VS_POSE_LOG) { double* pl = b.pose_log + ((size_t)s * VS_POSE_LOG + f) * 12; for (int k = 0; k < 12; ++k) pl[k] = c2w[k]; }
    st.dbg[8] += wall_clock64() - fc.t0;
  }
}

// recoverPoints on caller-provided lost points (vslam_stereo_recover): previous buffer 0 holds the lost points' descriptors,
// landmarks and landmark flags, the lost list is 0..n-1, survivors are appended to buffer 1 from its start.
struct RecoverAlone { double w2c[12]; double tau_track, tau_tri; int n; };
__global__ __launch_bounds__(VS_WG) void k_recover_alone(const DevCfg c, const DevBuf b, const RecoverAlone a) {
  __shared__ FrameShared sh;
  __shared__ __align__(16) unsigned char arena[VS_ARENA];
  __shared__ double w2c[12];
  const int s = b.s0, tid = threadIdx.x;
  if (tid < 12) w2c[tid] = a.w2c[tid];
  if (tid == 0) { sh.n_lost = a.n; sh.n_cur = 0; sh.flag = 0; }
  __syncthreads();
  wg_recover(c, b, s, sh, 0, 1, w2c, a.tau_track, a.tau_tri, arena);
  if (tid == 0) { b.st[s].n_cur = sh.n_cur; b.st[s].n_recovered = sh.flag; }
}

// ==============================================================================================
// Stage-granular entry points: the same device functions, one reference virtual per launch, with the
// control flow left to the caller (shim/proslam_hip_plugin.h keeps the reference's PoseTracker3D logic).
// ==============================================================================================
enum { VS_STAGE_TRACK = 1, VS_STAGE_ALIGN = 2, VS_STAGE_PRUNE_RECOVER = 3, VS_STAGE_UPDATE = 4, VS_STAGE_STEREO = 5, VS_STAGE_COMPUTE = 6 /* UPDATE then STEREO */,
       VS_STAGE_PRUNE_PROJECT = 7, VS_STAGE_RECOVER_APPEND = 8 /* PRUNE_RECOVER as two launches around the wide k_recover_brief */,
       VS_STAGE_STEREO_COUNT = 9 /* STEREO + the COUNT of active landmarks: their refinement runs beside the stage in the same launch (k_stage_lm) */ };

// WorldMap::createFrame + the bookkeeping PoseTracker3D::compute does before initialize() (:36-77)
// the caller's setters folded into a stage launch (StageIo): applied by one lane before anything reads the stream state
__device__ __forceinline__ void stage_apply_set(StreamState& st, const StageIo& io) {
  if (io.set_flags & 1) { st.status = io.status; st.win = io.win; st.tau_track = io.tau; for (int k = 0; k < 12; ++k) st.prior[k] = io.prior[k]; }
  if (io.set_flags & 2) { for (int k = 0; k < 12; ++k) st.pose[k] = io.pose[k]; }
}
__global__ __launch_bounds__(256) void k_begin(const DevCfg c, const DevBuf b, const StageIo io) {
  const int s = b.s0 + xcd_local_stream(blockIdx.x, gridDim.x, b.xcd_rot), tid = threadIdx.x;
  if (!vs_active(b, s)) return;
  StreamState& st = b.st[s];
  if (io.set_flags) { if (tid == 0) stage_apply_set(st, io); __syncthreads(); }
  const int f = st.frame_count;
  if (st.has_prev) {
    const PtView pv = pts_of(c, b, s, st.cur);
    const int P = *pv.n;
    for (int i = tid; i < P; i += blockDim.x) pv.meta[(size_t)i * META + M_NEXT] = 0;
  }
  if (tid == 0) {
    set_pose(c, b, s, f, st.pose);
    st.tau_tri = tau_tri_rule(c, st.status, b.n_kp[s * 2]);
    st.n_trk = 0; st.n_lost = 0; st.n_tracked_landmarks = 0; st.n_cur = 0; st.n_active = 0; st.al_n = 0;
    st.aligner_valid = 0; st.n_after_prune = 0; st.n_recovered = 0; st.n_new = 0; st.track_calls = 0;
    st.al_inliers = 0; st.al_outliers = 0; st.al_iterations = 0; st.al_converged = 0; st.al_total_error = 0;
    vslam_frame_info& info = b.info[s];
    info.status_at_start = st.status; info.fallback = 0; info.track_broken = 0;
  }
}

__device__ __forceinline__ void stage_body(const DevCfg& c, const DevBuf& b, int stage, int arg, const StageIo& io, FrameShared& sh, unsigned char* arena, int bx, int gx) {
  const int s = b.s0 + xcd_local_stream(bx, gx, b.xcd_rot), tid = threadIdx.x;
  if (!vs_active(b, s)) return;
  StreamState& st = b.st[s];
  vslam_frame_info& info = b.info[s];
  if (io.set_flags) { if (tid == 0) stage_apply_set(st, io); __syncthreads(); }
  const int f = st.frame_count;
  const int pb_prev = st.cur, pb_cur = st.cur ^ 1;
  const bool has_prev = st.has_prev != 0;
  if (tid == 0) {
    sh.n_trk = st.n_trk; sh.n_lost = st.n_lost; sh.n_lm = st.n_tracked_landmarks; sh.n_cur = st.n_cur; sh.n_cand = 0;
    sh.E = st.al_total_error; sh.inl = st.al_inliers; sh.outl = st.al_outliers; sh.its = 0; sh.conv = 0; sh.flag = 0;
  }
  __syncthreads();
  if (stage == VS_STAGE_TRACK && has_prev) {
    const double tau = st.tau_track;
    const unsigned long long t0 = wall_clock64();
    wg_track_resolve(c, b, s, sh, pb_prev, arena, st.win, tau, st.tau_tri, arg);
    if (tid == 0) {
      st.ticks[0] += wall_clock64() - t0;
      st.n_trk = sh.n_trk; st.n_lost = sh.n_lost; st.n_tracked_landmarks = sh.n_lm; st.aligner_valid = 0; st.tau_gen = tau;
      st.al_n = 0; st.track_calls += 1;
      info.n_tracked = sh.n_trk; info.n_lost = sh.n_lost; info.n_tracked_landmarks = sh.n_lm; info.track_attempts = st.track_calls;
      info.aligner_ran = 0;
    }
  } else if (stage == VS_STAGE_ALIGN && has_prev) {
    double T0[12];
    for (int k = 0; k < 12; ++k) T0[k] = st.prior[k];
    const unsigned long long t0 = wall_clock64();
    wg_align(c, b, s, sh, pb_prev, arg != 0, T0, arena, VS_ARENA);
    if (tid == 0) {
      st.ticks[1] += wall_clock64() - t0;
      st.al_n = sh.n_trk; st.al_inliers = sh.inl; st.al_outliers = sh.outl; st.al_iterations = sh.its; st.al_converged = sh.conv;
      st.al_total_error = sh.E; st.aligner_valid = 1;
      for (int k = 0; k < 12; ++k) st.al_T[k] = sh.T[k];
      for (int k = 0; k < 36; ++k) st.al_H[k] = sh.H[k];
      info.aligner_ran = 1; info.aligner_iterations = sh.its; info.aligner_converged = sh.conv; info.n_inliers = sh.inl;
      info.n_outliers = sh.outl; info.total_error = sh.E;
    }
  } else if (stage == VS_STAGE_PRUNE_RECOVER) {
    if (tid == 0) set_pose(c, b, s, f, st.pose);   // Frame::setRobotToWorld happened on the host side
    __syncthreads();
    if (has_prev) {
      wg_prune(c, b, s, sh, pb_prev, pb_cur, st.aligner_valid != 0);
      const int n_after = sh.n_cur;
      int n_rec = 0;
      const unsigned long long t0 = wall_clock64();
      if (arg) { wg_recover(c, b, s, sh, pb_prev, pb_cur, hpose_of(c, b, s, f) + 12, st.tau_gen, st.tau_tri, arena); n_rec = sh.flag; }
      if (tid == 0) {
        if (arg) st.ticks[2] += wall_clock64() - t0;
        st.n_cur = sh.n_cur; st.n_after_prune = n_after; st.n_recovered = n_rec;
        info.n_after_prune = n_after; info.n_recovered = n_rec; info.n_points = sh.n_cur;
      }
    }
  } else if (stage == VS_STAGE_PRUNE_PROJECT) {
    // _prunePoints, then the projection of the lost landmarks; their descriptors are computed by the wide k_recover_brief (one wavefront
    // per projected point over the whole chip instead of eight wavefronts behind one CU's memory pipe), which reads what it needs from fc
    if (tid == 0) set_pose(c, b, s, f, st.pose);   // Frame::setRobotToWorld happened on the host side
    __syncthreads();
    if (has_prev) {
      wg_prune(c, b, s, sh, pb_prev, pb_cur, st.aligner_valid != 0);
      const int n_after = sh.n_cur;
      __syncthreads();
      wg_recover_project(c, b, s, sh.n_lost, pb_prev, hpose_of(c, b, s, f) + 12);
      if (tid == 0) {
        st.n_cur = n_after; st.n_after_prune = n_after; st.n_recovered = 0;
        st.fc.n_lost = sh.n_lost; st.fc.tau_gen = st.tau_gen; st.fc.tau_tri = st.tau_tri;
        info.n_after_prune = n_after; info.n_recovered = 0; info.n_points = n_after;
      }
    } else if (tid == 0) {
      st.fc.n_lost = 0;
    }
  } else if (stage == VS_STAGE_RECOVER_APPEND) {
    if (has_prev) {
      const unsigned long long t0 = wall_clock64();
      wg_recover_append(c, b, s, sh, pb_prev, pb_cur);
      const int n_rec = sh.flag;
      if (tid == 0) {
        st.ticks[2] += wall_clock64() - t0;
        st.n_cur = sh.n_cur; st.n_recovered = n_rec;
        info.n_recovered = n_rec; info.n_points = sh.n_cur;
      }
    }
    if (arg & 2) {
      // the frame's point list is final: publish it to the history ring here, so that the landmark kernel of the next call (vslam_compute of a
      // one-stream context: lm_teams_body beside the stereo stage, k_stage_lm) finds what wg_update_points would have published first
      __syncthreads();
      wg_publish_history(c, b, s, sh.n_cur, pb_cur, f);
      if (tid == 0) { st.fc.n_cur = sh.n_cur; st.fc.lm_pb = pb_cur; st.fc.lm_f = f; }
    }
  } else if (stage == VS_STAGE_UPDATE || stage == VS_STAGE_STEREO || stage == VS_STAGE_COMPUTE || stage == VS_STAGE_STEREO_COUNT) {
    if (stage == VS_STAGE_STEREO_COUNT) {
      // _number_of_active_landmarks without the refinement: a point is active iff its track is long enough for a landmark
      const PtView cvc = pts_of(c, b, s, pb_cur);
      int active = 0;
      for (int i = tid; i < sh.n_cur; i += VS_WG) active += cvc.meta[(size_t)i * META + M_TLEN] >= c.c.minimum_track_length_for_landmark_creation ? 1 : 0;
      int total;
      block_exclusive_scan(active, sh.scan, &total);
      if (tid == 0) { st.n_active = total; info.n_active_landmarks = total; }
      __syncthreads();
    } else
    if (stage != VS_STAGE_STEREO) {
      const unsigned long long t0 = wall_clock64();
      wg_update_points(c, b, s, sh, pb_cur, f, arena);
      if (tid == 0) { st.n_active = sh.n_lm; info.n_active_landmarks = sh.n_lm; st.ticks[3] += wall_clock64() - t0; }
    }
    if (stage == VS_STAGE_COMPUTE) {     // the two launches of compute() in one: the shared scalars start over as a new launch would read them
      __syncthreads();
      if (tid == 0) { sh.n_lm = st.n_tracked_landmarks; sh.n_cand = 0; sh.flag = 0; sh.n_cur = st.n_cur; }
      __syncthreads();
    }
    if (stage != VS_STAGE_UPDATE) {
      const unsigned long long t0 = wall_clock64();
      wg_stereo(c, b, s, sh, pb_cur, st.tau_tri, f, arena, VS_ARENA);
      if (tid == 0) {
        st.ticks[4] += wall_clock64() - t0;
        const double* c2w = hpose_of(c, b, s, f);
        *pts_of(c, b, s, pb_cur).n = sh.n_cur;
        st.n_cur = sh.n_cur; st.n_new = sh.n_cand;
        st.n_tracked_landmarks_prev = st.n_active;
        st.frame_count = f + 1; st.has_prev = 1; st.cur = pb_cur;
        info.frame_index = f + 1; info.status = st.status;
        info.n_keypoints_left = b.n_kp[s * 2]; info.n_keypoints_right = b.n_kp[s * 2 + 1];
        int rl = 0, rr = 0;
        for (int r = 0; r < c.n_regions; ++r) { rl += b.iinfo[s].raw_count[0][r]; rr += b.iinfo[s].raw_count[1][r]; info.thresholds[r] = b.iinfo[s].thr_after[r]; }
        info.n_detected_left = rl; info.n_detected_right = rr;
        info.n_new_stereo = sh.n_cand; info.n_points = sh.n_cur; info.window_pixels = st.win; info.error_flags = st.error_flags;
        info.tau_track = st.tau_track; info.tau_triangulation = st.tau_tri;
        for (int k = 0; k < 12; ++k) { info.camera_left_to_world[k] = c2w[k]; info.previous_to_current[k] = st.prior[k]; }
        if (f < VS_POSE_LOG) { double* pl = b.pose_log + ((size_t)s * VS_POSE_LOG + f) * 12; for (int k = 0; k < 12; ++k) pl[k] = c2w[k]; }
      }
    }
  }
  if (io.report && s == io.report_stream) {
    // the stage's results for the caller, packed by this workgroup into the pinned host buffer (kernels_report.h): the host
    // synchronises the frame queue once and reads them there
    __threadfence();
    __syncthreads();
    report_body(c, b, s, io.report, io.report_in_progress, io.seq, io.L, io.out, (size_t)tid, (size_t)blockDim.x, true, tid, (int)blockDim.x);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) report_publish(io.out, io.seq);
  }
}
__global__ __launch_bounds__(VS_WG) void k_stage(const DevCfg c, const DevBuf b, int stage, int arg, const StageIo io) {
  __shared__ FrameShared sh;
  __shared__ __align__(16) unsigned char arena[VS_ARENA];
  stage_body(c, b, stage, arg, io, sh, arena, blockIdx.x, gridDim.x);
}
// vslam_compute of a context whose vslam_prune_recover has published the frame's history: the n stream workgroups run the stage (STEREO_COUNT: the
// stereo sweep with the active landmarks counted, not refined), G more workgroups per stream refine the landmarks beside it (lm_teams_body) — one launch
__global__ __launch_bounds__(VS_WG) void k_stage_lm(const DevCfg c, const DevBuf b, int stage, int arg, const StageIo io, int n, int G) {
  __shared__ FrameShared sh;
  __shared__ __align__(16) unsigned char arena[VS_ARENA];
  if ((int)blockIdx.x < n) { stage_body(c, b, stage, arg, io, sh, arena, blockIdx.x, n); return; }
  const int i = (int)blockIdx.x - n, sl = i / G;
  const int s = b.s0 + sl;
  if (!vs_active(b, s)) return;
  lm_teams_body(c, b, s, i - sl * G, G, 0, true, arena, sh.flag, sh.n_proj, sh.scan);
}

// vslam_reset_stream, asynchronous: the stream state has an image-pipeline half (the detector thresholds, written by k_emit
// and read by k_fast_box on the image stream) and a tracker half (everything else, frame stream); each half is reset by a
// one-thread kernel queued on the HIP stream that owns it, so a stream restarts between two frames without a host sync.
struct ResetList { int32_t n; int32_t ids[63]; };   // streams restarted between two frames, one launch per half of the state
__global__ void k_reset_stream_img(const DevCfg c, const DevBuf b, const ResetList l) {
  if ((int)threadIdx.x >= l.n) return;
  const int s = l.ids[threadIdx.x];
  StreamState& st = b.st[s];
  for (int r = 0; r < VSLAM_MAX_REGIONS; ++r) st.thr[r] = r < c.n_regions ? c.c.detector_threshold_minimum : 0;
  atomicAnd(&st.error_flags, ~1);     // bit 0 (keypoint capacity) is raised by k_emit on this HIP stream
}
__global__ void k_reset_stream_trk(const DevCfg c, const DevBuf b, const ResetList l) {
  if ((int)threadIdx.x >= l.n) return;
  const int s = l.ids[threadIdx.x];
  StreamState& st = b.st[s];
  st.status = VSLAM_LOCALIZING; st.win = c.c.maximum_projection_tracking_distance_pixels; st.frame_count = 0; st.has_prev = 0;
  st.n_tracked_landmarks_prev = 0; st.cur = 0; st.aligner_valid = 0; atomicAnd(&st.error_flags, 1);
  st.tau_track = c.c.minimum_descriptor_distance_tracking; st.tau_tri = 0.1 * 256;
  tf_identity(st.prior); tf_identity(st.pose);
  st.by_appearance = 0; st.n_trk = 0; st.n_lost = 0; st.n_tracked_landmarks = 0;
  st.al_n = 0; st.al_inliers = 0; st.al_outliers = 0; st.al_iterations = 0; st.al_converged = 0; st.al_wsize = 0; st.al_total_error = 0;
  st.tau_gen = 0; st.n_cur = 0; st.n_active = 0; st.n_after_prune = 0; st.n_recovered = 0; st.n_new = 0; st.track_calls = 0;
  b.n_points[s * 2] = 0; b.n_points[s * 2 + 1] = 0;
  vslam_frame_info& info = b.info[s];
  unsigned char* p = reinterpret_cast<unsigned char*>(&info);
  for (size_t k = 0; k < sizeof(vslam_frame_info); ++k) p[k] = 0;
}
// the current pose of every stream (camera_left_to_world of the frame just processed) -> dst[stream][12]
__global__ void k_gather_poses(const DevBuf b, int n, double* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 12) dst[i] = b.st[b.s0 + i / 12].pose[i % 12];
}

// setters of the tracker-owned state (one thread)
struct D12 { double v[12]; };   // a transform passed by value as a kernel argument (no staging buffer)
__global__ void k_set_tracker_state(const DevBuf b, int s, int status, int win, double tau, const D12 prior) {
  StreamState& st = b.st[s];
  st.status = status; st.win = win; st.tau_track = tau;
  for (int k = 0; k < 12; ++k) st.prior[k] = prior.v[k];
}
__global__ void k_set_pose(const DevBuf b, int s, const D12 pose) {
  StreamState& st = b.st[s];
  for (int k = 0; k < 12; ++k) st.pose[k] = pose.v[k];
}
