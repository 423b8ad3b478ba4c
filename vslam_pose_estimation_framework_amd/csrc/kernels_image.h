// kernels_image.h — per-image kernels: FAST-9/16 + NMS + 9x9 box image (K1), keypoint emission +
// CSR index + threshold controller (K2), BRIEF-32 (K3), N x M 2-NN (K6).  gfx950, wave64.
//
// Reference code replaced (paths relative to the reference root):
//   K1  cv::FastFeatureDetector::detect per detector region     base_framepoint_generator.cpp:12-25,362-367
//   K2  detectKeypoints bookkeeping + adjustDetectorThresholds    base_framepoint_generator.cpp:377-429,440-459
//       IntensityFeatureMatcher::setFeatures                      intensity_feature_matcher.cpp:48-70
//   K3  _descriptor_extractor->compute (BRIEF-32)                 base_framepoint_generator.cpp:431-438
//   K6  matcher->knnMatch(k=2)                                    stereo_framepoint_generator.cpp:168-206
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "../../include/vslam_brief_pattern.h"
#include "../../include/vslam_orb_pattern.h"

__constant__ int8_t c_brief[256][4] = VSLAM_BRIEF_PATTERN_INIT;
__constant__ int8_t c_orb[256][4] = VSLAM_ORB_PATTERN_INIT;

// ---- indexing helpers ------------------------------------------------------------------------
__device__ __forceinline__ size_t ix_side(const DevCfg& c, int s, int d) { return (size_t)s * 2 + d; }
__device__ __forceinline__ uint16_t* box_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.box + ix_side(c, s, d) * (size_t)c.c.rows * c.bstride;
}
__device__ __forceinline__ uint8_t* score_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.score8 + ix_side(c, s, d) * (size_t)c.c.rows * c.bstride;
}
__device__ __forceinline__ unsigned long long* mask_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.mask + ix_side(c, s, d) * (size_t)c.c.rows * c.TX;
}
__device__ __forceinline__ int16_t* kpxy_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.kp_xy + ix_side(c, s, d) * (size_t)c.NMAX * 2;
}
__device__ __forceinline__ uint8_t* kpscore_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.kp_score + ix_side(c, s, d) * (size_t)c.NMAX;
}
__device__ __forceinline__ uint8_t* desc_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.desc + ix_side(c, s, d) * (size_t)c.NMAX * 32;
}
__device__ __forceinline__ int32_t* rowcell_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.rowcell + ix_side(c, s, d) * (size_t)c.c.rows * (c.CW + 1);
}
__device__ __forceinline__ uint8_t* used_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.used + ix_side(c, s, d) * (size_t)c.NMAX;
}
__device__ __forceinline__ int32_t* kill_of(const DevCfg& c, const DevBuf& b, int s, int d) {
  return b.kill + ix_side(c, s, d) * (size_t)c.NMAX;
}

// ---- block-wide exclusive scan of one int per thread (blockDim.x multiple of 64, <= 1024) ------
// returns the exclusive prefix; *total receives the block sum.  sh must hold 17 ints.
__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // protect sh from a previous use
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < nw; ++i) { const int t = sh[i]; sh[i] = acc; acc += t; }
    sh[16] = acc;
  }
  __syncthreads();
  *total = sh[16];
  return sh[w] + inc - v;
}

// ==============================================================================================
// K1: FAST-9/16 score + strict 3x3 NMS -> 1 bit/pixel corner mask (+ sparse u8 scores), fused with
// the 9x9 box-sum image BRIEF samples.  One 64 x VS_TILE_H (64) output tile per 256-thread workgroup; the u8
// tile with a 4 px halo (FAST ring 3 + NMS 1 == box radius 4) is staged once in LDS.
// HBM traffic per pixel: 1 B read, 2 B box write, 1/8 B mask write, sparse scores.
// ==============================================================================================
// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2), so the linear block id
// is re-labelled such that (1) CONSECUTIVE tiles (x fastest, then y, then side) run on the same XCD — the halo lines neighbouring
// tiles share are fetched into one L2 only — and (2) ALL tiles of stream s run on XCD s mod 8, the XCD of the stream's k_emit
// workgroups, of its candidate / distance blocks (xcd_stream_block) and of its frame workgroup: what one kernel of the step
// leaves in that L2 (corner masks, keypoints, row / cell CSR, descriptors) is what the next one reads.  gridDim.z = 2 * streams
// (z = 2 * stream + side).  Speed only, never correctness.
// mono: one image per stream (RGB-D mode: gridDim.z = streams, the returned z is the stream) instead of two (z = 2 * stream + side)
__device__ __forceinline__ void xcd_tile(int* tx, int* ty, int* tz, int mono = 0, int rot = 0) {
  const int gx = gridDim.x, gy = gridDim.y;
  const int tps = mono ? gx * gy : gx * gy * 2;         // tiles of one stream (both images)
  const int ns = mono ? gridDim.z : gridDim.z >> 1;
  int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int per = (ns >> 3) * tps;                      // tiles per XCD among the first 8 * (ns / 8) streams
  // block lin runs on physical XCD (lin + q0) % 8 (q0: the queue's); rot = (q0 - s0) & 7 makes x the stream's residue class
  if (lin < (per << 3)) { const int x = (lin + rot) & 7, j = lin >> 3, q = j / tps; lin = (x + 8 * q) * tps + (j - q * tps); }
  *tx = lin % gx;
  const int r = lin / gx;
  *ty = r % gy;
  *tz = r / gy;
}

// threshold of the detector region whose FAST-valid area (ROI minus 3 px) contains (x,y), or -1
__device__ __forceinline__ int region_threshold(const DevCfg& c, const int32_t* thr, int x, int y) {
  for (int r = 0; r < c.n_regions; ++r) {
    const DevRegion& R = c.regions[r];
    if (x >= R.x + 3 && x < R.x + R.w - 3 && y >= R.y + 3 && y < R.y + R.h - 3) return thr[r];
  }
  return -1;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// FAST-9/16 without branches, two pixels per lane in packed i16 (v_pk_sub/min/max_i16): with
// A = max over the 16 nine-arcs of min(d), Bm = min over arcs of max(d) (d = centre - ring pixel), the pixel is
// a corner iff A > t or -Bm > t, and cornerScore = max(t, A, -Bm) - 1 — one sliding min/max table gives both.
__device__ __forceinline__ void fast_pair_scores(const uint8_t (*t)[80], int ly0, int lx0, int ly1, int lx1, int thr0, int thr1,
                                                 int* s0, int* s1) {
  const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const s16x2 v = {(short)t[ly0][lx0], (short)t[ly1][lx1]};
  s16x2 d[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const s16x2 p = {(short)t[ly0 + dy[k]][lx0 + dx[k]], (short)t[ly1 + dy[k]][lx1 + dx[k]]};
    d[k] = v - p;
  }
  s16x2 mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { mn2[k] = pk_min(d[k], d[(k + 1) & 15]); mx2[k] = pk_max(d[k], d[(k + 1) & 15]); }
#pragma unroll
  for (int k = 0; k < 16; ++k) { mn4[k] = pk_min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = pk_max(mx2[k], mx2[(k + 2) & 15]); }
  s16x2 A = {-1000, -1000}, Bm = {1000, 1000};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    A = pk_max(A, pk_min(pk_min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]));
    Bm = pk_min(Bm, pk_max(pk_max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]));
  }
  const int a0 = A.x, b0 = -Bm.x, a1 = A.y, b1 = -Bm.y;
  *s0 = (thr0 >= 0 && (a0 > thr0 || b0 > thr0)) ? max(thr0, max(a0, b0)) - 1 : 0;
  *s1 = (thr1 >= 0 && (a1 > thr1 || b1 > thr1)) ? max(thr1, max(a1, b1)) - 1 : 0;
}

// The candidate queue is DYNAMIC shared memory (VS_FB_DYN_LDS bytes at every launch): with all 22.6 KB declared statically the
// compiler sees an LDS-bound occupancy of 7 wavefronts per SIMD and allows itself 72 VGPRs (it used 65); with the queue out of
// sight it has to honour 8 wavefronts = 64 VGPRs (tools/probe/cosched.hip: a 64-VGPR neighbour is what lets a 128-VGPR wavefront
// of another kernel in when one workgroup leaves).
#define VS_FB_QCAP ((VS_TILE_H + 2) * 66)
#define VS_FB_HS_BYTES ((VS_TILE_H + 8) * VS_TILE_W * 2)
#define VS_FB_DYN_LDS (((VS_FB_QCAP + 256) * 2) > VS_FB_HS_BYTES ? ((VS_FB_QCAP + 256) * 2) : VS_FB_HS_BYTES)
__global__ __launch_bounds__(256, 8) void k_fast_box(const DevCfg c, const DevBuf b) {
  extern __shared__ __align__(16) unsigned char fb_dyn[];
  __shared__ __align__(16) uint8_t tile[VS_TILE_H + 8][80];
  __shared__ __align__(4) uint8_t sc[VS_TILE_H + 2][68];
  __shared__ int32_t s_thr[VSLAM_MAX_REGIONS];
  // candidate queue of the tile (packed: the scoring pass fills whole wavefronts — per-wavefront queues were measured slower,
  // they leave every wavefront a partly filled scoring pass); the 256 slots behind it take the stores of lanes without a
  // candidate, so that the four queue writes of a pretest pass need no exec-mask branches
  constexpr int QCAP = VS_FB_QCAP;
  uint16_t* queue = reinterpret_cast<uint16_t*>(fb_dyn);
  // the horizontal 9-sums of the box pass live in the SAME memory: the queue is dead once the NMS has read it, the sums are built after
  // that (one more barrier, 7-9 KB less LDS).  With that a 64-row tile costs 20 052 B, 8 workgroups = 32 wavefronts still share a CU, and the
  // taller tile stages 72 rows for 64 instead of 56 for 48 (measured: 0.309 -> 0.275 ms back to back at 160 streams; 32/48/80/96 rows slower)
  uint16_t (*hs)[VS_TILE_W] = reinterpret_cast<uint16_t (*)[VS_TILE_W]>(fb_dyn);
  __shared__ unsigned long long lmask[VS_TILE_H];
  __shared__ int qn;
  int tx, ty, tz;
  xcd_tile(&tx, &ty, &tz, c.mono, b.xcd_rot);
  const int s = b.s0 + (c.mono ? tz : (tz >> 1)), side = c.mono ? 0 : (tz & 1);
  if (!vs_active(b, s)) return;
  const int x0 = tx * VS_TILE_W, y0 = ty * VS_TILE_H;
  const int rows = c.c.rows, cols = c.c.cols;
  const uint8_t* img = b.img[side] + (size_t)s * b.img_stream_stride;
  const int stride = b.img_row_stride;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < c.n_regions) s_thr[tid] = min(max(b.st[s].thr[tid], 0), 255);
  if (tid == 0) qn = 0;
  if (tid < VS_TILE_H) lmask[tid] = 0ull;
  // scores of non-candidates are zero: clear the whole score tile with dword stores, candidates overwrite theirs
  for (int i = tid; i < (VS_TILE_H + 2) * 17; i += 256) reinterpret_cast<uint32_t*>(&sc[0][0])[i] = 0u;
  // ---- stage the (64+8) x (H+8) u8 tile.  Interior tiles: aligned dwords, all of a thread's loads in flight before
  // the first LDS store; tiles touching the left/right image border: clamped bytes. ------------------------------------
  const bool aligned = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0);
  constexpr int NST = ((VS_TILE_H + 8) * 18 + 255) / 256;
  if (aligned && x0 >= 4 && x0 + 68 <= cols) {
    uint32_t v[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 256 * u;
      const int r = i / 18, q = i - r * 18;
      const int gy = min(max(y0 - 4 + r, 0), rows - 1);
      v[u] = 0;
      if (i < (VS_TILE_H + 8) * 18) v[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(img + (size_t)gy * stride + (x0 - 4 + 4 * q)));
    }
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 256 * u;
      const int r = i / 18, q = i - r * 18;
      if (i < (VS_TILE_H + 8) * 18) *reinterpret_cast<uint32_t*>(&tile[r][4 * q]) = v[u];
    }
  } else {
    for (int i = tid; i < (VS_TILE_H + 8) * 18; i += 256) {
      const int r = i / 18, q = i - r * 18;
      const int gy = min(max(y0 - 4 + r, 0), rows - 1), gx0 = x0 - 4 + 4 * q;
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) v |= (uint32_t)img[(size_t)gy * stride + min(max(gx0 + k, 0), cols - 1)] << (8 * k);
      *reinterpret_cast<uint32_t*>(&tile[r][4 * q]) = v;
    }
  }
  // one detector region covers the whole 66 x (H+2) score region of most tiles: its threshold is then tile-uniform
  __syncthreads();
  // Exactly one detector region's FAST-valid area (ROI minus 3 px) reaches into the score region: its threshold holds for every
  // valid pixel of the tile, and validity is a rectangle [vx0, vx1) x [vy0, vy1) in region coordinates — the whole region for
  // interior tiles (uni_full), clipped for tiles on the image border or on a ROI edge.  Two or more regions: per-pixel lookup.
  int uni_thr = -2, vx0 = 0, vx1 = 0, vy0 = 0, vy1 = 0;
  bool uni_full = false;
  {
    int hits = 0;
    for (int r = 0; r < c.n_regions; ++r) {
      const DevRegion& R = c.regions[r];
      const int ax0 = max(R.x + 3, x0 - 1), ax1 = min(R.x + R.w - 3, x0 + 65), ay0 = max(R.y + 3, y0 - 1), ay1 = min(R.y + R.h - 3, y0 + VS_TILE_H + 1);
      if (ax0 < ax1 && ay0 < ay1) {
        if (hits == 0) { uni_thr = s_thr[r]; vx0 = ax0 - (x0 - 1); vx1 = ax1 - (x0 - 1); vy0 = ay0 - (y0 - 1); vy1 = ay1 - (y0 - 1); }
        ++hits;
      }
    }
    if (hits != 1) uni_thr = -2;
    uni_full = uni_thr != -2 && vx0 == 0 && vx1 == 66 && vy0 == 0 && vy1 == VS_TILE_H + 2;
  }
  // ---- FAST on the 66 x (H+2) score region (tile + 1 px NMS halo), two passes:
  //  A) every pixel: the high-speed test (a 9-arc of 16 contains two ADJACENT compass points, i.e. one of {N,S} and one
  //     of {E,W}, on the dark or on the bright side) -> ~1 pixel in 8 survives, queued in LDS (one atomic per wave-row)
  //  B) queued pixels only, two per lane in packed i16: exact corner test + cornerScore
  auto pretest = [&](int r, int cc, bool valid = true) {
    const int thr = !valid ? -1 : (uni_full ? uni_thr : region_threshold(c, s_thr, x0 - 1 + cc, y0 - 1 + r));
    bool cand = false;
    if (thr >= 0) {
      const int ly = r + 3, lx = cc + 3;
      const int v = tile[ly][lx];
      const int d0 = v - tile[ly + 3][lx], d4 = v - tile[ly][lx + 3], d8 = v - tile[ly - 3][lx], d12 = v - tile[ly][lx - 3];
      const int dk = min(max(d0, d8), max(d4, d12));
      const int br = max(min(d0, d8), min(d4, d12));
      cand = dk > thr || br < -thr;
    }
    const unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&qn, __popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (cand) queue[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)((r << 8) | cc);
    }
  };
#ifndef VS_PROBE
#define VS_PROBE 0
#endif
  if (VS_PROBE & 1) { /* probe build: no pretest, no candidates */ } else
  if (uni_thr != -2) {
    // uniform threshold (the usual tile): four pixels per lane from aligned dwords, the compass differences in packed
    // i16 (v_perm_b32 unpacks, v_pk_sub/min/max_i16).  16 lanes per region row (dwords 1..16 = region columns 1..64), four rows
    // per wavefront and pass: no division in the index arithmetic.  The four candidate predicates are the sign bits of the packed
    // differences, balloted as they are (v_cmp_lt_i16 / _i32): one queue reservation per wave-pass.
    const s16x2 thr2 = {(short)uni_thr, (short)uni_thr};
    // Tile heights that are a multiple of 16: the H tile rows (region rows 1 .. H) are whole passes of every wavefront, and the
    // NMS halo ring (region rows 0 and H + 1, region columns 0 and 65) goes through the one-pixel test, spread over all lanes.
    constexpr bool RING = (VS_TILE_H % 16) == 0;
    constexpr int RBASE = RING ? 1 : 0, RCOUNT = RING ? VS_TILE_H : VS_TILE_H + 2;
#pragma unroll 1
    for (int pass = 0; pass < (RCOUNT + 15) / 16; ++pass) {
      if (pass * 16 + w * 4 >= RCOUNT) continue;          // wave-uniform: nothing of this wavefront in the pass
      const int r = RBASE + pass * 16 + (tid >> 4), q = (tid & 15) + 1;
      uint32_t ng0 = 0, ng1 = 0;
      if (r < RBASE + RCOUNT) {
        const uint32_t* rowp = reinterpret_cast<const uint32_t*>(&tile[r + 3][0]);
        const uint32_t C = rowp[q], C0 = rowp[q - 1], C2 = rowp[q + 1];
        const uint32_t N = reinterpret_cast<const uint32_t*>(&tile[r][0])[q], S = reinterpret_cast<const uint32_t*>(&tile[r + 6][0])[q];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const uint32_t sel = half ? 0x0c030c02u : 0x0c010c00u;
          const uint32_t vv = __builtin_amdgcn_perm(0u, C, sel), nn = __builtin_amdgcn_perm(0u, N, sel), ss = __builtin_amdgcn_perm(0u, S, sel);
          const uint32_t ee = __builtin_amdgcn_perm(C2, C, half ? 0x0c060c05u : 0x0c040c03u);
          const uint32_t ww = __builtin_amdgcn_perm(C, C0, half ? 0x0c040c03u : 0x0c020c01u);
          // dark side: some of {N,S} and some of {E,W} darker than v - thr  <=>  max(min(n,s), min(e,w)) < v - thr
          // bright side: min(max(n,s), max(e,w)) > v + thr                  (values 0..255 +- thr: no i16 overflow)
          const s16x2 v = __builtin_bit_cast(s16x2, vv), n = __builtin_bit_cast(s16x2, nn), s_ = __builtin_bit_cast(s16x2, ss);
          const s16x2 e = __builtin_bit_cast(s16x2, ee), w_ = __builtin_bit_cast(s16x2, ww);
          const s16x2 lo = pk_max(pk_min(n, s_), pk_min(e, w_)), hi = pk_min(pk_max(n, s_), pk_max(e, w_));
          const uint32_t neg = __builtin_bit_cast(uint32_t, (s16x2)(lo - (v - thr2))) | __builtin_bit_cast(uint32_t, (s16x2)((v + thr2) - hi));
          if (half) ng1 = neg; else ng0 = neg;
        }
      }
      bool p0 = (short)(ng0 & 0xffffu) < 0, p1 = (int)ng0 < 0, p2 = (short)(ng1 & 0xffffu) < 0, p3 = (int)ng1 < 0;
      if (!uni_full) {   // clipped validity rectangle (wave-uniform branch): region columns 4q-3 .. 4q of region row r
        const bool rok = r >= vy0 && r < vy1;
        const int c0 = 4 * q - 3;
        p0 = p0 && rok && c0 >= vx0 && c0 < vx1;         p1 = p1 && rok && c0 + 1 >= vx0 && c0 + 1 < vx1;
        p2 = p2 && rok && c0 + 2 >= vx0 && c0 + 2 < vx1; p3 = p3 && rok && c0 + 3 >= vx0 && c0 + 3 < vx1;
      }
      const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1), m2 = __ballot(p2), m3 = __ballot(p3);
      const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
      if (n0 + n1 + n2 + n3) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&qn, n0 + n1 + n2 + n3);
        base = __builtin_amdgcn_readfirstlane(base);
        const int ent = (r << 8) | (4 * q - 3);
        auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
        const int trash = QCAP + tid;
        queue[p0 ? base + below(m0) : trash] = (uint16_t)ent;
        queue[p1 ? base + n0 + below(m1) : trash] = (uint16_t)(ent + 1);
        queue[p2 ? base + n0 + n1 + below(m2) : trash] = (uint16_t)(ent + 2);
        queue[p3 ? base + n0 + n1 + n2 + below(m3) : trash] = (uint16_t)(ent + 3);
      }
    }
    if (RING) {
      constexpr int NRING = 2 * 66 + 2 * VS_TILE_H;
      for (int i0 = 0; i0 < NRING; i0 += 256) {
        const int i = i0 + tid;
        int r = 0, cc = 0;
        if (i < 66) { r = 0; cc = i; }
        else if (i < 132) { r = VS_TILE_H + 1; cc = i - 66; }
        else { const int k = i - 132; r = 1 + (k >> 1); cc = (k & 1) ? 65 : 0; }
        pretest(r, cc, i < NRING);
      }
    } else {
      // region columns 0 and 65 (the NMS halo left and right of the tile): (H+2)*2 pixels on the wavefronts the last pass leaves idle
      if (tid >= 64 && tid < 64 + (VS_TILE_H + 2) * 2) { const int i = tid - 64; pretest(i >> 1, (i & 1) ? 65 : 0); }
    }
  } else {
    for (int r = w; r < VS_TILE_H + 2; r += 4) pretest(r, lane);
    // the two halo columns 64, 65 of every row: (H+2)*2 pixels
    if (tid < (VS_TILE_H + 2) * 2) pretest(tid >> 1, 64 + (tid & 1));
    static_assert((VS_TILE_H + 2) * 2 <= 256, "one pass over the halo columns");
  }
  __syncthreads();
  if (!(VS_PROBE & 2)) {
    const int nq = qn;
    for (int q = tid; 2 * q < nq; q += 256) {
      const int e0 = queue[2 * q], e1 = (2 * q + 1 < nq) ? queue[2 * q + 1] : e0;
      const int r0 = e0 >> 8, c0 = e0 & 255, r1 = e1 >> 8, c1 = e1 & 255;
      int sv0, sv1;
      // queued pixels are valid pixels: with one region in reach their threshold is that region's
      const int t0 = uni_thr != -2 ? uni_thr : region_threshold(c, s_thr, x0 - 1 + c0, y0 - 1 + r0);
      const int t1 = uni_thr != -2 ? uni_thr : region_threshold(c, s_thr, x0 - 1 + c1, y0 - 1 + r1);
      fast_pair_scores(tile, r0 + 3, c0 + 3, r1 + 3, c1 + 3, t0, t1, &sv0, &sv1);
      sc[r0][c0] = (uint8_t)sv0;
      if (2 * q + 1 < nq) sc[r1][c1] = (uint8_t)sv1;
    }
  }
  __syncthreads();
  // ---- strict 3x3 NMS, candidate-driven: only scored pixels look at their 8 neighbours; survivors set their bit in the
  // LDS row masks and write their (sparse) score ---------------------------------------------------------------------
  unsigned long long* mask = mask_of(c, b, s, side);
  uint8_t* score8 = score_of(c, b, s, side);
  uint16_t* box = box_of(c, b, s, side);
  {
    const int nq = qn;
    for (int q = tid; q < nq; q += 256) {
      const int e0 = queue[q];
      const int r = e0 >> 8, cc = e0 & 255;
      if (r < 1 || r > VS_TILE_H || cc < 1 || cc > 64) continue;      // halo pixels only serve as neighbours
      const int v = sc[r][cc];
      if (v == 0) continue;
      const bool keep = v > sc[r - 1][cc - 1] && v > sc[r - 1][cc] && v > sc[r - 1][cc + 1] && v > sc[r][cc - 1] &&
                        v > sc[r][cc + 1] && v > sc[r + 1][cc - 1] && v > sc[r + 1][cc] && v > sc[r + 1][cc + 1];
      if (keep) {
        atomicOr(&lmask[r - 1], 1ull << (cc - 1));
        const int gy = y0 + r - 1;
        if (gy < rows) score8[(size_t)gy * c.bstride + (x0 + cc - 1)] = (uint8_t)v;
      }
    }
  }
  __syncthreads();   // every reader of the candidate queue is done: its memory becomes the horizontal sums
  // ---- horizontal 9-sums, four outputs per thread from three aligned dwords -----------------------------------------
  const bool want_box = !(VS_PROBE & 4) && c.c.descriptor_type == VSLAM_DESCRIPTOR_BRIEF;   // ORB samples the Gaussian image instead
  if (want_box)
  for (int i = tid; i < (VS_TILE_H + 8) * 16; i += 256) {
    const int r = i >> 4, q = i & 15;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(&tile[r][4 * q]);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    const uint32_t s0 = __builtin_amdgcn_sad_u8(w0, 0u, __builtin_amdgcn_sad_u8(w1, 0u, w2 & 255u));
    const uint32_t s1 = s0 - (w0 & 255u) + ((w2 >> 8) & 255u);
    const uint32_t s2 = s1 - ((w0 >> 8) & 255u) + ((w2 >> 16) & 255u);
    const uint32_t s3 = s2 - ((w0 >> 16) & 255u) + (w2 >> 24);
    *reinterpret_cast<uint2*>(&hs[r][4 * q]) = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
  }
  __syncthreads();
  // ---- vertical 9-sums (sliding), two pixels per lane in packed u16 -> u16 box image: lanes 0-31 own the upper half of
  // the wave's rows, lanes 32-63 the lower half --------------------------------------------------------------------------
  if (want_box) {
    constexpr int RW = VS_TILE_H / 8;   // output rows per half-wave
    const int half = lane >> 5, px = 2 * (lane & 31);
    const int rbase = (w * 2 + half) * RW;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 acc = {0, 0};
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += *reinterpret_cast<const u16x2*>(&hs[rbase + k][px]);
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int r = rbase + j, gy = y0 + r;
      if (gy < rows && x0 + px < c.bstride) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, acc), reinterpret_cast<uint32_t*>(box + (size_t)gy * c.bstride + x0 + px));
      if (j < RW - 1) acc += *reinterpret_cast<const u16x2*>(&hs[r + 9][px]) - *reinterpret_cast<const u16x2*>(&hs[r][px]);
    }
  }
  __syncthreads();
  if (tid < VS_TILE_H && y0 + tid < rows) mask[(size_t)(y0 + tid) * c.TX + tx] = lmask[tid];
}

// ==============================================================================================
// K2: one workgroup per stream.  Scans the corner masks of both images in (row, x) order, applies
// the descriptor border filter (KeyPointsFilter::runByImageBorder(28)), writes keypoints
// row-major, builds the row/cell CSR, counts raw detections per detector region, then runs the
// threshold controller + adjustDetectorThresholds and the triangulation-distance rule
// (stereo_framepoint_generator.cpp:109-125).  `border` = 28 in the pipeline, 0 for stand-alone FAST.
// run_controller: 0 none, 1 over the stream's two images (stereo: grid (streams, 2)), 2 over ONE image (RGB-D mode, grid
// (streams, 1): adjustDetectorThresholds averages a single detection, depth_framepoint_generator.cpp:24-44).
// ==============================================================================================
__device__ __forceinline__ unsigned long long col_mask(int lo, int hi, int wx0) {
  // bits of a 64-px word starting at column wx0 whose column lies in [lo, hi)
  int a = max(lo - wx0, 0), e = min(hi - wx0, 64);
  if (e <= a) return 0ull;
  const unsigned long long hi_m = (e >= 64) ? ~0ull : ((1ull << e) - 1ull);
  return hi_m & ~((1ull << a) - 1ull);
}

#ifndef VS_EMIT_WPT
#define VS_EMIT_WPT 4    // mask words a thread keeps in registers per pass (one pass = 512 * 4 words = 131072 px; 96 VGPRs, no scratch).  Measured (round 4,
                         // 157 streams, alone on the chip): 4 -> 58 us, 8 -> 67 us, 16 -> 73 us: the kernel is bound by the per-lane emission loops, not by its four passes
#endif
__global__ __launch_bounds__(512, VS_EMIT_WPT > 8 ? 2 : 4) void k_emit(const DevCfg c, const DevBuf b, int border, int run_controller) {
  __shared__ int sh_wave[8], sh_total;
  __shared__ int sh_cnt[VSLAM_MAX_REGIONS];
  __shared__ int sh_last;
  // block (i, side) is the (side * gridDim.x + i)-th of the launch: both images of stream s on physical XCD s % 8 (dev_types.h xcd_rot)
  const int s = b.s0 + xcd_local_stream(blockIdx.x, gridDim.x, (b.xcd_rot + blockIdx.y * gridDim.x) & 7), tid = threadIdx.x;
  if (!vs_active(b, s)) return;
  const int rows = c.c.rows, cols = c.c.cols, TX = c.TX, CW = c.CW;
  StreamState& st = b.st[s];
  if (tid < VSLAM_MAX_REGIONS) sh_cnt[tid] = 0;
  __syncthreads();
  const int nwords = rows * TX;
  // one 512-thread workgroup per image (blockIdx.y = side).
  // Word order = keypoint order.  A pass covers 512 * VS_EMIT_WPT consecutive mask words: wave w owns 64 * VS_EMIT_WPT of them, [256 w, 256 w + 256)
  // of the pass, lane l its words l, l + 64, ... — every load and store of a wave covers consecutive words (coalesced),
  // and the prefix over the words is 8 wave scans + one 8-entry scan across waves.
  const int side = blockIdx.y, lane = tid & 63, w = tid >> 6;
  {
    const unsigned long long* mask = mask_of(c, b, s, side);
    const uint8_t* score8 = score_of(c, b, s, side);
    int16_t* kxy = kpxy_of(c, b, s, side);
    uint8_t* ksc = kpscore_of(c, b, s, side);
    int32_t* rowcell = rowcell_of(c, b, s, side);
    uint8_t* used = used_of(c, b, s, side);
    auto filtered = [&](unsigned long long m, int row, int t) -> unsigned long long {
      const bool row_ok = row >= border && row < rows - border;
      return row_ok ? (m & col_mask(border, cols - border, t * 64)) : 0ull;
    };
    int done = 0;   // keypoints of earlier passes
    const int q64 = 64 / TX, r64 = 64 - q64 * TX;   // (row, word-in-row) of word + 64 without a division
    for (int p0 = 0; p0 < nwords; p0 += 512 * VS_EMIT_WPT) {
      const int wbase = p0 + w * 64 * VS_EMIT_WPT + lane;
      const int row_b = wbase / TX, t_b = wbase - row_b * TX;
      unsigned long long mw[VS_EMIT_WPT];
#pragma unroll
      for (int j = 0; j < VS_EMIT_WPT; ++j) mw[j] = (wbase + 64 * j < nwords) ? mask[wbase + 64 * j] : 0ull;
      // raw detections per region (controller input): every corner bit lies in exactly one region
      for (int r = 0; r < c.n_regions; ++r) {
        const DevRegion& R = c.regions[r];
        int n = 0, row = row_b, t = t_b;
#pragma unroll
        for (int j = 0; j < VS_EMIT_WPT; ++j) {
          if (row >= R.y + 3 && row < R.y + R.h - 3) n += __popcll(mw[j] & col_mask(R.x + 3, R.x + R.w - 3, t * 64));
          t += r64; row += q64; if (t >= TX) { t -= TX; ++row; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
        if (lane == 0 && n) atomicAdd(&sh_cnt[r], n);
      }
      // exclusive prefix of the filtered counts in word order
      int ex[VS_EMIT_WPT];
      int carry = 0;
      int row = row_b, t = t_b;
#pragma unroll
      for (int j = 0; j < VS_EMIT_WPT; ++j) {
        const int wd = wbase + 64 * j;
        mw[j] = wd < nwords ? filtered(mw[j], row, t) : 0ull;   // from here on only the border-filtered bits matter
        t += r64; row += q64; if (t >= TX) { t -= TX; ++row; }
        const int cnt = __popcll(mw[j]);
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
        ex[j] = carry + inc - cnt;
        carry += __shfl(inc, 63, 64);
      }
      __syncthreads();   // sh_wave / sh_total of the previous pass are consumed
      if (lane == 0) sh_wave[w] = carry;
      __syncthreads();
      if (tid == 0) { int acc = 0; for (int i = 0; i < 8; ++i) { const int v = sh_wave[i]; sh_wave[i] = acc; acc += v; } sh_total = acc; }
      __syncthreads();
      const int wave_off = done + sh_wave[w];
      // keypoints (x, row) in row-major order + the row/cell CSR: stores only, consecutive lanes -> consecutive addresses
      row = row_b; t = t_b;
#pragma unroll
      for (int j = 0; j < VS_EMIT_WPT; ++j) {
        const int wd = wbase + 64 * j;
        if (wd < nwords) {
          unsigned long long f = mw[j];
          int off = wave_off + ex[j];
          int32_t* rc = rowcell + (size_t)row * (CW + 1) + t * 4;
          rc[0] = min(off, c.NMAX);
          rc[1] = min(off + __popcll(f & 0xFFFFull), c.NMAX);
          rc[2] = min(off + __popcll(f & 0xFFFFFFFFull), c.NMAX);
          rc[3] = min(off + __popcll(f & 0xFFFFFFFFFFFFull), c.NMAX);
          if (t == TX - 1) rc[4] = min(off + __popcll(f), c.NMAX);
          while (f) {
            const int bit = __ffsll((long long)f) - 1;
            f &= f - 1;
            if (off < c.NMAX) {
              *reinterpret_cast<int32_t*>(kxy + 2 * off) = (t * 64 + bit) | (row << 16);
              used[off] = 0;
            }
            ++off;
          }
        }
        t += r64; row += q64; if (t >= TX) { t -= TX; ++row; }
      }
      done += sh_total;
    }
    const int total = done;
    if (total > c.NMAX) { if (tid == 0) atomicOr(&st.error_flags, 1); }
    if (tid == 0) b.n_kp[s * 2 + side] = min(total, c.NMAX);
    __syncthreads();   // the workgroup's keypoint stores are visible to its own loads below
    // scores: one keypoint per thread, four gathers in flight
    const int nk = min(total, c.NMAX);
    for (int i0 = tid; i0 < nk; i0 += 4 * 512) {
      int px[4];
      uint8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 512 * u;
        px[u] = (i < nk) ? *reinterpret_cast<const int32_t*>(kxy + 2 * i) : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i0 + 512 * u < nk) ? score8[(size_t)(px[u] >> 16) * c.bstride + (px[u] & 0xFFFF)] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i0 + 512 * u < nk) ksc[i0 + 512 * u] = v[u];
    }
    __syncthreads();
  }
  // the second of the stream's two workgroups to get here runs the controller on both images' counts
  if (tid == 0) {
    for (int r = 0; r < c.n_regions; ++r) b.iinfo[s].raw_count[side][r] = sh_cnt[r];
    __threadfence();
    const int last_ticket = (int)gridDim.y - 1;
    const int ticket = atomicAdd(&b.iinfo[s].ticket, 1);
    sh_last = ticket == last_ticket;
    if (ticket == last_ticket) { __threadfence(); b.iinfo[s].ticket = 0; }
  }
  __syncthreads();
  if (tid == 0 && sh_last) {
    const int (*cnt)[VSLAM_MAX_REGIONS] = b.iinfo[s].raw_count;
    if (run_controller) {
      // detectKeypoints controller (base_framepoint_generator.cpp:382-415) for L then R with the
      // thresholds that were in effect, then adjustDetectorThresholds (:440-459)
      const double tol = c.c.target_number_of_keypoints_tolerance, maxchg = c.c.detector_threshold_maximum_change;
      const double tmin = c.c.detector_threshold_minimum, tmax = c.c.detector_threshold_maximum;
      const double target = (double)c.target_per_detector;
      const int n_sides = run_controller == 2 ? 1 : 2;
      for (int r = 0; r < c.n_regions; ++r) {
        double acc = 0;
        for (int sd = 0; sd < n_sides; ++sd) {
          double t = (double)st.thr[r];
          const double delta = ((double)__hip_atomic_load(&cnt[sd][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) / target;
          if (delta < -tol) {
            const double change = fmax(delta, -maxchg);
            t = t + fmin(change * t, -1.0);
            if (t < tmin) t = tmin;
          } else if (delta > tol) {
            const double change = fmin(delta, maxchg);
            t += fmax(change * t, 1.0);
            if (t > tmax) t = tmax;
          }
          acc += t;
        }
        st.thr[r] = (int)rint(acc / n_sides);
      }
    }
    for (int r = 0; r < c.n_regions; ++r) b.iinfo[s].thr_after[r] = st.thr[r];
  }
}

// ==============================================================================================
// K3: BRIEF-32.  One wavefront per keypoint (grid-strided): lane l evaluates tests l, 64+l, 128+l,
// 192+l on the u16 box image; each ballot is 8 descriptor bytes (bit-reversed per byte: MSB first).
// ==============================================================================================
__device__ __forceinline__ void brief_wave(const uint16_t* box, int bstride, int x, int y, int lane, uint8_t* out32) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = j * 64 + lane;
    const int a = box[(size_t)(y + c_brief[i][0]) * bstride + (x + c_brief[i][1])];
    const int bb = box[(size_t)(y + c_brief[i][2]) * bstride + (x + c_brief[i][3])];
    const unsigned long long m = __ballot(a < bb);
    if (lane == 0) {
      const unsigned long long wv = __builtin_bswap64(__brevll(m));
      reinterpret_cast<unsigned long long*>(out32)[j] = wv;
    }
  }
}

// Tiled form: one workgroup owns the keypoints of a 128 x 64 pixel tile (found through the row/cell CSR), stages
// the (128+48) x (64+48) u16 box region once in LDS with coalesced loads and evaluates the 256 tests from LDS,
// one wavefront per keypoint.  Replaces 512 scattered 2-byte global gathers per keypoint.
#ifndef VS_BT_W
#define VS_BT_W 128
#endif
#ifndef VS_BT_H
#define VS_BT_H 64      // <= 64: the per-row keypoint counts of a tile are scanned by one wavefront
#endif
#define VS_BT_RW (VS_BT_W + 2 * VSLAM_BRIEF_PATCH_HALF)   // 176
#define VS_BT_RH (VS_BT_H + 2 * VSLAM_BRIEF_PATCH_HALF)   // 80
__global__ __launch_bounds__(256) void k_brief(const DevCfg c, const DevBuf b) {
  __shared__ __align__(16) uint16_t reg[VS_BT_RH * VS_BT_RW];
  __shared__ int row_lo[VS_BT_H], row_off[VS_BT_H + 1];
  int tx, ty, tz;
  xcd_tile(&tx, &ty, &tz, c.mono, b.xcd_rot);
  const int s = b.s0 + (c.mono ? tz : (tz >> 1)), side = c.mono ? 0 : (tz & 1);
  if (!vs_active(b, s)) return;
  const int x0 = tx * VS_BT_W, y0 = ty * VS_BT_H;
  const int rows = c.c.rows;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t* rowcell = rowcell_of(c, b, s, side);
  // keypoints of the tile: per row the CSR range of cells [8*tx, 8*tx+8)
  if (tid < VS_BT_H) {
    const int r = y0 + tid;
    int lo = 0, n = 0;
    if (r < rows) {
      const int c0 = min((VS_BT_W / 16) * tx, c.CW), c1 = min((VS_BT_W / 16) * (tx + 1), c.CW);
      lo = rowcell[(size_t)r * (c.CW + 1) + c0];
      n = rowcell[(size_t)r * (c.CW + 1) + c1] - lo;
    }
    row_lo[tid] = lo;
    int inc = n;
#pragma unroll
    for (int o = 1; o < VS_BT_H; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (tid >= o) inc += t; }
    row_off[tid + 1] = inc;
    if (tid == 0) row_off[0] = 0;
  }
  __syncthreads();
  const int K = row_off[VS_BT_H];
  if (K == 0) return;
  const uint16_t* box = box_of(c, b, s, side);
  // stage the box region with 16-byte loads, ALL of a thread's loads in flight before the first LDS store (the loop
  // is otherwise a chain of dependent L2 round trips).  x0 - 24 is a multiple of 8 pixels and the row stride a multiple
  // of 64, so a chunk lies entirely inside or outside the padded row; outside chunks (never sampled by a keypoint that
  // passed the 28 px border filter) are zero.
  constexpr int CPR = VS_BT_RW / 8;                              // 16-byte chunks per region row (22)
  constexpr int NLD = (VS_BT_RH * CPR + 255) / 256;              // loads per thread (7)
  uint4 stage[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int i = tid + 256 * u;
    const int r = i / CPR, q = i - r * CPR;
    const int gy = min(max(y0 - VSLAM_BRIEF_PATCH_HALF + r, 0), rows - 1);
    const int gx = x0 - VSLAM_BRIEF_PATCH_HALF + 8 * q;
    stage[u] = make_uint4(0u, 0u, 0u, 0u);
    if (i < VS_BT_RH * CPR && gx >= 0 && gx + 8 <= c.bstride) stage[u] = *reinterpret_cast<const uint4*>(box + (size_t)gy * c.bstride + gx);
  }
  // lane l of wave w looks up keypoint w + 4*l of the tile (row by binary search in the CSR prefix, x from the keypoint
  // array) while the staging loads are in flight
  const int16_t* kxy = kpxy_of(c, b, s, side);
  auto lookup = [&](int k, int* base, int* idx) {
    *base = 0; *idx = -1;
    if (k < K) {
      int r = 0;
#pragma unroll
      for (int step = VS_BT_H / 2; step > 0; step >>= 1) if (row_off[r + step] <= k) r += step;
      const int id = row_lo[r] + (k - row_off[r]);
      *idx = id;
      *base = (r + VSLAM_BRIEF_PATCH_HALF) * VS_BT_RW + (kxy[2 * id] - x0 + VSLAM_BRIEF_PATCH_HALF);
    }
  };
  int my_base, my_idx;
  lookup(w + 4 * lane, &my_base, &my_idx);
  // the 4 test pairs of this lane as offsets inside the LDS region
  int off_a[4], off_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = j * 64 + lane;
    off_a[j] = c_brief[i][0] * VS_BT_RW + c_brief[i][1];
    off_b[j] = c_brief[i][2] * VS_BT_RW + c_brief[i][3];
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int i = tid + 256 * u;
    if (i < VS_BT_RH * CPR) *reinterpret_cast<uint4*>(&reg[8 * i]) = stage[u];
  }
  __syncthreads();
  uint8_t* desc = desc_of(c, b, s, side);
  for (int k0 = 0; k0 < K; k0 += 256) {
    if (k0 > 0) lookup(k0 + w + 4 * lane, &my_base, &my_idx);
    const int n_here = (min(K - k0, 256) - w + 3) >> 2;          // keypoints of this wave in the chunk
    for (int i = 0; i < n_here; ++i) {
      const int base = __builtin_amdgcn_readlane(my_base, i), idx = __builtin_amdgcn_readlane(my_idx, i);
      unsigned long long word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long m = __ballot(reg[base + off_a[j]] < reg[base + off_b[j]]);
        if (lane == j) word = m;
      }
      if (lane < 4) reinterpret_cast<unsigned long long*>(desc + (size_t)32 * idx)[lane] = __builtin_bswap64(__brevll(word));
    }
  }
}

// stand-alone BRIEF at caller keypoints (vslam_brief_describe): keep[] = inside the 28 px border
__global__ __launch_bounds__(256) void k_brief_at(const uint16_t* box, int bstride, int rows, int cols, int n,
                                                  const int16_t* xy, uint8_t* keep, uint8_t* desc) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  for (int i = wave; i < n; i += nwaves) {
    const int x = xy[2 * i], y = xy[2 * i + 1];
    const bool in = x >= VSLAM_BRIEF_BORDER && x < cols - VSLAM_BRIEF_BORDER && y >= VSLAM_BRIEF_BORDER && y < rows - VSLAM_BRIEF_BORDER;
    if (lane == 0) keep[i] = in ? 1 : 0;
    if (in) brief_wave(box, bstride, x, y, lane, desc + (size_t)32 * i);
    else if (lane < 4) reinterpret_cast<unsigned long long*>(desc + (size_t)32 * i)[lane] = 0ull;
  }
}

// ==============================================================================================
// ORB as descriptor extractor (cv::ORB::create()->compute on the detector's keypoints; base_framepoint_generator.cpp:190-196,
// 219-224, 431-438): 7x7 Gaussian (sigma 2, BORDER_REFLECT_101) in OpenCV's 8-bit fixed-point form — integer row pass with
// round(256 k), integer column pass, (v + 2^15) >> 16 — then 256 steered intensity tests per keypoint on the blurred image.
// The blurred u8 image lives in the memory of the (unused) box image, row stride c.bstride bytes.
// ==============================================================================================
struct Gauss7 { int32_t k[4]; };   // k[0] = centre tap ... k[3] = outermost (symmetric kernel)
__device__ __forceinline__ int reflect101(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }
__device__ __forceinline__ uint8_t* blur_of(const DevCfg& c, const DevBuf& b, int s, int d) { return reinterpret_cast<uint8_t*>(box_of(c, b, s, d)); }

// one 64 x H tile per workgroup: the (64+8) x (H+6) source pixels (columns x0-4 .. x0+67, reflected at the image border) staged in
// LDS — aligned dwords for tiles inside the image, bytes for border tiles —, row sums four per thread from three dwords
// (v_alignbyte + two v_dot4_u32_u8 per pixel; u16: 255 * 257 < 2^16), column sums four per thread (v_mad_u32_u24), one dword
// store per four pixels where the output row allows it.  Same integers as the serial form.
__device__ __forceinline__ void gauss7_tile(const uint8_t* img, int stride, int rows, int cols, int x0, int y0, const Gauss7& g, uint8_t* out, int ostride,
                                            uint8_t (*src)[72], uint16_t (*hs)[VS_TILE_W]) {
  const int tid = threadIdx.x;
  const bool in_dwords = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0) && x0 >= 4 && x0 + 68 <= cols;
  if (in_dwords) {
    constexpr int NST = ((VS_TILE_H + 6) * 18 + 255) / 256;   // all of a thread's loads in flight before the first LDS store
    uint32_t v[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 256 * u;
      const int r = i / 18, q = i - 18 * r;
      const int gy = reflect101(min(y0 - 3 + r, rows + 2), rows);
      v[u] = 0;
      if (i < (VS_TILE_H + 6) * 18) v[u] = *reinterpret_cast<const uint32_t*>(img + (size_t)gy * stride + (x0 - 4 + 4 * q));
    }
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 256 * u;
      const int r = i / 18, q = i - 18 * r;
      if (i < (VS_TILE_H + 6) * 18) *reinterpret_cast<uint32_t*>(&src[r][4 * q]) = v[u];
    }
  } else {
    for (int i = tid; i < (VS_TILE_H + 6) * 70; i += 256) {
      const int r = i / 70, q = i - 70 * r;
      src[r][q + 1] = img[(size_t)reflect101(min(y0 - 3 + r, rows + 2), rows) * stride + reflect101(min(x0 - 3 + q, cols + 2), cols)];
    }
  }
  __syncthreads();
  // source column x0 - 3 + q lives at src[.][q + 1]: outputs 4t .. 4t+3 of a row need bytes 4t+1 .. 4t+10 = dwords t, t+1, t+2
  const uint32_t klo = (uint32_t)g.k[3] | ((uint32_t)g.k[2] << 8) | ((uint32_t)g.k[1] << 16) | ((uint32_t)g.k[0] << 24);   // taps -3 .. 0
  const uint32_t khi = (uint32_t)g.k[1] | ((uint32_t)g.k[2] << 8) | ((uint32_t)g.k[3] << 16);                              // taps +1 .. +3
  for (int i = tid; i < (VS_TILE_H + 6) * 16; i += 256) {
    const int r = i >> 4, t = i & 15;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(&src[r][4 * t]);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = j == 3 ? w1 : __builtin_amdgcn_alignbyte(w1, w0, 1 + j), hi = j == 3 ? w2 : __builtin_amdgcn_alignbyte(w2, w1, 1 + j);
      o[j] = __builtin_amdgcn_udot4(lo, klo, __builtin_amdgcn_udot4(hi, khi, 0u, false), false);
    }
    *reinterpret_cast<uint2*>(&hs[r][4 * t]) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
  }
  __syncthreads();
  const bool out_dwords = ((ostride & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
  for (int i = tid; i < VS_TILE_H * 16; i += 256) {
    const int r = i >> 4, t = i & 15;
    if (y0 + r >= rows || x0 + 4 * t >= cols) continue;
    uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 7; ++d) {
      const uint2 v = *reinterpret_cast<const uint2*>(&hs[r + d][4 * t]);
      const uint32_t kk = (uint32_t)g.k[d < 3 ? 3 - d : d - 3];
      acc[0] += kk * (v.x & 0xFFFFu); acc[1] += kk * (v.x >> 16); acc[2] += kk * (v.y & 0xFFFFu); acc[3] += kk * (v.y >> 16);
    }
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) px[j] = min((acc[j] + (1u << 15)) >> 16, 255u);
    uint8_t* dst = out + (size_t)(y0 + r) * ostride + x0 + 4 * t;
    if (out_dwords && x0 + 4 * t + 4 <= (ostride > cols ? ostride : cols)) {
      *reinterpret_cast<uint32_t*>(dst) = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (x0 + 4 * t + j < cols) dst[j] = (uint8_t)px[j];
    }
  }
}
__global__ __launch_bounds__(256) void k_gauss7(const DevCfg c, const DevBuf b, const Gauss7 g) {
  __shared__ uint8_t src[VS_TILE_H + 6][72];
  __shared__ uint16_t hs[VS_TILE_H + 6][VS_TILE_W];
  int tx, ty, tz;
  xcd_tile(&tx, &ty, &tz, c.mono, b.xcd_rot);
  const int s = b.s0 + (c.mono ? tz : (tz >> 1)), side = c.mono ? 0 : (tz & 1);
  if (!vs_active(b, s)) return;
  gauss7_tile(b.img[side] + (size_t)s * b.img_stream_stride, b.img_row_stride, c.c.rows, c.c.cols, tx * VS_TILE_W, ty * VS_TILE_H, g,
              blur_of(c, b, s, side), c.bstride, src, hs);
}
// stand-alone form on an explicit image (vslam_gaussian_blur7_u8, vslam_orb_describe)
__global__ __launch_bounds__(256) void k_gauss7_plain(const uint8_t* img, int stride, int rows, int cols, const Gauss7 g, uint8_t* out, int ostride) {
  __shared__ uint8_t src[VS_TILE_H + 6][72];
  __shared__ uint16_t hs[VS_TILE_H + 6][VS_TILE_W];
  gauss7_tile(img, stride, rows, cols, blockIdx.x * VS_TILE_W, blockIdx.y * VS_TILE_H, g, out, ostride, src, hs);
}

// computeOrbDescriptors (WTA_K = 2) for one keypoint by one wavefront: lane l evaluates tests l, 64+l, 128+l, 192+l; the
// rotated, rounded tap offsets depend on the angle only (one angle per launch: the FAST keypoints all carry -1)
struct OrbTaps { int off[4][2]; };
__device__ __forceinline__ OrbTaps orb_taps(int lane, float a, float bsin, int stride) {
  OrbTaps t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = j * 64 + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float px = (float)c_orb[i][2 * h], py = (float)c_orb[i][2 * h + 1];
      const float xf = px * a - py * bsin, yf = px * bsin + py * a;
      t.off[j][h] = (int)rintf(yf) * stride + (int)rintf(xf);
    }
  }
  return t;
}
__device__ __forceinline__ void orb_wave(const uint8_t* centre, const OrbTaps& t, unsigned long long d[4]) {
  int v[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j][0] = centre[t.off[j][0]]; v[j][1] = centre[t.off[j][1]]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j] = __ballot(v[j][0] < v[j][1]);   // bit l of word j = test 64 j + l: bytes LSB first
}
// Tiled like k_brief: one workgroup owns the keypoints of a 128 x 64 pixel tile (row / cell CSR), stages the (128+32) x (64+32)
// bytes of the blurred image around it in LDS with 16-byte loads and evaluates the 256 steered tests from LDS, one wavefront
// per keypoint (512 scattered one-byte global gathers per keypoint kept the texture-address unit busy for ~0.2 ms per launch).
// The rotated taps stay within 14 px of the keypoint (pattern radius 13, rounded after the rotation).
#define VS_OT_HALF 16
#define VS_OT_RW (VS_BT_W + 2 * VS_OT_HALF)   // 160
#define VS_OT_RH (VS_BT_H + 2 * VS_OT_HALF)   // 96
__global__ __launch_bounds__(256) void k_orb_describe(const DevCfg c, const DevBuf b, float a, float bsin) {
  __shared__ __align__(16) uint8_t reg[VS_OT_RH * VS_OT_RW];
  __shared__ int row_lo[VS_BT_H], row_off[VS_BT_H + 1];
  int tx, ty, tz;
  xcd_tile(&tx, &ty, &tz, c.mono, b.xcd_rot);
  const int s = b.s0 + (c.mono ? tz : (tz >> 1)), side = c.mono ? 0 : (tz & 1);
  if (!vs_active(b, s)) return;
  const int x0 = tx * VS_BT_W, y0 = ty * VS_BT_H;
  const int rows = c.c.rows;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t* rowcell = rowcell_of(c, b, s, side);
  if (tid < VS_BT_H) {
    const int r = y0 + tid;
    int lo = 0, n = 0;
    if (r < rows) {
      const int c0 = min((VS_BT_W / 16) * tx, c.CW), c1 = min((VS_BT_W / 16) * (tx + 1), c.CW);
      lo = rowcell[(size_t)r * (c.CW + 1) + c0];
      n = rowcell[(size_t)r * (c.CW + 1) + c1] - lo;
    }
    row_lo[tid] = lo;
    int inc = n;
#pragma unroll
    for (int o = 1; o < VS_BT_H; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (tid >= o) inc += t; }
    row_off[tid + 1] = inc;
    if (tid == 0) row_off[0] = 0;
  }
  __syncthreads();
  const int K = row_off[VS_BT_H];
  if (K == 0) return;
  const uint8_t* blur = blur_of(c, b, s, side);
  // x0 - 16 is a multiple of 16 and the row stride a multiple of 64 bytes: a chunk lies inside or outside the padded row
  constexpr int CPR = VS_OT_RW / 16;                             // 16-byte chunks per region row (10)
  constexpr int NLD = (VS_OT_RH * CPR + 255) / 256;              // loads per thread (4)
  uint4 stage[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int i = tid + 256 * u;
    const int r = i / CPR, q = i - r * CPR;
    const int gy = min(max(y0 - VS_OT_HALF + r, 0), rows - 1);
    const int gx = x0 - VS_OT_HALF + 16 * q;
    stage[u] = make_uint4(0u, 0u, 0u, 0u);
    if (i < VS_OT_RH * CPR && gx >= 0 && gx + 16 <= c.bstride) stage[u] = *reinterpret_cast<const uint4*>(blur + (size_t)gy * c.bstride + gx);
  }
  const int16_t* kxy = kpxy_of(c, b, s, side);
  auto lookup = [&](int k, int* base, int* idx) {
    *base = 0; *idx = -1;
    if (k < K) {
      int r = 0;
#pragma unroll
      for (int step = VS_BT_H / 2; step > 0; step >>= 1) if (row_off[r + step] <= k) r += step;
      const int id = row_lo[r] + (k - row_off[r]);
      *idx = id;
      *base = (r + VS_OT_HALF) * VS_OT_RW + (kxy[2 * id] - x0 + VS_OT_HALF);
    }
  };
  int my_base, my_idx;
  lookup(w + 4 * lane, &my_base, &my_idx);
  const OrbTaps t = orb_taps(lane, a, bsin, VS_OT_RW);
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int i = tid + 256 * u;
    if (i < VS_OT_RH * CPR) *reinterpret_cast<uint4*>(&reg[16 * i]) = stage[u];
  }
  __syncthreads();
  uint8_t* desc = desc_of(c, b, s, side);
  for (int k0 = 0; k0 < K; k0 += 256) {
    if (k0 > 0) lookup(k0 + w + 4 * lane, &my_base, &my_idx);
    const int n_here = (min(K - k0, 256) - w + 3) >> 2;
    for (int i = 0; i < n_here; ++i) {
      const int base = __builtin_amdgcn_readlane(my_base, i), idx = __builtin_amdgcn_readlane(my_idx, i);
      unsigned long long word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long m = __ballot(reg[base + t.off[j][0]] < reg[base + t.off[j][1]]);   // bit l of word j = test 64 j + l
        if (lane == j) word = m;
      }
      if (lane < 4) reinterpret_cast<unsigned long long*>(desc + (size_t)32 * idx)[lane] = word;
    }
  }
}
// stand-alone ORB at caller keypoints (vslam_orb_describe): keep[] = inside the 31 px border
__global__ __launch_bounds__(256) void k_orb_at(const uint8_t* blur, int stride, int rows, int cols, int n, const int16_t* xy, float a, float bsin,
                                                uint8_t* keep, uint8_t* desc) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
  const OrbTaps t = orb_taps(lane, a, bsin, stride);
  for (int i = wave; i < n; i += nwaves) {
    const int x = xy[2 * i], y = xy[2 * i + 1];
    const bool in = x >= VSLAM_ORB_BORDER && x < cols - VSLAM_ORB_BORDER && y >= VSLAM_ORB_BORDER && y < rows - VSLAM_ORB_BORDER;
    if (lane == 0) keep[i] = in ? 1 : 0;
    unsigned long long d[4] = {0ull, 0ull, 0ull, 0ull};
    if (in) orb_wave(blur + (size_t)y * stride + x, t, d);
    const unsigned long long dv = lane == 0 ? d[0] : (lane == 1 ? d[1] : (lane == 2 ? d[2] : d[3]));
    if (lane < 4) reinterpret_cast<unsigned long long*>(desc + (size_t)32 * i)[lane] = dv;
  }
}

// ORB::compute on keypoints with an octave and an angle (vslam_orb_describe_keypoints): one wavefront per keypoint, the tests steered by the
// keypoint's own rotation (cos / sin evaluated on the host as OpenCV does), sampled from the Gaussian image of the keypoint's pyramid level
struct OrbLevels { const uint8_t* blur[16]; int32_t stride[16], rows[16], cols[16]; };
__global__ __launch_bounds__(256) void k_orb_at_levels(const OrbLevels L, int n, const int32_t* pos, const float* ab, uint8_t* keep, uint8_t* desc) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
  for (int i = wave; i < n; i += nwaves) {
    const int lv = pos[3 * i + 2];          // -1: removed on the host (border filter)
    unsigned long long d[4] = {0ull, 0ull, 0ull, 0ull};
    const bool in = lv >= 0;
    if (in) {
      const int stride = L.stride[lv];
      const OrbTaps t = orb_taps(lane, ab[2 * i], ab[2 * i + 1], stride);
      orb_wave(L.blur[lv] + (size_t)pos[3 * i + 1] * stride + pos[3 * i], t, d);
    }
    if (lane == 0) keep[i] = in ? 1 : 0;
    const unsigned long long dv = lane == 0 ? d[0] : (lane == 1 ? d[1] : (lane == 2 ? d[2] : d[3]));
    if (lane < 4) reinterpret_cast<unsigned long long*>(desc + (size_t)32 * i)[lane] = dv;
  }
}

// ==============================================================================================
// K6: brute-force N x M 2-nearest-neighbours on 32-byte rows — matcher->knnMatch(query, train, k = 2) of the use_matches block
// (stereo_framepoint_generator.cpp:168-206).  The reference converts the descriptors to CV_32F and picks the norm with the
// matcher type (:175-197): BRUTEFORCE = L2, BRUTEFORCE_L1, BRUTEFORCE_SL2 (squared L2), BRUTEFORCE_HAMMING on the bits.
// All four distances are exact integers on bytes (L2^2 <= 32*255^2 < 2^24, so the float accumulation upstream is exact too);
// L2 takes the float square root at the end.
//
// Layout: a workgroup owns 16 queries; every query is searched by a 16-lane group (four queries per wavefront), lane l of the
// group taking train rows l, l+16, ... of each 256-row tile.  A tile is staged in LDS by one coalesced 32-byte row load per
// thread (the next tile's loads are in flight while the current one is searched); the two 16-byte halves of a row are swapped
// for rows 8..15 of every 16 so that the 16 lanes of a group read 16 rows from distinct banks, and the four groups of a
// wavefront read the same addresses (broadcast).  Per pair: Hamming 8 x (xor, v_bcnt accumulate); L1 8 x v_sad_u8; L2 / SL2
// ||q||^2 + ||t||^2 - 2 q.t with 8 x v_dot4_u32_u8 (the row norms are computed once while staging).  Each lane keeps its
// best two as 64-bit keys (distance << 32 | row: the lowest index wins ties); the 16 lanes merge with 4 xor-shuffles.
// ==============================================================================================
enum { VS_KNN_HAMMING = 0, VS_KNN_L2 = 1, VS_KNN_L1 = 2, VS_KNN_SL2 = 3 };
#define VS_KNN_TILE 256
__device__ __forceinline__ void knn_merge(unsigned long long& b0, unsigned long long& b1, unsigned long long c0, unsigned long long c1) {
  const unsigned long long lo = b0 < c0 ? b0 : c0, hi = b0 < c0 ? c0 : b0, m = b1 < c1 ? b1 : c1;
  b0 = lo; b1 = hi < m ? hi : m;
}
template <int NORM>
__device__ __forceinline__ void knn2_body(int nq, const uint8_t* __restrict__ q, int nt, const uint8_t* __restrict__ t, int32_t* idx,
                                          float* dist, uint4 (*tile)[2], uint32_t* tnorm) {
  const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
  const int i = blockIdx.x * 16 + grp;
  const bool dotn = NORM == VS_KNN_L2 || NORM == VS_KNN_SL2;
  uint32_t qv[8];
  uint32_t qn = 0;
  {
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)32 * (i < nq ? i : nq - 1));
    const uint4 a = qp[0], c = qp[1];
    qv[0] = a.x; qv[1] = a.y; qv[2] = a.z; qv[3] = a.w; qv[4] = c.x; qv[5] = c.y; qv[6] = c.z; qv[7] = c.w;
    if (dotn) {
#pragma unroll
      for (int k = 0; k < 8; ++k) qn = __builtin_amdgcn_udot4(qv[k], qv[k], qn, false);
    }
  }
  unsigned long long b0 = ~0ull, b1 = ~0ull;
  uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
  if (tid < nt) { const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)32 * tid); n0 = tp[0]; n1 = tp[1]; }
  for (int j0 = 0; j0 < nt; j0 += VS_KNN_TILE) {
    __syncthreads();                                   // the previous tile has been searched
    {
      const int sw = (tid >> 3) & 1;
      tile[tid][sw] = n0; tile[tid][sw ^ 1] = n1;
      if (dotn) {
        uint32_t tn = 0;
        tn = __builtin_amdgcn_udot4(n0.x, n0.x, tn, false); tn = __builtin_amdgcn_udot4(n0.y, n0.y, tn, false);
        tn = __builtin_amdgcn_udot4(n0.z, n0.z, tn, false); tn = __builtin_amdgcn_udot4(n0.w, n0.w, tn, false);
        tn = __builtin_amdgcn_udot4(n1.x, n1.x, tn, false); tn = __builtin_amdgcn_udot4(n1.y, n1.y, tn, false);
        tn = __builtin_amdgcn_udot4(n1.z, n1.z, tn, false); tn = __builtin_amdgcn_udot4(n1.w, n1.w, tn, false);
        tnorm[tid] = tn;
      }
    }
    const int jn = j0 + VS_KNN_TILE;
    if (jn + tid < nt) { const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)32 * (jn + tid)); n0 = tp[0]; n1 = tp[1]; }
    __syncthreads();
    const int cnt = min(VS_KNN_TILE, nt - j0);
    for (int r = sub; r < cnt; r += 16) {
      const int sw = (r >> 3) & 1;
      const uint4 a = tile[r][sw], c = tile[r][sw ^ 1];
      const uint32_t tv[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      uint32_t d = 0;
      if (NORM == VS_KNN_HAMMING) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(qv[k] ^ tv[k]);
      } else if (NORM == VS_KNN_L1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d = __builtin_amdgcn_sad_u8(qv[k], tv[k], d);
      } else {
        uint32_t dot = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) dot = __builtin_amdgcn_udot4(qv[k], tv[k], dot, false);
        d = qn + tnorm[r] - 2u * dot;
      }
      const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)(j0 + r);
      if (key < b0) { b1 = b0; b0 = key; }
      else if (key < b1) { b1 = key; }
    }
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    const unsigned long long c0 = __shfl_xor(b0, m, 64), c1 = __shfl_xor(b1, m, 64);
    knn_merge(b0, b1, c0, c1);
  }
  if (i < nq && sub == 0) {
    const bool h0 = b0 != ~0ull, h1 = b1 != ~0ull;
    idx[2 * i] = h0 ? (int32_t)(b0 & 0xFFFFFFFFull) : -1;
    idx[2 * i + 1] = h1 ? (int32_t)(b1 & 0xFFFFFFFFull) : -1;
    const float d0 = (float)(uint32_t)(b0 >> 32), d1 = (float)(uint32_t)(b1 >> 32);
    dist[2 * i] = h0 ? (NORM == VS_KNN_L2 ? sqrtf(d0) : d0) : 0.f;
    dist[2 * i + 1] = h1 ? (NORM == VS_KNN_L2 ? sqrtf(d1) : d1) : 0.f;
  }
}
__global__ __launch_bounds__(256) void k_knn2(int norm, int nq, const uint8_t* __restrict__ q, int nt,
                                              const uint8_t* __restrict__ t, int32_t* idx, float* dist) {
  __shared__ uint4 tile[VS_KNN_TILE][2];
  __shared__ uint32_t tnorm[VS_KNN_TILE];
  if (norm == VS_KNN_HAMMING) knn2_body<VS_KNN_HAMMING>(nq, q, nt, t, idx, dist, tile, tnorm);
  else if (norm == VS_KNN_L2) knn2_body<VS_KNN_L2>(nq, q, nt, t, idx, dist, tile, tnorm);
  else if (norm == VS_KNN_L1) knn2_body<VS_KNN_L1>(nq, q, nt, t, idx, dist, tile, tnorm);
  else knn2_body<VS_KNN_SL2>(nq, q, nt, t, idx, dist, tile, tnorm);
}
