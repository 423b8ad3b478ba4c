// cosched.hip — can a "frame-phase" workgroup of a given footprint get onto a CU WHILE a fine-grained image kernel floods the
// chip from another HIP stream, or does it wait until the flood's grid is exhausted?  (phases of k_frame
// with <= 128 VGPRs / <= 48 KB LDS "so image-kernel waves co-reside".)
//
// Flood F: many short workgroups shaped like k_fast_box (256 threads, <= 72 VGPRs, 22.6 KB LDS, ~12 us each; 7 per CU) or like
// k_brief (256 threads, 64 VGPRs, 39 KB LDS, ~40 us each; 4 per CU), enough of them to keep the chip busy for ~300 us.
// Mid M: 144 workgroups of the footprint under test, each spinning 100 us, launched on a second stream right after F.
// Reported: when M's workgroups START relative to F's first workgroup (100 MHz wall clock, global), and how long F takes with
// and without M.  A footprint that co-schedules starts within a few tens of us; one that starves starts when F drains.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void flood(unsigned long long ticks, unsigned long long* first, int* sink) {
  __shared__ unsigned char lds[LDS_BYTES];
  lds[threadIdx.x] = 1;
  asm volatile("v_mov_b32 v60, 0" ::: "v60");          // ~64 VGPRs
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x < 2048) atomicMin(first, t0);
  float a = threadIdx.x * 1e-3f;
  while (wall_clock64() - t0 < ticks) { for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f; }   // VALU-busy like the real kernel
  if (a == 123.f && lds[threadIdx.x ^ 1] == 7) *sink = 1;
}

template <int THREADS, int VGPRS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) void mid(unsigned long long ticks, unsigned long long* start, int* sink) {
  __shared__ unsigned char lds[LDS_BYTES];
  lds[threadIdx.x] = 1;
  if (VGPRS > 192) asm volatile("v_mov_b32 v250, 0" ::: "v250");
  else if (VGPRS > 128) asm volatile("v_mov_b32 v188, 0" ::: "v188");
  else if (VGPRS > 64) asm volatile("v_mov_b32 v124, 0" ::: "v124");
  else asm volatile("v_mov_b32 v60, 0" ::: "v60");
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) start[blockIdx.x] = t0;
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }     // latency-bound: mostly idle issue slots
  if (lds[threadIdx.x ^ 1] == 7) *sink = 1;
}

// A latency-bound chain like the frame kernel's (dependent fp64 FMAs, one wave per SIMD, an LDS round trip and a barrier every 64
// steps): how much longer does it take when its SIMDs are shared with the flood's VALU-busy waves, and does s_setprio help?
template <int PRIO>
__global__ __launch_bounds__(256) void chain(int steps, double* out, unsigned long long* dur) {
  __shared__ double x[256];
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  asm volatile("v_mov_b32 v124, 0" ::: "v124");          // a 128-VGPR allocation
  const unsigned long long t0 = wall_clock64();
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < steps; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = fma(a, b, 1e-9);
    x[threadIdx.x] = a;
    __syncthreads();
    a += x[(threadIdx.x + 1) & 255] * 1e-30;
  }
  if (threadIdx.x == 0) dur[blockIdx.x] = wall_clock64() - t0;
  if (a == 123.0) out[0] = a;
}
template <int FLDS, int PRIO>
void run_chain(const char* name, int flood_blocks, unsigned long long flood_ticks) {
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int* sink; unsigned long long *first, *dur; double* out;
  CK(hipMalloc(&sink, 4)); CK(hipMalloc(&first, 8)); CK(hipMalloc(&dur, 8 * 144)); CK(hipMalloc(&out, 8));
  double med[2];
  for (int with = 0; with < 2; ++with) {
    CK(hipMemset(dur, 0, 8 * 144));
    CK(hipDeviceSynchronize());
    if (with) hipLaunchKernelGGL(flood<FLDS>, dim3(flood_blocks), dim3(256), 0, sa, flood_ticks, first, sink);
    hipLaunchKernelGGL(chain<PRIO>, dim3(144), dim3(256), 0, sb, 150, out, dur);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> d(144);
    CK(hipMemcpy(d.data(), dur, 8 * 144, hipMemcpyDeviceToHost));
    std::sort(d.begin(), d.end());
    med[with] = d[72] * 0.01;
  }
  std::printf("%-10s chain of 150 x 64 dependent fp64 FMAs + LDS + barrier, s_setprio %d: %.1f us alone, %.1f us beside the flood (x %.2f)\n", name, PRIO, med[0], med[1], med[1] / med[0]);
  CK(hipFree(sink)); CK(hipFree(first)); CK(hipFree(dur)); CK(hipFree(out));
  CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
}

struct Result { float f_alone, f_with; double s_min, s_med, s_max; };

template <int FLDS, int THREADS, int VGPRS, int LDS_BYTES>
Result run(int flood_blocks, unsigned long long flood_ticks) {
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int* sink; unsigned long long *first, *start;
  CK(hipMalloc(&sink, 4)); CK(hipMalloc(&first, 8)); CK(hipMalloc(&start, 8 * 144));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned long long big = ~0ull;
  Result r{};
  for (int with = 0; with < 2; ++with) {
    CK(hipMemcpy(first, &big, 8, hipMemcpyHostToDevice));
    CK(hipMemset(start, 0, 8 * 144));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(flood<FLDS>, dim3(flood_blocks), dim3(256), 0, sa, flood_ticks, first, sink);
    CK(hipEventRecord(e1, sa));
    if (with) hipLaunchKernelGGL((mid<THREADS, VGPRS, LDS_BYTES>), dim3(144), dim3(THREADS), 0, sb, 10000ull, start, sink);
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    (with ? r.f_with : r.f_alone) = ms;
  }
  unsigned long long f0; std::vector<unsigned long long> st(144);
  CK(hipMemcpy(&f0, first, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(st.data(), start, 8 * 144, hipMemcpyDeviceToHost));
  std::vector<double> d;
  for (auto v : st) d.push_back(((double)v - (double)f0) * 0.01);   // us
  std::sort(d.begin(), d.end());
  r.s_min = d.front(); r.s_med = d[72]; r.s_max = d.back();
  CK(hipFree(sink)); CK(hipFree(first)); CK(hipFree(start));
  CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  return r;
}

#define ROW(FL, T, V, L, NAME)                                                                                               \
  { Result r = run<FL, T, V, L>(fb, ft);                                                                                      \
    std::printf("%-10s mid %3d thr %3d VGPR %5.1f KB LDS : flood alone %.3f ms, with mid %.3f ms; mid starts after %7.1f / %7.1f / %7.1f us (min / median / max)\n", \
                NAME, T, V, L / 1024.0, r.f_alone, r.f_with, r.s_min, r.s_med, r.s_max); }

int main() {
  {
    const int fb = 256 * 7 * 25; const unsigned long long ft = 1200;     // fast_box-like: 12 us, 7 per CU, ~300 us in all
    std::printf("flood shaped like k_fast_box (256 thr, ~64 VGPR, 22.6 KB LDS, 12 us per workgroup, %d workgroups)\n", fb);
    ROW(23142, 256, 64, 20 * 1024, "fast_box")
    ROW(23142, 256, 128, 20 * 1024, "fast_box")
    ROW(23142, 256, 128, 44 * 1024, "fast_box")
    ROW(23142, 512, 128, 2 * 1024, "fast_box")
    ROW(23142, 512, 128, 44 * 1024, "fast_box")
    ROW(23142, 256, 192, 44 * 1024, "fast_box")
    ROW(23142, 256, 256, 8 * 1024, "fast_box")
    ROW(23142, 512, 256, 8 * 1024, "fast_box")
    ROW(23142, 512, 256, 140 * 1024, "fast_box")
    run_chain<23142, 0>("fast_box", fb, ft);
    run_chain<23142, 3>("fast_box", fb, ft);
  }
  {
    const int fb = 256 * 4 * 8; const unsigned long long ft = 4000;      // brief-like: 40 us, 4 per CU, ~320 us in all
    std::printf("flood shaped like k_brief (256 thr, ~64 VGPR, 39 KB LDS, 40 us per workgroup, %d workgroups)\n", fb);
    ROW(39936, 256, 128, 2 * 1024, "brief")
    ROW(39936, 512, 128, 2 * 1024, "brief")
    ROW(39936, 256, 256, 2 * 1024, "brief")
    ROW(39936, 512, 128, 44 * 1024, "brief")
    ROW(39936, 256, 128, 30 * 1024, "brief")
    ROW(39936, 512, 256, 8 * 1024, "brief")
  }
  return 0;
}
