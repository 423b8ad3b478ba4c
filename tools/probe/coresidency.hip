// coresidency.hip — which resource of a long-running "frame-like" workgroup keeps short "image-like" workgroups off
// its CU on MI355X?  Kernel A spins for a fixed wall time with a given footprint (threads, VGPRs, LDS, scratch);
// kernel B is many short workgroups.  B's duration is measured alone and while A occupies every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

template <int THREADS, int VGPRS, int LDS_KB, int SCRATCH>
__global__ __launch_bounds__(THREADS) void spin(unsigned long long ticks, int* sink, int* started) {
  __shared__ unsigned char lds[LDS_KB * 1024 + 16];
  volatile int priv[SCRATCH / 4 + 1];
  priv[threadIdx.x % (SCRATCH / 4 + 1)] = 1;
  lds[threadIdx.x] = 1;
  if (VGPRS > 128) asm volatile("v_mov_b32 v250, 0" ::: "v250");
  else if (VGPRS > 64) asm volatile("v_mov_b32 v120, 0" ::: "v120");
  if (threadIdx.x == 0) { __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
  if (lds[threadIdx.x ^ 1] == 7 && priv[0] == 9) *sink = 1;
}
__global__ __launch_bounds__(256) void shortwork(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
  if (a == 123.f) out[0] = a;
}
template <int THREADS, int VGPRS, int LDS_KB, int SCRATCH>
void run(const char* name, int blocksA) {
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int* sink; float* out; CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nb = 256 * 8 * 4, iters = 4000;
  // B alone
  shortwork<<<nb, 256, 0, sb>>>(out, iters); CK(hipStreamSynchronize(sb));
  CK(hipEventRecord(e0, sb)); shortwork<<<nb, 256, 0, sb>>>(out, iters); CK(hipEventRecord(e1, sb)); CK(hipStreamSynchronize(sb));
  float alone; CK(hipEventElapsedTime(&alone, e0, e1));
  // B while A holds the CUs (A spins 3 ms = 300000 ticks of 100 MHz)
  int* started; CK(hipHostMalloc(&started, 4, hipHostMallocMapped)); *started = 0;
  spin<THREADS, VGPRS, LDS_KB, SCRATCH><<<blocksA, THREADS, 0, sa>>>(300000ull, sink, started);
  int seen = 0;
  for (int spin_i = 0; spin_i < 2000000; ++spin_i) { seen = *(volatile int*)started; if (seen >= blocksA) break; }
  CK(hipEventRecord(e0, sb)); shortwork<<<nb, 256, 0, sb>>>(out, iters); CK(hipEventRecord(e1, sb));
  CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sa));
  float with; CK(hipEventElapsedTime(&with, e0, e1));
  std::printf("%-44s A=%3d blocks (%3d running when B was launched): B alone %.3f ms, B with A resident %.3f ms\n", name, blocksA, seen, alone, with);
  CK(hipFree(sink)); CK(hipFree(out));
}
int main() {
  run<512, 256, 74, 0>("512 thr, 256 VGPR, 74 KB LDS", 256);
  run<512, 256, 1, 0>("512 thr, 256 VGPR,  1 KB LDS", 256);
  run<512, 128, 74, 0>("512 thr, 128 VGPR, 74 KB LDS", 256);
  run<512, 128, 1, 0>("512 thr, 128 VGPR,  1 KB LDS", 256);
  run<256, 256, 74, 0>("256 thr, 256 VGPR, 74 KB LDS", 256);
  run<256, 256, 1, 0>("256 thr, 256 VGPR,  1 KB LDS", 256);
  run<512, 64, 1, 0>("512 thr,  64 VGPR,  1 KB LDS", 256);
  run<512, 128, 1, 2048>("512 thr, 128 VGPR,  1 KB LDS, 2 KB scratch", 256);
  run<512, 256, 74, 0>("512 thr, 256 VGPR, 74 KB LDS", 160);
  run<256, 256, 74, 0>("256 thr, 256 VGPR, 74 KB LDS (2/CU?)", 512);
  return 0;
}
