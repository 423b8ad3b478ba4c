// contention.hip — what slows a latency-bound kernel (dependent HBM/L2 loads, one 512-thread workgroup on 160 CUs, like
// the frame kernel) when another kernel shares the GPU: arithmetic on the other CUs, or memory traffic?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__global__ __launch_bounds__(512) void chase(const int* next, int steps, int* out) {
  asm volatile("v_mov_b32 v250, 0" ::: "v250");   // 256 VGPRs: the workgroup owns its CU
  int i = (blockIdx.x * 512 + threadIdx.x) * 97 % (1 << 22);
  for (int k = 0; k < steps; ++k) i = next[i];
  if (i == -1) out[0] = i;
}
__global__ __launch_bounds__(256) void valu_hog(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.3f, d = 0.7f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; c = c * b + a; d = d * b + c; }
  if (a + c + d == 123.f) out[0] = a;
}
__global__ __launch_bounds__(256) void mem_hog(const uint4* src, uint4* dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main() {
  const int N = 1 << 22;
  std::vector<int> h(N);
  unsigned long long x = 12345;
  for (int i = 0; i < N; ++i) { x = x * 6364136223846793005ull + 1442695040888963407ull; h[i] = (int)((x >> 33) % N); }
  int *next, *out; float* fo; uint4 *src, *dst;
  const size_t nb = 64ull << 20;   // 1 GiB each
  CK(hipMalloc(&next, N * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&fo, 4)); CK(hipMalloc(&src, nb * 16)); CK(hipMalloc(&dst, nb * 16));
  CK(hipMemcpy(next, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(src, 1, nb * 16));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_chase = [&](const char* name, int hog) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      if (hog == 1) valu_hog<<<256 * 64, 256, 0, sb>>>(fo, 200000);
      if (hog == 2) mem_hog<<<2048, 256, 0, sb>>>(src, dst, nb, 4);
      if (hog == 3) mem_hog<<<256, 256, 0, sb>>>(src, dst, nb / 8, 4);
      CK(hipEventRecord(e0, sa)); chase<<<160, 512, 0, sa>>>(next, 400, out); CK(hipEventRecord(e1, sa));
      CK(hipStreamSynchronize(sa));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipStreamSynchronize(sb));
      best = ms < best ? ms : best;
    }
    std::printf("%-46s chase of 400 dependent loads: %.3f ms (%.2f us per load)\n", name, best, best * 1e3f / 400);
  };
  chase<<<160, 512, 0, sa>>>(next, 10, out); CK(hipDeviceSynchronize());
  time_chase("alone", 0);
  time_chase("with a VALU-only kernel on the free CUs", 1);
  time_chase("with a streaming copy (2048 workgroups)", 2);
  time_chase("with a light streaming copy (256 workgroups)", 3);
  // how long the hogs take alone, for scale
  CK(hipEventRecord(e0, sb)); mem_hog<<<2048, 256, 0, sb>>>(src, dst, nb, 4); CK(hipEventRecord(e1, sb)); CK(hipStreamSynchronize(sb));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("streaming copy alone: %.3f ms = %.2f TB/s (read+write)\n", ms, 4.0 * 2 * nb * 16 / ms * 1e-9);
  return 0;
}
