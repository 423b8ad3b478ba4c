// dispatch_rate.hip — how fast does an MI355X start workgroups?  Empty / tiny kernels over a large grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)
template <int LDS>
__global__ void empty(int* out) {
  __shared__ int l[LDS / 4 + 1];
  if (LDS > 0) { l[threadIdx.x % (LDS / 4 + 1)] = 1; }
  if (out == nullptr && l[0] == 5) out[0] = 1;
}
template <int LDS>
float run(int grid, int block) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  empty<LDS><<<grid, block>>>(nullptr); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int k = 0; k < 10; ++k) empty<LDS><<<grid, block>>>(nullptr); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 10;
}
int main() {
  for (int block : {64, 128, 256, 512, 1024}) {
    const int grid = 76800 * 256 / block;
    const float a = run<0>(grid, block), b = run<15000>(grid, block);
    std::printf("block %4d grid %6d: no LDS %.3f ms (%.2f ns/WG, %.2f ns/wave)   15 KB LDS %.3f ms (%.2f ns/WG)\n", block, grid, a, a * 1e6 / grid,
                a * 1e6 / grid / (block / 64), b, b * 1e6 / grid);
  }
  return 0;
}
