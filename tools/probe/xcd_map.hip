// Which XCD does workgroup b of a launch run on?  (s_getreg HW_REG_XCC_ID)
//   part 1: a few grid shapes on two HIP streams, one launch at a time          -> block b runs on XCD (q0 + b) % 8, q0 a property of the queue
//   part 2: launches queued back to back on one stream (no host sync between)    -> does q0 hold from launch to launch?
//   part 3: the same while a long kernel keeps another stream busy
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/xcd_map.bin tools/probe/xcd_map.hip && tools/probe/xcd_map.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(int* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)(v & 0xF);
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
__global__ void busy(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }
int main() {
  int* d; CK(hipMalloc(&d, 16 * 4096 * 4));
  std::vector<int> h(16 * 4096);
  hipStream_t s1, s2, s3; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2)); CK(hipStreamCreate(&s3));
  struct G { int x, y, z; } grids[] = {{1, 1, 1}, {2, 1, 1}, {20, 6, 2}, {3, 5, 1}};
  std::printf("part 1: one launch at a time\n");
  for (auto g : grids)
    for (int rep = 0; rep < 4; ++rep) {
      hipStream_t st = rep & 1 ? s2 : s1;
      hipLaunchKernelGGL(k, dim3(g.x, g.y, g.z), dim3(256), 0, st, d, 0);
      CK(hipStreamSynchronize(st));
      const int n = g.x * g.y * g.z;
      CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
      int rr = 1;
      for (int i = 1; i < n; ++i) rr &= (h[i] == ((h[0] + i) & 7));
      std::printf("  stream %d grid (%d,%d,%d): block 0 on XCD %d, %s\n", rep & 1, g.x, g.y, g.z, h[0], rr ? "round robin from there" : "NOT round robin");
    }
  for (int part = 2; part <= 3; ++part) {
    std::printf("part %d: 12 launches queued back to back on stream 0%s\n", part, part == 3 ? ", stream 2 busy with 300 blocks" : "");
    if (part == 3) hipLaunchKernelGGL(busy, dim3(300), dim3(256), 0, s3, 20000);
    const int sizes[12] = {1, 157, 3, 240, 1, 8, 9, 1, 314, 1, 2, 1};
    for (int q = 0; q < 12; ++q) hipLaunchKernelGGL(k, dim3(sizes[q]), dim3(256), 0, s1, d + q * 4096, q % 3 == 1 ? 50 : 0);
    CK(hipStreamSynchronize(s1));
    CK(hipMemcpy(h.data(), d, 12 * 4096 * 4, hipMemcpyDeviceToHost));
    for (int q = 0; q < 12; ++q) {
      int rr = 1;
      for (int i = 1; i < sizes[q]; ++i) rr &= (h[q * 4096 + i] == ((h[q * 4096] + i) & 7));
      std::printf("  launch %2d (%3d blocks): block 0 on XCD %d, %s\n", q, sizes[q], h[q * 4096], rr ? "round robin" : "NOT round robin");
    }
    CK(hipDeviceSynchronize());
  }
  return 0;
}
