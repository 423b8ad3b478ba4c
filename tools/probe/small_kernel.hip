// small_kernel.hip — why do short kernels over ~5000 workgroups take ~70 us?  Variants of "load one int, maybe exit".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)
struct Big { int a[300]; const int* p[60]; };
__global__ __launch_bounds__(256) void k_small(const int* n, int* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n[blockIdx.y]) return;
  out[i] = i;
}
__global__ __launch_bounds__(256) void k_big(const Big g, int* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.p[7][blockIdx.y] + g.a[123]) return;
  out[i] = i;
}
__global__ __launch_bounds__(256) void k_chain(const int* n, const int* idx, int* out, int depth) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n[blockIdx.y]) return;
  int v = i;
  for (int d = 0; d < depth; ++d) v = idx[(v * 97 + d) & 0xFFFFF];
  out[i] = v;
}
int main() {
  int *n, *out, *idx; CK(hipMalloc(&n, 4 * 160)); CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&idx, 4 << 20));
  int hn[160]; for (int i = 0; i < 160; ++i) hn[i] = 2600;
  CK(hipMemcpy(n, hn, sizeof hn, hipMemcpyHostToDevice)); CK(hipMemset(idx, 0, 4 << 20));
  Big g; for (int i = 0; i < 300; ++i) g.a[i] = 0; for (int i = 0; i < 60; ++i) g.p[i] = n;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto t = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); std::printf("%-64s %.1f us\n", name, ms * 100);
  };
  t("grid (32,160) x 256, pointer args, 11 of 32 blocks/stream useful", [&] { k_small<<<dim3(32, 160), 256>>>(n, out); });
  t("grid (11,160) x 256, pointer args", [&] { k_small<<<dim3(11, 160), 256>>>(n, out); });
  t("grid (32,160) x 256, 1.7 KB struct by value", [&] { k_big<<<dim3(32, 160), 256>>>(g, out); });
  t("grid (11,160) x 256, dependent chain of 4 loads", [&] { k_chain<<<dim3(11, 160), 256>>>(n, idx, out, 4); });
  t("grid (11,160) x 256, dependent chain of 16 loads", [&] { k_chain<<<dim3(11, 160), 256>>>(n, idx, out, 16); });
  t("grid (44,160) x 64,  dependent chain of 16 loads", [&] { k_chain<<<dim3(44, 160), 64>>>(n, idx, out, 16); });
  return 0;
}
