// How long hipMalloc / hipFree / first touch take on this box, by size (why the CLI
// reserves its batch scratch before the mapping span).
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void Touch(char *p, size_t n, size_t stride) {
  for (size_t i = (blockIdx.x * 256ull + threadIdx.x) * stride; i < n; i += (size_t)gridDim.x * 256 * stride) p[i] = 1;
}
int main() {
  hipFree(0);
  for (double gb : {0.25, 1.0, 4.0, 16.0, 64.0}) {
    const size_t n = (size_t)(gb * (1ull << 30));
    void *p = nullptr;
    double t0 = now();
    if (hipMalloc(&p, n) != hipSuccess) { printf("%.2f GB: hipMalloc failed\n", gb); continue; }
    double t1 = now();
    hipLaunchKernelGGL(Touch, dim3(4096), dim3(256), 0, 0, (char *)p, n, (size_t)4096);
    hipDeviceSynchronize();
    double t2 = now();
    hipLaunchKernelGGL(Touch, dim3(4096), dim3(256), 0, 0, (char *)p, n, (size_t)4096);
    hipDeviceSynchronize();
    double t3 = now();
    hipFree(p);
    double t4 = now();
    printf("%6.2f GB: hipMalloc %8.2f ms  first touch (1 B / 4 KB) %8.2f ms  second %8.2f ms  hipFree %8.2f ms\n", gb,
           (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
  }
  void *h = nullptr;
  double t0 = now();
  hipHostMalloc(&h, 1ull << 30, hipHostMallocDefault);
  double t1 = now();
  hipHostFree(h);
  printf("hipHostMalloc 1 GB: %.1f ms\n", (t1 - t0) * 1e3);
  return 0;
}
