#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  for (double gb : {2.0, 3.9, 4.1, 6.0, 8.0}) {
    std::vector<void*> ps; double t0 = now();
    for (int i = 0; i < 8; ++i) { void *p = nullptr; if (hipMalloc(&p, (size_t)(gb * (1ull << 30))) != hipSuccess) break; ps.push_back(p); }
    double t1 = now();
    printf("8 x %.1f GB: %.1f ms total\n", gb, (t1 - t0) * 1e3);
    for (void *p : ps) hipFree(p);
  }
  // VMM: reserve 32 GB of address space, map 8 x 4 GB physical chunks
  size_t gran = 0; hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  printf("granularity %zu\n", gran);
  void *va = nullptr; size_t chunk = 4ull << 30, total = 8 * chunk;
  double t0 = now();
  hipError_t e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
  printf("reserve: %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    for (int i = 0; i < 8; ++i) {
      hipMemGenericAllocationHandle_t h; e = hipMemCreate(&h, chunk, &prop, 0); if (e != hipSuccess) { printf("create: %s\n", hipGetErrorString(e)); break; }
      e = hipMemMap((char*)va + i * chunk, chunk, 0, h, 0); if (e != hipSuccess) { printf("map: %s\n", hipGetErrorString(e)); break; }
    }
    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, total, &ad, 1); printf("setaccess: %s\n", hipGetErrorString(e));
    printf("VMM 32 GB: %.1f ms\n", (now() - t0) * 1e3);
    e = hipMemset(va, 1, total); hipDeviceSynchronize(); printf("memset: %s %.1f ms\n", hipGetErrorString(e), (now() - t0) * 1e3);
  }
  return 0;
}
