// Where does hipMalloc get slow?  Allocate chunks of argv[1] GiB (default 2) up to argv[2] GiB (default 200) and
// print every call that took more than 5 ms together with the total allocated so far.
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const double chunk_gb = argc > 1 ? atof(argv[1]) : 2.0, limit_gb = argc > 2 ? atof(argv[2]) : 200.0;
  (void)hipFree(0);
  std::vector<void *> ps;
  double total = 0, slow = 0;
  const double t_all = now();
  while (total + chunk_gb <= limit_gb) {
    void *p = nullptr;
    const double t0 = now();
    if (hipMalloc(&p, (size_t)(chunk_gb * (1ull << 30))) != hipSuccess) { printf("failed at %.1f GiB\n", total); break; }
    const double ms = (now() - t0) * 1e3;
    total += chunk_gb;
    ps.push_back(p);
    if (ms > 5) { printf("  call %zu (total %.1f GiB): %.1f ms\n", ps.size(), total, ms); slow += ms; }
  }
  printf("%zu x %.2f GiB: %.1f ms in all, %.1f ms in slow calls\n", ps.size(), chunk_gb, (now() - t_all) * 1e3, slow);
  for (void *p : ps) (void)hipFree(p);
  return 0;
}
