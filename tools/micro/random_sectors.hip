// Microbenchmark (documentation of the walker's memory ceiling): rate of independent
// random 64-byte sector reads from a table far larger than L2 / larger than the Infinity Cache, issued
// the way the walker issues them (four lanes of a quad fetch the four 16-byte parts of one sector),
// with K sectors in flight per quad.  Prints GB/s of sectors and G sectors/s.
//   hipcc --offload-arch=gfx950 -O3 -o random_sectors random_sectors.hip && ./random_sectors
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int K>
__global__ __launch_bounds__(256) void Gather(const uint4 *__restrict__ table, uint64_t n_sectors, uint32_t iters,
                                              uint32_t *__restrict__ sink) {
  const uint32_t quad = (blockIdx.x * 256 + threadIdx.x) >> 2, part = threadIdx.x & 3;
  uint64_t state = 0x9E3779B97F4A7C15ull * (quad + 1);
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t sec = (state >> 20) % n_sectors;
      v[k] = table[sec * 4 + part];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k].x ^ v[k].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int K>
static void Run(const uint4 *table, uint64_t n_sectors, uint32_t *sink, const char *label, int blocks_per_cu) {
  const uint32_t iters = 2048 / K, blocks = 256 * blocks_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(Gather<K>, dim3(blocks), dim3(256), 0, 0, table, n_sectors, iters, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(Gather<K>, dim3(blocks), dim3(256), 0, 0, table, n_sectors, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double sectors = (double)blocks * 64 * iters * K;
  printf("%s  %2d sectors in flight per quad, %d workgroups/CU: %7.1f GB/s  %6.2f G sectors/s\n", label, K,
         blocks_per_cu, sectors * 64 / ms / 1e6, sectors / ms / 1e6);
}

int main() {
  uint32_t *sink;
  hipMalloc(&sink, 4);
  const double sizes_mb[] = {2, 64, 200, 1300, 8000};
  for (double mb : sizes_mb) {
    const uint64_t n_sectors = (uint64_t)(mb * 1e6 / 64);
    uint4 *table;
    if (hipMalloc(&table, n_sectors * 64) != hipSuccess) return 1;
    hipMemset(table, 1, n_sectors * 64);
    char label[64];
    snprintf(label, sizeof label, "table %6.0f MB:", mb);
    Run<1>(table, n_sectors, sink, label, 4);
    Run<4>(table, n_sectors, sink, label, 4);
    Run<4>(table, n_sectors, sink, label, 8);
    Run<16>(table, n_sectors, sink, label, 8);
    hipFree(table);
  }
  return 0;
}
