// Microbenchmark: rate of independent random reads of aligned records of 64, 128 and 256 bytes (a group of
// 4 / 8 / 16 lanes fetches the 16-byte parts of one record), K records in flight per group.  Answers
// whether a 128-byte record costs one request or two in the regime the tree walker runs in.
//   hipcc --offload-arch=gfx950 -O3 -o random_lines random_lines.hip && ./random_lines
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int K, int PARTS>
__global__ __launch_bounds__(256) void Gather(const uint4 *__restrict__ table, uint64_t n_records, uint32_t iters,
                                              uint32_t *__restrict__ sink) {
  const uint32_t grp = (blockIdx.x * 256 + threadIdx.x) / PARTS, part = threadIdx.x % PARTS;
  uint64_t state = 0x9E3779B97F4A7C15ull * (grp + 1);
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t rec = (state >> 20) % n_records;
      v[k] = table[rec * PARTS + part];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k].x ^ v[k].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int K, int PARTS>
static void Run(const uint4 *table, uint64_t bytes, uint32_t *sink, double mb) {
  const uint64_t n_records = bytes / (16 * PARTS);
  const uint32_t iters = 1024 / K, blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((Gather<K, PARTS>), dim3(blocks), dim3(256), 0, 0, table, n_records, iters, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((Gather<K, PARTS>), dim3(blocks), dim3(256), 0, 0, table, n_records, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double recs = (double)blocks * (256 / PARTS) * iters * K;
  printf("table %6.0f MB  record %3d B  %2d in flight: %7.1f GB/s  %6.2f G records/s\n", mb, 16 * PARTS, K,
         recs * 16 * PARTS / ms / 1e6, recs / ms / 1e6);
}

int main() {
  uint32_t *sink;
  hipMalloc(&sink, 4);
  const double sizes_mb[] = {2, 200, 1300};
  for (double mb : sizes_mb) {
    const uint64_t bytes = (uint64_t)(mb * 1e6) & ~255ull;
    uint4 *table;
    if (hipMalloc(&table, bytes) != hipSuccess) return 1;
    hipMemset(table, 1, bytes);
    Run<4, 4>(table, bytes, sink, mb);
    Run<4, 8>(table, bytes, sink, mb);
    Run<4, 16>(table, bytes, sink, mb);
    Run<16, 4>(table, bytes, sink, mb);
    Run<16, 8>(table, bytes, sink, mb);
    Run<16, 16>(table, bytes, sink, mb);
    hipFree(table);
  }
  return 0;
}
