// Microbenchmark: does a cache-policy modifier on the load change what a random 64-byte record costs beyond L2?
// Same gather as random_lines.hip (groups of 4 lanes fetch one aligned 64-byte record, 16 records in flight per
// group, 1.3 GB table), the load a buffer load with the modifier under test.  Run once plainly for the rates
// and once under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum` (kernel-trace only)
// for the sizes of the requests each variant sends to the fabric.
//   hipcc --offload-arch=gfx950 -O3 -o random_policy random_policy.hip && ./random_policy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// (buffer loads: the policy bits travel in the builtin's aux operand -- bit 0 sc0, bit 1 nt, bit 4 sc1 -- and the
// compiler tracks the loads itself)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int AUX>
struct Pol {
  static constexpr int kAux = AUX;
  static const char *Name() {
    static const char *names[32] = {};
    switch (AUX) {
      case 0: return "(none)";
      case 1: return "sc0";
      case 2: return "nt";
      case 3: return "sc0 nt";
      case 16: return "sc1";
      case 17: return "sc0 sc1";
      case 18: return "sc1 nt";
      case 19: return "sc0 sc1 nt";
    }
    return names[0] ? names[0] : "?";
  }
};
typedef Pol<0> PolNone;
typedef Pol<2> PolNt;
typedef Pol<1> PolSc0;
typedef Pol<16> PolSc1;
typedef Pol<17> PolSc01;
typedef Pol<18> PolSc1Nt;
typedef Pol<19> PolAll;

template <class P, int K, int PARTS>
__global__ __launch_bounds__(256) void GatherPol(const uint4 *__restrict__ table, uint64_t n_records, uint32_t iters,
                                                 uint32_t *__restrict__ sink) {
  const uint32_t grp = (blockIdx.x * 256 + threadIdx.x) / PARTS, part = threadIdx.x % PARTS;
  uint64_t state = 0x9E3779B97F4A7C15ull * (grp + 1);
  uint32_t acc = 0;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, (int)(n_records * PARTS * 16), 0x00020000);
  for (uint32_t it = 0; it < iters; ++it) {
    u32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t rec = (state >> 20) % n_records;
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((rec * PARTS + part) * 16), 0, P::kAux);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k].x ^ v[k].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <class Pol, int K, int PARTS>
static void Run(const uint4 *table, uint64_t bytes, uint32_t *sink) {
  const uint64_t n_records = bytes / (16 * PARTS);
  const uint32_t iters = 1024 / K, blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((GatherPol<Pol, K, PARTS>), dim3(blocks), dim3(256), 0, 0, table, n_records, iters, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((GatherPol<Pol, K, PARTS>), dim3(blocks), dim3(256), 0, 0, table, n_records, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double recs = (double)blocks * (256 / PARTS) * iters * K;
  printf("record %3d B  policy %-12s: %7.1f GB/s  %6.2f G records/s\n", 16 * PARTS, Pol::Name(),
         recs * 16 * PARTS / ms / 1e6, recs / ms / 1e6);
}

int main() {
  uint32_t *sink;
  hipMalloc(&sink, 4);
  const uint64_t bytes = 1300000000ull & ~255ull;
  uint4 *table;
  if (hipMalloc(&table, bytes) != hipSuccess) return 1;
  hipMemset(table, 1, bytes);
  Run<PolNone, 16, 4>(table, bytes, sink);
  Run<PolNt, 16, 4>(table, bytes, sink);
  Run<PolSc0, 16, 4>(table, bytes, sink);
  Run<PolSc1, 16, 4>(table, bytes, sink);
  Run<PolSc01, 16, 4>(table, bytes, sink);
  Run<PolSc1Nt, 16, 4>(table, bytes, sink);
  Run<PolAll, 16, 4>(table, bytes, sink);
  Run<PolNone, 16, 8>(table, bytes, sink);
  Run<PolNt, 16, 8>(table, bytes, sink);
  Run<PolAll, 16, 8>(table, bytes, sink);
  // 32-byte records (two lanes each)
  Run<PolNone, 16, 2>(table, bytes, sink);
  Run<PolAll, 16, 2>(table, bytes, sink);
  hipFree(table);
  return 0;
}
