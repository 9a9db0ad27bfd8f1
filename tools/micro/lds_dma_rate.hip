// Microbenchmark (round 6): what does the set walk's record traffic through the LDS cost per CU?  A wave fetches 64
// records of 112 bytes the way the walker does -- seven lanes per record, nine records per `global_load_lds_dwordx4`,
// eight instructions -- from a table that stays in the L2 (no fabric traffic: this measures the CU side only), and
// reads its lane's record back with seven `ds_read_b128`.  Variants:
//   dma+read : both (the walker's fetch and read-back)
//   dma      : the LDS-DMA loads only
//   read     : the read-back only
//   vgpr+write+read : the same bytes through registers -- `global_load_dwordx4` into VGPRs, `ds_write_b128`, then the
//                     read-back (what the fetch would be without LDS-DMA)
// for 2 .. 16 waves per CU (workgroups of 128 threads with the walker's 20 KB of LDS).  Prints bytes per clock and CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_rate lds_dma_rate.hip && ./lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kThreads = 128;

template <int kMode>
__global__ __launch_bounds__(kThreads) void Kernel(const uint4 *__restrict__ table, uint32_t n_records, uint32_t iters,
                                                   uint32_t *__restrict__ sink, unsigned long long *__restrict__ clocks) {
  __shared__ uint4 stage[kThreads / 64][448];
  __shared__ uint32_t pad[2560];  // (the walker's other 10 KB per workgroup: same occupancy limit)
  const uint32_t lane = threadIdx.x & 63u;
  uint4 *stage_w = stage[threadIdx.x >> 6];
  if (threadIdx.x == 0) pad[0] = 0;
  const uint32_t grp = lane / 7u, part = lane - 7u * grp;
  uint32_t state = (blockIdx.x * kThreads + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  const unsigned long long t0 = clock64();
  for (uint32_t it = 0; it < iters; ++it) {
    state = state * 1664525u + 1013904223u;
    const uint32_t want = (state >> 8) % n_records;  // this lane's record
    if (kMode == 0 || kMode == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t src = (uint32_t)r * 9u + grp;
        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)want);
        if (lane < 63u && src < 64u) {
          const uint4 *g = table + (size_t)v * 8 + part;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                           (__attribute__((address_space(3))) void *)(stage_w + r * 63), 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (kMode == 3) {
      uint4 v4[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t src = (uint32_t)r * 9u + grp;
        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)want);
        v4[r] = make_uint4(0, 0, 0, 0);
        if (lane < 63u && src < 64u) v4[r] = table[(size_t)v * 8 + part];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t src = (uint32_t)r * 9u + grp;
        if (lane < 63u && src < 64u) stage_w[r * 63 + lane] = v4[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
    if (kMode == 0 || kMode == 2 || kMode == 3) {
      const uint4 *mine = stage_w + lane * 7;
      const uint4 a = mine[0], b = mine[1], c = mine[2], d = mine[3], e = mine[4], f = mine[5], g = mine[6];
      acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
    state += acc & 1u;
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) atomicAdd(clocks, t1 - t0);
  if (acc == 0x12345678u) sink[0] = acc + pad[1];
}

template <int kMode>
static void Run(const char *name, const uint4 *table, uint32_t n_records, uint32_t *sink, unsigned long long *clocks, int groups_per_cu) {
  const uint32_t iters = 2000, blocks = 256 * groups_per_cu;
  hipMemset(clocks, 0, 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((Kernel<kMode>), dim3(blocks), dim3(kThreads), 0, 0, table, n_records, 200, sink, clocks);
  hipMemset(clocks, 0, 8);
  hipEventRecord(a);
  hipLaunchKernelGGL((Kernel<kMode>), dim3(blocks), dim3(kThreads), 0, 0, table, n_records, iters, sink, clocks);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long h = 0;
  hipMemcpy(&h, clocks, 8, hipMemcpyDeviceToHost);
  const double waves = (double)blocks * (kThreads / 64);
  const double ticks_per_iter = (double)h / waves / iters;
  const double bytes_per_iter = 64.0 * 112.0;  // per wave and iteration, each direction
  const double waves_per_cu = waves / 256.0;
  printf("%-16s %2.0f waves/CU: %8.1f ticks per wave-iteration, %6.1f us total, %6.1f B/tick/CU of records\n", name, waves_per_cu,
         ticks_per_iter, ms * 1e3, bytes_per_iter * waves_per_cu / ticks_per_iter);
}

int main() {
  uint32_t *sink;
  unsigned long long *clocks;
  hipMalloc(&sink, 4);
  hipMalloc(&clocks, 8);
  const uint32_t n_records = 8192;  // 1 MB: L2-resident
  uint4 *table;
  hipMalloc(&table, (size_t)n_records * 128);
  hipMemset(table, 1, (size_t)n_records * 128);
  for (int g : {1, 2, 4, 6, 7, 8}) {
    Run<0>("dma+read", table, n_records, sink, clocks, g);
    Run<1>("dma", table, n_records, sink, clocks, g);
    Run<2>("read", table, n_records, sink, clocks, g);
    Run<3>("vgpr+write+read", table, n_records, sink, clocks, g);
  }
  return 0;
}
