// Calibration of rocprofv3's memory-side counters for the access shapes of the tree walker
// (seed_lookup.hip).  MI355X_MICROARCH.md: FETCH_SIZE is exact only for what it was calibrated on
// (it reads half the bytes of a wide coalesced stream); "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern".  Every kernel below touches a KNOWN
// number of bytes of a table far larger than L2 + Infinity Cache, each byte once (a random
// permutation without repeats inside a launch, so no cache level can absorb a re-read), in one of
// the walker's shapes; run under
//   rocprofv3 --pmc <counter> --kernel-trace ... -- ./calib_traffic
// (one counter set per pass) and compare the counters per kernel with the "known bytes" this
// program prints (tools/calib_summarize.py does that and writes profiles/r2_calibration.json).
//
//   hipcc --offload-arch=gfx950 -O3 -o calib_traffic calib_traffic.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x)                                                              \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

// A bijection of [0, n) for n a power of two: odd multiplier, xor-shift, odd multiplier.
__device__ __forceinline__ uint32_t Perm(uint32_t i, uint32_t mask) {
  i = (i * 0x9E3779B1u) & mask;
  i ^= i >> 15;
  i = (i * 0x85EBCA6Bu + 0x1234567u) & mask;
  i ^= i >> 13;
  i = (i * 0xC2B2AE35u) & mask;
  return i;
}
// (xor-shifts by >= half the width of a mask are bijections only on the full word: the host checks
//  that the map is a bijection for the mask used, and the known byte counts rely on that.)

// (1) stream: 16 B per lane, coalesced -- the shape the guide calibrated (expect FETCH_SIZE = 1/2).
__global__ __launch_bounds__(256) void CalibStreamRead(const uint4 *__restrict__ t, uint64_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = t[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// the same stream under another name: the eviction pass of (4), so that the per-kernel counter means of
// CalibStreamRead are those of the 2 GiB pass alone
__global__ __launch_bounds__(256) void CalibEvictRead(const uint4 *__restrict__ t, uint64_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = t[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// (2) the walker's node fetch: the four lanes of a quad fetch the four 16-byte parts of one random
// 64-byte record with global_load_lds_dwordx4 (16 records per instruction), records never repeated.
__global__ __launch_bounds__(256) void CalibQuadRecord64(const uint4 *__restrict__ t, uint32_t mask, uint32_t per_quad,
                                                         uint32_t *sink) {
  __shared__ uint4 stage[4][256];
  const uint32_t lane = threadIdx.x & 63u, quad = (blockIdx.x * 256u + threadIdx.x) >> 2;
  uint4 *stage_w = stage[threadIdx.x >> 6];
  uint32_t acc = 0;
  for (uint32_t it = 0; it < per_quad; ++it) {
    const uint32_t rec = Perm(quad * per_quad + it, mask);
    const uint4 *g = t + (size_t)rec * 4 + (lane & 3u);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)stage_w, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    { const uint4 a = stage_w[lane]; acc += a.x ^ a.y ^ a.z ^ a.w; }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// (3) the walker's leaf scan: 21 lanes per leaf block, three blocks per wave instruction; lane t reads
// the first half at block + t and the second half at block + roundup(n, 8) + t (n = 14: the mean
// leaf of the 12 Mb index; block = 2 * 16 slots = 512 B, 128-byte aligned).  Useful bytes 2*14*16 =
// 448 B per block; 64-byte sectors touched: 4 + 4 = 8 (512 B); 128-byte lines touched: 2 + 2.
__global__ __launch_bounds__(256) void CalibLeafBlock(const uint4 *__restrict__ t, uint32_t mask, uint32_t per_group,
                                                      uint32_t n, uint32_t *sink) {
  const uint32_t lane = threadIdx.x & 63u, grp = lane / 21u, tt = lane - grp * 21u;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  uint32_t acc = 0;
  const uint32_t rn = (n + 7u) & ~7u;
  for (uint32_t it = 0; it < per_group; ++it) {
    if (grp < 3 && tt < n) {
      const uint32_t blk = Perm((wave * 3u + grp) * per_group + it, mask);
      const uint4 *fh = t + (size_t)blk * 32u + tt;  // 32 slots of 16 B = 512-byte blocks
      const uint4 a = fh[0], b = fh[rn];
      acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// (4) the walker's stack spill: a lane writes 32 B (two float4) into its private strip
// ([lane][64 entries] x 32 B = 2 KB per lane), entry after entry; later read back the same way.
__global__ __launch_bounds__(256) void CalibSpillWrite(float4 *__restrict__ stk, uint32_t entries, uint32_t *sink) {
  const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
  float4 *s = stk + (size_t)gl * (entries * 2);
  for (uint32_t e = 0; e < entries; ++e) {
    s[2 * e] = make_float4(1.f, 2.f, (float)e, 4.f);
    s[2 * e + 1] = make_float4(5.f, 6.f, 7.f, (float)gl);
  }
}
__global__ __launch_bounds__(256) void CalibSpillRead(const float4 *__restrict__ stk, uint32_t entries, uint32_t *sink) {
  const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
  const float4 *s = stk + (size_t)gl * (entries * 2);
  float acc = 0;
  for (uint32_t e = entries; e-- > 0;) {
    const float4 a = s[2 * e], b = s[2 * e + 1];
    acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  }
  if (acc == 0.12345f) sink[0] = 1;
}

// (5) the walker's hit store: one 16-byte record per lane at a random 16-byte slot (never repeated).
__global__ __launch_bounds__(256) void CalibHitStore(uint4 *__restrict__ t, uint32_t mask, uint32_t per_lane) {
  const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
  for (uint32_t it = 0; it < per_lane; ++it) {
    const uint32_t slot = Perm(gl * per_lane + it, mask);
    t[slot] = make_uint4(gl, it, 3, 4);
  }
}

// (6) a lane-private 32-byte record read (the walker's query record fetch: two float4 of one lane).
__global__ __launch_bounds__(256) void CalibLaneRecord32(const uint4 *__restrict__ t, uint32_t mask, uint32_t per_lane,
                                                         uint32_t *sink) {
  const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < per_lane; ++it) {
    const uint32_t rec = Perm(gl * per_lane + it, mask);
    const uint4 a = t[(size_t)rec * 2], b = t[(size_t)rec * 2 + 1];
    acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void PermCheck(uint32_t mask, uint32_t *flags, uint32_t *bad) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= mask; i += (uint64_t)gridDim.x * 256) {
    const uint32_t p = Perm((uint32_t)i, mask);
    if (atomicOr(&flags[p >> 5], 1u << (p & 31)) & (1u << (p & 31))) atomicAdd(bad, 1u);
  }
}

static bool IsBijection(uint32_t mask) {
  uint32_t *flags, *bad, hb = 0;
  CHK(hipMalloc(&flags, ((size_t)mask + 1) / 8 + 4));
  CHK(hipMalloc(&bad, 4));
  CHK(hipMemset(flags, 0, ((size_t)mask + 1) / 8 + 4));
  CHK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(PermCheck, dim3(4096), dim3(256), 0, 0, mask, flags, bad);
  CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  CHK(hipFree(flags));
  CHK(hipFree(bad));
  return hb == 0;
}

int main() {
  uint32_t *sink;
  CHK(hipMalloc(&sink, 4));
  const uint64_t table_bytes = 4ull << 30;  // 4 GiB: 16x the Infinity Cache
  uint4 *table;
  CHK(hipMalloc(&table, table_bytes));
  CHK(hipMemset(table, 1, table_bytes));
  CHK(hipDeviceSynchronize());
  const uint32_t blocks = 256 * 8;
  printf("{\n");
  // (1) stream 2 GiB
  {
    const uint64_t n16 = (2ull << 30) / 16;
    hipLaunchKernelGGL(CalibStreamRead, dim3(blocks), dim3(256), 0, 0, table, n16, sink);
    CHK(hipDeviceSynchronize());
    printf(" \"CalibStreamRead\": {\"read_bytes\": %llu, \"write_bytes\": 0},\n", (unsigned long long)(n16 * 16));
  }
  // (2) 2^24 random 64-byte records of 2^26 (= 4 GiB): one quarter of the table, each once
  {
    const uint32_t mask = (1u << 26) - 1, per_quad = 128;  // 2048*64 quads * 128 = 2^24 records
    if (!IsBijection(mask)) { fprintf(stderr, "perm not a bijection at 26 bits\n"); return 1; }
    hipLaunchKernelGGL(CalibQuadRecord64, dim3(blocks), dim3(256), 0, 0, table, mask, per_quad, sink);
    CHK(hipDeviceSynchronize());
    const uint64_t recs = (uint64_t)blocks * 64 * per_quad;
    printf(" \"CalibQuadRecord64\": {\"read_bytes\": %llu, \"write_bytes\": 0, \"records\": %llu},\n",
           (unsigned long long)(recs * 64), (unsigned long long)recs);
  }
  // (3) leaf blocks: 2^23 blocks of 512 B = 4 GiB; each group reads per_group blocks
  {
    const uint32_t mask = (1u << 23) - 1, per_group = 128, n = 14;
    if (!IsBijection(mask)) { fprintf(stderr, "perm not a bijection at 23 bits\n"); return 1; }
    hipLaunchKernelGGL(CalibLeafBlock, dim3(blocks), dim3(256), 0, 0, table, mask, per_group, n, sink);
    CHK(hipDeviceSynchronize());
    const uint64_t blks = (uint64_t)blocks * 4 * 3 * per_group;  // 4 waves x 3 groups
    printf(" \"CalibLeafBlock\": {\"read_bytes\": %llu, \"sector_bytes\": %llu, \"line_bytes\": %llu, \"write_bytes\": 0, "
           "\"blocks\": %llu},\n",
           (unsigned long long)(blks * 2 * n * 16), (unsigned long long)(blks * 512), (unsigned long long)(blks * 512),
           (unsigned long long)blks);
  }
  // (4) spills: 2048*256 lanes x 64 entries x 32 B = 1 GiB written, then read
  {
    const uint32_t entries = 64;
    hipLaunchKernelGGL(CalibSpillWrite, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<float4 *>(table), entries, sink);
    CHK(hipDeviceSynchronize());
    // evict: stream over the other 3 GiB
    hipLaunchKernelGGL(CalibEvictRead, dim3(blocks), dim3(256), 0, 0, table + (1ull << 30) / 16, (3ull << 30) / 16, sink);
    hipLaunchKernelGGL(CalibSpillRead, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float4 *>(table), entries, sink);
    CHK(hipDeviceSynchronize());
    const uint64_t b = (uint64_t)blocks * 256 * entries * 32;
    printf(" \"CalibSpillWrite\": {\"read_bytes\": 0, \"write_bytes\": %llu},\n", (unsigned long long)b);
    printf(" \"CalibSpillRead\": {\"read_bytes\": %llu, \"write_bytes\": 0},\n", (unsigned long long)b);
  }
  // (5) hit stores: 2^24 random 16-byte slots of 2^28
  {
    const uint32_t mask = (1u << 28) - 1, per_lane = 32;
    if (!IsBijection(mask)) { fprintf(stderr, "perm not a bijection at 28 bits\n"); return 1; }
    hipLaunchKernelGGL(CalibHitStore, dim3(blocks), dim3(256), 0, 0, table, mask, per_lane);
    CHK(hipDeviceSynchronize());
    const uint64_t st = (uint64_t)blocks * 256 * per_lane;
    printf(" \"CalibHitStore\": {\"read_bytes\": 0, \"write_bytes\": %llu, \"stores\": %llu},\n",
           (unsigned long long)(st * 16), (unsigned long long)st);
  }
  // (6) lane-private 32-byte records: 2^24 of 2^27
  {
    const uint32_t mask = (1u << 27) - 1, per_lane = 32;
    if (!IsBijection(mask)) { fprintf(stderr, "perm not a bijection at 27 bits\n"); return 1; }
    hipLaunchKernelGGL(CalibLaneRecord32, dim3(blocks), dim3(256), 0, 0, table, mask, per_lane, sink);
    CHK(hipDeviceSynchronize());
    const uint64_t recs = (uint64_t)blocks * 256 * per_lane;
    printf(" \"CalibLaneRecord32\": {\"read_bytes\": %llu, \"write_bytes\": 0, \"records\": %llu}\n",
           (unsigned long long)(recs * 32), (unsigned long long)recs);
  }
  printf("}\n");
  return 0;
}
