// Wave-level bitonic network over 512 u32 keys (8 per lane) with DPP row operations: correctness against std::sort and
// throughput.  The network of sort.hip (BitonicWave32) was developed here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <random>

#define CE(a, b)                                   \
  do {                                             \
    const uint32_t lo_ = min(k[a], k[b]);          \
    const uint32_t hi_ = max(k[a], k[b]);          \
    k[a] = lo_;                                    \
    k[b] = hi_;                                    \
  } while (0)

__device__ __forceinline__ void InLane4(uint32_t (&k)[8]) { CE(0, 4); CE(1, 5); CE(2, 6); CE(3, 7); }
__device__ __forceinline__ void InLane2(uint32_t (&k)[8]) { CE(0, 2); CE(1, 3); CE(4, 6); CE(5, 7); }
__device__ __forceinline__ void InLane1(uint32_t (&k)[8]) { CE(0, 1); CE(2, 3); CE(4, 5); CE(6, 7); }
__device__ __forceinline__ void Mirror4(uint32_t (&k)[8]) { CE(0, 3); CE(1, 2); CE(4, 7); CE(5, 6); }
__device__ __forceinline__ void Mirror8(uint32_t (&k)[8]) { CE(0, 7); CE(1, 6); CE(2, 5); CE(3, 4); }

// exchange with the lane a quad permutation names; lower lanes keep the minimum
template <int CTRL, bool REV>
__device__ __forceinline__ void QuadStage(uint32_t (&k)[8], bool lower) {
  uint32_t o[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) o[r] = k[r];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o[REV ? 7 - r : r], CTRL, 0xF, 0xF, true);
    k[r] = lower ? min(o[r], p) : max(o[r], p);
  }
}
// exchange inside a row of 16 lanes where the lower / upper lanes are whole banks of four lanes: the
// minimum is computed in the lower banks and the maximum in the upper ones, one DPP instruction each
#define BANK_ASM4(CL, BL, CH, BH, r0, r1, r2, r3, s0, s1, s2, s3)                                             \
  asm volatile("s_nop 1\n\t"                                                                                   \
               "v_min_u32_dpp %0, %8, %4 " CL " row_mask:0xf bank_mask:" BL "\n\t"                             \
               "v_min_u32_dpp %1, %9, %5 " CL " row_mask:0xf bank_mask:" BL "\n\t"                             \
               "v_min_u32_dpp %2, %10, %6 " CL " row_mask:0xf bank_mask:" BL "\n\t"                            \
               "v_min_u32_dpp %3, %11, %7 " CL " row_mask:0xf bank_mask:" BL "\n\t"                            \
               "v_max_u32_dpp %0, %8, %4 " CH " row_mask:0xf bank_mask:" BH "\n\t"                             \
               "v_max_u32_dpp %1, %9, %5 " CH " row_mask:0xf bank_mask:" BH "\n\t"                             \
               "v_max_u32_dpp %2, %10, %6 " CH " row_mask:0xf bank_mask:" BH "\n\t"                            \
               "v_max_u32_dpp %3, %11, %7 " CH " row_mask:0xf bank_mask:" BH                                   \
               : "=&v"(k[r0]), "=&v"(k[r1]), "=&v"(k[r2]), "=&v"(k[r3])                                        \
               : "v"(o[r0]), "v"(o[r1]), "v"(o[r2]), "v"(o[r3]), "v"(o[s0]), "v"(o[s1]), "v"(o[s2]), "v"(o[s3]))
#define BANK_STAGE(CL, BL, CH, BH, REV)                        \
  do {                                                         \
    uint32_t o[8];                                             \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) o[r] = k[r]; \
    if (REV) {                                                 \
      BANK_ASM4(CL, BL, CH, BH, 0, 1, 2, 3, 7, 6, 5, 4);       \
      BANK_ASM4(CL, BL, CH, BH, 4, 5, 6, 7, 3, 2, 1, 0);       \
    } else {                                                   \
      BANK_ASM4(CL, BL, CH, BH, 0, 1, 2, 3, 0, 1, 2, 3);       \
      BANK_ASM4(CL, BL, CH, BH, 4, 5, 6, 7, 4, 5, 6, 7);       \
    }                                                          \
  } while (0)
template <bool REV>
__device__ __forceinline__ void PermStage(uint32_t (&k)[8], int addr, bool lower) {
  uint32_t o[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) o[r] = k[r];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t p = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)o[REV ? 7 - r : r]);
    k[r] = lower ? min(o[r], p) : max(o[r], p);
  }
}

__device__ __forceinline__ void BitonicWave32(uint32_t (&k)[8], uint32_t lane) {
  const bool l1 = (lane & 1u) == 0, l2 = (lane & 2u) == 0, l4 = (lane & 4u) == 0, l8 = (lane & 8u) == 0;
  const bool l16 = (lane & 16u) == 0, l32 = (lane & 32u) == 0;
  InLane1(k);                         // k = 2
  Mirror4(k); InLane1(k);             // 4
  Mirror8(k); InLane2(k); InLane1(k); // 8
  // 16: lanes in pairs
  QuadStage<0xB1, true>(k, l1);
  InLane4(k); InLane2(k); InLane1(k);
  // 32: lanes in quads
  QuadStage<0x1B, true>(k, l2);
  QuadStage<0xB1, false>(k, l1);
  InLane4(k); InLane2(k); InLane1(k);
  // 64: 8 lanes
  BANK_STAGE("row_half_mirror", "0x5", "row_half_mirror", "0xa", true);
  QuadStage<0x4E, false>(k, l2);
  QuadStage<0xB1, false>(k, l1);
  InLane4(k); InLane2(k); InLane1(k);
  // 128: 16 lanes
  BANK_STAGE("row_mirror", "0x3", "row_mirror", "0xc", true);
  BANK_STAGE("row_shl:4", "0x5", "row_shr:4", "0xa", false);
  QuadStage<0x4E, false>(k, l2);
  QuadStage<0xB1, false>(k, l1);
  InLane4(k); InLane2(k); InLane1(k);
  // 256: 32 lanes
  PermStage<true>(k, (int)((lane ^ 31u) << 2), l16);
  BANK_STAGE("row_shl:8", "0x3", "row_shr:8", "0xc", false);
  BANK_STAGE("row_shl:4", "0x5", "row_shr:4", "0xa", false);
  QuadStage<0x4E, false>(k, l2);
  QuadStage<0xB1, false>(k, l1);
  InLane4(k); InLane2(k); InLane1(k);
  // 512: 64 lanes
  PermStage<true>(k, (int)((lane ^ 63u) << 2), l32);
  PermStage<false>(k, (int)((lane ^ 16u) << 2), l16);
  BANK_STAGE("row_shl:8", "0x3", "row_shr:8", "0xc", false);
  BANK_STAGE("row_shl:4", "0x5", "row_shr:4", "0xa", false);
  QuadStage<0x4E, false>(k, l2);
  QuadStage<0xB1, false>(k, l1);
  InLane4(k); InLane2(k); InLane1(k);
  (void)l4; (void)l8;
}

__global__ __launch_bounds__(64) void TestKernel(const uint32_t *in, uint32_t *out, int reps) {
  const uint32_t lane = threadIdx.x;
  const uint32_t *src = in + (size_t)blockIdx.x * 512;
  uint32_t k[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) k[r] = src[lane * 8 + r];
  for (int i = 0; i < reps; ++i) {
    BitonicWave32(k, lane);
    if (i + 1 < reps) {
#pragma unroll
      for (int r = 0; r < 8; ++r) k[r] = k[r] * 2654435761u + (uint32_t)i;  // rescramble
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) out[(size_t)blockIdx.x * 512 + lane * 8 + r] = k[r];
}

int main(int argc, char **argv) {
  const int nb = 4096 * 4, reps = argc > 1 ? atoi(argv[1]) : 1;
  std::vector<uint32_t> h((size_t)nb * 512), o(h.size());
  std::mt19937 g(1);
  for (auto &x : h) x = g();
  uint32_t *d_in, *d_out;
  hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, h.size() * 4);
  hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  TestKernel<<<nb, 64>>>(d_in, d_out, 1);
  hipMemcpy(o.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (int b = 0; b < nb; ++b) {
    std::vector<uint32_t> ref(h.begin() + (size_t)b * 512, h.begin() + (size_t)(b + 1) * 512);
    std::sort(ref.begin(), ref.end());
    for (int i = 0; i < 512; ++i) bad += ref[i] != o[(size_t)b * 512 + i];
  }
  printf("mismatches: %zu\n", bad);
  const int R = 64;
  hipEventRecord(e0);
  TestKernel<<<nb, 64>>>(d_in, d_out, R);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%d networks of 512 keys: %.3f ms -> %.2f G keys/s, %.1f SIMD-cycles per network at 2.4 GHz\n", nb * R, ms,
         (double)nb * R * 512 / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / ((double)nb * R));
  return bad != 0;
}
