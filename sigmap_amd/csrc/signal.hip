// signal.hip -- SignalBatch::AddSignal's conversion on the device (signal_batch.cc:139-151, the
// same loop for BLOW5 records at :195-207): raw ADC counts -> pA, samples outside (30, 200) pA are
// dropped, the survivors are compacted in order.  The reference then cuts its 4000-sample chunks out
// of the FILTERED array (sigmap.cc:637-648), which is why the whole read is converted before any
// chunk of it can be addressed.
//
//   scale = range / digitisation                (f32 division)
//   pa    = ((float)raw + offset) * scale       (f32 add, then f32 multiply: no contraction)
//   keep  iff pa > 30 && pa < 200
//
// Reads are cut into segments of kSeg raw samples, one workgroup per segment, two passes: count the
// survivors per segment; (one lane per read) exclusive scan of the read's segment counts; recompute
// and scatter.  Only the first `keep` filtered samples of a read are stored (its resident slot holds
// `keep` floats), but all of them are counted: the filtered length is the read length the PAF row
// reports (sigmap.cc:731, GetSignalLengthAt).
#include "common.h"

namespace smhip {
namespace {
constexpr uint32_t kSeg = 4096;  // raw samples per workgroup

__device__ __forceinline__ uint32_t FindRead(const uint32_t *__restrict__ seg_start, uint32_t n_reads, uint32_t s) {
  uint32_t lo = 0, hi = n_reads;  // seg_start[lo] <= s < seg_start[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (seg_start[mid] <= s) lo = mid;
    else hi = mid;
  }
  return lo;
}

// kScatter = false: seg_count[s] = survivors of segment s.  kScatter = true: the survivors are
// written to the read's slot at seg_base[s] + rank (ranks in sample order).
template <bool kScatter>
__global__ __launch_bounds__(256) void RawConvertKernel(const int16_t *__restrict__ raw, const RawReadMeta *__restrict__ meta,
                                                        const uint32_t *__restrict__ seg_start, uint32_t n_reads,
                                                        uint32_t *__restrict__ seg_count,
                                                        const uint32_t *__restrict__ seg_base, SlotView pa) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t r = FindRead(seg_start, n_reads, s);
  const RawReadMeta m = meta[r];
  const uint32_t first = (s - seg_start[r]) * kSeg;
  const uint32_t n = min(kSeg, m.n_raw - first);
  const int16_t *x = raw + m.raw_off + first;
  const float scale = m.range / m.digitisation;
  const uint32_t keep = pa.keep;
  float *out = kScatter ? pa.At(m.slot) : nullptr;
  uint32_t run = kScatter ? seg_base[s] : 0u;  // survivors before the current round (of the read / of the segment)
  for (uint32_t i0 = 0; i0 < n; i0 += 256) {
    const uint32_t i = i0 + tid;
    float v = 0.0f;
    bool ok = false;
    if (i < n) {
      v = ((float)x[i] + m.offset) * scale;
      ok = v > 30.0f && v < 200.0f;
    }
    const uint64_t b = __ballot(ok);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    const uint32_t c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
    if (kScatter) {
      const uint32_t before = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
      const uint32_t k = run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      if (ok && k < keep) out[k] = v;
    }
    run += c0 + c1 + c2 + c3;
    __syncthreads();
  }
  if (!kScatter && tid == 0) seg_count[s] = run;
}

__global__ __launch_bounds__(64) void RawSegScanKernel(const RawReadMeta *__restrict__ meta,
                                                       const uint32_t *__restrict__ seg_start, uint32_t n_reads,
                                                       const uint32_t *__restrict__ seg_count,
                                                       uint32_t *__restrict__ seg_base, uint32_t *__restrict__ slot_len,
                                                       uint32_t *__restrict__ read_len) {
  const uint32_t r = blockIdx.x * 64 + threadIdx.x;
  if (r >= n_reads) return;
  uint32_t acc = 0;
  for (uint32_t s = seg_start[r]; s < seg_start[r + 1]; ++s) {
    seg_base[s] = acc;
    acc += seg_count[s];
  }
  read_len[r] = acc;
  slot_len[meta[r].slot] = acc;
}

// Batch descriptors from (slot, chunk index): chunk c = the slot's filtered samples
// [chunk_index * chunk_size, + chunk_size) (sigmap.cc:647-652), clipped to what the slot holds.
__global__ void SelectChunksKernel(const uint32_t *__restrict__ slots, const uint32_t *__restrict__ chunk_index,
                                   uint32_t n_chunks, uint32_t chunk_size, SlotView pa,
                                   const uint32_t *__restrict__ slot_len, uint64_t *__restrict__ cbegin,
                                   uint32_t *__restrict__ clen) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const uint32_t slot = slots[c];
  const uint64_t b = (uint64_t)chunk_index[c] * chunk_size;
  const uint32_t have = min(slot_len[slot], pa.keep);
  // (the batch's chunk source is the null pointer: begin = the chunk's address in floats)
  cbegin[c] = (uint64_t)(reinterpret_cast<uintptr_t>(pa.At(slot)) / sizeof(float)) + b;
  clen[c] = b < have ? (uint32_t)min<uint64_t>(chunk_size, have - b) : 0u;
}
}  // namespace

uint32_t RawSegments(uint32_t n_raw) { return n_raw ? (n_raw + kSeg - 1) / kSeg : 1u; }

hipError_t LaunchRawConvert(smhip_ctx *ctx, const int16_t *raw, const RawReadMeta *meta, const uint32_t *seg_start,
                            uint32_t n_reads, uint32_t n_segs, uint32_t *seg_count, uint32_t *seg_base, SlotView pa,
                            uint32_t *slot_len, uint32_t *read_len) {
  if (n_reads == 0) return hipSuccess;
  hipLaunchKernelGGL(RawConvertKernel<false>, dim3(n_segs), dim3(256), 0, ctx->stream, raw, meta, seg_start, n_reads,
                     seg_count, nullptr, pa);
  hipLaunchKernelGGL(RawSegScanKernel, dim3((n_reads + 63) / 64), dim3(64), 0, ctx->stream, meta, seg_start, n_reads,
                     seg_count, seg_base, slot_len, read_len);
  hipLaunchKernelGGL(RawConvertKernel<true>, dim3(n_segs), dim3(256), 0, ctx->stream, raw, meta, seg_start, n_reads,
                     seg_count, seg_base, pa);
  return hipGetLastError();
}

hipError_t LaunchSelectChunks(smhip_ctx *ctx, const uint32_t *slots, const uint32_t *chunk_index, uint32_t n_chunks,
                              uint32_t chunk_size, SlotView pa, const uint32_t *slot_len, uint64_t *cbegin,
                              uint32_t *clen) {
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(SelectChunksKernel, dim3((n_chunks + 255) / 256), dim3(256), 0, ctx->stream, slots, chunk_index,
                     n_chunks, chunk_size, pa, slot_len, cbegin, clen);
  return hipGetLastError();
}
}  // namespace smhip
