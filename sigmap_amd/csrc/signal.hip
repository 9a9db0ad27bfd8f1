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

// ---- pieces of reads pulled over PCIe by the GPU itself (smhip_reads_upload_pieces).  The host keeps the reads'
// ADC counts in PINNED memory, which the device can address: workgroup r copies the n samples of piece r into the
// staging buffer, where the conversion kernels above find them.  A piece keeps its phase inside a 16-byte unit
// (sample i of a piece whose first sample sits at byte a of its unit lands at sample 8 dst + a / 2 + i), so that
// all of it but a few samples at either end moves as aligned 16-byte loads and stores, a kilobyte per wave and
// instruction; nothing outside [src, src + 2 n) is read.  One launch for any number of scattered pieces, no staging
// copy on the host.
__global__ __launch_bounds__(256) void GatherPiecesKernel(const PieceSrc *__restrict__ src, int16_t *__restrict__ staged) {
  const PieceSrc p = src[blockIdx.x];
  const uint32_t a = (uint32_t)(p.src & 15ull) >> 1;                     // samples before the piece in its first unit
  const int16_t *from = reinterpret_cast<const int16_t *>(p.src);
  int16_t *to = staged + (size_t)p.dst * 8 + a;
  const uint32_t head = min(p.n, (8u - a) & 7u);                         // samples up to the first unit boundary
  const uint32_t units = (p.n - head) >> 3, tail0 = head + units * 8;
  if (threadIdx.x < head) to[threadIdx.x] = from[threadIdx.x];
  const uint4 *f4 = reinterpret_cast<const uint4 *>(from + head);
  uint4 *t4 = reinterpret_cast<uint4 *>(to + head);
  for (uint32_t u = threadIdx.x; u < units; u += 256) t4[u] = f4[u];
  if (threadIdx.x < p.n - tail0) to[tail0 + threadIdx.x] = from[tail0 + threadIdx.x];
}

// filtered length of every piece against what the host counted (SignalBatch's load phase): a difference means the two
// do not agree on the (30, 200) pA filter -- the chunk would not be the reference's chunk.  The flag is a word of
// pinned host memory, only ever raised to 1: a plain system-scope STORE (a read-modify-write atomic on host memory needs
// PCIe atomics end to end; where a slot lacks them it is dropped or undefined, and the check would never fire).
__global__ void CheckPieceLenKernel(const uint32_t *__restrict__ read_len, const uint32_t *__restrict__ expect, uint32_t n,
                                    uint32_t *host_flag) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && read_len[r] != expect[r]) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

hipError_t LaunchGatherPieces(smhip_ctx *ctx, const PieceSrc *src, uint32_t n, void *staged) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(GatherPiecesKernel, dim3(n), dim3(256), 0, ctx->stream, src, reinterpret_cast<int16_t *>(staged));
  return hipGetLastError();
}

hipError_t LaunchCheckPieceLen(smhip_ctx *ctx, const uint32_t *read_len, const uint32_t *expect, uint32_t n, uint32_t *host_flag) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(CheckPieceLenKernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, read_len, expect, n, host_flag);
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
