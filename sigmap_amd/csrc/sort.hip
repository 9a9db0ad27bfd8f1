// sort.hip -- anchor decode + per-(reference, strand) ordering, one workgroup per chunk.
//
// Replaces spatial_index.cc:371-417: each of the first min(hits, 5000) results of a seed
// becomes SignalAnchor{target_position, query_position, distance} in bucket
// [strand][reference] (position layout: sigmap_adaptor.h:7-17 / spatial_index.cc:48-52),
// anchors carried over from the previous chunk's chains come first (spatial_index.cc:303-322),
// and every bucket is std::sort'ed by (target_position, query_position, distance).
// The chaining loop then visits buckets in the order ref0+, ref0-, ref1+, ...
// (spatial_index.cc:421-423), so the device produces ONE array per chunk ordered by
// (group = ref*2 + strand, target, query, distance) plus the list of group boundaries.
//
// Method: 64-bit keys [group | target_position | index].  A chunk's anchors are cut into PARTS of ~384 by
// ranges of the dense key group * span + target (SortPartitionKernel: two passes over the hits, count and
// scatter); one WAVE sorts a part in registers with a bitonic network (SortPartWaveKernel); parts that a skewed
// target distribution makes larger than 1024 keys go to a workgroup (SortPartKernel: two-level LDS sort -
// counting sort into 1024 buckets, rank sort of every bucket - or a bitonic network in memory beyond 8192).
// Runs of equal (group, target) - short - are then ordered by (query, distance) in place.  Every comparison
// key is total, so the result is the unique sorted sequence the reference's std::sort produces.
#include "common.h"

namespace smhip {
namespace {

constexpr int kSortThreads = 1024;
constexpr uint32_t kMaxQueriesPerChunk = (kFeatStride - 5) / 1 + 1;  // step >= 1
constexpr uint32_t kBucketBits = 10;
constexpr uint32_t kBuckets = 1u << kBucketBits;  // buckets of the two-level sort (~9 keys each: four per wave pass)
constexpr uint32_t kMaxBucket = 512;  // largest bucket one wave rank-sorts (8 keys per lane)
constexpr uint32_t kPartTarget = 384;   // keys per part aimed at: one WAVE sorts a part of up to 512 keys in registers
constexpr uint32_t kMaxParts = 1024;    // parts per chunk (chunks of up to ~400 000 anchors; more: bigger parts)
constexpr uint32_t kPartBits = 10;      // work item = (chunk << kPartBits) | part
static_assert((1u << kPartBits) == kMaxParts, "part field of a work item");
constexpr uint32_t kPartTab = kMaxParts + 4;  // per chunk: part starts [P + 1], then [kMaxParts + 1] = P, [+ 2] = max target
// parts by size: <= 512 keys (8 per lane) and <= 1024 keys (16 per lane) are sorted by one wave each
// (SortPartWaveKernel), larger ones -- a skewed target distribution -- by a workgroup (SortPartKernel)
constexpr uint32_t kWaveCap0 = 512, kWaveCap1 = 1024;

#ifdef SMHIP_PHASE_PROF
__device__ unsigned long long g_phase_prof[32];  // (the PHASE_* macros of common.h)
#endif

template <typename KeyPtr>
__device__ __forceinline__ void BitonicSort(KeyPtr keys, uint32_t npad, uint32_t tid) {
  for (uint32_t k = 2; k <= npad; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < npad / 2; t += kSortThreads) {
        // t enumerates the pairs (i, i ^ j) with i < (i ^ j)
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint32_t l = i | j;
        const uint64_t a = keys[i], b = keys[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          keys[i] = b;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
  }
}


// Two-level LDS sort of n <= PT * kSortThreads keys that the threads hold in registers (mykeys[it] = key of
// element tid + it * kSortThreads, ~0 beyond n).  dense_lo / range: the dense keys
// group * span + target of these keys lie in [dense_lo, dense_lo + range).  On return lds_keys[0, n) is
// sorted.  Every thread of the workgroup calls it; hist / bstart / big_list / s_maxb / n_big are the
// caller's shared scratch.
template <uint32_t PT>
__device__ __forceinline__ void LdsSortKeys(uint64_t (&mykeys)[PT], uint32_t n, uint64_t *lds_keys, uint32_t *hist,
                                            uint32_t *bstart, uint32_t *big_list, uint32_t &s_maxb, uint32_t &n_big,
                                            uint64_t span, uint64_t dense_lo, uint64_t range, uint32_t ibits,
                                            uint32_t tid) {
  constexpr uint32_t kPerThread = PT;  // keys per thread: n <= PT * kSortThreads
  // counting sort of the keys into kBuckets buckets of the dense key (monotone in the key), then every
  // bucket - about ten keys - is rank-sorted (keys are unique: rank = number of smaller keys), without
  // workgroup barriers.  A key set with a heavy bucket (many equal targets) takes the bitonic path.
  for (uint32_t i = tid; i < kBuckets + 1; i += kSortThreads) hist[i] = 0;
  if (tid == 0) {
    n_big = 0;
    s_maxb = 0;
  }
  __syncthreads();
  const uint32_t rbits = 64u - (uint32_t)__clzll((long long)(range - 1ull) | 1ll);
  const uint32_t shift = rbits > kBucketBits ? rbits - kBucketBits : 0u;
  bool bucketed = false;
  uint32_t myb[kPerThread];
#pragma unroll
  for (uint32_t it = 0; it < kPerThread; ++it) {
    if (tid + it * kSortThreads >= n) continue;
    const uint64_t key = mykeys[it];
    const uint64_t dense = (key >> (32 + ibits)) * span + ((key >> ibits) & 0xFFFFFFFFull) - dense_lo;
    myb[it] = (uint32_t)(dense >> shift);
    atomicAdd(&hist[myb[it]], 1u);
  }
  __syncthreads();
  // exclusive scan of the bucket sizes by one wave (16 buckets per lane, shuffles): two workgroup
  // barriers instead of the twenty-two of a Hillis-Steele scan over 1024 threads
  if (tid < 64) {
    constexpr uint32_t kPer = kBuckets / 64;
    uint32_t h[kPer], sum = 0, mx = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      h[k] = hist[kPer * tid + k];
      sum += h[k];
      mx = max(mx, h[k]);
    }
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
      if ((int)tid >= d) incl += o;
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    }
    uint32_t run = incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      bstart[kPer * tid + k] = run;
      run += h[k];
      hist[kPer * tid + k] = 0;  // becomes the insertion cursor
    }
    if (tid == 0) {
      s_maxb = mx;
      bstart[kBuckets] = n;
    }
  }
  __syncthreads();
  bucketed = s_maxb <= kMaxBucket;
  if (bucketed) {
#pragma unroll
    for (uint32_t it = 0; it < kPerThread; ++it)
      if (tid + it * kSortThreads < n) lds_keys[bstart[myb[it]] + atomicAdd(&hist[myb[it]], 1u)] = mykeys[it];
    __syncthreads();
      const uint32_t lane = tid & 63u, wave = tid >> 6;
    // Four buckets per pass of a wave, sixteen lanes each: a rank sort reads every key of a bucket
    // once as an LDS broadcast, and a 64-lane 8-byte read occupies the LDS for four clocks however
    // many lanes use it -- with one small bucket per wave the phase was bound by exactly that.
    const uint32_t grp16 = lane >> 4, gl = lane & 15u;
    for (uint32_t b4 = wave * 4; b4 < kBuckets; b4 += (kSortThreads / 64) * 4) {
      const uint32_t b = b4 + grp16;
      const uint32_t b0 = bstart[b], m = bstart[b + 1] - b0;
      const bool small = m <= 16;
      const int msm = (int)(small ? m : 0u);  // (uniform per 16 lanes: four scalar reads, no LDS shuffle)
      const uint32_t mmax = (uint32_t)max(max(__builtin_amdgcn_readlane(msm, 0), __builtin_amdgcn_readlane(msm, 16)),
                                          max(__builtin_amdgcn_readlane(msm, 32), __builtin_amdgcn_readlane(msm, 48)));
      if (mmax >= 2) {
        const uint32_t ms = small ? m : 0u;  // groups with a larger bucket sit this pass out
        const uint64_t mine = gl < ms ? lds_keys[b0 + gl] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t x = 0; x < mmax; x += 4) {
          uint64_t o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = lds_keys[b0 + min(x + k, ms ? ms - 1 : 0u)];
#pragma unroll
          for (int k = 0; k < 4; ++k) rank += (x + k < ms && o[k] < mine) ? 1u : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();  // all reads of the buckets precede the permuting writes
        if (gl < ms) lds_keys[b0 + rank] = mine;
      }
      // buckets of 17..64 keys: one at a time by the whole wave, one key per lane
      const uint64_t medium = __ballot(m > 16 && m <= 64);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (!((medium >> (16 * g)) & 1ull)) continue;
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, 16 * g);
        const uint32_t cm = (uint32_t)__builtin_amdgcn_readlane((int)m, 16 * g);
        const uint64_t mine = lane < cm ? lds_keys[c0 + lane] : ~0ull;
        uint32_t rank = 0;
        uint32_t x = 0;
        for (; x + 8 <= cm; x += 8) {  // LDS broadcasts, eight in flight
          uint64_t o[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = lds_keys[c0 + x + k];
#pragma unroll
          for (int k = 0; k < 8; ++k) rank += o[k] < mine ? 1u : 0u;
        }
        for (; x < cm; ++x) rank += lds_keys[c0 + x] < mine ? 1u : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (lane < cm) lds_keys[c0 + rank] = mine;
      }
      // (a heavy bucket -- the true locus of a read: hundreds of keys -- would keep one wave busy
      // for tens of microseconds while the other fifteen wait at the barrier: left to the whole
      // workgroup below)
    }
    if (tid < kBuckets && bstart[tid + 1] - bstart[tid] > 64) big_list[atomicAdd(&n_big, 1u)] = tid;
    __syncthreads();
      static_assert(kMaxBucket <= kSortThreads, "one key per thread in a heavy bucket");
    for (uint32_t bi = 0; bi < n_big; ++bi) {
      const uint32_t bb = big_list[bi];
      const uint32_t b0 = bstart[bb], m = bstart[bb + 1] - b0;
      const uint64_t mine = tid < m ? lds_keys[b0 + tid] : ~0ull;
      uint32_t rank = 0;
      if (tid < ((m + 63u) & ~63u)) {  // the waves that hold keys
        uint32_t x = 0;
        for (; x + 8 <= m; x += 8) {  // LDS broadcasts, eight in flight
          uint64_t o[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = lds_keys[b0 + x + k];
#pragma unroll
          for (int k = 0; k < 8; ++k) rank += o[k] < mine ? 1u : 0u;
        }
        for (; x < m; ++x) rank += lds_keys[b0 + x] < mine ? 1u : 0u;
      }
      __syncthreads();  // all reads of the bucket precede the permuting writes
      if (tid < m) lds_keys[b0 + rank] = mine;
      __syncthreads();
    }
    __syncthreads();
    }
  if (!bucketed) {
    uint32_t npad = 1;
    while (npad < n) npad <<= 1;
#pragma unroll
    for (uint32_t it = 0; it < kPerThread; ++it)
      if (tid + it * kSortThreads < n) lds_keys[tid + it * kSortThreads] = mykeys[it];
    for (uint32_t i = n + tid; i < npad; i += kSortThreads) lds_keys[i] = ~0ull;
    __syncthreads();
    BitonicSort(lds_keys, npad, tid);
  }
}

// The sorted keys `keys[0, n)` (LDS or global) -> the sorted anchor arrays at s_*[0, n) (gathering the
// query position and distance of the unsorted anchor the key's low bits name), then runs of equal
// (group, target) are ordered by (query, distance): insertion sort by the thread that owns the run's
// first element.
// (u_qpos / u_dist are read at index from * pay_stride: 1 for columns, 4 for the {key, qpos, dist} records of a part)
template <typename KeyPtr>
__device__ __forceinline__ void EmitSorted(KeyPtr keys, uint32_t n, uint32_t ibits, const uint32_t *u_qpos,
                                           const float *u_dist, uint32_t pay_stride, uint32_t *s_tpos,
                                           uint32_t *s_qpos, uint32_t *s_grp, float *s_dist, uint32_t tid) {
  const uint64_t idx_mask = (1ull << ibits) - 1;
  constexpr int kSc = 4;  // elements per pass: one wait per gather level
  for (uint32_t i0 = tid; i0 < n; i0 += kSc * kSortThreads) {
    uint64_t key[kSc];
    uint32_t gq[kSc];
    float gd[kSc];
#pragma unroll
    for (int u = 0; u < kSc; ++u) key[u] = keys[min(i0 + u * kSortThreads, n - 1)];
#pragma unroll
    for (int u = 0; u < kSc; ++u) {
      const uint32_t from = (uint32_t)(key[u] & idx_mask);
      gq[u] = u_qpos[(size_t)from * pay_stride];
      gd[u] = u_dist[(size_t)from * pay_stride];
    }
#pragma unroll
    for (int u = 0; u < kSc; ++u) {
      const uint32_t i = i0 + u * kSortThreads;
      if (i >= n) continue;
      s_tpos[i] = (uint32_t)(key[u] >> ibits);
      s_grp[i] = (uint32_t)(key[u] >> (32 + ibits));
      s_qpos[i] = gq[u];
      s_dist[i] = gd[u];
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kSortThreads) {
    const uint64_t gt = keys[i] >> ibits;
    const bool run_start = i == 0 || (keys[i - 1] >> ibits) != gt;
    if (!run_start) continue;
    uint32_t e = i + 1;
    while (e < n && (keys[e] >> ibits) == gt) ++e;
    for (uint32_t x = i + 1; x < e; ++x) {
      const uint32_t kq = s_qpos[x];
      const float kd = s_dist[x];
      uint32_t y = x;
      while (y > i && (s_qpos[y - 1] > kq || (s_qpos[y - 1] == kq && s_dist[y - 1] > kd))) {
        s_qpos[y] = s_qpos[y - 1];
        s_dist[y] = s_dist[y - 1];
        --y;
      }
      s_qpos[y] = kq;
      s_dist[y] = kd;
    }
  }
}

struct BigSortWork {   // global scratch of the large-chunk path (LaunchSort)
  uint32_t *part_tab;  // [n_chunks][kPartTab]
  uint4 *recs[2];      // parts of the two wave classes: {A lo, A hi | m << 16, dlo lo, dlo hi}, A = first element of the
                       // part (m elements) in the sorted anchor columns, dlo = smallest dense key
                       // group * span + target the part can hold
  uint64_t *rin[2];    // ... and where the part's m records {key, qpos, dist} lie in the record scratch (a.keys)
  uint32_t *items;     // parts of the workgroup class: (chunk << kPartBits) | part
  uint32_t *redo;      // chunks the whole-line route gave up on (a part beyond its slice): [n_chunks]
  uint32_t *counters;  // [0..2] items per class, [3] chunks in redo
};

// The record scratch (a.keys as 16-byte records) of chunk c starts at RecOff(anchor_off[c], c).  The whole-line
// route gives every part a fixed slice of kSlice records.  Parts aim at kPartTarget = 384 anchors and a read's true
// locus adds a hundred or two to ONE part of its chunk: with slices of 512 every seventh chunk of a 250 Mb reference
// outgrew one and took the exact route; 640 leaves that to skewed target distributions (a part of 513 .. 640 keys is
// sorted by the wave kernel's 64-bit network: one per chunk).  A chunk of n anchors may need
// (n / kPartTarget + 3) * kSlice = 5 n / 3 + 1920 records: consecutive chunks' regions differ by at least
// 5 (n / 3 - 2 / 3) + 5 + 2048.  (LaunchSort sizes a.keys for RecOff(total anchors, n_chunks).)
constexpr uint32_t kSlice = 640;
__host__ __device__ __forceinline__ uint64_t RecOff(uint64_t anchor_off, uint64_t c) {
  return 5ull * (anchor_off / 3ull + c) + 2048ull * c;
}

// ---- decode: anchor i of a chunk = carried anchor i, or hit (i - ncar - qoff[k]) of the chunk's query k
struct DecodeCtx {
  const uint32_t *qoff;  // LDS: exclusive offsets of the queries' anchor runs, [nq + 1]
  uint32_t q0, nq, ncar, n, qso;
  uint64_t car0;
  const Hit *seg;  // the hit-slot segment that holds all of the chunk's queries (the rule), or null
  __device__ DecodeCtx(const SortArgs &a, const uint32_t *qoff_, uint32_t q0_, uint32_t nq_, uint32_t ncar_, uint32_t n_,
                       uint32_t c)
      : qoff(qoff_), q0(q0_), nq(nq_), ncar(ncar_), n(n_), qso(a.qstart ? a.qstart[c] : 0u),
        car0(a.carried_off ? a.carried_off[c] : 0), seg(nullptr) {
    if (nq && (q0 >> a.region.qshift) == ((q0 + nq - 1) >> a.region.qshift)) seg = a.region.seg[q0 >> a.region.qshift];
  }
};
struct Decoded {
  uint32_t grp, tpos, qpos;
  float dist;
};
// Exclusive offsets of the queries' anchor runs into LDS: qoff[k] = sum over j < k of min(hits of query q0 + j,
// num_nearest_points), qoff[nq] = the total.  Every thread takes ceil(nq / threads) consecutive queries; scan
// inside the waves by shuffles, the wave totals through `wtot`.  Ends with a barrier.
template <uint32_t NT>
__device__ __forceinline__ void ScanQueryOffsets(const SortArgs &a, uint32_t q0, uint32_t nq, uint32_t *qoff,
                                                 uint32_t *wtot, uint32_t tid) {
  const uint32_t per = (nq + NT - 1) / NT;
  const uint32_t qb = min(nq, tid * per), qe = min(nq, qb + per);
  uint32_t sum = 0;
  for (uint32_t k = qb; k < qe; ++k) sum += min(a.q_count[q0 + k], a.nnp);
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  const uint32_t incl = (uint32_t)WaveScanAdd((int)sum);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t k = 0; k < wave; ++k) before += wtot[k];
  uint32_t run = before + incl - sum;
  for (uint32_t k = qb; k < qe; ++k) {
    qoff[k] = run;
    run += min(a.q_count[q0 + k], a.nnp);
  }
  if (tid == NT - 1) qoff[nq] = run;
  __syncthreads();
}


// ---- every chunk is cut into PARTS by ranges of the dense key group * span + target, span = *a.span_bound
// (largest target of the index or of a carried anchor, + 1: known before the decode, so that an anchor's part is
// known the moment it is decoded); the key's index field becomes the position INSIDE the part, and the payload
// travels with the key in one 16-byte record {key, query position, distance}: a part's sorter reads its own slice
// once.  Work items for the part sorters: one record per part, by size class.  One workgroup per chunk at a time,
// two per CU.  Two routes:
//
//  * kLines (the rule): ONE pass over the hits.  A part gets a fixed slice of kSlice records in the chunk's region
//    of the record scratch; the workgroup takes the hits in TILES of up to 2048 -- a wave step is 64 hits of each of
//    two queries --, bins the tile's records by part in LDS (count, scan, place) and copies the binned tile out:
//    the records of one part are consecutive in the tile AND in memory, so the stores of neighbouring lanes fall
//    into the same 128-byte lines.  The scattered 16-byte store per anchor of the exact route costs a 32-byte
//    write request each (profiles/r3_calibration.json), and with its separate counting pass the kernel moved 64
//    bytes per anchor at the rate of fabric REQUESTS (0.9e8 reads + 3.6e8 writes per 3.6e8 anchors); this route
//    reads the hits once and writes whole runs.  A chunk in which a part outgrows its slice (a skewed target
//    distribution: repeats) is listed (w.redo) and mapped by a launch of the exact route.
//  * exact (kLines == false; the fallback, for the chunks of redo_list; knob sort_lines = 0: for every chunk): count
//    the parts' sizes in a pass over the hits of its own, then decode again and write every record into its part's
//    exact slice, one scattered store per anchor.
constexpr uint32_t kBigThreads = 1024;  // (256 / 512: 4.2 / 4.4 ms against 3.6 ms at 32 Mb)
constexpr uint32_t kGroupsLds = 1024;  // (reference, strand) groups counted in LDS by the partition kernel; more: GroupStartKernel
// records per tile of the whole-line route = RPT per thread: 2 (2048: two workgroups per CU), or -- large chunks, whose
// hundreds of parts get only a few records each per tile -- 4 (4096: runs twice as long, one workgroup per CU)
constexpr size_t TileLdsBytes(int rpt) { return (size_t)rpt * kBigThreads * (sizeof(uint4) + sizeof(uint32_t)); }  // records + destinations
template <bool kLines, int RPT>
__global__ __launch_bounds__(kBigThreads, RPT == 2 ? 8 : 4) void SortPartitionKernel(SortArgs a, uint32_t n_chunks, uint32_t gbits,
                                                                       uint32_t ibits, BigSortWork w,
                                                                       const uint32_t *__restrict__ redo_list) {
  __shared__ uint32_t qoff[kMaxQueriesPerChunk + 2];
  __shared__ uint32_t part[kBigThreads / 64];
  __shared__ uint32_t pcount[kMaxParts + 64], pstart[kMaxParts + 1];
  __shared__ uint32_t cls_n[4], cls_base[4];
  // anchors per group: whole parts are added part by part, the anchors of the few parts that straddle a group
  // boundary (strad: one bit per part) one by one; the group boundaries of the sorted chunk are the prefix sums
  __shared__ uint32_t gcnt[kGroupsLds];
  __shared__ uint32_t strad[kMaxParts / 32];
  // whole-line route: per-tile counts / starts / running positions of the parts, the query hand-out, the verdict
  __shared__ uint32_t tcount[kLines ? kMaxParts + 64 : 1], tstart[kLines ? kMaxParts + 64 : 1], tgpos[kLines ? kMaxParts + 64 : 1];
  __shared__ uint32_t s_overflow;
  constexpr uint32_t kTile = (uint32_t)RPT * kBigThreads;
  extern __shared__ __align__(16) unsigned char tile_mem[];  // kLines: uint4 tile_rec[kTile], uint32_t tile_dst[kTile]
  uint4 *tile_rec = reinterpret_cast<uint4 *>(tile_mem);
  uint32_t *tile_dst = reinterpret_cast<uint32_t *>(tile_mem + (size_t)kTile * sizeof(uint4));
  const bool fold_groups = (1u << gbits) <= kGroupsLds;
  const uint32_t tid = threadIdx.x;
  const uint64_t span = (uint64_t)*a.span_bound + 1ull;
  const uint64_t range = span << gbits;  // dense keys are < range
  const uint32_t rbits = 64u - (uint32_t)__clzll((long long)(range - 1ull) | 1ll);
  const uint64_t idx_mask = (1ull << ibits) - 1ull;
  PHASE_DECL();
  // (redo_list: the exact route for the chunks the whole-line launch gave up on, w.counters[3] of them)
  const uint32_t n_todo = redo_list ? w.counters[3] : n_chunks;
  for (uint32_t ci = blockIdx.x; ci < n_todo; ci += gridDim.x) {
    const uint32_t c = redo_list ? redo_list[ci] : ci;
    PHASE_INIT();
    const uint64_t base = a.anchor_off[c];
    const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
    const uint32_t q0 = a.q_off[c], nq = a.q_off[c + 1] - q0;
    const uint32_t ncar = a.carried_off ? a.carried_off[c + 1] - a.carried_off[c] : 0;
    // the key cannot index this many anchors / queries / chunks: flagged, no anchors
    const bool cap = n > (1u << ibits) || nq > kMaxQueriesPerChunk || c >= (1u << (32 - kPartBits));
    if (tid == 0) {
      a.results[c].n_anchors = cap ? 0u : n;
      a.n_groups[c] = 0;
      if (cap) a.results[c].status |= SMHIP_ST_SORT_CAP;
    }
    if (n == 0 || cap) continue;  // (uniform)
    __syncthreads();  // the previous chunk is done with the shared arrays
    for (uint32_t k = tid; k < kMaxParts + 64; k += kBigThreads) {
      pcount[k] = 0;
      if (kLines) tcount[k] = 0;
    }
    for (uint32_t k = tid; k < kGroupsLds; k += kBigThreads) gcnt[k] = 0;
    if (tid < kMaxParts / 32) strad[tid] = 0;
    if (tid == 0) s_overflow = 0;
    ScanQueryOffsets<kBigThreads>(a, q0, nq, qoff, part, tid);
    const DecodeCtx dc(a, qoff, q0, nq, ncar, n, c);
    PHASE_MARK(8);
    // as many parts as n / kPartTarget of them are non-empty: dense keys only reach range / 2^rbits of the scale
    const uint64_t want = ((uint64_t)(n + kPartTarget - 1) / kPartTarget << rbits) / range + 1;
    const uint32_t P = (uint32_t)min<uint64_t>(want, kMaxParts);
    auto part_of = [&](const Decoded &dv) {  // floor(dense * P / 2^rbits), dense < range <= 2^rbits: monotone in the key
      const uint64_t dense = (uint64_t)dv.grp * span + min((uint64_t)dv.tpos, span - 1ull);
      return min((uint32_t)((dense * P) >> rbits), P - 1u);  // (the clamps only ever act on invalid carried anchors)
    };
    // parts whose key range holds a group boundary: their anchors are counted into the groups one by one
    constexpr uint32_t kPartsPerThread = kMaxParts / kBigThreads;
    if (fold_groups) {
#pragma unroll
      for (uint32_t j = 0; j < kPartsPerThread; ++j) {
        const uint32_t pi = tid + j * kBigThreads;
        if (pi < P) {
          const uint64_t lo = (((uint64_t)pi << rbits) + P - 1) / P;  // dense keys of the part: [lo, hi)
          const uint64_t hi = min<uint64_t>(((((uint64_t)pi + 1) << rbits) + P - 1) / P, range);
          if (lo < hi && (uint32_t)(lo / span) != (uint32_t)((hi - 1) / span)) atomicOr(&strad[pi >> 5], 1u << (pi & 31u));
        }
      }
    }
    __syncthreads();
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // one hit of query k (cnt anchors, slot address and query position wave-uniform): fast slots or the geometric
    // overflow segments (seed_lookup.hip AllocOverflowSegment; a missing segment reads as zeros)
    auto load_hit = [&](uint32_t qi, const Hit *slots, uint32_t h) {
      if (h < a.fast_cap) return *reinterpret_cast<const uint4 *>(slots + h);
      const uint32_t kk = h - a.fast_cap;
      const uint32_t level = OvfLevel(kk, a.fast_cap);
      const uint32_t first = a.fast_cap * ((1u << level) - 1u);
      const uint32_t seg = level < (uint32_t)kOvfLevels ? a.ovf_tab[(size_t)qi * kOvfLevels + level] : 0xFFFFFFFFu;
      return seg != 0xFFFFFFFFu ? *reinterpret_cast<const uint4 *>(a.ovf_pool + (size_t)seg + (kk - first)) : make_uint4(0, 0, 0, 0);
    };
    auto decode_hit = [&](const uint4 &raw, uint32_t qpos) {  // Hit {uint64 position, dist, slot}
      const uint64_t position = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
      return Decoded{(uint32_t)((position >> 33) << 1) | (uint32_t)(position & 1), (uint32_t)(position >> 1), qpos,
                     __uint_as_float(raw.z)};
    };
    auto slots_of = [&](uint32_t qi) {
      return dc.seg ? dc.seg + (size_t)(qi & ((1u << a.region.qshift) - 1u)) * a.fast_cap : a.region.At(qi, 0);
    };
    // The exact route visits the anchors QUERY BY QUERY, a wave per query, its lanes over the query's hits (a large
    // chunk has ~100 hits per query): no search for the query of an anchor, slot addresses and query position
    // are wave-uniform, the hit records of a query are read by consecutive lanes.  fn(decoded anchor).
    auto for_each_anchor = [&](auto &&fn) {
      if constexpr (kLines) return;  // (the whole-line launch leaves the exact route to a launch of its own)
      for (uint32_t i = tid; i < ncar; i += kBigThreads) {  // smhip_carried_anchor {ref_strand, target, query, distance}
        const uint4 raw = *reinterpret_cast<const uint4 *>(a.carried + dc.car0 + i);
        fn(Decoded{raw.x, raw.y, raw.z, __uint_as_float(raw.w)});
      }
      for (uint32_t k = wave; k < nq; k += kBigThreads / 64) {
        const uint32_t cnt = qoff[k + 1] - qoff[k];
        const uint32_t qi = q0 + k;
        const Hit *slots = slots_of(qi);
        const uint32_t qpos = (k + 1) * a.step + dc.qso;  // position + query_start_offset
        for (uint32_t h0 = 0; h0 < cnt; h0 += 128) {  // two hits per lane in flight
          uint4 raw[2];
          bool live[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t h = h0 + 64 * u + lane;
            live[u] = h < cnt;
            raw[u] = make_uint4(0, 0, 0, 0);
            if (live[u]) raw[u] = load_hit(qi, slots, h);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (live[u]) fn(decode_hit(raw[u], qpos));
        }
      }
    };
    const uint64_t rec0 = RecOff(base, c);
    uint4 *recs_out = reinterpret_cast<uint4 *>(a.keys) + rec0;
    if constexpr (kLines) {
      // ---- the whole-line pass: the chunk's anchors in their decode order (carried anchors, then the hits query by
      // query) are cut into tiles of 2048, two per thread (anchor T * 2048 + 1024 u + tid: neighbouring lanes read
      // neighbouring hits of a query).  The query of an anchor is found in the LDS offsets (the previous tile's
      // answer bounds the search from below).  A tile's records are fetched while the tile before it is binned.
      const uint32_t n_tiles = (n + kTile - 1u) / kTile;
      const uint32_t scan_per = (P + 63u) / 64u;  // parts per lane of the tile scan (one wave)
      uint32_t k_hint[RPT] = {};
      uint4 raw_nx[RPT];
      uint32_t qpos_nx[RPT];  // query position of a hit; ~0: a carried anchor (the record holds its own)
      auto fetch = [&](uint32_t T) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const uint32_t i = T * kTile + (uint32_t)u * kBigThreads + tid;
          raw_nx[u] = make_uint4(0, 0, 0, 0);
          qpos_nx[u] = 0xFFFFFFFFu;
          if (i >= n) continue;
          if (i < ncar) {  // smhip_carried_anchor {ref_strand, target, query, distance}
            raw_nx[u] = *reinterpret_cast<const uint4 *>(a.carried + dc.car0 + i);
            continue;
          }
          const uint32_t j = i - ncar;
          uint32_t lo = k_hint[u], hi = nq;  // the query k with qoff[k] <= j < qoff[k + 1]
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (qoff[mid] <= j) lo = mid; else hi = mid;
          }
          k_hint[u] = lo;
          const uint32_t qi = q0 + lo;
          raw_nx[u] = load_hit(qi, slots_of(qi), j - qoff[lo]);
          qpos_nx[u] = (lo + 1u) * a.step + dc.qso;  // position + query_start_offset
        }
      };
      fetch(0);
      for (uint32_t T = 0; T < n_tiles; ++T) {
        Decoded dv[RPT];
        bool live[RPT];
        uint32_t pb[RPT] = {}, lpos[RPT] = {};
        // count: position of the record among the tile's records of its part
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          live[u] = T * kTile + (uint32_t)u * kBigThreads + tid < n;
          if (!live[u]) continue;
          dv[u] = qpos_nx[u] == 0xFFFFFFFFu ? Decoded{raw_nx[u].x, raw_nx[u].y, raw_nx[u].z, __uint_as_float(raw_nx[u].w)}
                                            : decode_hit(raw_nx[u], qpos_nx[u]);
          pb[u] = part_of(dv[u]);
          if (dv[u].grp >> gbits) {  // a (reference, strand) group the batch's n_refs does not have: an invalid carried
            live[u] = false;         // anchor (or an index that does not match n_refs).  Never placed: its part may lie
            s_overflow = 2u;         // beyond the chunk's slices; the chunk is flagged below
            continue;
          }
          lpos[u] = atomicAdd(&tcount[pb[u]], 1u);
        }
        PHASE_MARK(12);  // decode + count (waits for the tile's loads)
        if (T + 1 < n_tiles) fetch(T + 1);  // (in flight while this tile is binned and copied out)
        __syncthreads();
        PHASE_MARK(13);  // query search + issue of the next tile's loads + barrier
        if (tid < 64) {  // exclusive scan of the tile's part counts by one wave; the parts' running positions move on
          uint32_t tot = 0;
          for (uint32_t k = 0; k < scan_per; ++k) tot += tcount[scan_per * tid + k];
          const uint32_t incl = (uint32_t)WaveScanAdd((int)tot);
          uint32_t run = incl - tot;
          for (uint32_t k = 0; k < scan_per; ++k) {
            const uint32_t b = scan_per * tid + k;  // (< 64 scan_per <= kMaxParts + 63: the tables have that room)
            const uint32_t h = tcount[b];
            tstart[b] = run;
            run += h;
            const uint32_t g = pcount[b];
            tgpos[b] = g;
            pcount[b] = g + h;
            tcount[b] = 0;
          }
        }
        __syncthreads();
        PHASE_MARK(14);  // scan of the tile's part counts (one wave) + barrier
        // place: the tile's records sorted by part in LDS, each with its destination in the part's slice
#pragma unroll
        for (int u = 0; u < RPT; ++u)
          if (live[u]) {
            const uint32_t b = pb[u];
            const uint32_t pos = tgpos[b] + lpos[u];  // position inside the part = the key's index field
            const uint32_t slot = tstart[b] + lpos[u];
            if ((strad[b >> 5] >> (b & 31u)) & 1u) atomicAdd(&gcnt[min(dv[u].grp, kGroupsLds - 1u)], 1u);
            const uint64_t key = (((uint64_t)dv[u].grp << (32 + ibits)) | ((uint64_t)dv[u].tpos << ibits)) | ((uint64_t)pos & idx_mask);
            tile_rec[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), dv[u].qpos, __float_as_uint(dv[u].dist));
            tile_dst[slot] = pos < kSlice ? b * kSlice + pos : 0xFFFFFFFFu;  // (a full slice: counted, not written)
          }
        const uint32_t tile_n = min(kTile, n - T * kTile);
        __syncthreads();
        // copy out: consecutive threads take consecutive records of the binned tile -- runs of one part, i.e.
        // consecutive addresses
        PHASE_MARK(15);  // placement in LDS + barrier
        for (uint32_t sl = tid; sl < tile_n; sl += kBigThreads) {
          const uint32_t d = tile_dst[sl];
          if (d != 0xFFFFFFFFu) recs_out[d] = tile_rec[sl];
        }
        PHASE_MARK(7);  // copy out
      }
      for (uint32_t k = tid; k < kMaxParts; k += kBigThreads)
        if (pcount[k] > kSlice) atomicOr(&s_overflow, 1u);
      __syncthreads();
      if (s_overflow == 1u) {  // (uniform) a part outgrew its slice (skewed targets): the chunk goes to the exact route's launch
        if (tid == 0) w.redo[atomicAdd(&w.counters[3], 1u)] = c;
        continue;
      }
    } else {
      // ---- pass 1 of the exact route: part sizes
      for_each_anchor([&](const Decoded &dv) {
        if (dv.grp >> gbits) s_overflow = 2u;
        atomicAdd(&pcount[part_of(dv)], 1u);
      });
      __syncthreads();
    }
    if (s_overflow & 2u) {  // (uniform) an anchor outside the batch's (reference, strand) groups: flagged, no anchors
      if (tid == 0) {
        a.results[c].n_anchors = 0u;
        a.results[c].status |= SMHIP_ST_SORT_CAP;
      }
      continue;
    }
    PHASE_MARK(9);
    if (tid < 64) {  // exclusive scan of the part sizes by one wave
      constexpr uint32_t kPer = kMaxParts / 64;
      uint32_t h[kPer], tot = 0;
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        h[k] = pcount[kPer * tid + k];
        tot += h[k];
      }
      uint32_t incl = tot;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)tid >= d) incl += o;
      }
      uint32_t run = incl - tot;
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        pstart[kPer * tid + k] = run;
        run += h[k];
        pcount[kPer * tid + k] = 0;  // becomes the insertion cursor of the exact route
      }
      if (tid == 63) pstart[kMaxParts] = run;  // == n
    }
    if (tid < 3) cls_n[tid] = 0;
    __syncthreads();
    // ---- work items, one list per size class.  Narrow parts: dense key minus the part's lower bound, shifted
    // over a 9-bit index, fits 32 bits
    uint32_t *tab = w.part_tab + (size_t)c * kPartTab;
    for (uint32_t k = tid; k <= P; k += kBigThreads) tab[k] = pstart[k];
    const bool narrow = ((1ull << rbits) / P) + 2ull <= (1ull << 23);
    constexpr bool slices = kLines;  // the records lie in the parts' fixed slices (else at the exact offsets)
    uint32_t my_cls[kPartsPerThread], my_pos[kPartsPerThread], my_m[kPartsPerThread];
#pragma unroll
    for (uint32_t j = 0; j < kPartsPerThread; ++j) {
      const uint32_t pi = tid + j * kBigThreads;
      my_cls[j] = 3;
      my_pos[j] = my_m[j] = 0;
      if (pi < P) {
        my_m[j] = pstart[pi + 1] - pstart[pi];
        if (my_m[j]) {
          my_cls[j] = (my_m[j] <= kWaveCap0 && narrow) ? 0u : my_m[j] <= kWaveCap1 ? 1u : 2u;
          my_pos[j] = atomicAdd(&cls_n[my_cls[j]], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 3) cls_base[tid] = cls_n[tid] ? atomicAdd(&w.counters[tid], cls_n[tid]) : 0u;
    if (tid == 0) {
      tab[kMaxParts + 1] = P;
      tab[kMaxParts + 2] = (uint32_t)(span - 1ull);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kPartsPerThread; ++j) {
      const uint32_t pi = tid + j * kBigThreads;
      if (my_cls[j] == 2) {
        w.items[cls_base[2] + my_pos[j]] = (c << kPartBits) | pi;  // (only ever on the exact route: m > 1024 > kSlice)
      } else if (my_cls[j] < 2) {
        const uint64_t A = base + pstart[pi];
        const uint64_t dlo = (((uint64_t)pi << rbits) + P - 1) / P;  // part_of(.) == pi  <=>  dlo <= dense < the next dlo
        const uint32_t at = cls_base[my_cls[j]] + my_pos[j];
        (my_cls[j] ? w.recs[1] : w.recs[0])[at] =
            make_uint4((uint32_t)A, (uint32_t)(A >> 32) | (my_m[j] << 16), (uint32_t)dlo, (uint32_t)(dlo >> 32));
        (my_cls[j] ? w.rin[1] : w.rin[0])[at] = rec0 + (slices ? (uint64_t)pi * kSlice : (uint64_t)pstart[pi]);
      }
    }
    if (fold_groups) {  // whole parts inside one group: their sizes
#pragma unroll
      for (uint32_t j = 0; j < kPartsPerThread; ++j) {
        const uint32_t pi = tid + j * kBigThreads;
        if (pi < P && my_m[j] && !((strad[pi >> 5] >> (pi & 31u)) & 1u)) {
          const uint64_t lo = (((uint64_t)pi << rbits) + P - 1) / P;
          atomicAdd(&gcnt[min((uint32_t)(lo / span), kGroupsLds - 1u)], my_m[j]);
        }
      }
    }
    __syncthreads();
    PHASE_MARK(10);
    if constexpr (!kLines) {
      // ---- pass 2 of the exact route: decode again, into the parts' exact slices of the chunk's records (16 bytes,
      // ONE scattered store per anchor: three separate columns cost 4.0 ms instead of 3.0 at 32 Mb - the pass is
      // bound by write requests, not bytes)
      for_each_anchor([&](const Decoded &dv) {
        const uint32_t b = part_of(dv);
        const uint32_t pos = atomicAdd(&pcount[b], 1u);
        const uint32_t dst = pstart[b] + pos;
        if ((strad[b >> 5] >> (b & 31u)) & 1u) atomicAdd(&gcnt[min(dv.grp, kGroupsLds - 1u)], 1u);
        const uint64_t key = (((uint64_t)dv.grp << (32 + ibits)) | ((uint64_t)dv.tpos << ibits)) | ((uint64_t)pos & idx_mask);
        recs_out[dst] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), dv.qpos, __float_as_uint(dv.dist));
      });
    }
    if (fold_groups) {  // group boundaries of the sorted chunk: exclusive prefix sums of the non-empty groups' sizes
      __syncthreads();
      if (tid < 64) {
        constexpr uint32_t kPer = kGroupsLds / 64;
        uint32_t h[kPer], tot = 0, ne = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
          h[k] = gcnt[kPer * tid + k];
          tot += h[k];
          ne += h[k] ? 1u : 0u;
        }
        uint32_t incl = tot, incl_ne = ne;
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64), o2 = (uint32_t)__shfl_up((int)incl_ne, d, 64);
          if ((int)tid >= d) {
            incl += o;
            incl_ne += o2;
          }
        }
        uint32_t run = incl - tot, rank = incl_ne - ne;
        uint32_t *gs = a.group_start + base;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k)
          if (h[k]) {
            gs[rank++] = run;
            run += h[k];
          }
        if (tid == 63) a.n_groups[c] = incl_ne;
      }
    }
    PHASE_MARK(11);
  }
  PHASE_FLUSH();
}

// Largest target position of the resident index (once per index: LaunchSort's span bound), and of a batch's
// carried anchors on top of it (every batch)
__global__ void IndexMaxTargetKernel(const uint32_t *__restrict__ leaf_tab, const float4 *__restrict__ leaf_pts,
                                     uint64_t n_leaves, uint32_t *__restrict__ out) {
  uint32_t mx = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_leaves; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t tab = leaf_tab[k];
    const uint32_t n = tab >> 27;
    const uint64_t boff = (uint64_t)(tab & 0x07FFFFFFu) << 3;  // first 32-byte slot of the leaf's block
    for (uint32_t t = 0; t < n; ++t) {
      const float4 v = leaf_pts[2 * (boff + t) + 1];  // {v4, v5, position lo, position hi}
      const uint64_t pos = (uint64_t)__float_as_uint(v.z) | ((uint64_t)__float_as_uint(v.w) << 32);
      mx = max(mx, (uint32_t)(pos >> 1));
    }
  }
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  if ((threadIdx.x & 63u) == 0 && mx) atomicMax(out, mx);
}
__global__ void CarriedMaxTargetKernel(const smhip_carried_anchor *__restrict__ carried, uint64_t n_carried,
                                       const uint32_t *__restrict__ index_max, uint32_t *__restrict__ out) {
  uint32_t mx = *index_max;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_carried; k += (uint64_t)gridDim.x * blockDim.x)
    mx = max(mx, carried[k].target_position);
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  if ((threadIdx.x & 63u) == 0) atomicMax(out, mx);
}

// One part of a large chunk per workgroup pass: keys gk2[b0, b0 + m) -> sorted outputs [b0, b0 + m).
constexpr uint32_t kPartPerThread = 8;                              // keys per thread of a part ...
constexpr uint32_t kPartLdsKeys = kPartPerThread * kSortThreads;    // ... 8192 keys = 64 KB of LDS: two workgroups per CU
__global__ __launch_bounds__(kSortThreads, 8) void SortPartKernel(SortArgs a, uint32_t gbits, uint32_t ibits,
                                                                  BigSortWork w) {
  extern __shared__ __align__(16) uint64_t lds_keys[];
  __shared__ uint32_t hist[kBuckets + 1], bstart[kBuckets + 1];
  __shared__ uint32_t s_maxb, n_big;
  __shared__ uint32_t big_list[kBuckets];
  __shared__ unsigned long long s_dmin, s_dmax;
  const uint32_t tid = threadIdx.x;
  const uint32_t n_items = w.counters[2];
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const uint32_t item = w.items[it], c = item >> kPartBits, pi = item & (kMaxParts - 1u);
    const uint32_t *tab = w.part_tab + (size_t)c * kPartTab;
    const uint32_t b0 = tab[pi], m = tab[pi + 1] - b0;
    if (m == 0) continue;  // (uniform)
    const uint64_t base = a.anchor_off[c];
    const uint64_t span = (uint64_t)tab[kMaxParts + 2] + 1ull;
    uint4 *recs = reinterpret_cast<uint4 *>(a.keys) + RecOff(base, c) + b0;  // the part's records {key, qpos, dist} (SortPartitionKernel, exact route)
    const uint32_t *u_qpos = reinterpret_cast<const uint32_t *>(recs) + 2;
    const float *u_dist = reinterpret_cast<const float *>(recs) + 3;
    uint32_t *s_tpos = a.s_tpos + base + b0, *s_qpos = a.s_qpos + base + b0, *s_grp = a.s_grp + base + b0;
    float *s_dist = a.s_dist + base + b0;
    __syncthreads();  // the previous item is done with the shared arrays
#ifdef SMHIP_PHASE_PROF
    if (tid == 0) atomicAdd(&g_phase_prof[m <= kPartLdsKeys ? 20 : 21], 1ull);  // coverage: parts of the workgroup paths
#endif
    if (m <= kPartLdsKeys) {
      if (tid == 0) {
        s_dmin = ~0ull;
        s_dmax = 0ull;
      }
      __syncthreads();
      uint64_t mykeys[kPartPerThread];
      unsigned long long lmin = ~0ull, lmax = 0ull;
#pragma unroll
      for (uint32_t k = 0; k < kPartPerThread; ++k) {
        const uint32_t i = tid + k * kSortThreads;
        mykeys[k] = i < m ? *reinterpret_cast<const uint64_t *>(recs + i) : ~0ull;
        if (i < m) {
          const unsigned long long dense = (mykeys[k] >> (32 + ibits)) * span + ((mykeys[k] >> ibits) & 0xFFFFFFFFull);
          lmin = min(lmin, dense);
          lmax = max(lmax, dense);
        }
      }
      for (int d = 32; d > 0; d >>= 1) {  // wave minima / maxima, then one LDS atomic per wave
        lmin = min(lmin, (unsigned long long)__shfl_xor((long long)lmin, d, 64));
        lmax = max(lmax, (unsigned long long)__shfl_xor((long long)lmax, d, 64));
      }
      if ((tid & 63u) == 0) {
        atomicMin(&s_dmin, lmin);
        atomicMax(&s_dmax, lmax);
      }
      __syncthreads();
      const uint64_t dlo = s_dmin, range = s_dmax - s_dmin + 1ull;
          LdsSortKeys(mykeys, m, lds_keys, hist, bstart, big_list, s_maxb, n_big, span, dlo, range, ibits, tid);
      EmitSorted(lds_keys, m, ibits, u_qpos, u_dist, 4u, s_tpos, s_qpos, s_grp, s_dist, tid);
    } else {
      // a part beyond the LDS (skewed targets): bitonic network over the keys where they lie in the records,
      // padded (npad < 2m) into the part's slices of two columns nobody uses at this point
      struct Split {
        uint4 *rec;
        uint64_t *pad_a, *pad_b;
        uint32_t m, half;
        __device__ uint64_t &operator[](uint32_t i) const {
          if (i < m) return *reinterpret_cast<uint64_t *>(rec + i);
          i -= m;
          return i < half ? pad_a[i] : pad_b[i - half];
        }
      };
      uint32_t npad = 1;
      while (npad < m) npad <<= 1;
      const Split keys{recs, reinterpret_cast<uint64_t *>(a.a_qpos + base + b0),
                       reinterpret_cast<uint64_t *>(a.a_dist + base + b0), m, m / 2};
      for (uint32_t i = m + tid; i < npad; i += kSortThreads) keys[i] = ~0ull;
      __syncthreads();
      BitonicSort(keys, npad, tid);
      EmitSorted(keys, m, ibits, u_qpos, u_dist, 4u, s_tpos, s_qpos, s_grp, s_dist, tid);
    }
  }
}

// ---- one part per WAVE: bitonic network over KPL keys per lane (element e = lane * KPL + r), the payload
// (query position, distance) staged in LDS, no workgroup barriers.  Strides below KPL are compare-exchanges
// between registers of a lane, larger ones between lanes (ds_bpermute).
template <int KPL>
__device__ __forceinline__ void BitonicWave(uint64_t (&k)[KPL], uint32_t lane) {
  constexpr int N = 64 * KPL;
#pragma unroll
  for (int kk = 2; kk <= N; kk <<= 1) {
    // ascending block?  (element index & kk) == 0: a register bit below KPL, a lane bit from there on
    const bool up_lane = kk >= N ? true : (((lane * (uint32_t)KPL) & (uint32_t)kk) == 0u);
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j >= KPL) {
        const int lj = j / KPL;
        const bool keep_min = ((lane & (uint32_t)lj) == 0u) == up_lane;
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
          const uint64_t o = (uint64_t)__shfl_xor((long long)k[r], lj, 64);
          const bool lt = k[r] < o;
          k[r] = (lt == keep_min) ? k[r] : o;
        }
      } else {
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
          if ((r & j) != 0) continue;
          const bool up = kk < KPL ? ((r & kk) == 0) : up_lane;
          const uint64_t x = k[r], y = k[r | j];
          const bool sw = (x > y) == up;
          k[r] = sw ? y : x;
          k[r | j] = sw ? x : y;
        }
      }
    }
  }
}

// ---- the same network on 32-bit keys, 8 per lane (512 keys), written out stage by stage in its "mirror" form
// (the first step of every merge compares element i with its mirror image in the block, all later steps i with
// i ^ j: every comparison then keeps the minimum at the lower index).  Exchanges between lanes of a row of 16
// are DPP operands of v_min_u32 / v_max_u32 -- where the lower / upper lanes are whole banks of four lanes, one
// instruction each -- and only the three steps that cross rows go through ds_bpermute: 664 instructions per
// 512 keys against ~1800 for the generic 64-bit version (tools/micro/bitonic_wave.hip: 416 G keys/s).
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

// s_waitcnt for the records fetched by inline assembly (the prefetch of SortPartWaveKernel); the operands tie
// every later use of them to it
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void WaitRecs(u32x4 (&k)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7])::"memory");
}

// LDS column index of element e: one pad word per 32, so that the lanes' blocked accesses (stride KPL words)
// spread over the banks
__device__ __forceinline__ uint32_t Pad(uint32_t e) { return e + (e >> 5); }

template <int KPL, bool NARROW>
__global__ __launch_bounds__(64) void SortPartWaveKernel(SortArgs a, uint32_t ibits, BigSortWork w, int cls) {
  static_assert(!NARROW || KPL == 8, "the 32-bit network is written for 8 keys per lane");
  constexpr uint32_t CAP = 64 * KPL, COL = CAP + CAP / 32;
  // columns of the part in LDS; the keys pass through the room of the target / group columns on their way
  // from the coalesced loads (element lane + 64 t) to the network's layout (element lane * KPL + r), one pad
  // slot per KPL keys so that both access shapes spread over the banks
  constexpr uint32_t KEYCOL = CAP + 64;
  __shared__ uint32_t col_q[COL], col_d[COL];
  __shared__ __align__(8) uint32_t col_tg[2 * KEYCOL];
  static_assert(2 * KEYCOL >= 2 * COL, "target and group columns fit the key staging room");
  uint32_t *col_t = col_tg, *col_g = col_tg + COL;
  uint64_t *key_stage = reinterpret_cast<uint64_t *>(col_tg);
  const uint32_t lane = threadIdx.x;
  const uint32_t n_items = w.counters[cls];
  const uint4 *recs = cls ? w.recs[1] : w.recs[0];
  const uint64_t *rin = cls ? w.rin[1] : w.rin[0];  // where each part's records lie (SortPartitionKernel)
  const uint64_t idx_mask = (1ull << ibits) - 1ull;
  uint32_t it = blockIdx.x;
  if (it >= n_items) return;
  PHASE_DECL();
  // A part = m records {key, query position, distance} (SortPartitionKernel) at record index A of the key scratch.
  // KPL == 8 (nearly every part) is software-pipelined: the records of the NEXT part are in flight while the
  // current one runs through the network, part descriptors are fetched two parts ahead; no branch around any
  // load (a join would make the compiler wait for everything in flight): indices beyond the part are clamped
  // into it and the values replaced after the fact.
  constexpr bool PREFETCH = KPL == 8;
  const uint4 *rec_base = reinterpret_cast<const uint4 *>(a.keys);
  const uint64_t span = (uint64_t)*a.span_bound + 1ull;
  auto prefetch = [&](const uint4 &rc, uint64_t a_in, u32x4 (&dst)[8]) {
    const uint4 *src0 = rec_base + a_in;
    const uint32_t m = rc.y >> 16;
    // (as inline assembly, or the compiler sinks the loads below the network to save registers; the matching
    // wait is WaitRecs)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint4 *src = src0 + min(lane + 64u * t, m - 1u);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[t]) : "v"(src) : "memory");
    }
  };
  const uint32_t last = n_items - 1u;
  // (the descriptors are wave-uniform, and the compiler would move each into scalar registers the moment it is
  // loaded - a wait for everything in flight; an opaque zero in the index keeps them in vector registers)
  uint32_t opaque0;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
  uint4 rec = recs[it + opaque0];
  uint4 rec_next = recs[min(it + gridDim.x, last) + opaque0];
  uint64_t ain = rin[it + opaque0], ain_next = rin[min(it + gridDim.x, last) + opaque0];
  u32x4 kn[8];
  if (PREFETCH) prefetch(rec, ain, kn);
  for (;;) {
    const uint64_t A = (uint64_t)rec.x | ((uint64_t)(rec.y & 0xFFFFu) << 32);
    const uint32_t m = rec.y >> 16;  // 0 < m <= CAP
#ifdef SMHIP_PHASE_PROF
    if (lane == 0) atomicAdd(&g_phase_prof[NARROW ? 22 : (m <= 512u ? 23 : 24)], 1ull);  // coverage: parts per wave path
#endif
    PHASE_INIT();
    // records -> LDS: keys into the staging room (then read back in the network's layout), payload columns
    if (PREFETCH) WaitRecs(kn);
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
      const uint32_t i = lane + 64u * t;
      uint4 rv;
      if (PREFETCH) rv = make_uint4(kn[t & 7].x, kn[t & 7].y, kn[t & 7].z, kn[t & 7].w);
      else rv = rec_base[ain + min(i, m - 1u)];
      key_stage[i + i / KPL] = i < m ? ((uint64_t)rv.x | ((uint64_t)rv.y << 32)) : ~0ull;
      col_q[Pad(i)] = rv.z;  // (slots beyond the part get the clamped copies: never read)
      col_d[Pad(i)] = rv.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    uint64_t k[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) k[r] = key_stage[lane * (KPL + 1) + r];
    uint32_t k32[KPL];
    if (NARROW) {  // (dense key - the part's lower bound) << 9 | element: unique, ordered like the 64-bit keys
      const uint64_t dlo = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
#pragma unroll
      for (int r = 0; r < KPL; ++r) {
        const uint32_t e = lane * KPL + r;
        const uint64_t dense = (k[r] >> (32 + ibits)) * span + ((k[r] >> ibits) & 0xFFFFFFFFull);
        k32[r] = e < m ? (((uint32_t)(dense - dlo) << 9) | e) : 0xFFFFFFFFu;
      }
    }
    PHASE_MARK(5);
    const uint32_t it_next = it + gridDim.x;
    const bool more = it_next < n_items;
    if (PREFETCH) prefetch(rec_next, ain_next, kn);  // (past the end: the last part again, unused)
    const uint4 rec_next2 = recs[min(it_next + gridDim.x, last) + opaque0];
    const uint64_t ain_next2 = rin[min(it_next + gridDim.x, last) + opaque0];
    __builtin_amdgcn_sched_barrier(0);  // all of the above is ISSUED before the network starts
    PHASE_MARK(0);
    if constexpr (NARROW) BitonicWave32(k32, lane);
    else BitonicWave<KPL>(k, lane);
    __builtin_amdgcn_sched_barrier(0);  // ... and nothing that waits for them is hoisted into it
    PHASE_MARK(1);
    // payload of the sorted keys (index field = position inside the part), then everything back into the
    // columns in sorted order
    uint32_t q[KPL], d[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      // (padding keys: any slot)
      const uint32_t from = NARROW ? (k32[r] & (CAP - 1u)) : min((uint32_t)(k[r] & idx_mask), CAP - 1u);
      q[r] = col_q[Pad(from)];
      d[r] = col_d[Pad(from)];
      if (NARROW) k[r] = key_stage[from + from / KPL];  // the 64-bit key of that element: its target and group
      // (!NARROW: the sorted 64-bit keys are in registers; the staging room is simply overwritten below)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      const uint32_t e = Pad(lane * KPL + r);
      col_q[e] = q[r];
      col_d[e] = d[r];
      col_t[e] = (uint32_t)(k[r] >> ibits);
      col_g[e] = (uint32_t)(k[r] >> (32 + ibits));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    PHASE_MARK(2);
    // runs of equal (group, target): ordered by (query, distance) by the lane that owns the run's first element
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      const uint32_t e = lane * KPL + r;
      if (e + 1 >= m) continue;
      const uint32_t t = col_t[Pad(e)], g = col_g[Pad(e)];
      if (col_t[Pad(e + 1)] != t || col_g[Pad(e + 1)] != g) continue;        // no run here
      if (e > 0 && col_t[Pad(e - 1)] == t && col_g[Pad(e - 1)] == g) continue;  // not its first element
      uint32_t end = e + 2;
      while (end < m && col_t[Pad(end)] == t && col_g[Pad(end)] == g) ++end;
      for (uint32_t x = e + 1; x < end; ++x) {
        const uint32_t kq = col_q[Pad(x)];
        const float kd = __uint_as_float(col_d[Pad(x)]);
        uint32_t y = x;
        while (y > e && (col_q[Pad(y - 1)] > kq ||
                         (col_q[Pad(y - 1)] == kq && __uint_as_float(col_d[Pad(y - 1)]) > kd))) {
          col_q[Pad(y)] = col_q[Pad(y - 1)];
          col_d[Pad(y)] = col_d[Pad(y - 1)];
          --y;
        }
        col_q[Pad(y)] = kq;
        col_d[Pad(y)] = __float_as_uint(kd);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    PHASE_MARK(3);
    uint32_t *s_tpos = a.s_tpos + A, *s_qpos = a.s_qpos + A, *s_grp = a.s_grp + A;
    uint32_t *s_dist = reinterpret_cast<uint32_t *>(a.s_dist + A);
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
      const uint32_t i = lane + 64 * t;
      if (i < m) {
        s_tpos[i] = col_t[Pad(i)];
        s_grp[i] = col_g[Pad(i)];
        s_qpos[i] = col_q[Pad(i)];
        s_dist[i] = col_d[Pad(i)];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();  // the columns are free for the next part
    PHASE_MARK(4);
    if (!more) break;
    it = it_next;
    rec = rec_next;
    rec_next = rec_next2;
    ain = ain_next;
    ain_next = ain_next2;
  }
  PHASE_FLUSH();
}

// Group boundaries of the large chunks, from the sorted group column: every element that starts a
// group appends its index to a list (coalesced, order-free), then the list - one entry per
// (reference, strand) present - is rank-sorted; more groups than the list holds: counted and written
// in index order.
__global__ __launch_bounds__(kSortThreads) void GroupStartKernel(SortArgs a, uint32_t n_chunks) {
  __shared__ uint32_t part[kSortThreads];
  __shared__ uint32_t s_cnt;
  const uint32_t tid = threadIdx.x;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const uint64_t base = a.anchor_off[c];
    const uint32_t n = a.results[c].n_anchors;  // (0 for a flagged chunk)
    if (n == 0) continue;  // (uniform)
    const uint32_t *grp = a.s_grp + base;
    uint32_t *gs = a.group_start + base;
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kSortThreads)
      if (i == 0 || grp[i - 1] != grp[i]) {
        const uint32_t k = atomicAdd(&s_cnt, 1u);
        if (k < kSortThreads) part[k] = i;
      }
    __syncthreads();
    const uint32_t G = s_cnt;
    if (G <= kSortThreads) {
      if (tid < G) {
        const uint32_t v = part[tid];
        uint32_t rank = 0;
        for (uint32_t x = 0; x < G; ++x) rank += part[x] < v ? 1u : 0u;
        gs[rank] = v;
      }
      if (tid == 0) a.n_groups[c] = G;
      continue;
    }
    const uint32_t per_a = (n + kSortThreads - 1) / kSortThreads;
    const uint32_t ab = min(n, tid * per_a), ae = min(n, ab + per_a);
    uint32_t cnt = 0;
    for (uint32_t i = ab; i < ae; ++i) cnt += (i == 0 || grp[i - 1] != grp[i]) ? 1 : 0;
    __syncthreads();
    part[tid] = cnt;
    __syncthreads();
    for (uint32_t d = 1; d < kSortThreads; d <<= 1) {
      const uint32_t v = tid >= d ? part[tid - d] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    uint32_t wpos = tid ? part[tid - 1] : 0;
    for (uint32_t i = ab; i < ae; ++i)
      if (i == 0 || grp[i - 1] != grp[i]) gs[wpos++] = i;
    if (tid == kSortThreads - 1) a.n_groups[c] = part[tid];
  }
}

// anchors per chunk = carried + sum over its queries of min(hits, num_nearest_points); also the
// chunk's search counters (nodes visited, points tested, hits) from the per-query values
__global__ __launch_bounds__(256) void AnchorCountKernel(const uint32_t *__restrict__ q_off,
                                                         const uint32_t *__restrict__ q_count,
                                                         const uint32_t *__restrict__ q_nodes,
                                                         const uint32_t *__restrict__ q_points,
                                                         const uint32_t *__restrict__ carried_off,
                                                         uint32_t nnp, uint32_t *__restrict__ chunk_cnt,
                                                         smhip_chunk_result *__restrict__ results) {
  __shared__ unsigned long long red[4][256];
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  const uint32_t q0 = q_off[c], q1 = q_off[c + 1];
  unsigned long long s = 0, sn = 0, sp = 0, sh = 0;
  for (uint32_t q = q0 + tid; q < q1; q += 256) {
    const uint32_t h = q_count[q];
    s += min(h, nnp);
    sh += h;
    sn += q_nodes[q];
    sp += q_points[q];
  }
  red[0][tid] = s;
  red[1][tid] = sn;
  red[2][tid] = sp;
  red[3][tid] = sh;
  __syncthreads();
  for (uint32_t d = 128; d > 0; d >>= 1) {
    if (tid < d) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + d];
    }
    __syncthreads();
  }
  if (tid == 0) {
    chunk_cnt[c] = (uint32_t)red[0][0] + (carried_off ? carried_off[c + 1] - carried_off[c] : 0);
    results[c].nodes_visited = red[1][0];
    results[c].points_tested = red[2][0];
    results[c].hits_total = red[3][0];
  }
}
}  // namespace

hipError_t LaunchAnchorCount(smhip_ctx *ctx, const smhip_params &p, const uint32_t *q_off,
                             const uint32_t *q_count, const uint32_t *q_nodes, const uint32_t *q_points,
                             const uint32_t *carried_off, uint32_t n_chunks, uint32_t *chunk_cnt,
                             smhip_chunk_result *results) {
  hipLaunchKernelGGL(AnchorCountKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, q_off, q_count, q_nodes, q_points,
                     carried_off, (uint32_t)p.num_nearest_points, chunk_cnt, results);
  return hipGetLastError();
}

hipError_t LaunchIndexMaxTarget(smhip_ctx *ctx, uint32_t *out) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t), ctx->stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(IndexMaxTargetKernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->index.leaf_tab,
                     reinterpret_cast<const float4 *>(ctx->index.leaf_pts), ctx->index.n_leaves, out);
  return hipGetLastError();
}
hipError_t LaunchCarriedMaxTarget(smhip_ctx *ctx, const smhip_carried_anchor *carried, uint64_t n_carried,
                                  const uint32_t *index_max, uint32_t *out) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t), ctx->stream);
  if (e != hipSuccess) return e;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n_carried + 255) / 256 + 1, 1024);
  hipLaunchKernelGGL(CarriedMaxTargetKernel, dim3(grid), dim3(256), 0, ctx->stream, carried, n_carried, index_max, out);
  return hipGetLastError();
}

hipError_t InitSortAttributes() {  // per device (smhip_init): the workgroup part sort keeps up to 8192 keys in LDS,
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(SortPartKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)((size_t)kPartLdsKeys * sizeof(uint64_t)));
  if (e != hipSuccess) return e;  // ... the partition kernel's whole-line route a tile of 2048 records beside its tables
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(SortPartitionKernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)TileLdsBytes(2));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(SortPartitionKernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)TileLdsBytes(4));
}

// 16-byte records the sort needs in a.keys for n_anchors_total anchors in n_chunks chunks (RecOff)
uint64_t SortRecordRoom(uint64_t n_anchors_total, uint32_t n_chunks) { return RecOff(n_anchors_total, n_chunks) + 4; }

hipError_t LaunchSort(smhip_ctx *ctx, const SortArgs &a, uint32_t n_chunks, uint64_t n_anchors_total) {
  uint32_t gbits = 1;
  while ((1ull << gbits) < 2ull * a.n_refs) ++gbits;
  if (gbits > 24) return hipErrorInvalidValue;
  const uint32_t ibits = 64 - 32 - gbits;
  if (n_chunks == 0) return hipSuccess;
  // scratch: part tables, part descriptors / work items per size class (at most kMaxParts per chunk, and no
  // more than the anchors allow), three counters
  const uint64_t max_items = std::min<uint64_t>((uint64_t)n_chunks * kMaxParts,
                                                2 * (n_anchors_total / kPartTarget) + 3ull * n_chunks + 4);
  const size_t words = (size_t)n_chunks * (kPartTab + 1) + 13 * max_items + 16;
  hipError_t e = ctx->sort_tmp.Reserve(words * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  BigSortWork w;
  w.counters = ctx->sort_tmp.as<uint32_t>();
  w.part_tab = w.counters + 4;
  w.recs[0] = reinterpret_cast<uint4 *>(w.part_tab + (((size_t)n_chunks * kPartTab + 3) & ~(size_t)3));  // (16-byte aligned:
  w.recs[1] = w.recs[0] + max_items;                                                                       // counters is)
  w.rin[0] = reinterpret_cast<uint64_t *>(w.recs[1] + max_items);
  w.rin[1] = w.rin[0] + max_items;
  w.items = reinterpret_cast<uint32_t *>(w.rin[1] + max_items);
  w.redo = w.items + max_items;
  if ((e = hipMemsetAsync(w.counters, 0, 4 * sizeof(uint32_t), ctx->stream)) != hipSuccess) return e;
  const uint32_t part_blocks = std::min<uint32_t>(n_chunks, 256 * (2048 / kBigThreads));
  if (ctx->knobs.sort_lines != 0) {  // (knob sort_lines = 0: the exact two-pass route for every chunk, A/B)
    // tiles of 4096 for batches of large chunks (beyond ~100 parts a tile of 2048 gives a part fewer than 20 records);
    // knob sort_lines = 2 / 4: that tile size whatever the batch
    const bool big_tiles = ctx->knobs.sort_lines == 4 || (ctx->knobs.sort_lines != 2 && n_anchors_total / n_chunks > 40000);
    if (big_tiles)
      hipLaunchKernelGGL((SortPartitionKernel<true, 4>), dim3(std::min<uint32_t>(n_chunks, 256)), dim3(kBigThreads), TileLdsBytes(4),
                         ctx->stream, a, n_chunks, gbits, ibits, w, (const uint32_t *)nullptr);
    else
      hipLaunchKernelGGL((SortPartitionKernel<true, 2>), dim3(part_blocks), dim3(kBigThreads), TileLdsBytes(2), ctx->stream, a,
                         n_chunks, gbits, ibits, w, (const uint32_t *)nullptr);
    // the chunks it listed (none, as a rule: the launch then ends at once)
    hipLaunchKernelGGL((SortPartitionKernel<false, 2>), dim3(std::min<uint32_t>(part_blocks, 256)), dim3(kBigThreads), 0, ctx->stream,
                       a, n_chunks, gbits, ibits, w, (const uint32_t *)w.redo);
  } else {
    hipLaunchKernelGGL((SortPartitionKernel<false, 2>), dim3(part_blocks), dim3(kBigThreads), 0, ctx->stream, a, n_chunks, gbits,
                       ibits, w, (const uint32_t *)nullptr);
  }
  // one wave per part (persistent: the grid fills the chip, every wave takes parts in turn); the parts beyond
  // a wave's registers - rare - go to the workgroup version
  const uint32_t wave_grid = (uint32_t)std::min<uint64_t>(max_items, 256 * 18);
  hipLaunchKernelGGL((SortPartWaveKernel<8, true>), dim3(wave_grid), dim3(64), 0, ctx->stream, a, ibits, w, 0);
  hipLaunchKernelGGL((SortPartWaveKernel<16, false>), dim3(std::min<uint32_t>(wave_grid, 256 * 9)), dim3(64), 0,
                     ctx->stream, a, ibits, w, 1);
  const uint32_t part_grid = (uint32_t)std::min<uint64_t>(max_items, 1024);
  hipLaunchKernelGGL(SortPartKernel, dim3(part_grid), dim3(kSortThreads), (size_t)kPartLdsKeys * sizeof(uint64_t),
                     ctx->stream, a, gbits, ibits, w);
  if ((1u << gbits) > kGroupsLds)  // (else the partition kernel has listed the group boundaries)
    hipLaunchKernelGGL(GroupStartKernel, dim3(std::min<uint32_t>(n_chunks, 1024)), dim3(kSortThreads), 0, ctx->stream, a,
                       n_chunks);
  return hipGetLastError();
}
}  // namespace smhip

#ifdef SMHIP_PHASE_PROF
extern "C" int smhip_debug_phase_prof(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(smhip::g_phase_prof), 32 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smhip::g_phase_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
