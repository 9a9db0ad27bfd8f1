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
// Method: 64-bit keys [group | target_position | unsorted index].  Up to 16384 keys are sorted in
// LDS in two levels (counting sort into 1024 buckets of a dense monotone key, then a rank sort of
// every bucket); a key set with a heavy bucket falls back to a bitonic network in LDS.  A chunk with
// more keys is first cut into parts of that size by key range (below).  Runs of equal (group, target)
// - short - are then ordered by (query, distance) in place.  Every comparison key is total, so the
// result is the unique sorted sequence the reference's std::sort produces.
#include "common.h"

namespace smhip {
namespace {

constexpr int kSortThreads = 1024;
constexpr uint32_t kLdsKeys = 16384;  // 128 KB of LDS
constexpr uint32_t kMaxQueriesPerChunk = (kFeatStride - 5) / 1 + 1;  // step >= 1
constexpr uint32_t kBucketBits = 10;
constexpr uint32_t kBuckets = 1u << kBucketBits;  // buckets of the two-level sort (~9 keys each: four per wave pass)
constexpr uint32_t kMaxBucket = 512;  // largest bucket one wave rank-sorts (8 keys per lane)
// A chunk with more than kLdsKeys anchors (large references: 23 000 per chunk at 32 Mb, 181 000 at
// 250 Mb) is cut into PARTS by ranges of the dense key group * (max target + 1) + target -- one counting
// pass over its keys -- and every part, about kPartTarget keys, is then sorted like a small chunk, in
// LDS, by a workgroup of SortPartKernel: the per-anchor cost of the large-reference sort is the LDS
// path's plus one more read and write of the key.  A part that does not fit the LDS (a skewed target
// distribution) is sorted by a bitonic network in its own slice of the key scratch.
constexpr uint32_t kPartTarget = 5632;  // keys per part aimed at (a part sorts in LDS up to 8192 keys)
constexpr uint32_t kMaxParts = 256;     // parts per chunk (chunks of up to ~1.5 M anchors; more: bigger parts)
constexpr uint32_t kPartTab = kMaxParts + 4;  // per chunk: part starts [P + 1], then [kMaxParts + 1] = P, [+ 2] = max target
constexpr uint32_t kPerThread = (kLdsKeys + kSortThreads - 1) / kSortThreads;  // 16 keys per thread in registers

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


// Two-level LDS sort of n <= kLdsKeys keys that the threads hold in registers (mykeys[it] = key of
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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
template <typename KeyPtr>
__device__ __forceinline__ void EmitSorted(KeyPtr keys, uint32_t n, uint32_t ibits, const uint32_t *u_qpos,
                                           const float *u_dist, uint32_t *s_tpos, uint32_t *s_qpos, uint32_t *s_grp,
                                           float *s_dist, uint32_t tid) {
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
      gq[u] = u_qpos[from];
      gd[u] = u_dist[from];
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

// part of a key: floor(dense * P / 2^rbits), dense < range <= 2^rbits (monotone in the key)
__device__ __forceinline__ uint32_t PartOf(uint64_t key, uint64_t span, uint32_t ibits, uint32_t P, uint32_t rbits) {
  const uint64_t dense = (key >> (32 + ibits)) * span + ((key >> ibits) & 0xFFFFFFFFull);
  return (uint32_t)((dense * P) >> rbits);
}

struct BigSortWork {   // global scratch of the large-chunk path (LaunchSort)
  uint32_t *part_tab;  // [n_chunks][kPartTab]
  uint32_t *items;     // (chunk << 8) | part
  uint32_t *big_chunks;
  uint32_t *counters;  // [0] items, [1] big chunks
};

__global__ __launch_bounds__(kSortThreads) void SortKernel(SortArgs a, uint32_t gbits, uint32_t ibits, BigSortWork w) {
  extern __shared__ __align__(16) uint64_t lds_keys[];
  __shared__ uint32_t part[kSortThreads];
  __shared__ uint32_t hist[kBuckets + 1], bstart[kBuckets + 1];
  __shared__ uint32_t s_maxt, s_maxb, n_big;
  __shared__ uint32_t big_list[kBuckets];
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint64_t base = a.anchor_off[c];
  const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
  const uint32_t q0 = a.q_off[c], nq = a.q_off[c + 1] - q0;
  const uint32_t ncar = a.carried_off ? a.carried_off[c + 1] - a.carried_off[c] : 0;
  if (tid == 0) {
    a.results[c].n_anchors = n;
    a.n_groups[c] = 0;
    s_maxt = 0;
    s_maxb = 0;
    n_big = 0;
  }
  if (n == 0) return;
  if (n > (1u << ibits) || nq > kMaxQueriesPerChunk || c >= (1u << 24)) {  // the key cannot index this many anchors
    if (tid == 0) {
      a.results[c].status |= SMHIP_ST_SORT_CAP;
      a.results[c].n_anchors = 0;
    }
    return;
  }
  const bool in_lds = n <= kLdsKeys;
  uint64_t *gkeys = a.keys + 2 * base;  // per-chunk scratch of 2n keys
  uint32_t *u_qpos = a.a_qpos + base;  // (target and group travel in the key)
  float *u_dist = a.a_dist + base;
  // ---- exclusive offsets of the queries' anchor runs (block scan over nq counts) into LDS
  // (the key array is not in use yet: its tail holds the nq+1 offsets during the decode)
  uint32_t *qoff = reinterpret_cast<uint32_t *>(lds_keys) + 2 * kLdsKeys - (kMaxQueriesPerChunk + 2);
  const uint32_t per = (nq + kSortThreads - 1) / kSortThreads;
  const uint32_t qb = min(nq, tid * per), qe = min(nq, qb + per);
  uint32_t s = 0;
  for (uint32_t k = qb; k < qe; ++k) s += min(a.q_count[q0 + k], a.nnp);
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < kSortThreads; d <<= 1) {
    const uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  {
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t k = qb; k < qe; ++k) {
      qoff[k] = run;
      run += min(a.q_count[q0 + k], a.nnp);
    }
    if (tid == kSortThreads - 1) qoff[nq] = part[tid];
  }
  __syncthreads();
  // ---- decode: carried anchors, then each query's hits in traversal order.  One thread per
  // anchor (binary search of its query in the LDS offsets), so the hit records of a query are
  // read by consecutive lanes.  Keys go to registers first: they overwrite the offsets' LDS.
  const uint32_t qso = a.qstart ? a.qstart[c] : 0;
  uint64_t mykeys[kPerThread];
  const bool reg_keys = in_lds;  // LDS path: n <= kLdsKeys, so kPerThread registers suffice
  uint32_t local_maxt = 0;
  uint32_t search_steps = 0;  // binary-search steps that cover nq + 1 offsets
  while ((1u << search_steps) < nq + 1) ++search_steps;
  const uint64_t car0 = a.carried_off ? a.carried_off[c] : 0;
  // kDec anchors per pass: their query searches advance together, then the 16-byte records (a hit,
  // or a carried anchor: same size) are fetched with one wait, then the keys.
  constexpr int kDec = 4;
  auto decode4 = [&](uint32_t i0, uint64_t (&keys4)[kDec]) {
    uint32_t lo[kDec], kq[kDec];
    const uint4 *src[kDec];
    bool live[kDec], zero[kDec];
#pragma unroll
    for (int u = 0; u < kDec; ++u) {
      lo[u] = 0;
      live[u] = i0 + u * kSortThreads < n;
    }
    for (uint32_t st = search_steps; st-- > 0;) {  // largest k with qoff[k] <= h0 (qoff[0] = 0)
#pragma unroll
      for (int u = 0; u < kDec; ++u) {
        const uint32_t i = i0 + u * kSortThreads;
        const uint32_t h0 = i >= ncar ? i - ncar : 0;
        const uint32_t mid = lo[u] + (1u << st);
        if (mid < nq && qoff[mid] <= h0) lo[u] = mid;
      }
    }
#pragma unroll
    for (int u = 0; u < kDec; ++u) {
      const uint32_t i = i0 + u * kSortThreads;
      zero[u] = false;
      kq[u] = lo[u];
      src[u] = reinterpret_cast<const uint4 *>(a.ovf_pool);  // harmless default for lanes past the end
      if (live[u]) {
        if (i < ncar) {
          src[u] = reinterpret_cast<const uint4 *>(a.carried + car0 + i);
        } else {
          const uint32_t h = i - ncar - qoff[lo[u]];
          const uint32_t qi = q0 + lo[u];
          if (h < a.fast_cap) {
            src[u] = reinterpret_cast<const uint4 *>(a.region.At(qi, h));
          } else {  // geometric overflow segments (seed_lookup.hip EmitOverflow)
            const uint32_t kk = h - a.fast_cap;
            const uint32_t level = 31u - (uint32_t)__clz((int)(kk / a.fast_cap + 1));
            const uint32_t first = a.fast_cap * ((1u << level) - 1u);
            const uint32_t seg = level < (uint32_t)kOvfLevels ? a.ovf_tab[(size_t)qi * kOvfLevels + level] : 0xFFFFFFFFu;
            if (seg != 0xFFFFFFFFu) src[u] = reinterpret_cast<const uint4 *>(a.ovf_pool + (size_t)seg + (kk - first));
            else zero[u] = true;
          }
        }
      }
    }
    uint4 raw[kDec];
#pragma unroll
    for (int u = 0; u < kDec; ++u) raw[u] = *src[u];
#pragma unroll
    for (int u = 0; u < kDec; ++u) {
      const uint32_t i = i0 + u * kSortThreads;
      if (!live[u]) continue;
      uint32_t tpos, qpos, grp;
      float dist;
      if (i < ncar) {  // smhip_carried_anchor {ref_strand, target_position, query_position, distance}
        grp = raw[u].x;
        tpos = raw[u].y;
        qpos = raw[u].z;
        dist = __uint_as_float(raw[u].w);
      } else {  // Hit {uint64 position, dist, slot}
        const uint64_t position = zero[u] ? 0ull : ((uint64_t)raw[u].x | ((uint64_t)raw[u].y << 32));
        const uint32_t ref = (uint32_t)(position >> 33);
        tpos = (uint32_t)(position >> 1);
        grp = (ref << 1) | (uint32_t)(position & 1);
        qpos = (kq[u] + 1) * a.step + qso;  // position + query_start_offset
        dist = zero[u] ? 0.0f : __uint_as_float(raw[u].z);
      }
      local_maxt = max(local_maxt, tpos);
      u_qpos[i] = qpos;
      u_dist[i] = dist;
      keys4[u] = ((uint64_t)grp << (32 + ibits)) | ((uint64_t)tpos << ibits) | i;
    }
  };
  if (reg_keys) {  // statically indexed registers (a run-time index would put the array in scratch)
#pragma unroll
    for (uint32_t bt = 0; bt < kPerThread / kDec; ++bt) {
      uint64_t k4[kDec];
#pragma unroll
      for (int u = 0; u < kDec; ++u) k4[u] = ~0ull;
      if (tid + bt * kDec * kSortThreads < n) decode4(tid + bt * kDec * kSortThreads, k4);
#pragma unroll
      for (int u = 0; u < kDec; ++u) mykeys[bt * kDec + u] = k4[u];
    }
  } else {
    for (uint32_t i0 = tid; i0 < n; i0 += kDec * kSortThreads) {
      uint64_t k4[kDec];
      decode4(i0, k4);
#pragma unroll
      for (int u = 0; u < kDec; ++u)
        if (i0 + u * kSortThreads < n) gkeys[i0 + u * kSortThreads] = k4[u];
    }
  }
  atomicMax(&s_maxt, local_maxt);
  __syncthreads();  // everyone is done with the offsets; s_maxt ready
  const uint64_t span = (uint64_t)s_maxt + 1ull;
  const uint64_t range = span << gbits;  // dense keys are < range
  if (in_lds) {
    LdsSortKeys(mykeys, n, lds_keys, hist, bstart, big_list, s_maxb, n_big, span, 0ull, range, ibits, tid);
    EmitSorted(lds_keys, n, ibits, u_qpos, u_dist, a.s_tpos + base, a.s_qpos + base, a.s_grp + base, a.s_dist + base,
               tid);
  } else {
    // ---- large chunk: cut the keys (in gkeys[0, n)) into P parts by dense-key range, parts
    // contiguous in gkeys[n, 2n); SortPartKernel sorts the parts, GroupStartKernel lists the groups
    uint32_t *pcount = reinterpret_cast<uint32_t *>(lds_keys);  // [kMaxParts] counts, then cursors
    uint32_t *pstart = pcount + kMaxParts;                       // [kMaxParts + 1]
    const uint32_t rbits = 64u - (uint32_t)__clzll((long long)(range - 1ull) | 1ll);
    // as many parts as n / kPartTarget of them are non-empty: dense keys only reach range / 2^rbits of the scale
    const uint64_t want = ((uint64_t)(n + kPartTarget - 1) / kPartTarget << rbits) / range + 1;
    const uint32_t P = (uint32_t)min<uint64_t>(want, kMaxParts);
    if (tid < kMaxParts) pcount[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kSortThreads) atomicAdd(&pcount[PartOf(gkeys[i], span, ibits, P, rbits)], 1u);
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the part sizes by one wave
      constexpr uint32_t kPer = kMaxParts / 64;
      uint32_t h[kPer], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        h[k] = pcount[kPer * tid + k];
        sum += h[k];
      }
      uint32_t incl = sum;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)tid >= d) incl += o;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        pstart[kPer * tid + k] = run;
        run += h[k];
        pcount[kPer * tid + k] = 0;  // becomes the insertion cursor
      }
      if (tid == 63) pstart[kMaxParts] = run;  // == n
    }
    __syncthreads();
    uint64_t *gk2 = gkeys + n;
    for (uint32_t i = tid; i < n; i += kSortThreads) {
      const uint64_t key = gkeys[i];
      const uint32_t b = PartOf(key, span, ibits, P, rbits);
      gk2[pstart[b] + atomicAdd(&pcount[b], 1u)] = key;
    }
    uint32_t *tab = w.part_tab + (size_t)c * kPartTab;
    if (tid <= P) tab[tid] = pstart[tid];
    if (tid == 0) {
      tab[kMaxParts + 1] = P;
      tab[kMaxParts + 2] = s_maxt;
      s_maxb = atomicAdd(&w.counters[0], P);  // (reused as the base of this chunk's work items)
      w.big_chunks[atomicAdd(&w.counters[1], 1u)] = c;
    }
    __syncthreads();
    if (tid < P) w.items[s_maxb + tid] = (c << 8) | tid;
    return;
  }
  // ---- group boundaries, in order: every element that starts a group appends its index to the
  // list (coalesced, order-free), then the list - one entry per (reference, strand) present - is
  // rank-sorted.  Chunks with more groups than the rank sort handles count and write in index order.
  {
    const uint32_t gsh = 32 + ibits;
    auto grp_of = [&](uint32_t i) { return (uint32_t)(lds_keys[i] >> gsh); };
    uint32_t *gs = a.group_start + base;
    __syncthreads();
    if (tid == 0) s_maxb = 0;  // reused as the list length
    __syncthreads();
    uint32_t mine_cnt = 0;
    for (uint32_t i = tid; i < n; i += kSortThreads) mine_cnt += (i == 0 || grp_of(i - 1) != grp_of(i)) ? 1u : 0u;
    if (mine_cnt) atomicAdd(&s_maxb, mine_cnt);
    __syncthreads();
    const uint32_t G = s_maxb;
    if (G <= kSortThreads) {
      __syncthreads();
      if (tid == 0) s_maxt = 0;  // reused as the append cursor
      __syncthreads();
      for (uint32_t i = tid; i < n; i += kSortThreads)
        if (i == 0 || grp_of(i - 1) != grp_of(i)) part[atomicAdd(&s_maxt, 1u)] = i;
      __syncthreads();
      if (tid < G) {
        const uint32_t v = part[tid];
        uint32_t rank = 0;
        for (uint32_t x = 0; x < G; ++x) rank += part[x] < v ? 1u : 0u;
        gs[rank] = v;
      }
      if (tid == 0) a.n_groups[c] = G;
    } else {
      const uint32_t per_a = (n + kSortThreads - 1) / kSortThreads;
      const uint32_t ab = min(n, tid * per_a), ae = min(n, ab + per_a);
      uint32_t cnt = 0;
      for (uint32_t i = ab; i < ae; ++i) cnt += (i == 0 || grp_of(i - 1) != grp_of(i)) ? 1 : 0;
      __syncthreads();
      part[tid] = cnt;
      __syncthreads();
      for (uint32_t d = 1; d < kSortThreads; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
      }
      uint32_t w = tid ? part[tid - 1] : 0;
      for (uint32_t i = ab; i < ae; ++i)
        if (i == 0 || grp_of(i - 1) != grp_of(i)) gs[w++] = i;
      if (tid == kSortThreads - 1) a.n_groups[c] = part[tid];
    }
  }
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
  const uint32_t n_items = w.counters[0];
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const uint32_t item = w.items[it], c = item >> 8, pi = item & 0xFFu;
    const uint32_t *tab = w.part_tab + (size_t)c * kPartTab;
    const uint32_t b0 = tab[pi], m = tab[pi + 1] - b0;
    if (m == 0) continue;  // (uniform)
    const uint64_t base = a.anchor_off[c];
    const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
    const uint64_t span = (uint64_t)tab[kMaxParts + 2] + 1ull;
    uint64_t *gk1 = a.keys + 2 * base + b0;  // the same index range in the first half: dead, this part's scratch
    uint64_t *gk2 = a.keys + 2 * base + n + b0;
    const uint32_t *u_qpos = a.a_qpos + base;
    const float *u_dist = a.a_dist + base;
    uint32_t *s_tpos = a.s_tpos + base + b0, *s_qpos = a.s_qpos + base + b0, *s_grp = a.s_grp + base + b0;
    float *s_dist = a.s_dist + base + b0;
    __syncthreads();  // the previous item is done with the shared arrays
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
        mykeys[k] = i < m ? gk2[i] : ~0ull;
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
      EmitSorted(lds_keys, m, ibits, u_qpos, u_dist, s_tpos, s_qpos, s_grp, s_dist, tid);
    } else {
      // a part beyond the LDS (skewed targets): bitonic network over gk2[b0, b0 + m) padded into the dead
      // first-half slice gk1[b0, b0 + m) (npad < 2m)
      struct Split {
        uint64_t *lo, *hi;
        uint32_t m;
        __device__ uint64_t &operator[](uint32_t i) const { return i < m ? lo[i] : hi[i - m]; }
      };
      uint32_t npad = 1;
      while (npad < m) npad <<= 1;
      const Split keys{gk2, gk1, m};
      for (uint32_t i = m + tid; i < npad; i += kSortThreads) keys[i] = ~0ull;
      __syncthreads();
      BitonicSort(keys, npad, tid);
      EmitSorted(gk2, m, ibits, u_qpos, u_dist, s_tpos, s_qpos, s_grp, s_dist, tid);
    }
  }
}

// Group boundaries of the large chunks, from the sorted group column: every element that starts a
// group appends its index to a list (coalesced, order-free), then the list - one entry per
// (reference, strand) present - is rank-sorted; more groups than the list holds: counted and written
// in index order.
__global__ __launch_bounds__(kSortThreads) void GroupStartKernel(SortArgs a, BigSortWork w) {
  __shared__ uint32_t part[kSortThreads];
  __shared__ uint32_t s_cnt;
  const uint32_t tid = threadIdx.x;
  const uint32_t n_bigc = w.counters[1];
  for (uint32_t it = blockIdx.x; it < n_bigc; it += gridDim.x) {
    const uint32_t c = w.big_chunks[it];
    const uint64_t base = a.anchor_off[c];
    const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
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

hipError_t InitSortAttributes() {  // per device (smhip_init): the sorts keep up to kLdsKeys keys in LDS
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(SortKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)((size_t)kLdsKeys * sizeof(uint64_t)));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(SortPartKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)((size_t)kPartLdsKeys * sizeof(uint64_t)));
}

hipError_t LaunchSort(smhip_ctx *ctx, const SortArgs &a, uint32_t n_chunks, uint64_t n_anchors_total) {
  uint32_t gbits = 1;
  while ((1ull << gbits) < 2ull * a.n_refs) ++gbits;
  if (gbits > 24) return hipErrorInvalidValue;
  const uint32_t ibits = 64 - 32 - gbits;
  const size_t lds = (size_t)kLdsKeys * sizeof(uint64_t);
  // scratch of the large-chunk path: part tables, work items (at most kMaxParts per chunk, and no more
  // than the anchors allow), the list of large chunks, two counters
  const uint64_t max_items = std::min<uint64_t>((uint64_t)n_chunks * kMaxParts,
                                                2 * (n_anchors_total / kPartTarget) + 3ull * n_chunks + 4);
  const size_t words = (size_t)n_chunks * kPartTab + max_items + n_chunks + 4;
  hipError_t e = ctx->sort_tmp.Reserve(words * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  BigSortWork w;
  w.counters = ctx->sort_tmp.as<uint32_t>();
  w.part_tab = w.counters + 4;
  w.items = w.part_tab + (size_t)n_chunks * kPartTab;
  w.big_chunks = w.items + max_items;
  if ((e = hipMemsetAsync(w.counters, 0, 4 * sizeof(uint32_t), ctx->stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(SortKernel, dim3(n_chunks), dim3(kSortThreads), lds, ctx->stream, a, gbits, ibits, w);
  if (n_anchors_total > kLdsKeys) {  // some chunk may be large
    const uint32_t part_grid = (uint32_t)std::min<uint64_t>(max_items, 4096);
    hipLaunchKernelGGL(SortPartKernel, dim3(part_grid), dim3(kSortThreads), (size_t)kPartLdsKeys * sizeof(uint64_t),
                       ctx->stream, a, gbits, ibits, w);
    hipLaunchKernelGGL(GroupStartKernel, dim3(std::min<uint32_t>(n_chunks, 1024)), dim3(kSortThreads), 0, ctx->stream, a, w);
  }
  return hipGetLastError();
}
}  // namespace smhip
