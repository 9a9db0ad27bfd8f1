// chain.hip -- chaining DP, traceback, primary-chain selection and MAPQ.  Kernels: C0 classify (a wave per
// 64-anchor tile: which anchors have no predecessor that passes the gap tests -- their predecessor windows laid
// out as dense (anchor, predecessor) pairs --, and the chaining of those whose passing predecessors are all of
// that kind), C0b pending (a workgroup per chunk: the anchors with a passing predecessor listed for the end
// search, those still unsettled listed and cut into independent items, one descriptor each), C1 lane DP (a lane per
// short item) beside C1b wave DP (a wave per long item, on the context's auxiliary stream), C2 ends (a wave
// per chunk: chain ends, traceback), C3 finalize (a wave per chunk: primary chains, MAPQ, statistics).
//
// Replaces spatial_index.cc:419-576 (+ TracebackChains :165-220, GeneratePrimaryChains
// :222-253, ComputeMAPQ :255-274) and the chain-0 statistics of sigmap.cc:714-729.
//
// The DP of one (reference, strand) bucket is sequential (score[i] depends on earlier
// scores and on the order-dependent num_skips early exit, spatial_index.cc:461-540), but an anchor whose
// predecessor scan meets no predecessor that passes the gap tests keeps its own start score whatever the
// others do (C0), and runs of unsettled anchors farther apart than max_target_gap_length never read each
// other's scores (C0b): the sequential part shrinks to short independent items.  The only
// cross-bucket coupling in the reference is the running `max_chaining_score`
// (spatial_index.cc:420,542-549): it never influences scores, only which anchors qualify
// as chain ends, so it is re-applied after the DP as (prefix max over earlier buckets) +
// (running max inside the bucket), which is the same sequence of values.
#include <float.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace smhip {
namespace {

struct Cand {
  float score;
  uint32_t idx;  // bucket-local anchor index
};

// spatial_index.cc:11-20: score descending, then index descending
__device__ __forceinline__ bool CandBefore(const Cand &a, const Cand &b) {
  if (a.score > b.score) return true;
  if (a.score == b.score) return a.idx > b.idx;
  return false;
}

// spatial_index.h:37-44 operator> on (score, num_anchors, direction, ref, start, end)
__device__ __forceinline__ bool ChainGreater(const ChainTmp &a, const ChainTmp &b) {
  if (a.score != b.score) return a.score > b.score;
  if (a.n_anchors != b.n_anchors) return a.n_anchors > b.n_anchors;
  if (a.direction != b.direction) return a.direction > b.direction;
  if (a.ref != b.ref) return a.ref > b.ref;
  if (a.start != b.start) return a.start > b.start;
  return a.end > b.end;
}

constexpr uint32_t kTile = 64;    // anchors per tile of the classify kernel: one per lane
constexpr uint32_t kRing = 128;   // its LDS ring per wave: the tile and the 64 anchors before it
constexpr uint32_t kDensePairs = 512;  // (anchor, predecessor) pairs of a tile the dense route takes; more: lockstep scans
// The classifier's window count is a bisection with steps kWindowStep0, kWindowStep0 / 2, .., 1: it saturates at
// kWindowSat = "that many or more".  The dense route is only valid for windows it has counted EXACTLY, so the largest
// window it may take (kDenseKMax, the cap of dense_k in LaunchChain) lies strictly below the saturation value; a pair's
// table entry packs {lane + 1 : 16 bits | first pair : 16 bits} and a passing pair's {lane : 8 | ring position : 8+}.
constexpr uint32_t kWindowStep0 = 16;
constexpr uint32_t kWindowSat = 2 * kWindowStep0 - 1;  // 31
constexpr uint32_t kDenseKMax = 30;
static_assert((kWindowStep0 & (kWindowStep0 - 1)) == 0, "the bisection halves its step down to 1");
static_assert(kDenseKMax < kWindowSat, "a saturated window count must never pass for a dense window");
static_assert(kWindowSat < kTile, "a window must fit the 64 anchors the ring holds before the tile");
static_assert(kDensePairs <= 0xFFFFu, "first-pair field of a pair-table entry is 16 bits");
static_assert(kDensePairs >= kTile, "the passing-pair list is written into the consumed part of the pair table");
static_assert(kRing == 2 * kTile && kRing <= 256, "ring positions relative to t0 - 64 fit the entry's 8-bit field");
constexpr int kPf = 4;            // predecessors fetched ahead of the sequential scan: one 16-byte load per column
static_assert(kPf == 4, "the lane DP fetches a block of predecessors as one uint4");
constexpr int kBigPendingMin = 24; // buckets with more unsettled anchors than this are chained by a whole wave ...
constexpr int kBigGroupCap = 512;  // ... staged in LDS when they fit, else through an LDS ring of that many anchors

// ---- C0: which anchors can be settled without the sequential DP?  One thread per anchor.
// The predecessor scan of anchor i (spatial_index.cc:461-540) only looks at predecessor SCORES
// when a predecessor passes the gap tests (then cur = score[j] + matching); every other evaluated
// predecessor yields cur = 0, which never beats the anchor's own positive start score and simply
// counts one skip.  So if the scan -- run on positions alone, counting skips -- reaches one of its
// exits (26 skips, target gap > 5000, band, bucket start) before meeting a gap-test-passing
// predecessor, the anchor's result is its start score with itself as predecessor, independent of
// any other anchor.  Those anchors ("isolated", ~94 % of a noise bucket) are finished here in
// parallel (score, predecessor, bucket maximum); the rest are flagged (used = 2) and listed in
// index order for the sequential kernels.
// One WAVE per TILE of 64 anchors (any chunk): the tile and the 64 anchors before it are staged in an LDS ring of
// the wave (the backward scans rarely leave that window; older predecessors are read from memory), so the
// parallelism follows the number of anchors, not of chunks -- a 250 Mb reference has 181 000 anchors per chunk
// in two buckets -- and no wave waits for another.
__device__ __forceinline__ uint32_t ChunkOfFlat(const uint32_t *__restrict__ off, uint32_t n_chunks, uint32_t x) {
  uint32_t lo = 0, hi = n_chunks;  // chunk c with off[c] <= x < off[c + 1]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void ChainTileCountKernel(ChainArgs a, uint32_t n_chunks, uint32_t *__restrict__ tiles) {
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += gridDim.x * 256) {
    const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - a.anchor_off[c]);
    tiles[c] = (n + kTile - 1u) / kTile;
    // bucket maxima of the chunk (positive floats order like their bit patterns): zeroed for the atomics below
    uint32_t *gmax_g = reinterpret_cast<uint32_t *>(a.scratch + 2 * a.anchor_off[c]);
    const uint32_t G = a.n_groups[c];
    for (uint32_t g = 0; g < G; ++g) gmax_g[g] = 0;
  }
}

#ifdef SMHIP_PHASE_PROF
__device__ unsigned long long g_phase_prof[32];  // (the PHASE_* macros of common.h)
#endif

// The gap tests of a predecessor that lies within max_target_gap_length on the target (spatial_index.cc:483-497):
// gap_length < max_gap_length && 0.75 < gap_scale < 5 with gap_scale = (float)qd / (float)td.  For
// 0 < td < 2^21 and 0 <= qd the two float comparisons are EXACTLY the integer ones, 4 qd > 3 td and qd < 5 td:
// near either bound qd < 2^24, so both operands convert exactly, and their correctly rounded quotient can
// only reach 5 (or 0.75) from the other side when it lies within half an ulp of it, i.e. when 1 / td < 2^-22
// (resp. 1 / (4 td) < 2^-25); a qd beyond 2^24 gives a quotient above 8 either way.  Checked against the float
// form for every td < 2^21 at and around both bounds, and for every (td, qd) with td <= 6000.  td is at most
// max_target_gap_length where this is called; kIntGap = false keeps the division for parameters beyond 2^21.
template <bool kIntGap>
__device__ __forceinline__ bool GapTestsPass(int32_t td, int32_t qd, int max_gap_length) {
  const int gap_length = abs(td - qd);
  // (0 < td < 2^21: 5 td and 3 td fit 32 bits; 4 qd > 3 td  <=>  qd > floor(3 td / 4))
  if (kIntGap) return gap_length < max_gap_length && qd < 5 * td && qd > ((3 * td) >> 2);
  const float gap_scale = td > 0 ? (float)qd / (float)td : 1.0f;
  return gap_length < max_gap_length && gap_scale < 5.0f && gap_scale > 0.75f;
}

// (float)(1 - 0.2 * (double)dist / (double)radius), spatial_index.cc:437-438 -- the start score's factor, wanted for 94 %
// of the anchors.  A double division costs ~30 instruction slots; x * RN(1 / R) is within 3 units of 2^-53 of x / R,
// so with |q| <= 4 |y| the difference 1 - q is within 34 ulp of the reference's y -- and the float result can only
// differ when y lies that close to the midpoint of two floats (the low 29 mantissa bits of the double = 0x10000000):
// those (2.4e-7 of the values), and everything outside the range where the bound holds, take the division.
__device__ __forceinline__ float StartCoef(float dist, double radius, double inv_radius) {
  const double x = 0.2 * (double)dist;
  const double q = x * inv_radius, y = 1.0 - q;
  const uint32_t low = (uint32_t)__double_as_longlong(y) & 0x1FFFFFFFu;
  const uint32_t off = low > 0x10000000u ? low - 0x10000000u : 0x10000000u - low;
  const double ay = fabs(y);
  const bool safe = fabs(q) <= 4.0 * ay && off > 64u && ay >= 0x1p-100 && ay <= 0x1p100;
  if (safe) return (float)y;
  return (float)(1 - x / radius);
}

__device__ __forceinline__ uint32_t WaveScanMaxU32(uint32_t v) {  // inclusive, like common.h's scans
#define SM_STEP(CTRL, ROWS) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false))
  SM_STEP(0x111, 0xF); SM_STEP(0x112, 0xF); SM_STEP(0x114, 0xF); SM_STEP(0x118, 0xF); SM_STEP(0x142, 0xA); SM_STEP(0x143, 0xC);
#undef SM_STEP
  return v;
}

template <bool kIntGap>
__global__ __launch_bounds__(256) void ChainClassifyKernel(ChainArgs a, uint32_t n_chunks,
                                                           const uint32_t *__restrict__ tile_off, float radius, int dim,
                                                           int max_gap_length, int max_target_gap_length,
                                                           int chaining_band_length, int max_num_skips,
                                                           uint32_t dense_k, uint32_t resolve) {
  // Every WAVE works on its own: no workgroup barrier anywhere (the 256-anchor tiles of the first version spent a
  // fifth of their time waiting at the barrier behind the wave with the longest backward scan).
  __shared__ uint32_t s_gmax_all[4][kTile + 2];  // maxima of the buckets the tile touches, relative to its first bucket
  __shared__ uint32_t s_head_all[4][kDensePairs];  // pair e -> the anchor whose window it belongs to (dense route); 0 between tiles
  __shared__ uint32_t s_found_all[4][kTile];       // ... and whether one of an anchor's pairs passed; 0 between tiles
  // chaining inside the classifier (below): an anchor's best passing predecessor {score bits, ring position}, whether
  // one of its passing predecessors is itself waiting for the DP; both 0 between tiles.  And the ring's fourth
  // column: an anchor's start-score factor.
  __shared__ unsigned long long s_best_all[4][kTile];
  __shared__ uint32_t s_unres_all[4][kTile];
  __shared__ float s_ring_c_all[4][kRing];
  // The ring as three 4-byte COLUMNS (target, query, bucket): a scan step reads target and query only -- whether a
  // predecessor lies in another bucket is an index comparison with the start of the anchor's own bucket -- i.e. two
  // ds_read_b32 whose lanes hit 64 different banks while they run in step, where the packed {target, query, bucket, -}
  // record of round 3 was one ds_read_b128: four passes through the banks, entries 16 apart on the same banks
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.53).
  __shared__ uint32_t s_ring_t_all[4][kRing], s_ring_q_all[4][kRing], s_ring_g_all[4][kRing];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const double inv_radius = 1.0 / (double)radius;
  uint32_t *s_gmax = s_gmax_all[wave], *s_head = s_head_all[wave], *s_found = s_found_all[wave];
  for (uint32_t e = lane; e < kDensePairs; e += 64) s_head[e] = 0;
  s_found[lane] = 0;
  unsigned long long *s_best = s_best_all[wave];
  uint32_t *s_unres = s_unres_all[wave];
  float *s_ring_c = s_ring_c_all[wave];
  s_best[lane] = 0ull;
  s_unres[lane] = 0;
  // which anchors of the PREVIOUS tile were left to the DP kernels (bit l: anchor t0 - 64 + l); all of them when the
  // previous tile was another wave's
  unsigned long long fmask_prev = 0ull;
  uint32_t *s_ring_t = s_ring_t_all[wave], *s_ring_q = s_ring_q_all[wave], *s_ring_g = s_ring_g_all[wave];
  // Every wave takes a contiguous range of the flat tile sequence (tiles of chunk 0, of chunk 1, ...): one
  // search for its first chunk, then tile after tile.  Inside a chunk the LDS ring (anchor p in slot p & 127)
  // already holds the previous tile, so a tile loads only its own 64 records.
  const uint32_t total_tiles = tile_off[n_chunks];
  const uint32_t n_waves = gridDim.x * 4u, gw = blockIdx.x * 4u + wave;
  const uint32_t per = (total_tiles + n_waves - 1) / n_waves;
  uint32_t tile = gw * per;
  const uint32_t tile_end = min(total_tiles, tile + per);
  if (tile >= tile_end) return;
  uint32_t c = ChunkOfFlat(tile_off, n_chunks, tile);
  bool ring_valid = false;  // the ring holds [t0 - 64, t0) of chunk c
  uint32_t ord_carry = 0;   // bucket ordinal of the last anchor of the previous tile (ring_valid)
  uint32_t bstart_carry = 0;  // ... and the index of that bucket's first anchor
  // the records (and distance) of the next tile of the same chunk are fetched into registers while the
  // current one is processed
  bool have_nx = false;
  uint32_t nx_t = 0, nx_q = 0, nx_g = 0;
  float nx_d = 0.0f;
  PHASE_DECL();
  for (; tile < tile_end; ++tile) {
    PHASE_INIT();
    while (tile >= tile_off[c + 1]) {  // next chunk (chunks without anchors have no tiles)
      ++c;
      ring_valid = false;
      have_nx = false;
    }
    const uint32_t t0 = (tile - tile_off[c]) * kTile;
    const uint64_t base = a.anchor_off[c];
    const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
    const uint32_t G = a.n_groups[c];
    const uint32_t *tpos = a.s_tpos + base, *qpos = a.s_qpos + base, *grp = a.s_grp + base;
    const uint32_t *gstart = a.group_start + base;
    const float *dist = a.s_dist + base;
    float *score = a.score + base;
    uint32_t *pred = a.pred + base;
    uint8_t *used = a.used + base;
    uint32_t *gmax_g = reinterpret_cast<uint32_t *>(a.scratch + 2 * base);
    const uint32_t w0 = t0 >= kTile ? t0 - kTile : 0;
    const uint32_t ai = t0 + lane;
    float my_dist = 0.0f;
    if (ai < n) {
      if (!have_nx) {
        nx_t = tpos[ai];
        nx_q = qpos[ai];
        nx_g = grp[ai];
        nx_d = dist[ai];
      }
      s_ring_t[ai & (kRing - 1)] = nx_t;
      s_ring_q[ai & (kRing - 1)] = nx_q;
      s_ring_g[ai & (kRing - 1)] = nx_g;
      my_dist = nx_d;
    }
    const float my_coef = StartCoef(my_dist, (double)radius, inv_radius);
    if (ai < n) s_ring_c[ai & (kRing - 1)] = my_coef;
    const uint32_t my_t = nx_t, my_q = nx_q, my_g = nx_g;
    have_nx = tile + 1 < tile_end && tile + 1 < tile_off[c + 1];
    if (have_nx) {
      const uint32_t gi = ai + kTile;
      if (gi < n) {
        nx_t = tpos[gi];
        nx_q = qpos[gi];
        nx_g = grp[gi];
        nx_d = dist[gi];
      }
    }
    if (!ring_valid && t0 > 0) {  // first tile of this wave in the middle of a chunk
      const uint32_t p = w0 + lane;
      s_ring_t[p & (kRing - 1)] = tpos[p];
      s_ring_q[p & (kRing - 1)] = qpos[p];
      s_ring_g[p & (kRing - 1)] = grp[p];
    }
    if (!ring_valid) fmask_prev = t0 > 0 ? ~0ull : 0ull;  // (another wave's tile: nothing is known about its anchors)
    s_gmax[lane] = 0;
    if (lane < 2) s_gmax[kTile + lane] = 0;
    // bucket ordinal of the tile's first anchor: carried over from the previous tile, or one search
    // (last bucket start <= t0); the others follow from the bucket changes inside the tile
    uint32_t first_ord = ord_carry;
    if (!ring_valid) {
      uint32_t lo = 0, hi = G;  // (wave-uniform)
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (gstart[mid] <= t0) lo = mid; else hi = mid;
      }
      first_ord = lo;
      bstart_carry = G ? gstart[lo] : 0u;  // the bucket that holds anchor t0 starts there
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    PHASE_MARK(0);
    uint32_t ord = 0, ord_last = 0;
    uint32_t bstart = 0;  // index of the first anchor of this anchor's bucket: a predecessor below it is another bucket's
    {
      // (with a valid ring the anchor before the tile is in LDS too: its bucket starts the count)
      const bool change = ai < n && (ai > t0 || ring_valid) && my_g != s_ring_g[(ai - 1) & (kRing - 1)];
      const unsigned long long m = __ballot(change);
      const unsigned long long upto = m & ((2ull << lane) - 1ull);        // changes up to and including ai
      ord = first_ord + (uint32_t)__popcll(upto);
      ord_last = first_ord + (uint32_t)__popcll(m);                       // ordinal of the tile's last anchor
      first_ord += (uint32_t)(m & 1ull);                                  // ordinal of the tile's first anchor
      bstart = upto ? t0 + (63u - (uint32_t)__clzll((long long)upto)) : bstart_carry;
      if (m) bstart_carry = t0 + (63u - (uint32_t)__clzll((long long)m));  // the bucket of the tile's last anchor
    }
    PHASE_MARK(1);
    const int32_t ct = (int32_t)my_t, cq = (int32_t)my_q;
    bool found = false;
    bool settled = false;  // found, and chained here
    float settled_score = 0.0f;
    uint32_t settled_pred = 0;
    // ---- WINDOWS.  The predecessors an anchor's scan can evaluate are the anchors of its bucket before it that are
    // not beyond the target gap: a contiguous window [lo, ai) (targets ascend inside a bucket).  While a window holds
    // no more than max_num_skips + 1 (and chaining_band_length) anchors the scan cannot end early except by finding a
    // predecessor that passes: `found` is then the OR over the window, in any order -- so the pairs (anchor,
    // predecessor) of the tile are laid out densely over the lanes (1.8 per anchor at the default parameters, where
    // the lockstep scans run as long as the longest of 64 windows: 13 predecessors for 3 useful ones per lane).
    uint32_t k = 0;
    if (dense_k) {
      // k = how many of ai - 1, ai - 2, ... are near: bisection on the count (five ring reads; 31 = "31 or more")
      const uint32_t dmax = (ai < n && ai > 0) ? min(ai - max(bstart, w0), kWindowSat) : 0u;
      // (round 5) ... and not beyond the target gap at which a predecessor can still PASS: the gap-scale test wants
      // 4 qd > 3 td, and qd = cq - pq <= cq (query positions are unsigned), so a predecessor with td >= 4 cq / 3 fails
      // whatever its query position.  Predecessors beyond that only ever count skips, after every one that could pass
      // has been looked at (the scan goes from the nearest outwards), so `found` -- and the chaining pass below, which
      // needs every PASSING predecessor -- are the same over this shorter window, which is the one that has to fit
      // dense_k.  A read's first chunk has query positions below ~450: the window shrinks from 5 000 to <= 600 targets
      // and the pairs of a noise tile from ~118 to ~15.  (Integer gap tests only: the float form keeps the full window.)
      const int32_t reach = kIntGap ? min(max_target_gap_length, (int32_t)min((uint32_t)cq, 0x1FFFFFFFu) / 3 * 4 + 4) : max_target_gap_length;
#pragma unroll
      for (uint32_t step = kWindowStep0; step >= 1; step >>= 1) {
        const uint32_t cand = k + step;
        const bool near = (int32_t)s_ring_t[(ai - cand) & (kRing - 1)] + reach >= ct;
        if (cand <= dmax && near) k = cand;
      }
    }
    PHASE_MARK(6);
    const uint32_t s_incl = (uint32_t)WaveScanAdd((int)k);
    const uint32_t n_pairs = (uint32_t)__builtin_amdgcn_readlane((int)s_incl, 63);
    if (dense_k && !__ballot(k > dense_k) && n_pairs <= kDensePairs) {
      const uint32_t s_mine = s_incl - k;  // my pairs are [s_mine, s_mine + k): predecessors ai - 1, ai - 2, ...
#ifdef SMHIP_PHASE_PROF
      if (threadIdx.x == 0) {
        atomicAdd(&g_phase_prof[20], 1ull);
        atomicAdd(&g_phase_prof[22], (unsigned long long)((n_pairs + 63) / 64));
        atomicAdd(&g_phase_prof[23], (unsigned long long)n_pairs);
      }
#endif
      if (k) s_head[s_mine] = ((lane + 1u) << 16) | s_mine;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      uint32_t carry = 0;
      uint32_t n_pass = 0;  // passing pairs so far (wave-uniform): listed in the part of the pair table already consumed
      for (uint32_t e0 = 0; e0 < n_pairs; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t h = s_head[e];
        s_head[e] = 0;
        h = max(WaveScanMaxU32(h), carry);  // the last head at or before e
        carry = (uint32_t)__builtin_amdgcn_readlane((int)h, 63);
        const uint32_t il = (h >> 16) - 1u, ia = t0 + il, pj = ia - 1u - (e - (h & 0xFFFFu));
        const int32_t it_ = (int32_t)s_ring_t[ia & (kRing - 1)], iq = (int32_t)s_ring_q[ia & (kRing - 1)];
        const int32_t pt = (int32_t)s_ring_t[pj & (kRing - 1)], pq = (int32_t)s_ring_q[pj & (kRing - 1)];
        const int32_t td = it_ - pt, qd = iq - pq;
        const bool pass = e < n_pairs && pq != iq && pt != it_ && qd >= 0 && GapTestsPass<kIntGap>(td, qd, max_gap_length);
        const unsigned long long pm = __ballot(pass);
        if (pass) {  // (few do)
          s_found[il & 63u] = 1u;
          // {anchor's lane, predecessor's position in [t0 - 64, t0 + 64)}: at most e0 + 64 pairs have been looked at,
          // so the slot lies in the part of the table this loop has read and cleared
          if (resolve) s_head[n_pass + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = (il & 63u) | ((pj + 64u - t0) << 8);
        }
        n_pass += (uint32_t)__popcll(pm);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      found = s_found[lane] != 0;
      s_found[lane] = 0;
      // ---- CHAINING HERE what does not need the DP kernels.  The DP of an anchor (spatial_index.cc:436-541) takes the
      // best of its own start score and, over the predecessors that pass, their score + the matching term -- the
      // first one met (the nearest) among equals, and only a strict improvement replaces the start score --; failing
      // predecessors only count skips, and inside a window of <= max_num_skips + 1 the skip limit cannot end the
      // scan.  So when every PASSING predecessor of an anchor is itself isolated (its score is its start score, known
      // here), the anchor's score and predecessor are a maximum over its passing pairs: taken with one 64-bit LDS
      // atomic per pair, {score bits | predecessor position} (scores are positive).  Two thirds of the anchors with a
      // passing predecessor are settled this way; an anchor with a passing predecessor that waits for the DP (this
      // tile's or the previous one's found mask; anything about another wave's tile counts as waiting) waits too.
      const unsigned long long fcur = __ballot(found);
      if (resolve && n_pass) {
        for (uint32_t b = 0; b < n_pass; b += 64) {
          const uint32_t idx = b + lane;
          const uint32_t ent = s_head[idx];
          s_head[idx] = 0;
          const uint32_t il = ent & 63u, rel = ent >> 8;  // (lanes past the list read 0: harmless addresses)
          const bool waits = rel < 64u ? ((fmask_prev >> rel) & 1ull) != 0 : ((fcur >> (rel & 63u)) & 1ull) != 0;
          const uint32_t ia = t0 + il, pj = t0 + rel - 64u;
          const int32_t it_ = (int32_t)s_ring_t[ia & (kRing - 1)], iq = (int32_t)s_ring_q[ia & (kRing - 1)];
          const int32_t pt = (int32_t)s_ring_t[pj & (kRing - 1)], pq = (int32_t)s_ring_q[pj & (kRing - 1)];
          const float ci = s_ring_c[ia & (kRing - 1)], cj = s_ring_c[pj & (kRing - 1)];
          const int32_t m = min(min(it_ - pt, iq - pq), dim);
          const float curv = cj * (float)dim + (float)m * ci;  // ps[kk] + matching
          if (idx < n_pass) {
            if (waits || !(curv > 0.0f)) s_unres[il] = 1u;
            else atomicMax(&s_best[il], ((unsigned long long)__float_as_uint(curv) << 32) | rel);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        if (found && s_unres[lane] == 0) {
          const unsigned long long bb = s_best[lane];
          const float sc0 = my_coef * (float)dim;
          if (sc0 > 0.0f) {  // (the maximum above compared score bits: positive scores only)
            const float curv = __uint_as_float((uint32_t)(bb >> 32));
            settled = true;
            settled_score = curv > sc0 ? curv : sc0;
            settled_pred = curv > sc0 ? t0 + ((uint32_t)bb & 0xFFu) - 64u : ai;
          }
        }
        s_unres[lane] = 0;
        s_best[lane] = 0ull;
      }
      PHASE_MARK(7);
    } else {
#ifdef SMHIP_PHASE_PROF
      if (threadIdx.x == 0) atomicAdd(&g_phase_prof[21], 1ull);
#endif
      // ---- the backward scans of the tile's 64 anchors, in lockstep and WITHOUT BRANCHES while the predecessors
      // are in the ring: a step is one 16-byte LDS read and ~25 predicated VALU operations for all lanes (the
      // per-lane branches of the sequential form cost three times that per step).  `found`: a predecessor passed
      // the gap tests (the anchor is not isolated); `active` ends at a bucket start, beyond the target gap, after
      // max_num_skips + 1 failed candidates, with `found`, or when the band is used up.
      bool active = ai < n && ai > 0;
      uint32_t pj = ai;  // the next predecessor is pj - 1
      int32_t skips = 0;
      uint32_t steps = 0;
      // (two predecessors per round: both ring reads are issued before either is looked at -- the second one's address
      // does not depend on the first one's outcome, only whether it counts does -- so the ballot, the branch and the
      // LDS round trip are paid once per pair)
      auto step1 = [&](uint32_t r_t, uint32_t r_q) {
        const bool go = active && pj > w0 && steps < (uint32_t)chaining_band_length;
        const int32_t pt = (int32_t)r_t, pq = (int32_t)r_q;
        const bool other_bucket = pj - 1u < bstart;  // (go: pj > w0 >= 0)
        const bool same_pos = pq == cq || pt == ct;                     // `continue`: not evaluated, no skip counted
        const bool far = pt + max_target_gap_length < ct;               // `break`
        const int32_t td = ct - pt, qd = cq - pq;
        const bool evaluated = !other_bucket && !same_pos && !far && qd >= 0;
        const bool pass = evaluated && GapTestsPass<kIntGap>(td, qd, max_gap_length);
        const bool fail = evaluated && !pass;
        if (go) {
          found = pass;
          skips += fail ? 1 : 0;
          // (a far predecessor ends the scan also when it shares the query position: every older predecessor of
          // the bucket is farther still, so the reference's loop `continue`s or `break`s its way out of the bucket
          // without evaluating anything)
          active = !(other_bucket || far || pass || skips > max_num_skips);
          --pj;
          ++steps;
        }
      };
      for (;;) {
        const bool go = active && pj > w0 && steps < (uint32_t)chaining_band_length;
        if (!__ballot(go)) break;
        const uint32_t t1 = s_ring_t[(pj - 1u) & (kRing - 1)], q1 = s_ring_q[(pj - 1u) & (kRing - 1)];
        const uint32_t t2 = s_ring_t[(pj - 2u) & (kRing - 1)], q2 = s_ring_q[(pj - 2u) & (kRing - 1)];  // (counts only if the
        step1(t1, q1);                                                                                 // first one leaves the
        step1(t2, q2);                                                                                 // lane active)
      }
      // lanes whose scan leaves the ring (64+ predecessors back: dense true-locus regions): the sequential form,
      // predecessors from memory
      if (active && pj > 0 && pj <= w0 && steps < (uint32_t)chaining_band_length) {
        for (; pj-- > 0 && steps < (uint32_t)chaining_band_length; ++steps) {
          if (pj < bstart) break;  // bucket start
          const int32_t pt = (int32_t)tpos[pj], pq = (int32_t)qpos[pj];
          if (pq == cq || pt == ct) continue;
          if (pt + max_target_gap_length < ct) break;
          const int32_t td = ct - pt, qd = cq - pq;
          if (qd < 0) continue;
          if (GapTestsPass<kIntGap>(td, qd, max_gap_length)) {
            found = true;
            break;
          }
          if (++skips > max_num_skips) break;
        }
      }
    }
    PHASE_MARK(2);
    fmask_prev = __ballot(found);  // (for the next tile: an anchor settled here still counts -- its score is not its start score)
    uint32_t sc_bits = 0;  // score bits of an anchor finished here (positive floats order like their bits); 0: none
    if (ai < n) {
      if (!found || settled) {
        const float sc = settled ? settled_score : my_coef * (float)dim;
        score[ai] = sc;
        pred[ai] = settled ? settled_pred : ai;
        // (chained here: listed by ChainPendingKernel for the end search -- its score can reach min_chaining_score --
        // but not for the DP; that kernel clears the flag)
        used[ai] = settled ? 3 : 0;
        sc_bits = __float_as_uint(sc);
      } else {
        used[ai] = 2;  // listed for the sequential kernels by ChainPendingKernel
        pred[ai] = ord;  // (until the DP writes the predecessor: the anchor's bucket ordinal, which the DP needs)
      }
    }
    // bucket maxima of the tile.  Most tiles lie inside ONE bucket: 64 lanes raising the same LDS word were 64
    // serialised passes of the atomic unit (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.39-0.40 for this kernel,
    // profiles/r4_*_pmc_per_launch.csv); the maximum of the wave is six DPP steps.
    if (ord_last == first_ord) {  // (wave-uniform)
      const uint32_t mx = (uint32_t)__builtin_amdgcn_readlane((int)WaveScanMaxU32(sc_bits), 63);
      if (lane == 0) s_gmax[0] = mx;
    } else if (sc_bits) {
      atomicMax(&s_gmax[ord - first_ord], sc_bits);  // (a tile of 64 anchors touches <= 64 buckets)
    }
    PHASE_MARK(3);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    PHASE_MARK(4);
    if (lane <= ord_last - first_ord && s_gmax[lane] != 0) atomicMax(&gmax_g[first_ord + lane], s_gmax[lane]);
    ord_carry = ord_last;
    ring_valid = true;
    PHASE_MARK(5);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();  // the ring half and the maxima are free for the next tile
  }
  PHASE_FLUSH();
}

// The anchors that wait for the DP kernels, listed per chunk in index order: in the free half of the chunk's region of
// the sort's key buffer (16 bytes per anchor: bucket maxima and their prefix maxima take the first 8).
__device__ __forceinline__ uint32_t *DpList(const ChainArgs &a, uint64_t base, uint32_t n) {
  return reinterpret_cast<uint32_t *>(a.scratch + 2 * base) + 2 * (size_t)n;
}

// ---- C0b: the anchors left for the sequential DP, listed per chunk in index order (one workgroup per
// chunk: an ordered compaction of the flags), and cut into ITEMS: maximal runs of listed anchors of one
// bucket in which consecutive ones lie within max_target_gap_length on the target.  The predecessor
// scan of an anchor stops at the first predecessor farther than that (spatial_index.cc:476-479, targets
// are sorted), so an anchor never reads the score of a listed anchor of an earlier item: items are
// independent DP problems.  A noise bucket of a large reference falls into thousands of items of two or
// three anchors; the true locus of a read is one long item.
__global__ __launch_bounds__(256) void ChainPendingKernel(ChainArgs a, int max_target_gap_length,
                                                          uint32_t *__restrict__ n_pending,
                                                          uint32_t *__restrict__ n_items) {
  __shared__ uint32_t s_wave[4], s_wave8[8][4], s_wave8d[8][4];
  __shared__ uint32_t s_run, s_run_d;
  const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t base = a.anchor_off[c];
  const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
  const uint32_t *tpos = a.s_tpos + base, *grp = a.s_grp + base;
  uint8_t *used = a.used + base;
  // TWO lists, both in index order: `scored` = every anchor with a passing predecessor (flags 2 and 3: what the end
  // search has to look at -- an isolated anchor's score is below min_chaining_score), and `pending` = those of them
  // that wait for the DP kernels (flag 2; 3 = chained by the classifier: the flag is cleared here)
  uint32_t *scored = a.pending + base;
  uint32_t *pending = DpList(a, base, n);
  uint32_t *items = a.lists + base;  // (ChainEndsKernel reuses this array for the chains' anchor lists afterwards)
  if (tid == 0) s_run = s_run_d = 0;
  __syncthreads();
  // 8192 flags per round: a thread takes one aligned word (four consecutive anchors) of each of eight
  // 1024-flag rows; the eight loads are in flight together and the eight rows are scanned side by side, so a
  // round costs one memory round trip and three barriers (a row at a time: nine dependent rounds for a 12 Mb
  // chunk, twenty-three for a 32 Mb one)
  const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(used) & 3u);  // flags are read as aligned words
  const uint32_t *used_w = reinterpret_cast<const uint32_t *>(used - shift);
  const uint32_t n_words = (n + shift + 3u) / 4u;
  for (uint32_t w0 = 0; w0 < n_words; w0 += 8 * 256) {
    uint32_t f[8], fd[8], cnt[8], incl[8], cntd[8], incld[8];
    bool any_d = false;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint32_t wi = w0 + (uint32_t)p * 256u + tid;
      const uint32_t w4 = wi < n_words ? used_w[wi] : 0u;
      const int64_t i0 = (int64_t)wi * 4 - (int64_t)shift;  // anchor of the word's first byte
      // (flags are 0, 2 or 3: bit 1 = listed, bit 0 = chained already)
      uint32_t bits = ((w4 >> 1) & 1u) | ((w4 >> 8) & 2u) | ((w4 >> 15) & 4u) | ((w4 >> 22) & 8u);
      const uint32_t done = (w4 & 1u) | ((w4 >> 7) & 2u) | ((w4 >> 14) & 4u) | ((w4 >> 21) & 8u);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + k < 0 || i0 + k >= (int64_t)n) bits &= ~(1u << k);  // bytes before the first / after the last anchor
      f[p] = bits;
      fd[p] = bits & ~done;
      cnt[p] = (uint32_t)__popc(bits);
      incl[p] = cnt[p];
      cntd[p] = (uint32_t)__popc(fd[p]);
      incld[p] = cntd[p];
      any_d = any_d || (bits & done) != 0;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      incl[p] = (uint32_t)WaveScanAdd((int)incl[p]);
      incld[p] = (uint32_t)WaveScanAdd((int)incld[p]);
    }
    if (lane == 63) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        s_wave8[p][wave] = incl[p];
        s_wave8d[p][wave] = incld[p];
      }
    }
    __syncthreads();
    uint32_t row_base = s_run, total = 0, row_base_d = s_run_d, total_d = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      uint32_t off = row_base + incl[p] - cnt[p], offd = row_base_d + incld[p] - cntd[p];
      for (uint32_t w2 = 0; w2 < wave; ++w2) {
        off += s_wave8[p][w2];
        offd += s_wave8d[p][w2];
      }
      const uint32_t row_total = s_wave8[p][0] + s_wave8[p][1] + s_wave8[p][2] + s_wave8[p][3];
      const uint32_t row_total_d = s_wave8d[p][0] + s_wave8d[p][1] + s_wave8d[p][2] + s_wave8d[p][3];
      const uint32_t i0 = (w0 + (uint32_t)p * 256u + tid) * 4u - shift;  // (only used where a flag is set: then in range)
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        if (f[p] & (1u << k)) scored[off++] = i0 + k;
        if (fd[p] & (1u << k)) pending[offd++] = i0 + k;
        else if (any_d && (f[p] & (1u << k))) used[i0 + k] = 0;  // chained by the classifier: an ordinary anchor from here on
      }
      row_base += row_total;
      total += row_total;
      row_base_d += row_total_d;
      total_d += row_total_d;
    }
    __syncthreads();
    if (tid == 0) {
      s_run += total;
      s_run_d += total_d;
    }
    __syncthreads();
  }
  const uint32_t npend = s_run_d;  // (of the DP list: what the items below are cut from)
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    n_pending[c] = s_run;  // (of the scored list: ChainEndsKernel's)
    s_run = 0;
  }
  __syncthreads();
  for (uint32_t p0 = 0; p0 < npend; p0 += 256) {
    const uint32_t pi = p0 + tid;
    bool start = false;
    if (pi < npend) {
      if (pi == 0) {
        start = true;
      } else {
        const uint32_t x = pending[pi - 1], y = pending[pi];
        start = grp[x] != grp[y] || tpos[x] + (uint32_t)max_target_gap_length < tpos[y];
      }
    }
    const unsigned long long m = __ballot(start);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = s_run;
    for (uint32_t w2 = 0; w2 < wave; ++w2) off += s_wave[w2];
    if (start) items[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pi;
    __syncthreads();
    if (tid == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  // ---- one DESCRIPTOR per item, so that the DP kernel finds everything about an item with one load instead of
  // a chain of dependent ones (item bounds, first anchor, its bucket): lo = chunk | first pending index << 32,
  // hi = pending anchors | bucket ordinal << 32.  An item's first listed anchor has an unlisted predecessor that
  // no other item shares, so a chunk of n anchors has at most n / 2 items: chunk c owns the entries
  // [anchor_off[c] / 2, anchor_off[c + 1] / 2) of the descriptor arrays (one 8-byte word per two anchors each) --
  // no atomics (one atomic per workgroup on a common counter cost 0.13 ms: 2 048 workgroups arrive together).
  const uint32_t n_it = s_run;
  const uint32_t *pred = a.pred + base;
  uint64_t *d_lo = a.item_lo + base / 2, *d_hi = a.item_hi + base / 2;
  __syncthreads();
  for (uint32_t k = tid; k < n_it; k += 256) {
    const uint32_t p0 = items[k];
    const uint32_t npl = (k + 1 < n_it ? items[k + 1] : npend) - p0;
    const uint32_t g = pred[pending[p0]];  // the bucket's ordinal, left there by the classify kernel
    d_lo[k] = (uint64_t)c | ((uint64_t)p0 << 32);
    d_hi[k] = (uint64_t)npl | ((uint64_t)g << 32);
  }
  if (tid == 0) n_items[c] = n_it;
}

// ---- C1: chaining DP (spatial_index.cc:436-541), ONE LANE PER ITEM, flattened over the whole batch.
// Predecessor records are fetched kPf at a time so the dependent chain sees one memory latency
// per block instead of one per predecessor.
template <bool kIntGap>
__device__ __forceinline__ void DpSmallItems(const ChainArgs &a, const uint32_t *__restrict__ ioff, uint32_t n_chunks,
                                             uint32_t first_block, uint32_t n_blocks, uint32_t (*s_ids)[64], float radius,
                                             int dim, int max_gap_length, int max_target_gap_length,
                                             int chaining_band_length, int max_num_skips) {
  const uint32_t total_items = ioff[n_chunks];
  PHASE_DECL();
  // grid-stride over items (their total is only known on the device)
  for (uint32_t gi = (blockIdx.x - first_block) * 64 + threadIdx.x; gi < total_items; gi += n_blocks * 64) {
  PHASE_INIT();
  // chunk of the item: chunks hold similar numbers of items, so a proportional guess and a few steps either way
  // replace the 14 dependent loads of a binary search (which remains for a guess that is far off)
  uint32_t c = (uint32_t)(((uint64_t)gi * n_chunks) / total_items);
  {
    uint32_t steps = 0;
    while (c > 0 && ioff[c] > gi && steps < 8) { --c; ++steps; }
    while (c + 1 < n_chunks && ioff[c + 1] <= gi && steps < 8) { ++c; ++steps; }
    if (ioff[c] > gi || ioff[c + 1] <= gi) c = ChunkOfFlat(ioff, n_chunks, gi);
  }
  const uint64_t base = a.anchor_off[c];
  const uint64_t d_hi = a.item_hi[base / 2 + (gi - ioff[c])], d_lo = a.item_lo[base / 2 + (gi - ioff[c])];  // (ChainPendingKernel)
  const uint32_t p0 = (uint32_t)(d_lo >> 32), npl = (uint32_t)d_hi, g = (uint32_t)(d_hi >> 32);
  if (npl > (uint32_t)kBigPendingMin) continue;  // a long item (the true locus of a read): chained by a whole wave
  const uint32_t *tpos = a.s_tpos + base, *qpos = a.s_qpos + base;
  const float *dist = a.s_dist + base;
  const uint32_t *gstart = a.group_start + base;
  float *score = a.score + base;
  uint32_t *pred = a.pred + base;
  uint8_t *used = a.used + base;
  uint32_t *gmax = reinterpret_cast<uint32_t *>(a.scratch + 2 * base);
  const uint32_t *pending = DpList(a, base, (uint32_t)(a.anchor_off[c + 1] - base));
  const uint32_t s = gstart[g];
  PHASE_MARK(8);
  // The anchors of the slice are chained in order (each may read the scores of the ones before),
  // but everything an anchor needs EXCEPT those scores -- its id, its own record, the positions of
  // its first kPf predecessors -- is fetched while the previous anchor is being chained, so an
  // anchor costs one memory round trip (the scores) instead of four dependent ones.
  // (npl <= kBigPendingMin; most items hold one or two anchors: 1.14 on average)
  for (uint32_t k = 0; k < npl; ++k) s_ids[k][threadIdx.x] = pending[p0 + k];
  struct Pre {
    uint32_t ai;
    float d;
    int32_t ct, cq, pt[kPf], pq[kPf];
  };
  auto issue = [&](uint32_t k, Pre &P) {
    const uint32_t ai = s_ids[k][threadIdx.x];
    P.ai = ai;
    P.d = dist[ai];
    P.ct = (int32_t)tpos[ai];
    P.cq = (int32_t)qpos[ai];
    const int32_t aidx = (int32_t)(ai - s);
    const int32_t sidx = aidx > chaining_band_length ? aidx - chaining_band_length : 0;
    // (the kernel is bound by the NUMBER of scattered requests - a lane per item, items anywhere in the chunk:
    // the kPf predecessors of an anchor are its neighbours in the sorted columns, so one unaligned 16-byte load
    // per column fetches them all; near the bucket's start the clamped single loads remain)
    if (aidx - kPf >= sidx) {
      const uint4 vt = *reinterpret_cast<const uint4 *>(tpos + s + aidx - kPf);
      const uint4 vq = *reinterpret_cast<const uint4 *>(qpos + s + aidx - kPf);
      P.pt[0] = (int32_t)vt.w; P.pt[1] = (int32_t)vt.z; P.pt[2] = (int32_t)vt.y; P.pt[3] = (int32_t)vt.x;
      P.pq[0] = (int32_t)vq.w; P.pq[1] = (int32_t)vq.z; P.pq[2] = (int32_t)vq.y; P.pq[3] = (int32_t)vq.x;
    } else {
#pragma unroll
      for (int j = 0; j < kPf; ++j) {
        const int32_t pj = max(max(aidx - 1 - j, sidx), 0);  // clamped: always a valid address
        P.pt[j] = (int32_t)tpos[s + pj];
        P.pq[j] = (int32_t)qpos[s + pj];
      }
    }
  };
#ifdef SMHIP_PHASE_PROF  // lane utilisation of the lane-per-item DP: sum of item lengths / (64 x the longest of the wave)
  {
    uint32_t mx = npl;
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    atomicAdd(&g_phase_prof[16], (unsigned long long)npl);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_phase_prof[17], (unsigned long long)mx * 64ull);
    atomicAdd(&g_phase_prof[18], 1ull);
  }
#endif
  PHASE_MARK(9);
  float gm = 0.0f;
  Pre cur{}, nxt{};
  if (npl > 0) issue(0, cur);
  for (uint32_t k = 0; k < npl; ++k) {
   {
    if (k + 1 < npl) issue(k + 1, nxt);
    const uint32_t ai = cur.ai;
    const uint32_t anchor_index = ai - s;
    const float coef = (float)(1 - 0.2 * (double)cur.d / (double)radius);
    float best = coef * (float)dim;
    uint32_t best_pred = anchor_index;
    const int32_t ct = cur.ct, cq = cur.cq;
    int32_t start_idx = 0;
    if (anchor_index > (uint32_t)chaining_band_length) start_idx = (int32_t)anchor_index - chaining_band_length;
    int32_t num_skips = 0;
    bool done = false;
    bool first = true;
    for (int32_t pj0 = (int32_t)anchor_index - 1; pj0 >= start_idx && !done; pj0 -= kPf) {
      int32_t pt[kPf], pq[kPf];
      float ps[kPf];
      if (pj0 - (kPf - 1) >= start_idx) {  // a whole block inside the bucket / band: one 16-byte load per column
        const uint4 vs = *reinterpret_cast<const uint4 *>(score + s + pj0 - (kPf - 1));
        ps[0] = __uint_as_float(vs.w); ps[1] = __uint_as_float(vs.z); ps[2] = __uint_as_float(vs.y); ps[3] = __uint_as_float(vs.x);
        if (first) {
#pragma unroll
          for (int kk = 0; kk < kPf; ++kk) {
            pt[kk] = cur.pt[kk];
            pq[kk] = cur.pq[kk];
          }
        } else {
          const uint4 vt = *reinterpret_cast<const uint4 *>(tpos + s + pj0 - (kPf - 1));
          const uint4 vq = *reinterpret_cast<const uint4 *>(qpos + s + pj0 - (kPf - 1));
          pt[0] = (int32_t)vt.w; pt[1] = (int32_t)vt.z; pt[2] = (int32_t)vt.y; pt[3] = (int32_t)vt.x;
          pq[0] = (int32_t)vq.w; pq[1] = (int32_t)vq.z; pq[2] = (int32_t)vq.y; pq[3] = (int32_t)vq.x;
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < kPf; ++kk) {
          // unconditional loads from a clamped index: a conditional load compiles to a branch with
          // its own s_waitcnt, i.e. one memory round trip per predecessor instead of one per block
          const int32_t pj = max(pj0 - kk, start_idx);
          if (first) {
            pt[kk] = cur.pt[kk];
            pq[kk] = cur.pq[kk];
          } else {
            pt[kk] = (int32_t)tpos[s + pj];
            pq[kk] = (int32_t)qpos[s + pj];
          }
          ps[kk] = score[s + pj];
        }
      }
      first = false;
#pragma unroll
      for (int kk = 0; kk < kPf; ++kk) {
        const int32_t pj = pj0 - kk;
        if (done || pj < start_idx) break;
        if (pq[kk] == cq) continue;
        if (pt[kk] == ct) continue;
        if (pt[kk] + max_target_gap_length < ct) {
          done = true;
          break;
        }
        const int32_t td = ct - pt[kk];
        const int32_t qd = cq - pq[kk];
        float curv = 0;
        if (qd < 0) continue;
        {
          const int32_t m = min(min(td, qd), dim);
          const float matching = (float)m * coef;
          if (GapTestsPass<kIntGap>(td, qd, max_gap_length)) curv = ps[kk] + matching;
        }
        if (curv > best) {
          best = curv;
          best_pred = (uint32_t)pj;
          --num_skips;
        } else {
          ++num_skips;
          if (num_skips > max_num_skips) {
            done = true;
            break;
          }
        }
      }
    }
    score[ai] = best;
    pred[ai] = s + best_pred;  // chunk-local absolute index
    used[ai] = 0;
    gm = fmaxf(gm, best);
    cur = nxt;
   }
  }
  PHASE_MARK(10);
  atomicMax(&gmax[g], __float_as_uint(gm));  // (several items of a bucket run at the same time; scores are positive)
  PHASE_MARK(11);
  }
  PHASE_FLUSH();
}

// ---- C1b: the same DP for LONG buckets (the true locus of a read), one WAVE per bucket with the
// bucket staged in LDS.  The predecessor scan of one anchor is evaluated 64 predecessors at a
// time; the reference's sequential rules -- running best, `num_skips` +-1 per evaluated
// predecessor, exit when num_skips > 25, exit at the first predecessor farther than 5000 on the
// target (spatial_index.cc:461-540) -- are reproduced with an exclusive prefix max (running
// best), a prefix sum (num_skips) and ballots (first exit), which yields exactly the state the
// sequential loop would have at its exit.
template <bool kLds, bool kIntGap>
__device__ __forceinline__ void DpBigBucket(uint32_t lane, uint32_t ng, uint32_t s, const uint32_t *__restrict__ pend,
                                            uint32_t npend, int32_t *l_t, int32_t *l_q, float *l_s, float *l_d,
                                            const uint32_t *tpos, const uint32_t *qpos, const float *dist,
                                            float *score, uint32_t *pred, uint8_t *used, uint32_t *gmax_slot,
                                            float radius, int dim,
                                            int max_gap_length, int max_target_gap_length, int chaining_band_length,
                                            int max_num_skips) {
  float gm = 0.0f;
  // kLds: the whole bucket is staged (index = bucket-local anchor).  Otherwise the LDS arrays are a
  // RING over the bucket (slot = anchor & (kBigGroupCap - 1)) that follows the unsettled anchors:
  // before anchor ai is chained, the anchors up to ai that are not in the ring yet are staged, so
  // the kBigGroupCap anchors ending at ai -- where a predecessor scan normally stays -- are read
  // from LDS (with the scores of earlier unsettled anchors as they were computed); older
  // predecessors come from memory.
  constexpr uint32_t kMask = (uint32_t)kBigGroupCap - 1u;
  uint32_t staged = 0;
  // What an anchor needs before its scan can start -- its index, its own record, its start score (a double
  // division) -- does not depend on the anchors before it: lane k fetches and computes it for anchor pi0 + k of
  // the next 64, and the sequential loop takes it from there with a scalar lane select.  (One dependent global
  // load + an LDS read + the division per anchor were half of an anchor's time on the item's critical path.)
  uint32_t pre_ai = 0;
  int32_t pre_t = 0, pre_q = 0;
  float pre_coef = 0.0f;
  for (uint32_t pi = 0; pi < npend; ++pi) {
    if ((pi & 63u) == 0) {
      const uint32_t k = pi + lane;
      if (k < npend) {
        pre_ai = pend[k] - s;
        pre_t = (int32_t)tpos[pre_ai];
        pre_q = (int32_t)qpos[pre_ai];
        pre_coef = (float)(1 - 0.2 * (double)dist[pre_ai] / (double)radius);
      }
    }
    const int sel = (int)(pi & 63u);
    const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)pre_ai, sel);  // bucket-local index of the next unsettled anchor
    if (!kLds) {
      const uint32_t from = max(staged, ai + 1 > (uint32_t)kBigGroupCap ? ai + 1 - (uint32_t)kBigGroupCap : 0u);
      for (uint32_t i = from + lane; i <= ai; i += 64) {
        l_t[i & kMask] = (int32_t)tpos[i];
        l_q[i & kMask] = (int32_t)qpos[i];
        l_d[i & kMask] = dist[i];
        l_s[i & kMask] = score[i];
      }
      staged = ai + 1;
      __syncthreads();
    }
    const uint32_t am = kLds ? ai : (ai & kMask);
    const int32_t ring_lo = kLds ? 0 : (int32_t)ai - (int32_t)kBigGroupCap + 1;  // oldest anchor still in LDS
    const float coef = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pre_coef), sel));
    float best = coef * (float)dim;
    uint32_t best_pred = ai;
    const int32_t ct = __builtin_amdgcn_readlane(pre_t, sel), cq = __builtin_amdgcn_readlane(pre_q, sel);
    int32_t start_idx = 0;
    if (ai > (uint32_t)chaining_band_length) start_idx = (int32_t)ai - chaining_band_length;
    int32_t num_skips = 0;
    for (int32_t pj0 = (int32_t)ai - 1; pj0 >= start_idx; pj0 -= 64) {
      const int32_t pj = pj0 - (int32_t)lane;
      const bool in_range = pj >= start_idx;
      int cls = 0;  // 0 = end of range / target-gap exit, 1 = `continue`, 3 = evaluated
      float cur = 0.0f;
      if (in_range) {
        const bool in_ring = kLds || pj >= ring_lo;
        const uint32_t pm = kLds ? (uint32_t)pj : ((uint32_t)pj & kMask);
        const int32_t pt = in_ring ? l_t[pm] : (int32_t)tpos[pj], pq = in_ring ? l_q[pm] : (int32_t)qpos[pj];
        if (pq == cq || pt == ct) {
          cls = 1;
        } else if (pt + max_target_gap_length < ct) {
          cls = 0;
        } else {
          const int32_t td = ct - pt, qd = cq - pq;
          if (qd < 0) {
            cls = 1;
          } else {
            cls = 3;
            const int32_t m = min(min(td, qd), dim);
            const float matching = (float)m * coef;
            if (GapTestsPass<kIntGap>(td, qd, max_gap_length)) cur = (in_ring ? l_s[pm] : score[pj]) + matching;
          }
        }
      }
      const bool ev = cls == 3;
      // running best before each lane: max(best, evaluated candidates of earlier lanes).  The two wave scans of a
      // step are on the item's critical path (anchor after anchor): DPP row shifts and broadcasts, six dependent
      // VALU operations each, where six __shfl_up were six dependent trips through the LDS crossbar
      const float v = WaveScanMax(ev ? cur : -FLT_MAX);
      // (wave_shr:1: lane l reads lane l - 1, lane 0 keeps the identity)
      float before = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-FLT_MAX), __float_as_int(v), 0x138, 0xF, 0xF, false));
      before = fmaxf(before, best);
      const bool improved = ev && cur > before;
      const int delta = WaveScanAdd(ev ? (improved ? -1 : 1) : 0);
      const int ns_after = num_skips + delta;  // num_skips after this lane's step
      const unsigned long long m_end = __ballot(cls == 0);
      const unsigned long long m_skip = __ballot(ev && !improved && ns_after > max_num_skips);
      const int l_end = m_end ? __ffsll((long long)m_end) - 1 : 64;    // exclusive exit
      const int l_skip = m_skip ? __ffsll((long long)m_skip) - 1 : 64;  // inclusive exit
      const int last_valid = min(l_end - 1, l_skip);                    // last lane whose step counts
      unsigned long long m_imp = __ballot(improved);
      if (last_valid < 63) m_imp &= (last_valid < 0) ? 0ull : ((2ull << last_valid) - 1ull);
      if (m_imp) {
        const int l_last = 63 - __clzll((long long)m_imp);  // (wave-uniform: a scalar lane select)
        best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), l_last));
        best_pred = (uint32_t)(pj0 - l_last);
      }
      if (l_end < 64 || l_skip < 64) break;
      num_skips = __builtin_amdgcn_readlane(ns_after, 63);
    }
    if (lane == 0) {
      l_s[am] = best;
      score[ai] = best;
      pred[ai] = s + best_pred;  // chunk-local absolute index
      used[ai] = 0;
    }
    gm = fmaxf(gm, best);
    __syncthreads();  // (one wave per workgroup) the new score is visible to the next anchor's scan
    if (!kLds) __threadfence_block();
  }
  if (lane == 0) atomicMax(gmax_slot, __float_as_uint(gm));  // (other items of the bucket run at the same time)
}

template <bool kIntGap>
__device__ __forceinline__ void DpBigItems(const ChainArgs &a, const uint32_t *__restrict__ ioff, uint32_t n_chunks,
                                           uint32_t n_blocks, int32_t *l_t, int32_t *l_q, float *l_s, float *l_d,
                                           float radius, int dim, int max_gap_length, int max_target_gap_length,
                                           int chaining_band_length, int max_num_skips) {
  const uint32_t lane = threadIdx.x;
  // a workgroup (one wave) takes chunks in turn, looks through the chunk's item descriptors (a few hundred,
  // coalesced) and chains the long ones -- about one per chunk
  for (uint32_t c = blockIdx.x; c < n_chunks; c += n_blocks) {
   const uint32_t n_it = ioff[c + 1] - ioff[c];
   const uint64_t base = a.anchor_off[c];
   for (uint32_t k0 = 0; k0 < n_it; k0 += 64) {
    const uint32_t k = k0 + lane;
    const uint64_t my_hi = k < n_it ? a.item_hi[base / 2 + k] : 0ull;
    for (unsigned long long mb = __ballot((uint32_t)my_hi > (uint32_t)kBigPendingMin); mb; mb &= mb - 1ull) {
    const uint32_t kb = k0 + (uint32_t)__builtin_ctzll(mb);
    const uint64_t d_lo = a.item_lo[base / 2 + kb], d_hi = a.item_hi[base / 2 + kb];
    const uint32_t p0 = (uint32_t)(d_lo >> 32), p1 = p0 + (uint32_t)d_hi, g = (uint32_t)(d_hi >> 32);
    const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
    const uint32_t G = a.n_groups[c];
    const uint32_t *gstart = a.group_start + base;
    const uint32_t *pending = DpList(a, base, n);
    const uint32_t s = gstart[g], e = g + 1 < G ? gstart[g + 1] : n;
    const uint32_t ng = e - s;
    const uint32_t *tpos = a.s_tpos + base + s, *qpos = a.s_qpos + base + s;
    const float *dist = a.s_dist + base + s;
    float *score = a.score + base + s;
    uint32_t *pred = a.pred + base + s;
    uint8_t *used = a.used + base + s;
    uint32_t *gmax = reinterpret_cast<uint32_t *>(a.scratch + 2 * base);
    // a small bucket is staged whole; otherwise the LDS arrays are a ring that follows the item's anchors
    const bool in_lds = ng <= (uint32_t)kBigGroupCap;
    __syncthreads();
    if (in_lds)
      for (uint32_t i = lane; i < ng; i += 64) {
        l_t[i] = (int32_t)tpos[i];
        l_q[i] = (int32_t)qpos[i];
        l_d[i] = dist[i];
        l_s[i] = score[i];  // valid for the anchors the classify kernel settled (others are not read before they are set)
      }
    __syncthreads();
    if (in_lds)
      DpBigBucket<true, kIntGap>(lane, ng, s, pending + p0, p1 - p0, l_t, l_q, l_s, l_d, tpos, qpos, dist, score, pred, used,
                        &gmax[g], radius, dim, max_gap_length, max_target_gap_length, chaining_band_length, max_num_skips);
    else
      DpBigBucket<false, kIntGap>(lane, ng, s, pending + p0, p1 - p0, l_t, l_q, l_s, l_d, tpos, qpos, dist, score, pred, used,
                         &gmax[g], radius, dim, max_gap_length, max_target_gap_length, chaining_band_length,
                         max_num_skips);
    }
   }
  }
}

// ---- C1 and C1b are independent (different items, disjoint anchors; the bucket maxima are atomics): the few long
// items -- latency-bound, a wave each -- run on the context's auxiliary stream beside the many short ones
// (consecutive launches: 0.57 + 0.40 ms at 12 Mb; ONE launch with both bodies, 0.78 - 0.83 ms: the lane DP then
// runs at the wave DP's register count, five waves per SIMD instead of eight).
constexpr uint32_t kDpBigBlocks = 8192;
template <bool kIntGap>
__global__ __launch_bounds__(64) void ChainDpKernel(ChainArgs a, const uint32_t *__restrict__ ioff, uint32_t n_chunks,
                                                    float radius, int dim, int max_gap_length,
                                                    int max_target_gap_length, int chaining_band_length,
                                                    int max_num_skips) {
  __shared__ uint32_t s_ids[kBigPendingMin][64];  // each lane's pending anchor ids
  DpSmallItems<kIntGap>(a, ioff, n_chunks, 0u, gridDim.x, s_ids, radius, dim, max_gap_length, max_target_gap_length,
                        chaining_band_length, max_num_skips);
}
template <bool kIntGap>
__global__ __launch_bounds__(64) void ChainDpBigKernel(ChainArgs a, const uint32_t *__restrict__ ioff, uint32_t n_chunks,
                                                       float radius, int dim, int max_gap_length,
                                                       int max_target_gap_length, int chaining_band_length,
                                                       int max_num_skips) {
  __shared__ int32_t l_t[kBigGroupCap], l_q[kBigGroupCap];
  __shared__ float l_s[kBigGroupCap], l_d[kBigGroupCap];
  DpBigItems<kIntGap>(a, ioff, n_chunks, gridDim.x, l_t, l_q, l_s, l_d, radius, dim, max_gap_length, max_target_gap_length,
                      chaining_band_length, max_num_skips);
}

// ---- C2: one wave per chunk: running max_chaining_score across buckets (spatial_index.cc:420,
// 542-549), chain ends, traceback (spatial_index.cc:165-220, 552-568)
constexpr uint32_t kCoopScanMin = 48;     // buckets with more DP-scored anchors than this are scanned by the whole wave
constexpr float kLongChainScore = 60.0f;  // buckets whose best candidate scores at least this are traced by the whole wave
constexpr uint32_t kTraceWindow = 1024;  // (pred, used) entries staged in LDS below the anchor being traced

__global__ __launch_bounds__(64) void ChainEndsKernel(ChainArgs a, int min_num_anchors, int num_best_chains,
                                                      float min_chaining_score, uint32_t *__restrict__ n_chains_out,
                                                      const uint32_t *__restrict__ n_pending, int dim) {
  __shared__ uint32_t s_nchains;
  __shared__ uint32_t w_pred[kTraceWindow];
  __shared__ uint8_t w_used[kTraceWindow];
  const uint32_t c = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint64_t base = a.anchor_off[c];
  const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
  const uint32_t G = a.n_groups[c];
  if (lane == 0) {
    s_nchains = 0;
    a.out_anchor_cnt[c] = 0;
    n_chains_out[c] = 0;
  }
  __syncthreads();
  if (n == 0 || G == 0) return;
  const uint32_t *tpos = a.s_tpos + base, *grp = a.s_grp + base;
  const uint32_t *gstart = a.group_start + base;
  const float *score = a.score + base;
  const uint32_t *pred = a.pred + base;
  uint8_t *used = a.used + base;
  uint32_t *lists = a.lists + base;
  float *gmax = reinterpret_cast<float *>(a.scratch + 2 * base);
  float *pmax = gmax + n;
  const uint32_t *pending = a.pending + base;
  const uint32_t npend = n_pending[c];
  const bool pending_only = min_chaining_score > (float)dim;
  // the chunk's candidate chains: a slice of 3 per bucket (spatial_index.cc:555-568 keeps at most
  // num_best_chains <= 3 per (strand, contig) bucket), so no candidate is ever dropped
  ChainTmp *chain_out = a.chain_tmp + 3 * (size_t)a.chain_off[c];
  const uint32_t chain_cap = 3 * G;
  // max_chaining_score on entry to each bucket = max over all earlier buckets (order ref0+,
  // ref0-, ref1+, ...; initial value 0): exclusive prefix max by wave shuffles, 64 at a time.
  {
    float carry = 0.0f;
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
      const uint32_t g = g0 + lane;
      const float v = WaveScanMax(g < G ? gmax[g] : 0.0f);
      // (wave_shr:1: lane l reads lane l - 1, lane 0 keeps 0)
      const float ex = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
      if (g < G) pmax[g] = fmaxf(carry, ex);
      carry = fmaxf(carry, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)));
    }
  }
  __syncthreads();
  for (uint32_t g0 = 0; g0 < G; g0 += 64) {
    const uint32_t g = g0 + lane;
    uint32_t s = 0, lim = 0, grp_id = 0, p_lo = 0, p_hi = 0, ntop_l = 0;
    float max_after = 0.0f, run_max_l = 0.0f;
    bool coop = false;
    Cand top[3] = {};
    if (g < G) {
    s = gstart[g];
    const uint32_t e = g + 1 < G ? gstart[g + 1] : n;
    float run_max = pmax[g];  // max_chaining_score on entry to this bucket
    uint32_t ntop = 0;
    auto consider = [&](uint32_t ai, float sc) {
      if (sc > run_max) run_max = sc;
      if (sc >= min_chaining_score && sc > run_max / 2) {  // spatial_index.cc:545-549
        // keep the best three in sorted order (num_best_chains <= 3).  Constant indices only: with a position
        // computed at run time the three candidates lived in scratch memory, a round trip per comparison
        const Cand cnd{sc, ai - s};
        if (ntop < 1 || CandBefore(cnd, top[0])) {
          top[2] = top[1];
          top[1] = top[0];
          top[0] = cnd;
        } else if (ntop < 2 || CandBefore(cnd, top[1])) {
          top[2] = top[1];
          top[1] = cnd;
        } else if (ntop < 3 || CandBefore(cnd, top[2])) {
          top[2] = cnd;
        }
        if (ntop < 3) ++ntop;
      }
    };
    if (pending_only) {
      // Settled anchors score at most `dim` (their own start score) < min_chaining_score: none is
      // a chain end, and a running maximum they alone raised is <= dim, so it cannot fail the
      // `sc > max / 2` test of a candidate (sc >= min_chaining_score > dim).  Only the anchors the
      // DP kernels scored need to be looked at.
      uint32_t lo2 = 0, hi2 = npend;
      while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2) >> 1; if (pending[mid] < s) lo2 = mid + 1; else hi2 = mid; }
      uint32_t lo3 = lo2, hi3 = npend;
      while (lo3 < hi3) { const uint32_t mid = (lo3 + hi3) >> 1; if (pending[mid] < e) lo3 = mid + 1; else hi3 = mid; }
      p_lo = lo2;
      p_hi = lo3;
      // a bucket with many DP-scored anchors (the true locus) is scanned by the whole wave below
      coop = p_hi - p_lo > kCoopScanMin;
      // kPf list entries, then their kPf scores, per pair of memory round trips (clamped
      // unconditional loads: a conditional one would wait on its own)
      bool more = lo2 < npend && !coop;
      for (uint32_t pi = lo2; more; pi += kPf) {
        uint32_t ids[kPf];
        float sc8[kPf];
#pragma unroll
        for (int k = 0; k < kPf; ++k) ids[k] = pending[min(pi + k, npend - 1)];
#pragma unroll
        for (int k = 0; k < kPf; ++k) sc8[k] = score[ids[k]];
#pragma unroll
        for (int k = 0; k < kPf; ++k) {
          if (!more) break;
          if (pi + k >= npend || ids[k] >= e) {
            more = false;
            break;
          }
          consider(ids[k], sc8[k]);
        }
      }
    } else {
      for (uint32_t a0 = s; a0 < e; a0 += kPf) {
        float sc8[kPf];
#pragma unroll
        for (int k = 0; k < kPf; ++k) sc8[k] = a0 + k < e ? score[a0 + k] : 0.0f;
#pragma unroll
        for (int k = 0; k < kPf; ++k) {
          const uint32_t ai = a0 + k;
          if (ai >= e) break;
          consider(ai, sc8[k]);
        }
      }
    }
    run_max_l = run_max;
    ntop_l = ntop;
    }
    // Whole-wave scan of the buckets flagged above: 64 of the bucket's DP-scored anchors at a time;
    // the running maximum becomes an inclusive prefix maximum (shuffles) and the three best chain ends
    // under (score desc, index desc) -- what the sequential insertion keeps -- are picked by wave
    // maxima of the key score_bits << 32 | index (scores are positive: they order like their bits).
    for (uint64_t todo = __ballot(coop); todo; todo &= todo - 1) {
      const int src = __builtin_ctzll(todo);
      const uint32_t b_lo = (uint32_t)__shfl((int)p_lo, src), b_hi = (uint32_t)__shfl((int)p_hi, src);
      const uint32_t bs = (uint32_t)__shfl((int)s, src);
      float carry = __shfl(run_max_l, src);
      unsigned long long best3[3] = {0, 0, 0};  // descending; 0 = none (a candidate's key is > 0)
      for (uint32_t t0 = b_lo; t0 < b_hi; t0 += 64) {
        const uint32_t pi = t0 + lane;
        const bool valid = pi < b_hi;
        const uint32_t ai = valid ? pending[pi] : 0u;
        const float sc = valid ? score[ai] : 0.0f;
        float pm = WaveScanMax(sc);  // inclusive prefix maximum (scores are >= 0)
        pm = fmaxf(pm, carry);
        carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm), 63));
        unsigned long long key = 0;
        if (valid && sc >= min_chaining_score && sc > pm / 2)  // spatial_index.cc:545-549
          key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(ai - bs);
        for (int r = 0; r < 3 && __ballot(key != 0); ++r) {
          unsigned long long m = key;
          for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long o = __shfl_xor(m, d, 64);
            if (o > m) m = o;
          }
          if (key == m) key = 0;  // keys are distinct (distinct indices): exactly one lane
          // insert m into the descending best3
          if (m > best3[0]) { best3[2] = best3[1]; best3[1] = best3[0]; best3[0] = m; }
          else if (m > best3[1]) { best3[2] = best3[1]; best3[1] = m; }
          else if (m > best3[2]) { best3[2] = m; }
          else break;  // the tile's remaining candidates are smaller still
        }
      }
      if ((int)lane == src) {
        run_max_l = carry;
        ntop_l = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (best3[k]) {
            top[k].score = __uint_as_float((uint32_t)(best3[k] >> 32));
            top[k].idx = (uint32_t)best3[k];
            ntop_l = k + 1;
          }
      }
    }
    if (g < G) {
    float run_max = run_max_l;
    const uint32_t ntop = ntop_l;
    run_max = fmaxf(run_max, gmax[g]);  // (the bucket maximum includes the settled anchors)
    max_after = run_max;  // max_chaining_score after the bucket
    grp_id = grp[s];
    lim = min(ntop, (uint32_t)min(num_best_chains, 3));
    // A bucket whose best candidate scores below kLongChainScore has short chains (an anchor adds at
    // most `dim` to a score): its lane traces them itself, all such buckets of the wave at once.
    if (lim > 0 && top[0].score < kLongChainScore) {
      const uint32_t ref = grp_id >> 1;
      const uint32_t direction = (grp_id & 1) == 0 ? 1u : 0u;  // strand 0 = Positive
      uint32_t cursor = s;  // chain anchor lists are packed into this bucket's own slice of `lists`
      for (uint32_t k = 0; k < lim; ++k) {
        const uint32_t end_abs = s + (k == 0 ? top[0].idx : (k == 1 ? top[1].idx : top[2].idx));
        if (!used[end_abs]) {
          // (the reference's loop, with pred[p] fetched beside used[p]: one memory round trip per
          // chain anchor instead of two dependent ones)
          bool stop_at_used = false;
          uint32_t cur = end_abs;
          const uint32_t list_off = cursor;
          uint32_t num = 0;
          uint32_t pc = pred[cur];
          for (;;) {
            lists[cursor++] = cur;
            used[cur] = 1;
            ++num;
            if (pc == cur) break;
            const uint8_t up = used[pc];
            const uint32_t pp = pred[pc];
            if (up) {
              stop_at_used = true;
              break;
            }
            cur = pc;
            pc = pp;
          }
          if (num >= (uint32_t)min_num_anchors) {
            float adjusted = score[end_abs];
            if (stop_at_used) adjusted -= score[pc];  // pc == pred[cur]
            const uint32_t slot = atomicAdd(&s_nchains, 1u);
            if (slot < chain_cap) {  // (always: at most 3 candidates per bucket)
              ChainTmp ch;
              ch.score = adjusted;
              ch.ref = ref;
              ch.start = tpos[cur];
              ch.end = tpos[end_abs];
              ch.n_anchors = num;
              ch.direction = direction;
              ch.end_anchor = end_abs;
              ch.list_off = list_off;
              chain_out[slot] = ch;
            }
          }
        }
        if (score[end_abs] < max_after / 2) break;  // spatial_index.cc:563-567
      }
      lim = 0;
    }
    }
    // TracebackChains, spatial_index.cc:165-220 (pred[] holds chunk-local absolute indices).  A
    // traceback is a pointer chase -- one memory round trip per chain anchor, ~250 of them for the
    // true locus of a read -- so the buckets with a long candidate are taken one after the other by
    // the WHOLE wave: their owner's state is broadcast, a window of 1024 (pred, used) entries
    // below the current anchor is staged in LDS by all lanes, and the chase runs on LDS.
    for (uint64_t todo = __ballot(lim > 0); todo; todo &= todo - 1) {
      const int src = __builtin_ctzll(todo);
      const uint32_t bs = (uint32_t)__shfl((int)s, src), blim = (uint32_t)__shfl((int)lim, src);
      const float bmax_after = __shfl(max_after, src);
      const uint32_t bgrp = (uint32_t)__shfl((int)grp_id, src);
      uint32_t bidx[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) bidx[k] = (uint32_t)__shfl((int)top[k].idx, src);
      const uint32_t ref = bgrp >> 1;
      const uint32_t direction = (bgrp & 1) == 0 ? 1u : 0u;  // strand 0 = Positive
      uint32_t cursor = bs;  // chain anchor lists are packed into this bucket's own slice of `lists`
      uint32_t w_lo = 1, w_hi = 0;  // window [w_lo, w_hi) of the bucket staged in LDS (empty)
      auto ensure = [&](uint32_t p) {  // wave-uniform
        if (p >= w_lo && p < w_hi) return;
        __syncthreads();
        w_hi = p + 1;
        w_lo = w_hi - bs > kTraceWindow ? w_hi - kTraceWindow : bs;
        for (uint32_t i = lane; i < w_hi - w_lo; i += 64) {
          w_pred[i] = pred[w_lo + i];
          w_used[i] = used[w_lo + i];
        }
        __syncthreads();
      };
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if ((uint32_t)k >= blim) break;
        const uint32_t end_abs = bs + bidx[k];
        ensure(end_abs);
        if (!w_used[end_abs - w_lo]) {
          bool stop_at_used = false;
          uint32_t cur = end_abs;
          const uint32_t list_off = cursor;
          uint32_t num = 0;
          uint32_t pc = w_pred[cur - w_lo];
          for (;;) {
            if (lane == 0) {
              lists[cursor] = cur;
              used[cur] = 1;
            }
            w_used[cur - w_lo] = 1;  // (every lane writes the same value)
            ++cursor;
            ++num;
            if (pc == cur) break;
            ensure(pc);
            const uint8_t up = w_used[pc - w_lo];
            const uint32_t pp = w_pred[pc - w_lo];
            if (up) {
              stop_at_used = true;
              break;
            }
            cur = pc;
            pc = pp;
          }
          if (num >= (uint32_t)min_num_anchors) {
            float adjusted = score[end_abs];
            if (stop_at_used) adjusted -= score[pc];  // pc == pred[cur]
            if (lane == 0) {
              const uint32_t slot = atomicAdd(&s_nchains, 1u);
              if (slot < chain_cap) {  // (always: at most 3 candidates per bucket)
                ChainTmp ch;
                ch.score = adjusted;
                ch.ref = ref;
                ch.start = tpos[cur];
                ch.end = tpos[end_abs];
                ch.n_anchors = num;
                ch.direction = direction;
                ch.end_anchor = end_abs;
                ch.list_off = list_off;
                chain_out[slot] = ch;
              }
            }
          }
        }
        if (score[end_abs] < bmax_after / 2) break;  // spatial_index.cc:563-567
      }
    }
  }
  __syncthreads();
  if (lane == 0) n_chains_out[c] = s_nchains;
}

// ---- C3: GeneratePrimaryChains + ComputeMAPQ + chain statistics + carry-over anchors, one wave
// per chunk (spatial_index.cc:222-274, sigmap.cc:702-729).  The candidate chains are ranked and
// filtered by the whole wave; all lanes gather the chains' anchors through the recorded index lists;
// chain 0's statistics are added up in the reference's order (f32 sums are order-sensitive).

// kBig = false: up to kChainCap candidates staged, ranked and filtered in LDS (every chunk of a
// reference with up to 21 contigs, and almost every chunk otherwise).  kBig = true: any number of
// candidates (3 per bucket, 2 buckets per contig), the same steps on the chunk's global slices.
template <bool kBig>
__device__ __forceinline__ void FinalizeBody(const ChainArgs &a, uint32_t c, uint32_t lane, uint32_t nch,
                                             ChainTmp *ch /* LDS[kChainCap], or the global sorted slice */,
                                             const ChainTmp *gch) {
  const uint64_t base = a.anchor_off[c];
  const uint32_t *tpos = a.s_tpos + base, *qpos = a.s_qpos + base, *grp = a.s_grp + base;
  const float *dist = a.s_dist + base;
  const uint32_t *lists = a.lists + base;
  smhip_chunk_result &res = a.results[c];
  // Descending by operator> (spatial_index.cc:569-570), keeping the order of equal chains: every
  // lane ranks its own chains against all the others (broadcast reads) -- a one-lane insertion sort
  // of a few dozen 32-byte records took 100 000+ cycles per chunk.
  if (!kBig) {
    for (uint32_t i = lane; i < nch; i += 64) ch[i] = gch[i];
    __syncthreads();
    static_assert(kChainCap <= 128, "two chains per lane");
    ChainTmp mine[2] = {};
    uint32_t rank[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (lane + 64u * h < nch) mine[h] = ch[lane + 64u * h];
    for (uint32_t j = 0; j < nch; ++j) {
      const ChainTmp o = ch[j];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t self = lane + 64u * h;
        if (self < nch && j != self) {
          if (ChainGreater(o, mine[h])) ++rank[h];
          else if (!ChainGreater(mine[h], o) && j < self) ++rank[h];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (lane + 64u * h < nch) ch[rank[h]] = mine[h];
    __syncthreads();
  } else {
    for (uint32_t self = lane; self < nch; self += 64) {
      const ChainTmp mine = gch[self];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < nch; ++j) {
        if (j == self) continue;
        const ChainTmp o = gch[j];
        if (ChainGreater(o, mine)) ++rank;
        else if (!ChainGreater(mine, o) && j < self) ++rank;
      }
      ch[rank] = mine;
    }
    __threadfence_block();
    __syncthreads();
  }
  // spatial_index.cc:222-253: greedy, in rank order; primaries are compacted to the front of ch[].
  // The candidates are taken one after the other, but each is tested against all kept primaries at
  // once.  (With one lane doing both loops a chunk whose candidates all score alike cost 16 000
  // dependent LDS round trips, and the launch waited.)
  uint32_t np = 0;
  if (nch > 0) {
    if (!kBig) {  // lane l holds primaries l and l + 64 in registers
      uint32_t p_ref[2] = {0, 0}, p_start[2] = {0, 0}, p_end[2] = {0, 0};
      float last_score;
      {
        const ChainTmp x = ch[0];
        if (lane == 0) {
          p_ref[0] = x.ref;
          p_start[0] = x.start;
          p_end[0] = x.end;
        }
        last_score = x.score;
        np = 1;
      }
      for (uint32_t ci = 1; ci < nch; ++ci) {
        const ChainTmp x = ch[ci];
        if (x.score < last_score / 3) break;
        bool overlaps = false;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (lane + 64u * h < np && x.ref == p_ref[h] && !(max(x.start, p_start[h]) > min(x.end, p_end[h])))
            overlaps = true;
        if (__ballot(overlaps) == 0) {
          if (lane == (np & 63u)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
              if ((np >> 6) == (uint32_t)h) {
                p_ref[h] = x.ref;
                p_start[h] = x.start;
                p_end[h] = x.end;
              }
          }
          if (lane == 0) ch[np] = x;  // np <= ci: nothing unread is overwritten
          last_score = x.score;
          ++np;
        }
      }
    } else {  // the kept primaries are read back from the front of the slice, 64 at a time
      float last_score = ch[0].score;
      np = 1;
      for (uint32_t ci = 1; ci < nch; ++ci) {
        const ChainTmp x = ch[ci];
        if (x.score < last_score / 3) break;
        bool overlaps = false;
        for (uint32_t pi = lane; pi < np; pi += 64) {
          const ChainTmp pk = ch[pi];
          if (x.ref == pk.ref && !(max(x.start, pk.start) > min(x.end, pk.end))) overlaps = true;
        }
        if (__ballot(overlaps) == 0) {
          if (lane == 0) ch[np] = x;  // np <= ci: nothing unread is overwritten
          __threadfence_block();
          __builtin_amdgcn_wave_barrier();
          last_score = x.score;
          ++np;
        }
      }
    }
  }
  __syncthreads();
  if (lane == 0 && np > 0) {
    // spatial_index.cc:255-274
    uint32_t mapq;
    if (np == 1) {
      mapq = 60;
    } else {
      int m = (int)(40 * (1 - ch[1].score / ch[0].score));
      if (m > 60) m = 60;
      if (m < 0) m = 0;
      mapq = (uint32_t)m;
    }
    res.n_chains = np;
    float sum = 0;
    for (uint32_t i = 0; i < np; ++i) sum += ch[i].score;  // sigmap.cc:702-705
    res.sum_score = sum;
    for (uint32_t i = 0; i < np && i < SMHIP_TOPK; ++i) {
      smhip_chain o;
      o.score = ch[i].score;
      o.ref = ch[i].ref;
      o.start = ch[i].start;
      o.end = ch[i].end;
      o.n_anchors = ch[i].n_anchors;
      o.mapq = i == 0 ? mapq : 0;
      o.direction = ch[i].direction;
      o.pad = 0;
      res.chains[i] = o;
    }
  }
  if (np == 0) {
    if (lane == 0) a.out_anchor_cnt[c] = 0;
    return;
  }
  // every primary chain's record, in rank order, at the front of the chunk's ranked slice (where the large-chunk
  // route has them already): smhip_map_download_chains hands out the whole std::vector<SignalAnchorChain>, not only
  // the SMHIP_TOPK chains of the result record
  if (!kBig) {
    ChainTmp *ranked = a.chain_sorted + 3 * (size_t)a.chain_off[c];
    for (uint32_t i = lane; i < np; i += 64) ranked[i] = ch[i];
  }
  // anchors of every primary chain, end -> start (the order TracebackChains pushes them), for the
  // next chunk's carry-over
  uint32_t w = 0;
  smhip_carried_anchor *out = a.out_anchors ? a.out_anchors + base : nullptr;
  // chain 0's sums (sigmap.cc:714-729) are order-sensitive f32 additions.  The anchors stay in the
  // registers of the lanes that gathered them, the gaps to the previous anchor are formed in
  // parallel, and every lane adds the 64 values of a tile up in index order from broadcasts
  // (v_readlane): the same additions in the same order, without a memory round trip per anchor.
  float ad = 0, at = 0, aq = 0;
  uint32_t carry_t = 0, carry_q = 0;  // last anchor of the previous tile
  for (uint32_t i = 0; i < np; ++i) {
    const ChainTmp ci = ch[i];
    const uint32_t g_id = grp[ci.end_anchor];
    const uint32_t *lst = lists + ci.list_off;
    const uint32_t na = ci.n_anchors;
    for (uint32_t k0 = 0; k0 < na; k0 += 64) {
      const uint32_t k = k0 + lane;
      uint32_t t = 0, q = 0;
      float d = 0.0f;
      if (k < na) {
        const uint32_t ai = lst[k];
        t = tpos[ai];
        q = qpos[ai];
        d = dist[ai];
        if (out) {
          smhip_carried_anchor o;
          o.ref_strand = g_id;
          o.target_position = t;
          o.query_position = q;
          o.distance = d;
          out[w + k] = o;
        }
      }
      if (i != 0) continue;
      uint32_t pt = (uint32_t)__shfl_up((int)t, 1), pq = (uint32_t)__shfl_up((int)q, 1);
      if (lane == 0) {
        pt = carry_t;
        pq = carry_q;
      }
      const float gt = (float)(uint32_t)(pt - t), gq = (float)(uint32_t)(pq - q);
      const uint32_t kn = min(na - k0, 64u);
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        if ((uint32_t)j < kn) {
          ad += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), j));
          if (k0 + (uint32_t)j > 0) {
            at += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gt), j));
            aq += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gq), j));
          }
        }
      }
      if (k0 == 0) {
        const uint32_t q_first = (uint32_t)__builtin_amdgcn_readlane((int)q, 0);
        if (lane == 0) res.c0_first_qpos = q_first;
      }
      carry_t = (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
      carry_q = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
      if (k0 + 64 >= na) {
        const uint32_t q_last = (uint32_t)__shfl((int)q, (int)(na - 1 - k0));
        if (lane == 0) res.c0_last_qpos = q_last;
      }
    }
    w += na;
    if (i == 0 && lane == 0) {
      res.c0_sum_dist = ad;
      res.c0_sum_tgap = at;
      res.c0_sum_qgap = aq;
    }
  }
  if (lane == 0) a.out_anchor_cnt[c] = w;
}

__global__ __launch_bounds__(64) void ChainFinalizeKernel(ChainArgs a, uint32_t n_chunks,
                                                          const uint32_t *__restrict__ n_chains_in) {
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  __shared__ ChainTmp ch[kChainCap];  // the chunk's candidate chains
  const size_t slice = 3 * (size_t)a.chain_off[c];
  const ChainTmp *gch = a.chain_tmp + slice;
  uint32_t nch = n_chains_in[c];
  const uint32_t cap = 3 * a.n_groups[c];
  if (nch > cap) {  // cannot happen (3 candidates per bucket); reported, never silent
    if (lane == 0) a.results[c].status |= SMHIP_ST_CHAIN_CAP;
    nch = cap;
  }
  if (nch <= (uint32_t)kChainCap) FinalizeBody<false>(a, c, lane, nch, ch, gch);
  else FinalizeBody<true>(a, c, lane, nch, a.chain_sorted + slice, gch);
}
}  // namespace

hipError_t LaunchChain(smhip_ctx *ctx, const smhip_params &p, const ChainArgs &a_in, uint32_t n_chunks,
                       uint64_t n_anchors_total, uint32_t n_refs) {
  if (n_chunks == 0) return hipSuccess;
  ChainArgs a = a_in;
  // an item of the wave-parallel kernel holds > kBigPendingMin anchors
  const uint64_t max_big = n_anchors_total / kBigPendingMin + 1;
  hipError_t e = ctx->chain_cnt.Reserve(((size_t)n_chunks * 7 + 8 + max_big) * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  uint32_t *n_chains = ctx->chain_cnt.as<uint32_t>(), *goff = n_chains + n_chunks;
  uint32_t *n_pending = goff + n_chunks + 1;
  uint32_t *tiles = n_pending + n_chunks, *tile_off = tiles + n_chunks;
  uint32_t *n_items = tile_off + n_chunks + 1, *ioff = n_items + n_chunks;
  // (item descriptors: the unsorted SoA is dead after the sort -- a_dist / a_grp hold 4 bytes per anchor each,
  // i.e. one 8-byte word per two anchors, and a chunk has at most half as many items as anchors)
  if ((e = LaunchScanU32(ctx, a.n_groups, n_chunks, goff)) != hipSuccess) return e;
  // candidate chains: 3 per bucket, a bucket holds >= 1 anchor and there are <= 2 per contig; second
  // half = the ranked copy of the chunks with more than kChainCap candidates
  const uint64_t max_cand = 3 * std::min<uint64_t>((uint64_t)n_chunks * 2 * n_refs, std::max<uint64_t>(n_anchors_total, 1));
  if ((e = ctx->chain_tmp.Reserve(2 * max_cand * sizeof(ChainTmp))) != hipSuccess) return e;
  a.chain_tmp = ctx->chain_tmp.as<ChainTmp>();
  a.chain_sorted = a.chain_tmp + max_cand;
  a.chain_off = goff;
  ctx->dl_chain_ranked = a.chain_sorted;  // (smhip_map_download_chains)
  ctx->dl_chain_off = goff;
  hipLaunchKernelGGL(ChainTileCountKernel, dim3((n_chunks + 255) / 256), dim3(256), 0, ctx->stream, a, n_chunks, tiles);
  if ((e = LaunchScanU32(ctx, tiles, n_chunks, tile_off)) != hipSuccess) return e;
  const uint64_t max_tiles = n_anchors_total / kTile + n_chunks;
  const uint32_t tile_blocks = (uint32_t)std::min<uint64_t>((max_tiles + 3) / 4, 256u * 8u * 2u);  // four waves (tile ranges) each
  // (integer form of the gap-scale tests: exact for target gaps below 2^21, see GapTestsPass)
  // (SMHIP_FLOAT_GAP: test knob, read per call -- the float form also where the integer one is valid)
  const bool int_gap = p.max_target_gap_length >= 0 && p.max_target_gap_length < (1 << 21) && !ctx->knobs.float_gap;
  // windows of up to dense_k predecessors are evaluated as dense pairs (the scan cannot end early inside them);
  // knob classify_dense = 0: lockstep scans for every tile
  const uint32_t dense_k = ctx->knobs.classify_dense == 0 ? 0u
      : (uint32_t)std::max<int64_t>(0, std::min<int64_t>(std::min<int64_t>((int64_t)p.max_num_skips + 1, p.chaining_band_length), kDenseKMax));
  // (30, not 31: the kernel's window count saturates at 31 = "31 or more", which must always be beyond dense_k)
  // ... and an anchor whose passing predecessors are all isolated is chained there (knob classify_resolve = 0: left to the DP)
  const uint32_t resolve = ctx->knobs.classify_resolve == 0 ? 0u : 1u;
  if (int_gap)
    hipLaunchKernelGGL(ChainClassifyKernel<true>, dim3(tile_blocks), dim3(256), 0, ctx->stream, a, n_chunks, tile_off,
                       p.search_radius, kMaxDim, p.max_gap_length, p.max_target_gap_length, p.chaining_band_length,
                       p.max_num_skips, dense_k, resolve);
  else
    hipLaunchKernelGGL(ChainClassifyKernel<false>, dim3(tile_blocks), dim3(256), 0, ctx->stream, a, n_chunks, tile_off,
                       p.search_radius, kMaxDim, p.max_gap_length, p.max_target_gap_length, p.chaining_band_length,
                       p.max_num_skips, dense_k, resolve);
  hipLaunchKernelGGL(ChainPendingKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, a, p.max_target_gap_length, n_pending,
                     n_items);
  if ((e = LaunchScanU32(ctx, n_items, n_chunks, ioff)) != hipSuccess) return e;
  // fork: the wave DP of the long items on the auxiliary stream, the lane DP of the short ones here; join before C2
  if ((e = hipEventRecord(ctx->ev_fork, ctx->stream)) != hipSuccess) return e;
  if ((e = hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0)) != hipSuccess) return e;
  if (int_gap)
    hipLaunchKernelGGL(ChainDpBigKernel<true>, dim3(kDpBigBlocks), dim3(64), 0, ctx->aux_stream, a, ioff, n_chunks,
                       p.search_radius, kMaxDim, p.max_gap_length, p.max_target_gap_length, p.chaining_band_length,
                       p.max_num_skips);
  else
    hipLaunchKernelGGL(ChainDpBigKernel<false>, dim3(kDpBigBlocks), dim3(64), 0, ctx->aux_stream, a, ioff, n_chunks,
                       p.search_radius, kMaxDim, p.max_gap_length, p.max_target_gap_length, p.chaining_band_length,
                       p.max_num_skips);
  if ((e = hipEventRecord(ctx->ev_join, ctx->aux_stream)) != hipSuccess) return e;
  const uint32_t dp_blocks = (uint32_t)std::min<uint64_t>((n_anchors_total + 63) / 64 + 1, 16384);
  if (int_gap)
    hipLaunchKernelGGL(ChainDpKernel<true>, dim3(dp_blocks), dim3(64), 0, ctx->stream, a, ioff, n_chunks, p.search_radius,
                       kMaxDim, p.max_gap_length, p.max_target_gap_length, p.chaining_band_length, p.max_num_skips);
  else
    hipLaunchKernelGGL(ChainDpKernel<false>, dim3(dp_blocks), dim3(64), 0, ctx->stream, a, ioff, n_chunks, p.search_radius,
                       kMaxDim, p.max_gap_length, p.max_target_gap_length, p.chaining_band_length, p.max_num_skips);
  if ((e = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)) != hipSuccess) return e;
  hipLaunchKernelGGL(ChainEndsKernel, dim3(n_chunks), dim3(64), 0, ctx->stream, a, p.min_num_anchors,
                     p.num_best_chains, p.min_chaining_score, n_chains, n_pending, kMaxDim);
  hipLaunchKernelGGL(ChainFinalizeKernel, dim3(n_chunks), dim3(64), 0, ctx->stream, a, n_chunks, n_chains);
  return hipGetLastError();
}

namespace {
__global__ void ChainCountKernel(const smhip_chunk_result *__restrict__ results, uint32_t n_chunks, uint32_t *__restrict__ cnt) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_chunks) cnt[c] = results[c].n_chains;
}
// primary chains of every chunk, in rank order, as smhip_chain records (chain 0 carries the MAPQ)
__global__ void ChainGatherKernel(const smhip_chunk_result *__restrict__ results, const ChainTmp *__restrict__ ranked,
                                  const uint32_t *__restrict__ chain_off, const uint64_t *__restrict__ out_off,
                                  smhip_chain *__restrict__ out) {
  const uint32_t c = blockIdx.x;
  const uint32_t np = results[c].n_chains;
  const ChainTmp *src = ranked + 3 * (size_t)chain_off[c];
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
    const ChainTmp t = src[i];
    smhip_chain o;
    o.score = t.score;
    o.ref = t.ref;
    o.start = t.start;
    o.end = t.end;
    o.n_anchors = t.n_anchors;
    o.mapq = i == 0 ? results[c].chains[0].mapq : 0;
    o.direction = t.direction;
    o.pad = 0;
    out[out_off[c] + i] = o;
  }
}
}  // namespace
hipError_t LaunchChainCount(smhip_ctx *ctx, const smhip_chunk_result *results, uint32_t n_chunks, uint32_t *cnt) {
  hipLaunchKernelGGL(ChainCountKernel, dim3((n_chunks + 255) / 256), dim3(256), 0, ctx->stream, results, n_chunks, cnt);
  return hipGetLastError();
}
hipError_t LaunchChainGather(smhip_ctx *ctx, const smhip_chunk_result *results, uint32_t n_chunks, const uint64_t *out_off,
                             smhip_chain *out) {
  hipLaunchKernelGGL(ChainGatherKernel, dim3(n_chunks), dim3(64), 0, ctx->stream, results,
                     reinterpret_cast<const ChainTmp *>(ctx->dl_chain_ranked), ctx->dl_chain_off, out_off, out);
  return hipGetLastError();
}
}  // namespace smhip

// ---- smhip_debug_start_coef_sweep: the classifier's start-score factor (StartCoef: reciprocal multiply + guard band)
// against the reference's expression, on the device, for every float whose bit pattern lies in [lo_bits, hi_bits)
namespace smhip {
namespace {
__global__ void StartCoefSweepKernel(uint32_t lo_bits, uint64_t count, float radius, unsigned long long *mismatches) {
  const double inv_radius = 1.0 / (double)radius;
  unsigned long long bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
    const float d = __uint_as_float(lo_bits + (uint32_t)i);
    const float fast = StartCoef(d, (double)radius, inv_radius);
    const float want = (float)(1 - 0.2 * (double)d / (double)radius);  // spatial_index.cc:437-438
    bad += __float_as_uint(fast) != __float_as_uint(want) ? 1ull : 0ull;
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace
}  // namespace smhip
extern "C" int smhip_debug_start_coef_sweep(smhip_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, float radius,
                                            uint64_t *mismatches) {
  if (!ctx || !mismatches || hi_bits < lo_bits) return SMHIP_EINVAL;
  unsigned long long *dev = nullptr;
  if (hipMalloc(&dev, sizeof *dev) != hipSuccess) return SMHIP_ENOMEM;
  int rc = SMHIP_OK;
  if (hipMemsetAsync(dev, 0, sizeof *dev, ctx->stream) != hipSuccess) rc = SMHIP_EHIP;
  if (rc == SMHIP_OK && hi_bits > lo_bits)
    hipLaunchKernelGGL(smhip::StartCoefSweepKernel, dim3(4096), dim3(256), 0, ctx->stream, lo_bits,
                       (uint64_t)hi_bits - lo_bits, radius, dev);
  unsigned long long host = 0;
  if (rc == SMHIP_OK && (hipMemcpyAsync(&host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipStreamSynchronize(ctx->stream) != hipSuccess))
    rc = SMHIP_EHIP;
  (void)hipFree(dev);
  *mismatches = host;
  return rc;
}

#ifdef SMHIP_PHASE_PROF
extern "C" int smhip_debug_chain_prof(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(smhip::g_phase_prof), 32 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smhip::g_phase_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
