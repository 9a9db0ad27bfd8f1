// seed_lookup.hip -- batched exact radius search over the flattened k-d tree (gfx950).
//
// Replaces, per seed, nanoflann's radiusSearch -> findNeighbors -> computeInitialDistances
// -> searchLevel -> L2_Adaptor::evalMetric -> RadiusResultSet::addPoint
// (nanoflann.hpp:1278-1301, 1221-1242, 1006-1023, 1346-1410, 383-408, 249-253) as called
// from SpatialIndex::GenerateChains (spatial_index.cc:351-409).
//
// Data layout in HBM (built once per index from the pre-order smhip_node array + point cloud):
//   inner    : one 64-byte record (one sector) per INNER node, numbered in pre-order among the inner
//              nodes: {divlow, divhigh, ref1 | feat << 28, ref2} and the TIGHT bounding boxes of
//              BOTH CHILDREN, 24 x u16: child1 lo[6], hi[6], child2 lo[6], hi[6], quantised on a grid over
//              the root box (value = root_lo[d] + q * step, step = widest root extent / 65534) and
//              rounded OUTWARDS, so a quantised box contains the exact one.  A ref is an inner
//              index, or kLeafTag | leaf ordinal (tree order): leaves have no record of their own.
//   quads    : one 128-byte record (one whole line) per inner node at EVEN depth: boxes and references of the up to
//              four nodes two levels below it -- the order-free search's node array (BuildQuads, further down)
//   leaf_tab : uint32 per leaf: (first 32-byte slot of its point block / 8) | n << 27
//   leaf_pts : one 256-byte-aligned block of 32-byte slots per leaf at slot offset `boff`: n points
//              {v0, v1, v2, v3, v4, v5, uint64 Point.position}, four to a 128-byte line (a leaf of 10 .. 20 points
//              is 3 .. 5 lines; the coordinates' first four and last two words in two runs of eight-point lines
//              were 4 or 6).  This replaces <= 20 scattered (vind -> 6 strided floats) gathers per leaf
//              (sigmap_adaptor.h:94-97), and a hit needs no second lookup for its position.
//   leaf_vind: uint32[n_slots]: original point index of slot boff + t
//              (Hit.slot; only the per-stage test entry point reads it)
//   qrec     : one 32-byte record per query: float q[6]; uint32 chunk; uint32 k
//
// The recursion of searchLevel is replaced by an explicit stack of pending "other child" visits
// (node, mindistsq, dists[6] = 32 bytes per entry).  That reproduces nanoflann's visiting
// order and pruning arithmetic (`mindistsq + cut_dist - dists[idx]`, same operation order)
// exactly, so hits come out in the reference's traversal order; the k-th hit of a query is stored
// at slot k, and k >= num_nearest_points (5000, spatial_index.cc:371-372) is dropped exactly like
// the reference.
//
// What is NOT visited although nanoflann visits it (results are identical, argument in place):
// a node whose (outward-rounded) tight box is at squared distance >= radius from the query; the
// test is made at the PARENT, which holds both children's boxes, so such a node costs neither a
// visit nor a stack entry, and a leaf costs no record read at all.  The box distance is
// accumulated with nanoflann's own association order (evalMetric) and every f32 operation
// involved is monotone, so it is a lower bound of the f32 distance nanoflann would compute for
// any point below the node -- no point there can pass `dist < radius`, the subtree contributes no
// hit, and skipping it changes neither the hit list nor its order.  nanoflann's own rule is still
// applied with its own arithmetic on top (it must be: it is not implied bit-for-bit by the box rule).
// On the 12 Mb index this reads about 250 node records per query where nanoflann visits 470 nodes
// and tests 205 points instead of 1196 (what bounds the kernel is the number of 64-byte sectors it
// touches: DESIGN.md 4.2).  kStrict = true switches the box rule off for inner nodes and counts
// nanoflann's work (nodes visited, points it would test): that is the census the roofline's
// algorithmic bytes are computed from, and what the per-stage tests compare with the oracle.
//
// Mapping: ONE LANE PER QUERY.  A wave alternates between two uniform phases so that its
// instruction stream is one body at a time:
//   walk : every active lane loads its current node record (4 x 16 B), tests the box, and either
//          descends (inner), queues the leaf in its LDS FIFO (leaf), or pops its stack;
//   scan : when enough lanes are blocked (FIFO full, or query finished with leaves pending) all
//          lanes drain their FIFOs, four points of a leaf per iteration.
// Leaves of a query are scanned in the order they were reached, so hits keep traversal order.
// The two newest stack entries of a lane live in LDS, older ones spill to HBM ([lane][64] x 32 B).
// The mapping path does not need the order (the anchors are sorted afterwards) except for queries beyond
// the reference's 5000-hit cap: it searches with the kernel's order-free SET mode (no dists[] / mindistsq
// state, 4-byte pending visits, box rule only, quad records: the kWalk* modes at the kernel) and repeats those
// queries in the reference's order (LaunchSeedLookup).
// Queries are visited in the order of the leaf that contains them (QueryKey/QueryScatter: a
// stackless descent + a counting sort that permutes the query records), each XCD taking one
// contiguous eighth of that order, so the lanes of a wave and the waves of an XCD walk the same
// part of the tree and share its cache lines; results are indexed by query id, so the visiting
// order never shows in the output.
#include <stdlib.h>

#include "common.h"

namespace smhip {
namespace {

constexpr int kSeedThreads = 256;
#ifndef SMHIP_FIFO
#define SMHIP_FIFO 4
#endif
constexpr int kFifo = SMHIP_FIFO;  // pending leaves per lane (at most 7: the scan phase counts them in three ballots)
constexpr uint32_t kIdxMask = 0x07FFFFFFu;  // index part of a ref (inner index or leaf ordinal)
constexpr uint32_t kPassTag = 0x40000000u;  // stack entry / current node: the box test at the parent was passed

struct QueryRec {  // 32 B
  float q[6];
  uint32_t chunk, k;
};

struct Sink {
  HitSlots region;      // [n_queries][fast_cap], in segments
  uint32_t fast_cap;    // slots per query before the overflow pool
  uint32_t nnp;         // num_nearest_points
  uint32_t *q_count;    // total hits per query
  uint32_t *q_nodes;    // optional per-query stats
  uint32_t *q_points;
  smhip_chunk_result *results;  // unused (per-chunk sums are formed by AnchorCountKernel)
  uint32_t *counters;   // [0] stack overflow flag, [1] #overflow queries, [3] pool full, [24..25] pool cursor (u64)
  Hit *pool;            // overflow pool: hits beyond the fast slots, in geometric segments
  uint32_t *ovf_tab;    // [n_queries][kOvfLevels] pool offset of each segment
  uint64_t pool_cap;
  uint32_t stop_at_cap;  // second search of the queries beyond the cap: a query ends with its nnp-th hit, its totals
                         // (hits, nodes, points) stay those of the first search
  uint32_t store_all;    // set walk whose over-cap queries are RANKED afterwards (OverCapSelectKernel): every hit is
                         // stored, also beyond the nnp-th in this walk's own order -- which of them the reference
                         // keeps is decided from the whole set
};

// nanoflann.hpp:1006-1023 computeInitialDistances (low test, then high test, per dimension)
#define SM_INIT_DIM(k, dk)                                    \
  {                                                           \
    const float lo = root_box[2 * k], hi = root_box[2 * k + 1]; \
    if (q##k < lo) {                                          \
      dk = (q##k - lo) * (q##k - lo);                         \
      mind += dk;                                             \
    }                                                         \
    if (q##k > hi) {                                          \
      dk = (q##k - hi) * (q##k - hi);                         \
      mind += dk;                                             \
    }                                                         \
  }

enum : uint32_t { M_FETCH = 0, M_NODE = 1, M_DRAIN = 2, M_IDLE = 3, M_WAIT = 4, M_HDONE = 5 };  // (the last two: shared walks)

// Sum over the six dimensions of (gap in grid steps)^2, three packed pairs, saturating at 2^32 - 1
// (v_dot2_u32_u16 ... clamp): compared with radius / step^2 as an integer, no conversion to float.
#define SM_BOXSUM(g01, g23, g45) \
  __builtin_amdgcn_udot2(g45, g45, __builtin_amdgcn_udot2(g23, g23, __builtin_amdgcn_udot2(g01, g01, 0u, true), true), true)

// Hits number fast_cap and beyond of a query (0-based, in traversal order; the first fast_cap go to the query's own
// slots) live in segments of the shared pool whose sizes double (fast_cap, 2*fast_cap, ...), each allocated with
// one atomic when its first hit arrives, so a query with many hits never has to be searched twice.  This is the
// allocation of segment `level` of query `qi`, done by the lane that holds the segment's first hit; returns the
// segment's pool offset, 0xFFFFFFFF when the pool is exhausted (reported as SMHIP_ST_HIT_POOL_FULL).
__device__ __forceinline__ uint32_t AllocOverflowSegment(const Sink &sink, uint32_t qi, uint32_t level) {
  const uint32_t size = sink.fast_cap << level;
  // 64-bit cursor (counters[24..25]): it keeps counting past the end of the pool, so that the
  // host learns the exact demand of a launch that ran out of room
  const unsigned long long base =
      atomicAdd(reinterpret_cast<unsigned long long *>(sink.counters + 24), (unsigned long long)size);
  uint32_t seg = (uint32_t)base;
  if (level == 0) atomicAdd(&sink.counters[1], 1u);
  if (base + size > sink.pool_cap || level >= (uint32_t)kOvfLevels) {
    atomicOr(&sink.counters[3], 1u);
    seg = 0xFFFFFFFFu;
  }
  if (level < (uint32_t)kOvfLevels) sink.ovf_tab[(size_t)qi * kOvfLevels + level] = seg;
  return seg;
}

#ifdef SMHIP_PHASE_PROF  // tools/phase_prof.sh: shader-clock ticks per wave in the phases of the walker
__device__ unsigned long long g_walk_prof[16];  // [8..13]: maxima over the waves; [14] / [15]: first start / last end (wall clock)
#define WALK_T() clock64()
#else
#define WALK_T() 0ull
#endif

// kMode: kWalkExact  -- nanoflann's traversal order and pruning rule on top of the box rule: hits in the reference's
//                       order (the per-stage entry point, and the queries beyond the 5000-hit cap, where the order
//                       decides which hits survive, spatial_index.cc:371-372);
//        kWalkCensus -- exactly nanoflann's node set (reference_walk = 1: the census of its work);
//        kWalkSet    -- the hit SET of a query in whatever order is cheapest.  Every hit of a query with at most
//                       num_nearest_points hits becomes an anchor and the anchors are sorted afterwards
//                       (spatial_index.cc:411-417), so for those queries -- all of them on a 12 Mb reference, 99.4 %
//                       on a 250 Mb one -- only the set matters; the others are searched again in kWalkExact mode
//                       (LaunchSeedLookup).  nanoflann's own rule (mindistsq + cut_dist - dists[idx] <= radius) only
//                       ever prunes subtrees without hits -- its search is exact -- so the box rule alone finds the
//                       same set: the walk carries no dists[] / mindistsq state, a pending visit is a 4-byte
//                       reference instead of a 32-byte record, and it reads QUAD records (layout: BuildQuads below):
//                       one 128-byte record -- one whole fabric request -- holds the boxes of a node's four
//                       GRANDCHILDREN, so a visit is four box tests and descends two levels of the tree for the one
//                       line a binary record costs per level (the launch is bound by the rate of line requests).
enum : int { kWalkExact = 0, kWalkCensus = 1, kWalkSet = 2 };
constexpr int kSetThreads = 128;  // workgroup of the set walk: 10 KB of LDS per wave, eight workgroups per CU
constexpr int kSetStack = 8;  // pending visits per lane kept in LDS by the set walk (deeper: the lane's spill area; 6: +7 %, 4: +14 %)

//
// kShare (set walk of a SMALL batch only -- the rounds of a streaming run): lanes without a query of their own take
// over pending subtrees of the other lanes of their wave.  With a query or fewer per lane a launch lasts as long as the
// walk of its heaviest query, record after record on one lane (a thousand dependent fetches in a dense region while most
// lanes of the chip are through); the set walk is order-free, so any pending reference of a query can be walked by any
// lane as long as the query's hits are numbered in one place: the lane that was handed the query stays its MASTER (its
// row of the scan phase's tables holds the coordinates, the hit count, the slot address and the overflow segment; the
// helpers' scan entries name it as their owner), waits for its helpers when its own part is done (M_WAIT) and then
// writes the query's totals.  Speed only: the same hit set, stored once each.
template <int kMode, int kThreads, bool kShare = false>
__global__ __launch_bounds__(kThreads, 4) void SeedWalkKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                                                               const uint32_t *__restrict__ n_queries_dev,
                                                               uint32_t n_queries_host, Sink sink, float radius,
                                                               float4 *__restrict__ stack_mem,
                                                               const uint32_t *__restrict__ range_end,
                                                               uint32_t range_stride, uint32_t scan_threshold) {
  constexpr bool kStrict = kMode == kWalkCensus;
  // (the launch packs its thresholds into one argument: scan | thin hand-out of shared walks << 8 | hand-out << 16 |
  // queries per wave / 8 up to which that thin hand-out applies << 24)
  uint32_t hand_thr = max((scan_threshold >> 16) & 0xFFu, 1u);
  const uint32_t share_grab = (scan_threshold >> 8) & 0xFFu, share_upto = (scan_threshold >> 24) * 8u;
  scan_threshold &= 0xFFu;
  constexpr bool kSet = kMode == kWalkSet;
  static_assert(!kShare || kSet, "only the order-free walk can be shared between lanes");
  if (n_queries_dev && *n_queries_dev == 0) return;  // (the second search of a batch without a query beyond the cap)
  const uint32_t gl = blockIdx.x * kThreads + threadIdx.x;
  float4 *stk = stack_mem + (size_t)gl * (kStackDepth * 2);
  // (set walk: the same area as 4-byte references, a row per lane: consecutive levels of a lane share a line that
  // stays in L1 / L2; level-major rows and a block per wave were measured: no gain)
  uint32_t *stk32 = reinterpret_cast<uint32_t *>(stk);
  const uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  const uint4 *pts4 = reinterpret_cast<const uint4 *>(ix.leaf_pts);
  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD: speed only,
  // never correctness).  Each XCD starts on its own contiguous eighth of the visiting order so
  // that its private L2 only has to hold that part of the tree.
  // (range_end[(r + 1) * range_stride - 1] = end of range r of the visiting order.)
  const uint32_t n_ranges = range_end ? 8u : 1u;
  const uint32_t first_range = range_end ? (blockIdx.x & 7u) : 0u;
  uint32_t *cursors = sink.counters + 16;
  float root_box[12];  // wave-uniform: kept in scalar registers
#pragma unroll
  for (int k = 0; k < 12; ++k)
    root_box[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ix.bbox[k])));
  // grid of the children's quantised boxes (wave-uniform too): a base per dimension, ONE step for all of them
  float box_base[6];
#pragma unroll
  for (int k = 0; k < 6; ++k)
    box_base[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ix.bbox[12 + k])));
  const float box_step = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ix.bbox[18])));
  const float box_inv =
      __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(box_step > 0.0f ? 1.0f / box_step : 0.0f)));
  // a box is entered iff the sum of its squared per-dimension gaps, in grid steps, is <= box_limit (SM_BOXSUM)
  // (never 2^32 - 1: a saturated sum -- an unused slot of a quad record always is one -- fails the test)
  uint32_t box_limit = 0xFFFFFFFEu;
  if (box_step > 0.0f) {
    const double lim = floor((double)radius / ((double)box_step * (double)box_step)) + 1.0;
    if (lim < 4294967294.0) box_limit = (uint32_t)lim;
  }
  box_limit = (uint32_t)__builtin_amdgcn_readfirstlane((int)box_limit);
  uint32_t mode = M_FETCH;
  float q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
  float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
  float mind = 0;
  uint32_t node = 0, sp = 0, hits = 0, nn = 0, np = 0, qi = 0;
  // the query on the box grid, two dimensions per word like the records: the nearest grid index
  uint32_t qc01 = 0, qc23 = 0, qc45 = 0;
  // top-of-stack cache: two 32-byte entries per lane in LDS, [slot][half][thread]
  __shared__ float4 tos[kSet ? 1 : 2][2][kSet ? 1 : kThreads];
  // pending leaves of this lane's query, in the order they were reached (leaf ordinals: rows 0 .. kFifo - 1), and -- set
  // walk -- the newest pending references of its stack behind them: rows of one array, so that the visit's store of a
  // passing slot is one instruction whichever of the two it goes to
  __shared__ uint32_t lane_rows[kFifo + (kSet ? kSetStack : 0)][kThreads];
  uint32_t(*const fifo)[kThreads] = lane_rows;
  uint32_t(*const lstk)[kThreads] = lane_rows + (kSet ? kFifo : 0);
  __shared__ uint32_t ovf_lds[kSet ? 1 : kThreads];  // scan phase: overflow segment base of each lane's query
  // address of the first hit slot of each lane's query (HitSlots::At(qi, 0): a pointer-table load), taken when the
  // query is handed out: at emit time it is an LDS read -- a global load there made every scan iteration wait for
  // everything in flight (the next window's point loads included)
  __shared__ Hit *slot_base[kSet ? 1 : kThreads];
  // staging of the cooperative node fetch: 64 records x 64 B (quad walk: x 112 B, the used part of a record) per wave
  __shared__ uint4 stage[kThreads / 64][kSet ? 448 : 260];  // (scan phase: 4 160 bytes of tables)
  const uint32_t lane = threadIdx.x & 63u;
  uint4 *stage_w = stage[threadIdx.x >> 6];
  // the wave's [64] slices of the two scan-phase tables above; the quad walk keeps them in the upper part of its
  // staging area (free during a scan phase: the scan's own tables take the lower 4 KB) and a lane's slot address
  // in registers between the phases
  uint32_t *ovf_w = kSet ? reinterpret_cast<uint32_t *>(stage_w + 292) : &ovf_lds[kSet ? 0u : (threadIdx.x & ~63u)];
  Hit **sbase_w = kSet ? reinterpret_cast<Hit **>(stage_w + 260) : &slot_base[kSet ? 0u : (threadIdx.x & ~63u)];
  Hit *my_slots = nullptr;
  uint32_t tcnt = 0, thead = 0;
  uint32_t fifo_cnt = 0;
  uint32_t ovf_base = 0;  // pool offset of the current overflow segment of this lane's query
  uint32_t st_walk = 0, st_walk_lanes = 0, st_scan = 0, st_scan_lanes = 0;  // wave-uniform statistics
  // A load that would otherwise add a dependent round trip to an iteration is issued one
  // iteration early and retires with the next node fetch: the newest spilled stack entry,
  // whenever a pop leaves room in the LDS cache (rf_*).
  float4 rf_a = make_float4(0, 0, 0, 0), rf_b = rf_a;
  bool rf_pending = false;
  // Queries are handed out dynamically, 64 at a time per wave, from one cursor per XCD range
  // (a wave whose own range is exhausted takes from the other ranges), so that neither a lane nor
  // a wave nor an XCD idles while another still has a backlog.
  uint32_t pool_next = 0, pool_end = 0, victim = 0;

  // ... 64 at a time when there is work for every lane of the launch; a SMALL batch (the tail rounds of a streaming
  // run: a hundred chunks, 30 000 queries for 262 144 lanes) is spread thinly instead, 8 to 32 queries per wave: a
  // wave lasts as long as its slowest query, and the slowest of 8 ends well before the slowest of 64 (2.1 -> ms of a
  // 3 ms round: profiles/EXPERIMENTS.md, round 5).  Speed only.
  uint32_t grab = 64u;
  uint32_t mlane = lane, hcnt = 0;  // kShare: the lane that holds this lane's query; helpers at work on this lane's query
  {
    const uint32_t n_waves = gridDim.x * (uint32_t)(kThreads / 64);
    const uint32_t per_wave = (total + n_waves - 1u) / n_waves;
    if (per_wave <= 32u) grab = per_wave <= 8u ? 8u : per_wave <= 16u ? 16u : 32u;
    // (shared walks: the lanes without a query help the others, so fewer queries per wave at a time spread the heavy
    // ones -- the head of the visiting order -- over more waves)
    // (measured, ms per launch at 64 / 300 / 1 200 / 2 000 chunks: 4 at a time 0.51 / 0.88 / 1.90 / 2.79, 8: 0.64 / 0.82 / 1.62 /
    // 2.37, 16: 0.66 / 0.82 / 1.45 / 2.06, 32: 0.67 / 1.22 / 1.41 / 1.81, as unshared: 0.64 / 1.24 / 1.90 / 2.16)
    if (kShare) {
      const uint32_t half = per_wave / 2u;
      grab = half >= 64u ? 64u : half >= 32u ? 32u : half >= 16u ? 16u : half >= 8u ? 8u : 4u;
      if (share_grab && per_wave <= share_upto) grab = share_grab;
    }
  }
  if (lane >= grab) mode = M_IDLE;  // (the other lanes of such a wave never take a query)
  if (grab < 64u) {                 // ... and the two thresholds, which count lanes, shrink with the lanes in use
    if (!kShare) scan_threshold = max(1u, scan_threshold * grab / 64u);  // (shared: the helpers fill the wave)
    hand_thr = max(1u, hand_thr * grab / 64u);
  }
  // the quad fetch's lane constants (set walk): where a lane's eight ds_bpermutes read from, its part of a record as a
  // byte offset, the wave's staging area as an LDS byte address in a scalar register; quads_narrow: the quad array lies
  // below 4 GiB, a record is base + a 32-bit offset (wave-uniform)
  uint32_t bp_addr[8];
  {
    const uint32_t grp_ = lane / 7u;
#pragma unroll
    for (int r = 0; r < 8; ++r) bp_addr[r] = (((uint32_t)r * 9u + grp_) & 63u) * 4u;
  }
  const uint32_t part16 = (lane - 7u * (lane / 7u)) * 16u;
  const uint32_t stage_lds = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)stage_w);
  const bool quads_narrow = kSet && ix.n_quads < (1ull << 25);
  unsigned long long pt_scan = 0, pt_wait = 0, pt_fetchq = 0, pt_visit = 0, pt_scan_setup = 0, pt_scan_load = 0;
  const unsigned long long pt_begin = WALK_T();
#ifdef SMHIP_PHASE_PROF
  const unsigned long long pt_wall_begin = wall_clock64();
#endif
  for (;;) {
    const unsigned long long pt0 = WALK_T();
    if (kShare) {
      // helpers that are through report to their query's lane: its count of helpers, the statistics of their part
      for (uint64_t fin = __ballot(mode == M_HDONE); fin; fin &= fin - 1ull) {
        const int t = __builtin_ctzll(fin);
        const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mlane, t);
        const uint32_t tn = (uint32_t)__builtin_amdgcn_readlane((int)nn, t), tp = (uint32_t)__builtin_amdgcn_readlane((int)np, t);
        if (lane == m) {
          --hcnt;
          nn += tn;
          np += tp;
        }
      }
      if (mode == M_HDONE) {
        mode = M_IDLE;
        mlane = lane;
      }
      if (mode == M_WAIT && hcnt == 0) {  // the last helper of this lane's query is through: the query is complete
        sink.q_count[qi] = hits;
        if (sink.q_nodes) sink.q_nodes[qi] = nn;
        if (sink.q_points) sink.q_points[qi] = np;
        mode = M_FETCH;
      }
      // Lanes with nothing to do (no query left anywhere, or never given one: a thin hand-out) take the OLDEST pending
      // reference -- the largest subtree -- of a lane that has one, the k-th idle lane from the k-th such lane; the
      // reference, the query's lane and its grid coordinates (all a walk needs: the float coordinates are only read
      // in the scan phase, from the master's row) travel through the wave's staging area, which is free between two
      // iterations.
      const uint64_t idle_m = __ballot(mode == M_IDLE);
      const uint64_t rich_m = idle_m ? __ballot(mode == M_NODE && sp >= 1u && sp <= (uint32_t)kSetStack) : 0ull;
      if (rich_m) {
        const uint64_t lt = (1ull << lane) - 1ull;
        const uint32_t n_pairs = min((uint32_t)__popcll(idle_m), (uint32_t)__popcll(rich_m));
        const uint32_t ri = (uint32_t)__popcll(idle_m & lt), rr = (uint32_t)__popcll(rich_m & lt);
        const bool gives = ((rich_m >> lane) & 1ull) && rr < n_pairs, takes = ((idle_m >> lane) & 1ull) && ri < n_pairs;
        if (gives) {
          const uint32_t ref = lstk[0][threadIdx.x];
#pragma unroll
          for (int i = 0; i + 1 < kSetStack; ++i)
            if ((uint32_t)i + 1u < sp) lstk[i][threadIdx.x] = lstk[i + 1][threadIdx.x];
          --sp;
          stage_w[2u * rr] = make_uint4(ref, mlane, qc01, qc23);
          stage_w[2u * rr + 1u] = make_uint4(qc45, 0u, 0u, 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        if (takes) {
          const uint4 ra = stage_w[2u * ri], rb = stage_w[2u * ri + 1u];
          node = ra.x;
          mlane = ra.y;
          qc01 = ra.z;
          qc23 = ra.w;
          qc45 = rb.x;
          sp = 0;
          nn = 0;
          np = 0;
          mode = M_NODE;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        for (uint64_t tk = __ballot(takes); tk; tk &= tk - 1ull) {
          const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mlane, __builtin_ctzll(tk));
          if (lane == m) ++hcnt;
        }
      }
    }
    const bool blocked = fifo_cnt == (uint32_t)kFifo || mode == M_DRAIN;
    const bool active = !blocked && mode != M_IDLE && (!kShare || mode != M_WAIT);
    const uint64_t act = __ballot(active);
    const uint64_t pend = __ballot(fifo_cnt > 0);
    if (act == 0 && pend == 0) break;
    if (act == 0 || (uint32_t)__popcll(__ballot(blocked)) >= scan_threshold) {
      // ---- scan phase (nanoflann.hpp:1350-1369, 391-406): the pending leaves of all lanes are
      // listed in (lane, arrival) order and scanned ONE POINT PER LANE, so the loads of a leaf are
      // one contiguous run and every lane that works tests a point.  A hit's ordinal within its
      // query = the query's hits so far + hits of the same query in earlier leaves of this
      // iteration + its rank among the leaf's hits (ballot), i.e. exactly the order in which
      // searchLevel reports them.
      const uint32_t c = fifo_cnt;
      // (shared walks: only a scan with entries of helpers can see a query in several runs of a window)
      const bool multi = kShare && __ballot(mlane != lane && c > 0) != 0;
      const uint64_t lt = (1ull << lane) - 1ull;
      const uint64_t cb0 = __ballot(c & 1u), cb1 = __ballot(c & 2u), cb2 = __ballot(c & 4u);
      const uint32_t my_first = (uint32_t)__popcll(cb0 & lt) + 2u * (uint32_t)__popcll(cb1 & lt) +
                                4u * (uint32_t)__popcll(cb2 & lt);
      const uint32_t n_ent = (uint32_t)__popcll(cb0) + 2u * (uint32_t)__popcll(cb1) + 4u * (uint32_t)__popcll(cb2);
      // [256 + 8] {first 32-byte slot, owner lane | n << 8 | flat position << 13}; the eight entries after the last
      // are sentinels (flat position: the maximum), so that a window's look-ahead needs no bounds
      uint2 *ents = reinterpret_cast<uint2 *>(stage_w);
      float4 *own_a = reinterpret_cast<float4 *>(stage_w + 132);  // [64] q0..q3 of the owning lane
      uint4 *own_b = stage_w + 196;                                // [64] q4, q5, query id, hits so far
      uint32_t n_points;  // points of all pending leaves (wave-uniform)
      {
        uint32_t tab[kFifo];
#pragma unroll
        for (int e = 0; e < kFifo; ++e) tab[e] = (uint32_t)e < c ? ix.leaf_tab[fifo[e][threadIdx.x]] : 0u;
        // position of every entry's first point in the flat sequence of all pending points
        uint32_t mine_pts = 0;
#pragma unroll
        for (int e = 0; e < kFifo; ++e) mine_pts += tab[e] >> 27;
        const uint32_t incl = (uint32_t)WaveScanAdd((int)mine_pts);
        n_points = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t run = incl - mine_pts;
#pragma unroll
        for (int e = 0; e < kFifo; ++e)
          if ((uint32_t)e < c) {
            const uint32_t nleaf = tab[e] >> 27;
            // {first 16-byte slot of the leaf's block, owner lane | points << 8 | flat position << 13}
            ents[my_first + e] = make_uint2((tab[e] & 0x07FFFFFFu) << 3, (kShare ? mlane : lane) | (nleaf << 8) | (run << 13));
            run += nleaf;
            if (!kStrict) np += nleaf;  // (strict: counted when the leaf was visited)
          }
      }
      if (lane < 8u) ents[n_ent + lane] = make_uint2(0u, 0xFFFFFFFFu);
      // (every lane: the query's coordinates are read back from here after the scan -- they need not stay in
      // registers through it)
      own_a[lane] = make_float4(q0, q1, q2, q3);
      own_b[lane] = make_uint4(__float_as_uint(q4), __float_as_uint(q5), qi, hits);
      if (c > 0 || kShare) {  // (shared walks: a master's row is read by its helpers' entries too)
        ovf_w[lane] = ovf_base;
        if (kSet) sbase_w[lane] = my_slots;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      // The points of all pending leaves form one flat sequence (leaf after leaf in entry order), scanned 64
      // at a time, ONE POINT PER LANE, whatever the leaf boundaries: leaves hold 1..20 points (12.5 on average:
      // fixed groups of 21 lanes per leaf left 35 % of the lanes idle).  The window of an iteration starts in
      // entry eb; the flat positions of the next kWin entries are read as LDS broadcasts and a lane's entry is
      // the number of them at or below its point.  (A window reaches at most the start of entry eb + kWin.)
      constexpr int kWin = 8;
      const uint32_t hits_word = 3;
      pt_scan_setup += WALK_T() - pt0;
      // window -> {this lane's entry record, its entry's index relative to the window's first entry, has a point}
      struct Win {
        uint2 mine;
        uint32_t e_rel;
        bool pt;
      };
      auto lookup = [&](uint32_t eb_, uint32_t p0_) {
        const uint32_t p = p0_ + lane;
        // (an entry's word is flat position << 13 | ...: position <= p iff word < (p + 1) << 13 -- compared as it
        // comes from LDS, no shift per entry; positions stay below 2^19)
        const uint32_t pl = (p + 1u) << 13;
        uint32_t e_rel = 0;
        bool pt = p < n_points;
#pragma unroll
        for (int i = 1; i <= kWin; ++i) {
          const uint32_t y_i = ents[eb_ + (uint32_t)i].y;  // (uniform address; past the end: a sentinel, 0xFFFFFFFF)
          if (i < kWin) e_rel += y_i < pl ? 1u : 0u;
          else pt = pt && !(y_i < pl);  // (a window reaches at most the start of entry eb + kWin)
        }
        return Win{ents[min(eb_ + e_rel, 255u)], e_rel, pt};  // (pt: then t < n, the next entry starts above p)
      };
      uint32_t eb = 0, p0 = 0;
      // the window's point loads (the 32-byte point: v0..v3, then v4, v5, position), from a valid address
      // for the lanes without a point: no branch around a load, so nothing waits for it before its use
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (a 128-bit register tuple: SM_PIN's operand type)
      auto load_points = [&](const Win &w_, uint32_t p0_, u32x4 &rv_, u32x4 &sv_) {
        const uint32_t t_ = p0_ + lane - (w_.mine.y >> 13);
        // (the slot number in 32 bits -- 27 bits of block offset << 3, + t < 32 --, one 64-bit shift-and-add for the address)
        const uint32_t slot_ = w_.pt ? w_.mine.x + t_ : 0u;
        const u32x4 *fh = reinterpret_cast<const u32x4 *>(pts4 + 2 * (size_t)slot_);
        rv_ = fh[0];
        sv_ = fh[1];
      };
      // number of lanes with a point, and the first entry of the next window (from the end of this one: p0 + 64,
      // or the start of entry eb + kWin if that came first; its first entry is the last lane's, or the one after
      // it when the lane had the entry's last point)
      auto advance = [&](const Win &w_, uint32_t p0_) {
        const uint32_t n_ = (w_.mine.y >> 8) & 31u, t_ = p0_ + lane - (w_.mine.y >> 13);
        const uint32_t nl = (uint32_t)__popcll(__ballot(w_.pt));  // >= 1 (a prefix of the lanes)
        const uint32_t e_last = (uint32_t)__builtin_amdgcn_readlane((int)w_.e_rel, 63);
        const uint32_t full = (uint32_t)__builtin_amdgcn_readlane((int)(t_ + 1u == n_ ? 1u : 0u), 63);
        eb += nl == 64u ? e_last + full : (uint32_t)kWin;  // (nl < 64: the window ended at entry eb + kWin, or it was the last)
        return nl;
      };
      auto process = [&](const Win &w_, uint32_t p0_, const u32x4 &rv, const u32x4 &sv) {
        ++st_scan;
        const uint2 mine = w_.mine;
        const uint32_t owner = mine.y & 0xFFu, t = p0_ + lane - (mine.y >> 13);
        const bool pt = w_.pt;
        const uint64_t pb = __ballot(pt);
        st_scan_lanes += (uint32_t)__popcll(pb);
        bool hit = false;
        float res = 0.0f;
        uint32_t oqi = 0, hbase = 0;
        if (pt) {
          const float4 oa = own_a[owner];
          const uint4 ob = own_b[owner];
          oqi = ob.z;
          hbase = ob.w;
          const float f0 = oa.x - __uint_as_float(rv.x), f1 = oa.y - __uint_as_float(rv.y);
          const float f2 = oa.z - __uint_as_float(rv.z), f3 = oa.w - __uint_as_float(rv.w);
          res = f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3; /* 0 + s == s exactly */
          const float f4 = __uint_as_float(ob.x) - __uint_as_float(sv.x);
          res += f4 * f4;
          const float f5 = __uint_as_float(ob.y) - __uint_as_float(sv.y);
          res += f5 * f5;
          hit = res < radius; /* RadiusResultSet::addPoint */
        }
        if (kSet && __ballot(pt && hbase + 64u > sink.fast_cap) == 0) {
          // (round 6) The order-free walk does not care WHICH of its slots a hit of a query takes, only that no two hits
          // take the same: when no query of the window can leave its fast slots in it (64 points at most), a hit's
          // ordinal is simply what an LDS atomic on its query's count returns -- one DS instruction for the ~35 vector
          // instructions of the run logic below (previous owner, run starts, rank inside the run, the run's last lane
          // writing the count back), which stays for the windows that may reach the overflow pool, where a segment's
          // first hit must be found, and for the walks that number hits in the reference's order.  (Hits of a query in
          // several runs of a window -- shared walks -- need nothing special here.)
          uint32_t ord_a = 0;
          if (hit) ord_a = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(&own_b[owner]) + hits_word, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (hit && (ord_a < sink.nnp || sink.store_all)) {
            typedef __attribute__((address_space(1))) u32x4 GlobalU4;
            GlobalU4 *dst = (GlobalU4 *)(uintptr_t)(sbase_w[owner] + ord_a);
            const u32x4 bits = {sv.z, sv.w, __float_as_uint(res), mine.x + t};
            *dst = bits;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
          return;
        }
        const uint64_t hb = __ballot(hit);
        // runs of lanes with the same owner (the entries of an owner are consecutive); ordinal = the owner's
        // hits so far + the hits in the lanes of its run below this one (earlier leaves of the owner in this
        // iteration and the lower points of this leaf)
        // (wave_shr:1: lane l reads lane l - 1; lane 0's value is not used)
        const uint32_t prev_owner = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)owner, 0x138, 0xF, 0xF, false);
        const uint64_t rb = __ballot(pt && (lane == 0 || owner != prev_owner));  // first lanes of the runs
        const uint64_t lowmask = (2ull << lane) - 1ull, below = (1ull << lane) - 1ull;
        const uint32_t run_first = 63u - (uint32_t)__clzll((long long)((rb & lowmask) | 1ull));
        // (shared walks: the entries of a query come from several lanes' FIFOs, so its points may form SEVERAL runs of a
        // window; som = all lanes of the window with a point of this lane's query, one ballot per distinct query)
        uint64_t som = 0;
        if (multi) {
          for (uint64_t todo = rb; todo;) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)owner, __builtin_ctzll(todo));
            const uint64_t m = __ballot(pt && owner == o);
            if (pt && owner == o) som = m;
            todo &= ~m;
          }
        }
        const uint32_t in_run_below =
            multi ? (uint32_t)__popcll(hb & som & below) : (uint32_t)__popcll(hb & below & ~((1ull << run_first) - 1ull));
        const uint32_t ord = hbase + in_run_below;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();  // every lane has read its owner's count
        // the last lane of a run adds the run's hits to the owner's count
        const uint64_t next_is_first_or_idle = (rb | ~pb) >> 1;  // bit L: lane L + 1 starts a run or has no point
        if (multi ? (pt && lane == 63u - (uint32_t)__clzll((long long)som))
                  : (pt && (lane == 63u || ((next_is_first_or_idle >> lane) & 1ull) != 0))) {
          const uint32_t cnt = in_run_below + (hit ? 1u : 0u);
          if (cnt) *(reinterpret_cast<uint32_t *>(&own_b[owner]) + hits_word) = hbase + cnt;
        }
        Hit h;
        h.position = (uint64_t)sv.z | ((uint64_t)sv.w << 32);
        h.dist = res;
        h.slot = mine.x + t;
        const bool emit = hit && (ord < sink.nnp || sink.store_all);
        if (emit && ord < sink.fast_cap) {
          // (a GLOBAL store: through the generic pointer read back from LDS it was a flat store, which the LDS
          // accesses after it have to wait for)
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(1))) u32x4 GlobalU4;
          GlobalU4 *dst = (GlobalU4 *)(uintptr_t)(sbase_w[owner] + ord);
          const u32x4 bits = {(uint32_t)h.position, (uint32_t)(h.position >> 32), __float_as_uint(h.dist), h.slot};
          *dst = bits;
        }
        // Hits beyond the query's own slots (rare on a small reference; MOST hits on a 250 Mb one, where a query
        // averages 800): all lanes of one (query, segment) group store together.  The group's first lane holds its
        // lowest ordinal (ordinals ascend with the lanes of a query's run), so it is the one that allocates the
        // segment when the window holds the segment's first hit; otherwise the segment is the query's current one
        // (ovf_w).  One lane at a time, as this loop was at first, made the 250 Mb search 34 ms instead of 14.
        const bool ov = emit && ord >= sink.fast_cap;
        if (__ballot(ov)) {  // (wave-uniform: a window without such a hit skips all of it)
          const uint32_t k = ord - sink.fast_cap;
          const uint32_t level = OvfLevel(k, sink.fast_cap);
          const uint32_t first = sink.fast_cap * ((1u << level) - 1u);
          for (uint64_t ovm = __ballot(ov); ovm;) {
            const int leader = __builtin_ctzll(ovm);
            const uint32_t l_owner = (uint32_t)__builtin_amdgcn_readlane((int)owner, leader);
            const uint32_t l_level = (uint32_t)__builtin_amdgcn_readlane((int)level, leader);
            const bool mine = ov && owner == l_owner && level == l_level;
            uint32_t seg = 0;
            if (lane == (uint32_t)leader) {
              seg = ovf_w[owner];
              if (k == first) {
                seg = AllocOverflowSegment(sink, oqi, level);
                ovf_w[owner] = seg;
              }
            }
            seg = (uint32_t)__builtin_amdgcn_readlane((int)seg, leader);
            if (mine && seg != 0xFFFFFFFFu) sink.pool[(size_t)seg + (k - first)] = h;
            ovm &= ~__ballot(mine);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      };
      // Two windows in flight, A and B in turn (no register copies between them): while one is processed, the
      // other's entries have been looked up and its point loads are on their way.
      // SM_PIN: an empty statement every use of the window's loaded registers depends on, placed after the OTHER
      // window's loads have been issued.  Without it the scheduler hoists the copy of a point's position words (into
      // the registers of the hit store) to right behind the load's issue, where the wait for it stalls the wave before
      // the current window's arithmetic: the two windows never overlapped.
#define SM_PIN(rv_, sv_) asm volatile("" : "+v"(rv_), "+v"(sv_)::"memory")
      if (n_points > 0) {
        Win wA = lookup(0u, 0u), wB = wA;
        u32x4 rvA, svA, rvB, svB;
        load_points(wA, 0u, rvA, svA);
        for (;;) {
          const uint32_t nlA = advance(wA, p0);
          const bool moreB = p0 + nlA < n_points;
          wB = lookup(eb, p0 + nlA);
          load_points(wB, p0 + nlA, rvB, svB);  // (past the end: no lane has a point, all load slot 0)
          SM_PIN(rvA, svA);
          process(wA, p0, rvA, svA);
          p0 += nlA;
          if (!moreB) break;
          const uint32_t nlB = advance(wB, p0);
          const bool moreA = p0 + nlB < n_points;
          wA = lookup(eb, p0 + nlB);
          load_points(wA, p0 + nlB, rvA, svA);
          SM_PIN(rvB, svB);
          process(wB, p0, rvB, svB);
          p0 += nlB;
          if (!moreA) break;
        }
#undef SM_PIN
      }
      {
        const float4 ra = own_a[lane];
        const uint4 rb = own_b[lane];
        q0 = ra.x; q1 = ra.y; q2 = ra.z; q3 = ra.w;
        q4 = __uint_as_float(rb.x); q5 = __uint_as_float(rb.y);
        hits = rb.w;
      }
      if (c > 0 || kShare) ovf_base = ovf_w[lane];
      fifo_cnt = 0;
      if (kShare && mode == M_DRAIN && mlane != lane) {
        mode = M_HDONE;  // a helper's part of a query is through
      } else if (kShare && mode == M_DRAIN && hcnt > 0) {
        mode = M_WAIT;  // the query's own lane is through, helpers are not
      } else if (mode == M_DRAIN) {  // query complete (plain stores; per-chunk sums: AnchorCountKernel)
        if (!sink.stop_at_cap) {
          sink.q_count[qi] = hits;
          if (sink.q_nodes) sink.q_nodes[qi] = nn;
          if (sink.q_points) sink.q_points[qi] = np;
        }
        mode = M_FETCH;
      } else if (!kSet && sink.stop_at_cap && mode == M_NODE && hits >= sink.nnp) {
        // the first nnp hits in the reference's order are out: the rest of this traversal would only count
        sp = 0;
        tcnt = 0;
        rf_pending = false;
        mode = M_FETCH;
      }
      pt_scan += WALK_T() - pt0;
      continue;
    }
    // ---- walk phase: one node visit per active lane ----
    ++st_walk;
    st_walk_lanes += (uint32_t)__popcll(act);
    // Queries are handed out to SEVERAL idle lanes at a time: a hand-out is two dependent global round trips (the
    // query record, then the table entry of its hit slots) that stall the whole wave, whatever the number of lanes
    // served; lanes that finish wait until hand_thr of them are idle -- or no lane has a node to visit.
    uint64_t need = __ballot(active && mode == M_FETCH);
    if ((uint32_t)__popcll(need) < hand_thr && __ballot(active && mode == M_NODE) != 0) need = 0;
    for (; need;) {
      if (pool_next >= pool_end) {  // wave-uniform: take the next 64 queries of a range
        bool got = false;
        while (victim < n_ranges) {
          const uint32_t xx = (first_range + victim) % n_ranges;
          const uint32_t rb = (range_end && xx) ? range_end[xx * range_stride - 1] : 0u;
          const uint32_t re = range_end ? range_end[(xx + 1) * range_stride - 1] : total;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&cursors[xx], grab);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (base < re - rb) {
            pool_next = rb + base;
            pool_end = min(re, pool_next + grab);
            got = true;
            break;
          }
          ++victim;
        }
        if (!got) {  // nothing left anywhere
          if (active && mode == M_FETCH) mode = M_IDLE;
          break;
        }
      }
      const uint32_t take = min(pool_end - pool_next, (uint32_t)__popcll(need));
      const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
      if (((need >> lane) & 1ull) && rank < take) {
        // records are already permuted into visiting order; the record's last word is the query id
        const uint32_t ri = pool_next + rank;
        const float4 a = *reinterpret_cast<const float4 *>(&qrec[ri]);
        const float4 b = *(reinterpret_cast<const float4 *>(&qrec[ri]) + 1);
        qi = __float_as_uint(b.w);
        q0 = a.x; q1 = a.y; q2 = a.z; q3 = a.w; q4 = b.x; q5 = b.y;
        if (!kSet) {
          d0 = d1 = d2 = d3 = d4 = d5 = 0.0f;
          mind = 0.0f;
          SM_INIT_DIM(0, d0) SM_INIT_DIM(1, d1) SM_INIT_DIM(2, d2) SM_INIT_DIM(3, d3) SM_INIT_DIM(4, d4)
          SM_INIT_DIM(5, d5)
        }
        node = ix.root_ref | kPassTag; sp = 0; tcnt = 0; hits = 0; nn = 0; np = 0;
        if (kSet) my_slots = sink.region.At(qi, 0);
        else sbase_w[lane] = sink.region.At(qi, 0);
        {
          uint32_t gc[6];
#define SM_GRID(k) gc[k] = (uint32_t)fminf(fmaxf(rintf((q##k - box_base[k]) * box_inv), 0.0f), 65535.0f);
          SM_GRID(0) SM_GRID(1) SM_GRID(2) SM_GRID(3) SM_GRID(4) SM_GRID(5)
#undef SM_GRID
          qc01 = gc[0] | (gc[1] << 16); qc23 = gc[2] | (gc[3] << 16); qc45 = gc[4] | (gc[5] << 16);
        }
        mode = M_NODE;
      }
      pool_next += take;
      need = __ballot(active && mode == M_FETCH);
    }
    const unsigned long long pt1 = WALK_T();
    pt_fetchq += pt1 - pt0;
    if (kSet) {
      // quad records: SEVEN lanes fetch the seven used 16-byte parts of one 128-byte record (one line: one request),
      // nine records per instruction, the 64th record with an eighth instruction; instruction r
      // writes 64 x 16 bytes from unit 63 r on, so record s = 9 r + lane / 7 lands as the 112-byte row 7 s .. 7 s + 6
      // (a stride of 28 words: sixteen lanes reading the same part of their rows touch all 64 banks once)
      const uint32_t want = (active && mode == M_NODE && !(node & kLeafTag)) ? (node & kIdxMask) : 0xFFFFFFFFu;
      if (quads_narrow) {
        // The whole fetch as one block of assembly (round 6): as C++ every one of the eight loads came out as select +
        // compare + saveexec + BRANCH + a wave-uniform LDS address computed on the vector ALU and read back with
        // v_readfirstlane + a two-instruction 64-bit address: 91 vector and ~60 scalar instructions and eight branches per
        // visit.  Here: v_cmpx writes the exec mask straight from the distributed reference (-1: nothing to fetch), the
        // address is a 32-bit offset from the array's base in scalar registers (the array is below 4 GiB: quads_narrow),
        // M0 advances by a scalar add: 8 + 16 vector instructions, no branch.  Lane 63 (lane / 7 = 9, part 0) is not
        // masked in instructions 0 .. 6: it fetches part 0 of record 9 r + 9 into unit 63 r + 63 -- the very unit lane 0
        // of instruction r + 1 writes the same sixteen bytes to.  Instruction 7 fetches record 63 alone (lanes 0 .. 6).
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        unsigned long long save;
        asm volatile(
            "ds_bpermute_b32 %0, %9, %17\n\t"
            "ds_bpermute_b32 %1, %10, %17\n\t"
            "ds_bpermute_b32 %2, %11, %17\n\t"
            "ds_bpermute_b32 %3, %12, %17\n\t"
            "ds_bpermute_b32 %4, %13, %17\n\t"
            "ds_bpermute_b32 %5, %14, %17\n\t"
            "ds_bpermute_b32 %6, %15, %17\n\t"
            "ds_bpermute_b32 %7, %16, %17\n\t"
            "s_mov_b64 %8, exec\n\t"
            "s_mov_b32 m0, %20\n\t"
#define SM_QFETCH(k, T)                                   \
            "s_waitcnt lgkmcnt(" #k ")\n\t"                 \
            "v_cmpx_ne_u32_e32 vcc, -1, " T "\n\t"          \
            "v_lshl_add_u32 " T ", " T ", 7, %18\n\t"       \
            "global_load_lds_dwordx4 " T ", %19\n\t"        \
            "s_mov_b64 exec, %8\n\t"                        \
            "s_add_u32 m0, m0, 0x3f0\n\t"
            SM_QFETCH(7, "%0") SM_QFETCH(6, "%1") SM_QFETCH(5, "%2") SM_QFETCH(4, "%3") SM_QFETCH(3, "%4")
            SM_QFETCH(2, "%5") SM_QFETCH(1, "%6")
#undef SM_QFETCH
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_and_b64 exec, %8, 0x7f\n\t"
            "v_cmpx_ne_u32_e32 vcc, -1, %7\n\t"
            "v_lshl_add_u32 %7, %7, 7, %18\n\t"
            "global_load_lds_dwordx4 %7, %19\n\t"
            "s_mov_b64 exec, %8\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&s"(save)
            : "v"(bp_addr[0]), "v"(bp_addr[1]), "v"(bp_addr[2]), "v"(bp_addr[3]), "v"(bp_addr[4]), "v"(bp_addr[5]),
              "v"(bp_addr[6]), "v"(bp_addr[7]), "v"(want), "v"(part16), "s"(ix.quads), "s"(stage_lds)
            : "memory", "vcc", "scc");  // (and M0: a reserved register the compiler sets anew before every use of its own)
      } else {
      const uint32_t grp = lane / 7u, part = lane - 7u * grp;
      uint32_t nsrc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t src = (uint32_t)r * 9u + grp;
        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)want);
        nsrc[r] = (lane < 63u && src < 64u) ? v : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (nsrc[r] != 0xFFFFFFFFu) {
          const uint4 *g = ix.quads + (size_t)nsrc[r] * 8 + part;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                           (__attribute__((address_space(3))) void *)(stage_w + r * 63), 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA loads have landed
      }
    } else {
      const uint32_t want = (active && mode == M_NODE && !(node & kLeafTag)) ? (node & kIdxMask) : 0xFFFFFFFFu;
      uint32_t nsrc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        nsrc[r] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)r * 16u + (lane >> 2)) * 4u), (int)want);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t src = (uint32_t)r * 16u + (lane >> 2);
        if (nsrc[r] != 0xFFFFFFFFu) {
          const uint4 *g = ix.inner + (size_t)nsrc[r] * 4 + ((lane & 3u) ^ ((src >> 2) & 3u));
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                           (__attribute__((address_space(3))) void *)(stage_w + r * 64), 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA loads have landed
    }
    const unsigned long long pt2 = WALK_T();
    pt_wait += pt2 - pt1;
    if (!kSet && rf_pending) {  // the refill issued by the previous pop has landed too: it is the oldest cached entry
      const uint32_t slot = tcnt == 0 ? thead : thead ^ 1u;
      tos[slot][0][threadIdx.x] = rf_a;
      tos[slot][1][threadIdx.x] = rf_b;
      ++tcnt;
      rf_pending = false;
    }
    if (kSet) {
      if (active && mode == M_NODE) {
        // ---- set walk: test both children's boxes; the first that passes is visited next, a second one is left on
        // the stack; leaves go to the FIFO at once.  A reference on the stack has passed its box test.
        uint32_t cur = 0xFFFFFFFFu;  // the node to go on with (none: pop)
        if (!(node & kLeafTag)) {
          // a quad record: {ref[4]} then four boxes of six words {lo01, lo23, lo45, hi01, hi23, hi45}; an unused
          // slot has a box (lo 65535, hi 0 in every dimension) whose sum saturates for any query
          const uint4 *mine = stage_w + lane * 7;
          const uint4 rf = mine[0], p1 = mine[1], p2 = mine[2], p3 = mine[3], p4 = mine[4], p5 = mine[5], p6 = mine[6];
          ++nn;  // (this mode's nodes_visited: the quad records read)
          typedef unsigned short us2 __attribute__((ext_vector_type(2)));
          const us2 k3 = {3, 3};
          const us2 ql01 = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, qc01), k3),
                    ql23 = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, qc23), k3),
                    ql45 = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, qc45), k3);
          const us2 qh01 = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, qc01), k3),
                    qh23 = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, qc23), k3),
                    qh45 = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, qc45), k3);
#define SM_GAP(qlw, hiw, low, qhw)                                                                    \
  __builtin_elementwise_max(__builtin_elementwise_sub_sat(qlw, __builtin_bit_cast(us2, hiw)), \
                            __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, low), qhw))
#define SM_QBOX(lo01, lo23, lo45, hi01, hi23, hi45) \
  SM_BOXSUM(SM_GAP(ql01, hi01, lo01, qh01), SM_GAP(ql23, hi23, lo23, qh23), SM_GAP(ql45, hi45, lo45, qh45))
          const uint32_t s0 = SM_QBOX(p1.x, p1.y, p1.z, p1.w, p2.x, p2.y);
          const uint32_t s1 = SM_QBOX(p2.z, p2.w, p3.x, p3.y, p3.z, p3.w);
          const uint32_t s2 = SM_QBOX(p4.x, p4.y, p4.z, p4.w, p5.x, p5.y);
          const uint32_t s3 = SM_QBOX(p5.z, p5.w, p6.x, p6.y, p6.z, p6.w);
#undef SM_QBOX
#undef SM_GAP
          // a passing slot: a leaf goes to the FIFO at once while there is room, the first inner reference is the
          // record to go on with, everything else waits on the stack.  Without branches (round 6: as an if / else chain
          // every slot cost ~20 vector + ~35 scalar instructions and five to eight branches, all of them taken by some
          // lane of a wave): the FIFO and the LDS part of the stack are rows of ONE array, so a slot is three
          // predicates, one row number and one predicated LDS store; only a lane whose LDS stack is full leaves the
          // straight line (its spill area in memory).
          const uint32_t sp0 = sp;
          uint32_t stk_bits = 0;  // slots that went to the stack (bit i: slot i)
          auto take = [&](uint32_t sum, uint32_t ref, uint32_t bit) {
            const bool pass = sum <= box_limit;
            const bool leaf = (ref & kLeafTag) != 0;
            const bool to_fifo = pass && leaf && fifo_cnt < (uint32_t)kFifo;
            const bool to_cur = pass && !leaf && cur == 0xFFFFFFFFu;
            const bool to_stk = pass && !to_fifo && !to_cur;
            const uint32_t row = to_fifo ? fifo_cnt : (uint32_t)kFifo + sp;
            if (to_fifo || (to_stk && sp < (uint32_t)kSetStack)) lane_rows[row][threadIdx.x] = to_fifo ? (ref & kIdxMask) : ref;
            cur = to_cur ? ref : cur;
            fifo_cnt += to_fifo ? 1u : 0u;
            sp += to_stk ? 1u : 0u;
            stk_bits |= to_stk ? bit : 0u;
          };
          take(s0, rf.x, 1u);
          take(s1, rf.y, 2u);
          take(s2, rf.z, 4u);
          take(s3, rf.w, 8u);
          if (sp > (uint32_t)kSetStack) {
            // (rare: the LDS rows of this lane's stack are full) entries number kSetStack and beyond go to the lane's
            // spill area in memory -- the visit's stacked slots again, in order, from the depth the visit started at
            constexpr uint32_t kCap = (uint32_t)kSetStack + (uint32_t)kStackDepth * 8u;
            uint32_t k = sp0;
            const uint32_t refs[4] = {rf.x, rf.y, rf.z, rf.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (stk_bits & (1u << i)) {
                if (k >= (uint32_t)kSetStack && k < kCap) stk32[k - (uint32_t)kSetStack] = refs[i];
                ++k;
              }
            if (sp > kCap) {  // deeper than the spill area: flagged (SMHIP_ST_STACK_FULL), the entry is lost
              atomicOr(&sink.counters[0], 1u);
              sp = kCap;
            }
          }
        } else {
          cur = node & (kLeafTag | kIdxMask);  // a leaf left over from the previous visit (the FIFO was full)
        }
        // leaves go to the FIFO as they come up (until it is full: the lane then waits for the scan phase with the
        // leaf as its current node); the next inner node is the one to fetch
        for (;;) {
          if (cur == 0xFFFFFFFFu) {
            if (sp == 0) break;
            --sp;
            cur = sp < (uint32_t)kSetStack ? lstk[sp][threadIdx.x] : stk32[sp - (uint32_t)kSetStack];
          }
          if (!(cur & kLeafTag) || fifo_cnt == (uint32_t)kFifo) break;
          fifo[fifo_cnt++][threadIdx.x] = cur & kIdxMask;
          cur = 0xFFFFFFFFu;
        }
        if (cur == 0xFFFFFFFFu) {  // stack empty: traversal of this query complete
          if (fifo_cnt > 0) {
            mode = M_DRAIN;  // its pending leaves are scanned first
          } else if (kShare && mlane != lane) {
            mode = M_HDONE;
          } else if (kShare && hcnt > 0) {
            mode = M_WAIT;
          } else {
            sink.q_count[qi] = hits;
            if (sink.q_nodes) sink.q_nodes[qi] = nn;
            if (sink.q_points) sink.q_points[qi] = np;
            mode = M_FETCH;
          }
        } else {
          node = cur;
        }
      }
    } else if (active && mode == M_NODE) {
      bool pop = true;
      if (node & kLeafTag) {
        // a leaf visit of searchLevel (nanoflann.hpp:1350-1369): the root of a one-leaf tree, or a far
        // child popped from the stack; kPassTag = its box test (made at the parent) was passed
        const uint32_t leaf = node & kIdxMask;
        ++nn;
        if (kStrict) np += ix.leaf_tab[leaf] >> 27;  // the reference evaluates every point of a visited leaf
        if (node & kPassTag) fifo[fifo_cnt++][threadIdx.x] = leaf;
      } else {
        // an inner visit (nanoflann.hpp:1371-1408); hd = {divlow, divhigh, ref1 | feat << 28, ref2}
        const uint32_t rot = (lane >> 2) & 3u;
        const uint4 *mine = stage_w + lane * 4;
        const uint4 hd = mine[0 ^ rot], b0 = mine[1 ^ rot], b1 = mine[2 ^ rot], b2 = mine[3 ^ rot];
        ++nn;
        // Lower bounds of the squared distance from the query to the children's boxes, on the 16-bit
        // grid the boxes are stored on (two dimensions per word, packed u16 arithmetic; the SAME step in every
        // dimension): per dimension the gap in grid steps, g = max(ql - hi, lo - qh, 0) with saturating
        // subtractions; the child is entered iff sum g^2 <= box_limit = floor(radius / step^2) + 1, all in integers.
        // ql / qh = the query's nearest grid index -/+ 3 (saturating): the real per-dimension distance from the
        // query to any point of the box is more than (g + 1) steps when g > 0 (three steps of margin less the
        // rounding of the query's grid index, of the outward-rounded box values fma(k, step, base) and of
        // base itself: a step is 1.5e-5 of the widest extent, an f32 rounding 6e-8 of the value).  So a child with
        // sum g^2 > box_limit >= radius / step^2 has every point at a real squared distance above
        // step^2 * sum (g + 1)^2 >= radius * (1 + 2 / g_max), g_max <= sqrt(box_limit) ~ 2 000: three orders of
        // magnitude more than the f32 rounding of nanoflann's own sum, whose result is therefore >= radius too --
        // the child holds no hit.  b0 = c1.lo[0..5], c1.hi[0..1]; b1 = c1.hi[2..5], c2.lo[0..3];
        // b2 = c2.lo[4..5], c2.hi[0..5]
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define SM_GAP(qlw, hiw, low, qhw)                                                                    \
  __builtin_elementwise_max(__builtin_elementwise_sub_sat(qlw, __builtin_bit_cast(us2, hiw)), \
                            __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, low), qhw))
        const us2 k3 = {3, 3};
        const us2 ql01 = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, qc01), k3),
                  ql23 = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, qc23), k3),
                  ql45 = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, qc45), k3);
        const us2 qh01 = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, qc01), k3),
                  qh23 = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, qc23), k3),
                  qh45 = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, qc45), k3);
        uint32_t s1, s2;  // sums of the squared gaps in grid steps (saturating)
        {
          const us2 g01 = SM_GAP(ql01, b0.w, b0.x, qh01), g23 = SM_GAP(ql23, b1.x, b0.y, qh23),
                    g45 = SM_GAP(ql45, b1.y, b0.z, qh45);
          s1 = SM_BOXSUM(g01, g23, g45);
        }
        {
          const us2 g01 = SM_GAP(ql01, b2.y, b1.z, qh01), g23 = SM_GAP(ql23, b2.z, b1.w, qh23),
                    g45 = SM_GAP(ql45, b2.w, b2.x, qh45);
          s2 = SM_BOXSUM(g01, g23, g45);
        }
#undef SM_GAP
        const uint32_t idx = (hd.z >> 28) & 7u;
        const uint32_t ref1 = hd.z & (kLeafTag | kIdxMask), ref2 = hd.w & (kLeafTag | kIdxMask);
        // q[idx], dists[idx]: a balanced tree of selects on the bits of idx (5 per value)
        const bool i0 = (idx & 1u) != 0, i1 = (idx & 2u) != 0, i2 = (idx & 4u) != 0;
        const float val = i2 ? (i0 ? q5 : q4) : (i1 ? (i0 ? q3 : q2) : (i0 ? q1 : q0));
        const float dst = i2 ? (i0 ? d5 : d4) : (i1 ? (i0 ? d3 : d2) : (i0 ? d1 : d0));
        const float divlow = __uint_as_float(hd.x), divhigh = __uint_as_float(hd.y);
        const float diff1 = val - divlow, diff2 = val - divhigh;
        const bool low_side = (diff1 + diff2) < 0;
        const uint32_t next = low_side ? ref1 : ref2, other = low_side ? ref2 : ref1;
        const bool in_next = (low_side ? s1 : s2) <= box_limit, in_other = (low_side ? s2 : s1) <= box_limit;
        const float dcut = low_side ? diff2 : diff1;
        const float cut = dcut * dcut;
        const float nm = mind + cut - dst;
        if (nm <= radius && (kStrict || in_other)) {  // mindistsq * epsError <= worstDist(), epsError == 1
          if (sp + tcnt < (uint32_t)kStackDepth) {
            const uint32_t slot = thead ^ 1u;
            if (tcnt == 2) {  // spill the older cached entry
              stk[2 * sp] = tos[slot][0][threadIdx.x];
              stk[2 * sp + 1] = tos[slot][1][threadIdx.x];
              ++sp;
            } else {
              ++tcnt;
            }
            // the entry is the current dists with dists[idx] = cut: written whole, then one word
            // replaced (words 2..7 of the entry are dists[0..5])
            tos[slot][0][threadIdx.x] = make_float4(__uint_as_float(other | (in_other ? kPassTag : 0u)), nm, d0, d1);
            tos[slot][1][threadIdx.x] = make_float4(d2, d3, d4, d5);
            const uint32_t word = idx + 2u;
            reinterpret_cast<float *>(&tos[slot][word >> 2][threadIdx.x])[word & 3u] = cut;
            thead = slot;
          } else {
            atomicOr(&sink.counters[0], 1u);
          }
        }
        if (next & kLeafTag) {  // the near child is a leaf: visited here, it has no record
          if (kStrict || in_next) {
            ++nn;
            if (kStrict) np += ix.leaf_tab[next & kIdxMask] >> 27;
            if (in_next) fifo[fifo_cnt++][threadIdx.x] = next & kIdxMask;
          }
        } else if (kStrict || in_next) {
          node = next | kPassTag;
          pop = false;
        }
      }
      if (pop) {
        if (sp + tcnt == 0) {  // traversal of this query complete
          if (fifo_cnt > 0) {
            mode = M_DRAIN;  // its pending leaves are scanned first
          } else {
            if (!sink.stop_at_cap) {
              sink.q_count[qi] = hits;
              if (sink.q_nodes) sink.q_nodes[qi] = nn;
              if (sink.q_points) sink.q_points[qi] = np;
            }
            mode = M_FETCH;
          }
        } else {  // return from searchLevel into the innermost pending visit
          float4 a, b;
          if (tcnt > 0) {
            a = tos[thead][0][threadIdx.x];
            b = tos[thead][1][threadIdx.x];
            --tcnt;
            thead ^= 1u;
          } else {
            --sp;
            a = stk[2 * sp];
            b = stk[2 * sp + 1];
          }
          node = __float_as_uint(a.x);
          mind = a.y;
          d0 = a.z; d1 = a.w; d2 = b.x; d3 = b.y; d4 = b.z; d5 = b.w;
          if (tcnt == 0 && sp > 0) {  // cache empty: refill from the spill area, without waiting
            // (not at tcnt == 1: a pop / push pair around the top would then spill and refill every time)
            --sp;
            rf_a = stk[2 * sp];
            rf_b = stk[2 * sp + 1];
            rf_pending = true;
          }
        }
      }
    }
  }
#ifdef SMHIP_PHASE_PROF
  if (kSet && lane == 0) {  // (the production walk)
    const unsigned long long pt_all = clock64() - pt_begin;
    atomicAdd(&g_walk_prof[0], pt_all);
    atomicAdd(&g_walk_prof[1], pt_scan);
    atomicAdd(&g_walk_prof[2], pt_fetchq);
    atomicAdd(&g_walk_prof[3], pt_wait);
    atomicAdd(&g_walk_prof[4], pt_all - pt_scan - pt_fetchq - pt_wait);  // visit logic
    atomicAdd(&g_walk_prof[5], 1ull);
    atomicAdd(&g_walk_prof[6], pt_scan_setup);
    atomicAdd(&g_walk_prof[7], pt_scan_load);
    if (kShare) {  // (the maxima: of the shared walks only -- the census launch of a fresh index would hide a small batch's)
      atomicMax(&g_walk_prof[8], pt_all);
      atomicMax(&g_walk_prof[9], pt_scan);
      atomicMax(&g_walk_prof[10], pt_fetchq);
      atomicMax(&g_walk_prof[11], pt_wait);
      atomicMax(&g_walk_prof[12], (unsigned long long)st_walk);
      atomicMax(&g_walk_prof[13], (unsigned long long)st_scan);
      atomicMin(&g_walk_prof[14], pt_wall_begin);
      atomicMax(&g_walk_prof[15], (unsigned long long)wall_clock64());
    }
  }
#endif
  (void)pt_visit;
  (void)pt_scan_setup;
  (void)pt_scan_load;
  (void)pt_scan;
  (void)pt_wait;
  (void)pt_fetchq;
  (void)pt_begin;
  if (lane == 0) {  // occupancy statistics of the two phases (smhip_debug_counters)
    unsigned long long *st = reinterpret_cast<unsigned long long *>(sink.counters + 8);
    atomicAdd(&st[0], (unsigned long long)st_walk);
    atomicAdd(&st[1], (unsigned long long)st_walk_lanes);
    atomicAdd(&st[2], (unsigned long long)st_scan);
    atomicAdd(&st[3], (unsigned long long)st_scan_lanes);
  }
}
#undef SM_INIT_DIM

// Query windows of every chunk -> 32-byte query records (spatial_index.cc:351-368: seed k of a
// chunk is the window starting at feature (k+1)*step).
__global__ void QueryExpandKernel(const float *__restrict__ feat, uint32_t feat_stride,
                                  const uint32_t *__restrict__ q_off, int step, QueryRec *__restrict__ qrec) {
  const uint32_t c = blockIdx.x;
  const uint32_t q0 = q_off[c], nq = q_off[c + 1] - q0;
  const float *f = feat + (size_t)c * feat_stride;
  for (uint32_t k = threadIdx.x; k < nq; k += blockDim.x) {
    QueryRec r;
    const float *p = f + (size_t)(k + 1) * step;
#pragma unroll
    for (int d = 0; d < 6; ++d) r.q[d] = p[d];
    r.chunk = c;
    r.k = k;
    qrec[q0 + k] = r;
  }
}

__global__ void QueryExpandRawKernel(const float *__restrict__ queries, uint32_t n, QueryRec *__restrict__ qrec) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    QueryRec r;
#pragma unroll
    for (int d = 0; d < 6; ++d) r.q[d] = queries[(size_t)k * 6 + d];
    r.chunk = 0;
    r.k = k;
    qrec[k] = r;
  }
}

// One thread per leaf: writes the leaf's point block (layout in the file header) and its exact
// tight box {lo[6], hi[6]}.  v[d] = values[vind[i] + d] (sigmap_adaptor.h:94-97: k-d point i is the
// window values[i..i+dim)), position = Point.position of the window's first element.
__global__ void LeafGatherKernel(const float *__restrict__ values, const uint64_t *__restrict__ positions,
                                 const uint32_t *__restrict__ vind, const uint32_t *__restrict__ leaf_left,
                                 const uint32_t *__restrict__ leaf_tab, uint64_t n_leaves,
                                 float *__restrict__ leaf_pts, uint32_t *__restrict__ slot_vind,
                                 float *__restrict__ leaf_box) {
  float4 *o = reinterpret_cast<float4 *>(leaf_pts);
  for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_leaves;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t tab = leaf_tab[k];
    const uint32_t left = leaf_left[k], n = tab >> 27, boff = (tab & kIdxMask) << 3;  // (first 32-byte slot)
    float lo[6], hi[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      lo[d] = 3.402823466e38f;
      hi[d] = -3.402823466e38f;
    }
    for (uint32_t t = 0; t < n; ++t) {
      const uint32_t v = vind[left + t];
      const uint64_t pos = positions[v];
      float p[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        p[d] = values[(uint64_t)v + d];
        lo[d] = fminf(lo[d], p[d]);
        hi[d] = fmaxf(hi[d], p[d]);
      }
      slot_vind[(size_t)boff + t] = v;  // Hit.slot -> original point index
      o[2 * ((size_t)boff + t)] = make_float4(p[0], p[1], p[2], p[3]);
      o[2 * ((size_t)boff + t) + 1] = make_float4(p[4], p[5], __uint_as_float((uint32_t)pos),
                                                  __uint_as_float((uint32_t)(pos >> 32)));
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      leaf_box[k * 12 + d] = lo[d];
      leaf_box[k * 12 + 6 + d] = hi[d];
    }
  }
}

__device__ __forceinline__ const float *BoxOf(uint32_t ref, const float *leaf_box, const float *inner_box) {
  return (ref & kLeafTag) ? leaf_box + (size_t)(ref & kIdxMask) * 12 : inner_box + (size_t)(ref & kIdxMask) * 12;
}

// Exact boxes of the inner nodes of one tree level (deepest level first): the union of the children's.
__global__ void InnerBoxKernel(const uint4 *__restrict__ inner, const uint32_t *__restrict__ level_nodes,
                               uint32_t count, const float *__restrict__ leaf_box, float *__restrict__ inner_box) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
    const uint64_t i = level_nodes[t];
    const uint4 hd = inner[i * 4];
    const float *a = BoxOf(hd.z, leaf_box, inner_box), *b = BoxOf(hd.w, leaf_box, inner_box);
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      inner_box[i * 12 + d] = fminf(a[d], b[d]);
      inner_box[i * 12 + 6 + d] = fmaxf(a[6 + d], b[6 + d]);
    }
  }
}

// grid[0..5] = base, grid[6..11] = step: the root's tight box on a grid of ONE step for all dimensions (the
// widest extent cut into 65534 steps), so that the walker can add squared gaps of different dimensions as integers.
__global__ void BoxGridKernel(uint32_t root_ref, const float *__restrict__ leaf_box,
                              const float *__restrict__ inner_box, float *__restrict__ grid) {
  const int d = threadIdx.x;
  if (d >= 6) return;
  const float *r = BoxOf(root_ref, leaf_box, inner_box);
  float widest = 0.0f;
  for (int k = 0; k < 6; ++k) widest = fmaxf(widest, r[6 + k] - r[k]);
  grid[d] = r[d];
  grid[6 + d] = widest / 65534.0f;
}

// The children's boxes of every inner record, on the grid, rounded outwards: the quantised lo
// dequantises (with the walker's own expression) to a value <= the exact lo, hi to one >= the exact hi.
__global__ void PackBoxesKernel(uint4 *__restrict__ inner, uint64_t n_inner, const float *__restrict__ leaf_box,
                                const float *__restrict__ inner_box, const float *__restrict__ grid,
                                uint32_t *__restrict__ error_flag) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_inner;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 hd = inner[i * 4];
    uint32_t q[24];  // child1 lo[6], hi[6], child2 lo[6], hi[6]
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float *bx = BoxOf(c ? hd.w : hd.z, leaf_box, inner_box);
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const float base = grid[d], step = grid[6 + d], lo = bx[d], hi = bx[6 + d];
        int ql = 0, qh = 0;
        if (step > 0.0f) {
          ql = (int)fminf(fmaxf(floorf((lo - base) / step), 0.0f), 65535.0f);
          qh = (int)fminf(fmaxf(ceilf((hi - base) / step), 0.0f), 65535.0f);
        }
        while (ql > 0 && __fmaf_rn((float)ql, step, base) > lo) --ql;
        while (qh < 65535 && __fmaf_rn((float)qh, step, base) < hi) ++qh;
        if (!(__fmaf_rn((float)ql, step, base) <= lo) || !(__fmaf_rn((float)qh, step, base) >= hi)) bad = true;
        q[c * 12 + d] = (uint32_t)ql;
        q[c * 12 + 6 + d] = (uint32_t)qh;
      }
    }
    if (bad) atomicOr(error_flag, 1u);
    uint32_t w[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = q[2 * k] | (q[2 * k + 1] << 16);
    inner[i * 4 + 1] = make_uint4(w[0], w[1], w[2], w[3]);
    inner[i * 4 + 2] = make_uint4(w[4], w[5], w[6], w[7]);
    inner[i * 4 + 3] = make_uint4(w[8], w[9], w[10], w[11]);
  }
}

// ---- quad records (the set walk's own node array, derived from `inner`).  The inner nodes at even depth -- the root,
// and every inner grandchild of such a node -- are the HEADS; a head's record lists up to four SLOTS, the nodes two
// levels below it (a child that is a leaf takes one slot itself), each with its tight box exactly as the parents'
// binary records hold it and a reference: the quad number of a grandchild that is inner (a head again), or
// kLeafTag | leaf ordinal.  Quads are numbered in pre-order of their heads.  128 bytes: {ref[4]}, 4 x {lo01, lo23,
// lo45, hi01, hi23, hi45} (two u16 grid coordinates per word), 16 bytes unused.
__global__ void QuadMarkKernel(const uint4 *__restrict__ inner, uint64_t n_inner, uint32_t *__restrict__ head,
                               uint32_t round, uint32_t *__restrict__ changed) {
  for (uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; x < n_inner; x += (uint64_t)gridDim.x * blockDim.x) {
    if (head[x] != round) continue;
    const uint4 hd = inner[x * 4];
    const uint32_t c[2] = {hd.z & (kLeafTag | kIdxMask), hd.w & (kLeafTag | kIdxMask)};
    for (int i = 0; i < 2; ++i) {
      if (c[i] & kLeafTag) continue;
      const uint4 hc = inner[(uint64_t)c[i] * 4];
      const uint32_t g[2] = {hc.z & (kLeafTag | kIdxMask), hc.w & (kLeafTag | kIdxMask)};
      for (int j = 0; j < 2; ++j)
        if (!(g[j] & kLeafTag)) {
          head[g[j]] = round + 1u;
          *changed = 1u;
        }
    }
  }
}

__global__ void QuadFlagKernel(const uint32_t *__restrict__ head, uint64_t n_inner, uint32_t *__restrict__ flag) {
  for (uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; x < n_inner; x += (uint64_t)gridDim.x * blockDim.x)
    flag[x] = head[x] ? 1u : 0u;
}

// box of child `which` (0 / 1) of a binary record as six words {lo01, lo23, lo45, hi01, hi23, hi45}
__device__ __forceinline__ void ChildBoxWords(const uint4 *rec, int which, uint32_t *w) {
  const uint4 b0 = rec[1], b1 = rec[2], b2 = rec[3];
  if (which == 0) {
    w[0] = b0.x; w[1] = b0.y; w[2] = b0.z; w[3] = b0.w; w[4] = b1.x; w[5] = b1.y;
  } else {
    w[0] = b1.z; w[1] = b1.w; w[2] = b2.x; w[3] = b2.y; w[4] = b2.z; w[5] = b2.w;
  }
}

__global__ void QuadFillKernel(const uint4 *__restrict__ inner, uint64_t n_inner, const uint32_t *__restrict__ flag,
                               const uint32_t *__restrict__ quad_of, uint4 *__restrict__ quads) {
  for (uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; x < n_inner; x += (uint64_t)gridDim.x * blockDim.x) {
    if (!flag[x]) continue;
    uint32_t w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // unused slot: no reference, a box nothing is close to
      w[k] = 0xFFFFFFFFu;
      w[4 + 6 * k] = w[5 + 6 * k] = w[6 + 6 * k] = 0xFFFFFFFFu;
    }
    int slot = 0;
    const uint4 *rx = inner + x * 4;
    const uint4 hd = rx[0];
    const uint32_t c[2] = {hd.z & (kLeafTag | kIdxMask), hd.w & (kLeafTag | kIdxMask)};
    for (int i = 0; i < 2; ++i) {
      if (c[i] & kLeafTag) {
        w[slot] = c[i];
        ChildBoxWords(rx, i, w + 4 + 6 * slot);
        ++slot;
        continue;
      }
      const uint4 *rc = inner + (uint64_t)c[i] * 4;
      const uint4 hc = rc[0];
      const uint32_t g[2] = {hc.z & (kLeafTag | kIdxMask), hc.w & (kLeafTag | kIdxMask)};
      for (int j = 0; j < 2; ++j) {
        w[slot] = (g[j] & kLeafTag) ? g[j] : quad_of[g[j]];
        ChildBoxWords(rc, j, w + 4 + 6 * slot);
        ++slot;
      }
    }
    uint4 *o = quads + (uint64_t)quad_of[x] * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  }
}

// ---- visiting order of the queries.  Bucket = ordinal (tree order) of the leaf that contains the
// query point (a stackless best-child descent) >> shift; the buckets are cut into 8 equal ranges
// (one per XCD), and inside a range the queries are visited in 16 cost classes, most expensive
// first, each class in bucket order.  The class is |q|^2 (the windows are z-scored, so density -
// and with it the number of nodes a search visits - falls with the distance from the origin:
// measured correlation -0.8); visiting the cheap queries last keeps the tail of the launch, when
// lanes run out of work one by one, short.  Counting sort on (range, class, bucket in range).
constexpr uint32_t kCostClasses = 16;
constexpr uint32_t kLeafBucketBits = 16;  // resolution of the leaf order inside a class
// 1/16 ... 15/16 quantiles of |q|^2 over the query windows of simulated reads (speed only: a
// different signal distribution changes the sizes of the classes, never a result)
__constant__ float kClassThreshold[kCostClasses - 1] = {1.81f, 2.45f, 2.98f, 3.47f, 3.93f, 4.40f, 4.88f, 5.37f,
                                                        5.90f, 6.46f, 7.10f, 7.85f, 8.76f, 9.98f, 11.95f};
__global__ void QueryKeyKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                               const uint32_t *__restrict__ n_queries_dev, uint32_t n_queries_host,
                               uint32_t shift, uint32_t per_range, uint32_t classes,
                               uint32_t *__restrict__ keys, uint32_t *__restrict__ hist) {
  const uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const QueryRec r = qrec[q];
    // The descent stops as soon as the bucket is decided: the leaves below a node are the ordinals [lo, hi) --
    // leaves are numbered in tree order and inner records in pre-order, so the first child's subtree holds
    // (second child's inner index - this index) leaves, or (second child's leaf ordinal - lo) -- and once lo and
    // hi - 1 share a bucket the remaining ~6 of ~25 dependent record reads are not needed (same keys as a full
    // descent).
    uint32_t node = ix.root_ref, lo = 0, hi = (uint32_t)ix.n_leaves;
    while (!(node & kLeafTag) && (lo >> shift) != ((hi - 1u) >> shift)) {
      const uint4 rec = ix.inner[(size_t)node * 4];
      const float val = r.q[(rec.z >> 28) & 7u];
      const float diff1 = val - __uint_as_float(rec.x), diff2 = val - __uint_as_float(rec.y);
      const uint32_t ref2 = rec.w & (kLeafTag | kIdxMask);
      const uint32_t n1 = (ref2 & kLeafTag) ? (ref2 & kIdxMask) - lo : ref2 - node;  // leaves below the first child
      if ((diff1 + diff2) < 0) {
        hi = lo + n1;
        node = rec.z & (kLeafTag | kIdxMask);
      } else {
        lo += n1;
        node = ref2;
      }
    }
    const uint32_t lb = ((node & kLeafTag) ? (node & kIdxMask) : lo) >> shift, range = lb / per_range;
    const float r2 = r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3] + r.q[4] * r.q[4] +
                     r.q[5] * r.q[5];
    uint32_t cls = 0;  // number of thresholds <= r2 (binary search over 15 sorted values)
#pragma unroll
    for (uint32_t stepw = 8; stepw > 0; stepw >>= 1)
      if (r2 >= kClassThreshold[cls + stepw - 1]) cls += stepw;
    cls = cls * classes / kCostClasses;  // (SMHIP_CLASSES < 16: coarser classes, for experiments)
    const uint32_t key = (range * classes + cls) * per_range + (lb - range * per_range);
    keys[q] = key;
    atomicAdd(&hist[key], 1u);
  }
}

__global__ void QueryScatterKernel(const uint32_t *__restrict__ n_queries_dev, uint32_t n_queries_host,
                                   const uint32_t *__restrict__ keys, uint32_t *__restrict__ cursor,
                                   const QueryRec *__restrict__ qrec, QueryRec *__restrict__ sorted) {
  const uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    QueryRec r = qrec[q];
    r.k = q;  // the walker needs the query id; (chunk, k) stay available in the unsorted array
    sorted[atomicAdd(&cursor[keys[q]], 1u)] = r;
  }
}

constexpr int kLaneBlocksPerCU = 4;  // 40 KB of LDS per 256-thread workgroup
constexpr int kScanThreshold = 4;    // blocked lanes of a wave that trigger a scan phase
constexpr int kHandoutLanes = 4;     // idle lanes of a wave that trigger a hand-out of queries (1 / 2 / 4 / 8 / 16: 5.68 / 5.61 / 5.57 / 5.57 / 5.59 ms)
constexpr uint32_t kShareMaxChunks = 4096;  // batches up to this many chunks (~ four queries per lane of the launch) share walks
constexpr int kShareGrab = 0, kShareGrabUpto = 0;  // thin hand-out of shared walks: lanes per wave, up to this many queries per wave (0: as unshared)
constexpr int kSetScanThreshold = 8;  // ... of the set walk (2 / 4 / 8 / 16: 6.52 / 6.44 / 6.35 / 6.44 ms)

// The queries the set walk found more than num_nearest_points hits for are listed for a second search in the
// reference's traversal order: their records from the unsorted array (query i at [i]), the last word set to the
// query id like QueryScatterKernel does.
__global__ void OverCapCollectKernel(const QueryRec *__restrict__ qrec, const uint32_t *__restrict__ n_queries_dev,
                                     const uint32_t *__restrict__ q_count, uint32_t nnp, QueryRec *__restrict__ list,
                                     uint32_t *__restrict__ n_list) {
  const uint32_t total = *n_queries_dev;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (q_count[i] <= nnp) continue;
    QueryRec r = qrec[i];
    r.k = i;
    list[atomicAdd(n_list, 1u)] = r;
  }
}

// ---- Queries beyond the hit cap without a second search.  The set walk has every hit of such a query (H >
// num_nearest_points of them, in the query's slots and overflow segments); the reference keeps the FIRST
// num_nearest_points in its traversal order (spatial_index.cc:371-372).  That order is a depth-first walk that takes
// the child on the query's side of the split first (nanoflann.hpp:1384-1399: diff1 + diff2 < 0 -> child1) and scans a
// leaf's points in index order, so a hit's place in it is decided by the root-to-leaf path of its leaf: the string
// of "the far child was taken" bits, most significant first -- paths of different leaves part at some node, where
// the near one comes first -- and, inside a leaf, by the point's slot.  So: (A) every hit gets its leaf's path
// string as a 64-bit key, left aligned (a descent by leaf ordinal: the first child's subtree of an inner record
// holds second-child-index - own-index leaves), written over the hit's position word; (B) a radix select finds
// the key of the num_nearest_points-th hit; (C) the hits at or below it that lie beyond the first
// num_nearest_points records change places with the records they displace (the kept hits' ORDER does not matter: the
// anchors are sorted afterwards); (D) the kept records get their positions back from the point blocks.  One
// workgroup per query, everything in place.  A tree deeper than 64 levels does not fit the key: such a query is
// handed on to the search in the reference's order (list2), as before.
constexpr uint32_t kSelThreads = 256;
constexpr uint32_t kSelMaxKeep = 8192;  // room for the displaced records' indices; a larger num_nearest_points: the second search

__global__ void SlotLeafKernel(const uint32_t *__restrict__ leaf_tab, uint64_t n_leaves, uint32_t *__restrict__ slot_leaf) {
  for (uint64_t l = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; l < n_leaves; l += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t w = leaf_tab[l], n = w >> 27;
    const uint64_t b0 = w & 0x07FFFFFFu;  // first block of eight 32-byte slots
    for (uint32_t b = 0; b < (n + 7u) / 8u; ++b) slot_leaf[b0 + b] = (uint32_t)l;
  }
}

__device__ __forceinline__ Hit *HitAt(const Sink &sink, uint32_t qi, uint32_t i) {
  if (i < sink.fast_cap) return sink.region.At(qi, i);
  const uint32_t k = i - sink.fast_cap;
  const uint32_t level = OvfLevel(k, sink.fast_cap);
  const uint32_t first = sink.fast_cap * ((1u << level) - 1u);
  return sink.pool + (size_t)sink.ovf_tab[(size_t)qi * kOvfLevels + level] + (k - first);
}

// the path string of leaf `leaf` for query q (see above); false: deeper than 64 levels
__device__ __forceinline__ bool PathKey(const IndexView &ix, const QueryRec &r, uint32_t leaf, uint64_t &key, uint32_t &depth_out) {
  uint32_t node = ix.root_ref, lo = 0, depth = 0;
  uint64_t k = 0;
  while (!(node & kLeafTag)) {
    if (depth == 64u) return false;
    const uint4 rec = ix.inner[(size_t)node * 4];
    const uint32_t feat = (rec.z >> 28) & 7u;
    const float val = feat == 0 ? r.q[0] : feat == 1 ? r.q[1] : feat == 2 ? r.q[2] : feat == 3 ? r.q[3] : feat == 4 ? r.q[4] : r.q[5];
    const float diff1 = val - __uint_as_float(rec.x), diff2 = val - __uint_as_float(rec.y);
    const bool near_first = (diff1 + diff2) < 0;  // nanoflann.hpp:1384
    const uint32_t ref1 = rec.z & (kLeafTag | kIdxMask), ref2 = rec.w & (kLeafTag | kIdxMask);
    const uint32_t n1 = (ref2 & kLeafTag) ? (ref2 & kIdxMask) - lo : ref2 - node;  // leaves below the first child
    const bool go_first = leaf < lo + n1;
    k = (k << 1) | (go_first != near_first ? 1ull : 0ull);
    node = go_first ? ref1 : ref2;
    if (!go_first) lo += n1;
    ++depth;
  }
  key = depth ? k << (64u - depth) : 0ull;
  depth_out = depth;
  return true;
}

__global__ __launch_bounds__(kSelThreads) void OverCapSelectKernel(IndexView ix, const uint32_t *__restrict__ slot_leaf,
                                                                   const QueryRec *__restrict__ list,
                                                                   const uint32_t *__restrict__ n_list, Sink sink,
                                                                   QueryRec *__restrict__ list2, uint32_t *__restrict__ n_list2) {
  __shared__ uint32_t hist[256];
  __shared__ uint16_t posB[kSelMaxKeep];  // (indices below num_nearest_points <= 8192)
  __shared__ uint32_t eq_slot[32];
  __shared__ uint32_t s_bin, s_rem, s_neq, s_depth, s_wave[2][kSelThreads / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t total = *n_list;
  const uint4 *pts4 = reinterpret_cast<const uint4 *>(ix.leaf_pts);
  auto position_of = [&](uint32_t slot) {
    const uint4 v = pts4[2 * (size_t)slot + 1];  // {v4, v5, position lo, position hi}
    return (uint64_t)v.z | ((uint64_t)v.w << 32);
  };
  for (uint32_t it = blockIdx.x; it < total; it += gridDim.x) {
    const QueryRec r = list[it];
    const uint32_t qi = r.k, H = sink.q_count[qi], keep = sink.nnp;
    __syncthreads();  // the previous query is done with the shared arrays
    if (tid == 0) s_depth = 0;
    // a hit the pool had no room for (the batch is flagged and searched again with a larger pool): nothing to do here
    bool missing = false;
    if (H > sink.fast_cap && tid < (uint32_t)kOvfLevels) {
      const uint32_t last_level = OvfLevel(H - 1u - sink.fast_cap, sink.fast_cap);
      missing = last_level >= (uint32_t)kOvfLevels ||
                (tid <= last_level && sink.ovf_tab[(size_t)qi * kOvfLevels + tid] == 0xFFFFFFFFu);
    }
    if (__syncthreads_or(missing ? 1 : 0)) continue;
    // ---- A: the keys.  A thread takes a contiguous run of the hits: consecutive hits mostly share their leaf (a
    // leaf's hits are emitted together), and with it the key
    const uint32_t run = (H + kSelThreads - 1u) / kSelThreads;
    const uint32_t i0 = min(H, tid * run), i1 = min(H, i0 + run);
    bool bad = false;
    {
      uint32_t last_leaf = 0xFFFFFFFFu, max_depth = 0;
      uint64_t last_key = 0;
      for (uint32_t i = i0; i < i1; ++i) {
        Hit *hp = HitAt(sink, qi, i);
        const uint32_t leaf = slot_leaf[hp->slot >> 3];
        if (leaf != last_leaf) {
          uint32_t depth = 0;
          if (!PathKey(ix, r, leaf, last_key, depth)) bad = true;
          max_depth = max(max_depth, depth);
          last_leaf = leaf;
        }
        hp->position = last_key;
      }
      if (max_depth) atomicMax(&s_depth, max_depth);
    }
    if (__syncthreads_or(bad ? 1 : 0)) {  // a path beyond 64 levels: the positions back, the query to the ordered search
      for (uint32_t i = i0; i < i1; ++i) {
        Hit *hp = HitAt(sink, qi, i);
        hp->position = position_of(hp->slot);
      }
      if (tid == 0) list2[atomicAdd(n_list2, 1u)] = r;
      continue;
    }
    // ---- B: the key of the keep-th hit in key order (ties: the hits of one leaf), eight bits per pass
    // (no key has a bit below its path's length: with paths of at most s_depth levels the passes beyond
    // ceil(s_depth / 8) would all pick digit 0)
    uint64_t prefix = 0;
    uint32_t remaining = keep;
    const int n_pass = (int)max((s_depth + 7u) / 8u, 1u);
    for (int pass = 0; pass < n_pass; ++pass) {
      const int shift = 56 - 8 * pass;
      hist[tid] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < H; i += 4 * kSelThreads) {  // (four loads in flight)
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = HitAt(sink, qi, min(i + (uint32_t)u * kSelThreads, H - 1u))->position;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i + (uint32_t)u * kSelThreads < H && (pass == 0 || (key[u] >> (shift + 8)) == prefix))
            atomicAdd(&hist[(uint32_t)(key[u] >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          c[k] = hist[4 * tid + k];
          tot += c[k];
        }
        uint32_t before = (uint32_t)WaveScanAdd((int)tot) - tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (before < remaining && before + c[k] >= remaining) {  // exactly one bin: the matching keys number >= remaining
            s_bin = 4 * tid + k;
            s_rem = remaining - before;
          }
          before += c[k];
        }
      }
      __syncthreads();
      prefix = (prefix << 8) | s_bin;
      remaining = s_rem;
      __syncthreads();
    }
    const uint64_t kstar = prefix << (8 * (8 - n_pass));  // `remaining` of the hits with this key -- one leaf's -- are kept: the lowest slots
    if (tid == 0) s_neq = 0;
    __syncthreads();
    for (uint32_t i = tid; i < H; i += kSelThreads) {
      const Hit *hp = HitAt(sink, qi, i);
      if (hp->position == kstar) {
        const uint32_t e = atomicAdd(&s_neq, 1u);
        if (e < 32u) eq_slot[e] = hp->slot;
      }
    }
    __syncthreads();
    const uint32_t n_eq = min(s_neq, 32u);  // (a leaf holds at most 21 points)
    auto selected = [&](uint64_t key, uint32_t slot) {
      if (key != kstar) return key < kstar;
      uint32_t below = 0;
      for (uint32_t e = 0; e < n_eq; ++e) below += eq_slot[e] < slot ? 1u : 0u;
      return below < remaining;
    };
    // ---- C: kept hits beyond the first `keep` records take the places of the dropped ones among them (both in
    // index order; the dropped ones all come first: their indices are collected before any kept one needs them)
    uint32_t n_b = 0, n_a = 0;
    for (uint32_t base = 0; base < H; base += kSelThreads) {
      const uint32_t i = base + tid;
      const bool live = i < H;
      uint4 rec = make_uint4(0, 0, 0, 0);
      if (live) rec = *reinterpret_cast<const uint4 *>(HitAt(sink, qi, i));
      const bool sel = live && selected((uint64_t)rec.x | ((uint64_t)rec.y << 32), rec.w);
      const bool is_b = live && i < keep && !sel, is_a = live && i >= keep && sel;
      const uint64_t mb = __ballot(is_b), ma = __ballot(is_a);
      if (lane == 0) {
        s_wave[0][wave] = (uint32_t)__popcll(mb);
        s_wave[1][wave] = (uint32_t)__popcll(ma);
      }
      __syncthreads();
      uint32_t off_b = n_b, off_a = n_a, tot_b = 0, tot_a = 0;
#pragma unroll
      for (uint32_t w2 = 0; w2 < kSelThreads / 64; ++w2) {
        if (w2 < wave) {
          off_b += s_wave[0][w2];
          off_a += s_wave[1][w2];
        }
        tot_b += s_wave[0][w2];
        tot_a += s_wave[1][w2];
      }
      const uint64_t below = (1ull << lane) - 1ull;
      if (is_b) posB[off_b + (uint32_t)__popcll(mb & below)] = (uint16_t)i;
      __syncthreads();
      if (is_a) *reinterpret_cast<uint4 *>(HitAt(sink, qi, posB[off_a + (uint32_t)__popcll(ma & below)])) = rec;
      n_b += tot_b;
      n_a += tot_a;
      __syncthreads();
    }
    // ---- D: the kept records' positions back (the word held the key)
    for (uint32_t i = tid; i < keep; i += kSelThreads) {
      Hit *hp = HitAt(sink, qi, i);
      hp->position = position_of(hp->slot);
    }
  }
}

// small_batch: the host's knowledge that the launch has about a query per lane or fewer (the set walk then shares the
// walk of a query between the lanes of a wave: SeedWalkKernel, kShare)
hipError_t LaunchWalk(smhip_ctx *ctx, const smhip_params &p, int mode, const QueryRec *qrec, const uint32_t *n_queries_dev,
                      uint32_t n_queries_host, const Sink &sink, const uint32_t *range_end, uint32_t range_stride,
                      uint32_t blocks_override = 0, bool small_batch = false) {
  const int env_blocks = ctx->knobs.lane_blocks_per_cu, xcd_ranges = ctx->knobs.xcd_ranges,
            env_thr = ctx->knobs.scan_threshold;
  int blocks_per_cu = p.walk_blocks_per_cu ? (int)p.walk_blocks_per_cu : (env_blocks ? env_blocks : kLaneBlocksPerCU);
  if (blocks_per_cu < 1 || blocks_per_cu > 8) blocks_per_cu = kLaneBlocksPerCU;
  int thr = env_thr > 0 ? env_thr : kScanThreshold;
  // (the set walk's workgroups are half the size: the same knob counts them in pairs)
  uint32_t set_blocks_per_cu = 2u * (uint32_t)blocks_per_cu;
  if (ctx->knobs.set_groups_per_cu >= 1 && ctx->knobs.set_groups_per_cu <= 8) set_blocks_per_cu = (uint32_t)ctx->knobs.set_groups_per_cu;
  if (mode == kWalkSet && env_thr <= 0) thr = kSetScanThreshold;
  if (thr < 1) thr = 1;
  if (thr > 64) thr = 64;
  {
    int hand = ctx->knobs.handout > 0 ? ctx->knobs.handout : kHandoutLanes;  // (experiment knob)
    if (hand < 1 || hand > 64) hand = kHandoutLanes;
    thr |= hand << 16;  // (the thresholds travel in one kernel argument)
    const int sg = ctx->knobs.walk_grab > 0 ? ctx->knobs.walk_grab : kShareGrab;
    const int su = ctx->knobs.walk_grab_upto > 0 ? ctx->knobs.walk_grab_upto : kShareGrabUpto;
    thr |= (std::min(sg, 64) << 8) | (int)((uint32_t)std::min(su / 8, 255) << 24);
  }
  // (walk_share: -1 = small batches share, 1 = every set walk, 0 = none: the A/B reference.  Alone on the GPU sharing
  // wins at every size -- 51 % of the launch at 300 chunks, 16 % at 3 000, 5 % at 5 000, 1 % at 10 000 --, with four
  // batches of 10 000 in flight it loses 2 %: 9.94 against 9.74 ms per step)
  const bool share = mode == kWalkSet && ctx->knobs.walk_share != 0 && (small_batch || ctx->knobs.walk_share == 1);
  const uint32_t blocks = blocks_override ? blocks_override : 256 * blocks_per_cu;
  const size_t stack_bytes = (size_t)256 * 8 * kSeedThreads * kStackDepth * 2 * sizeof(float4);  // (any grid up to 8 per CU)
  hipError_t e = ctx->walk_stack.Reserve(stack_bytes);
  if (e != hipSuccess) return e;
  const uint32_t *re = xcd_ranges ? range_end : nullptr;
#define SM_WALK(MODE, THREADS, SHARE, BLOCKS)                                                                          \
  hipLaunchKernelGGL((SeedWalkKernel<MODE, THREADS, SHARE>), dim3(BLOCKS), dim3(THREADS), 0, ctx->stream, ctx->index,  \
                     qrec, n_queries_dev, n_queries_host, sink, p.search_radius, ctx->walk_stack.as<float4>(), re,     \
                     range_stride, (uint32_t)thr)
  if (mode == kWalkCensus) SM_WALK(kWalkCensus, kSeedThreads, false, blocks);
  else if (mode == kWalkSet && share) SM_WALK(kWalkSet, kSetThreads, true, blocks_override ? blocks_override : 256 * set_blocks_per_cu);
  else if (mode == kWalkSet) SM_WALK(kWalkSet, kSetThreads, false, blocks_override ? blocks_override : 256 * set_blocks_per_cu);
  else SM_WALK(kWalkExact, kSeedThreads, false, blocks);
#undef SM_WALK
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchLeafGather(smhip_ctx *ctx, int dim, const float *values, const uint64_t *positions,
                            const uint32_t *vind, const uint32_t *leaf_left, const uint32_t *leaf_tab,
                            uint64_t n_leaves, float *leaf_pts, uint32_t *slot_vind, float *leaf_box) {
  if (dim != 6) return hipErrorInvalidValue;
  hipLaunchKernelGGL(LeafGatherKernel, dim3(4096), dim3(256), 0, ctx->stream, values, positions, vind, leaf_left,
                     leaf_tab, n_leaves, leaf_pts, slot_vind, leaf_box);
  return hipGetLastError();
}

hipError_t LaunchInnerBoxes(smhip_ctx *ctx, const void *inner, const uint32_t *level_nodes, uint32_t count,
                            const float *leaf_box, float *inner_box) {
  if (count == 0) return hipSuccess;
  const uint32_t blocks = std::min<uint32_t>((count + 255) / 256, 4096);
  hipLaunchKernelGGL(InnerBoxKernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4 *>(inner),
                     level_nodes, count, leaf_box, inner_box);
  return hipGetLastError();
}

hipError_t LaunchPackBoxes(smhip_ctx *ctx, void *inner, uint64_t n_inner, uint32_t root_ref, const float *leaf_box,
                           const float *inner_box, float *grid, uint32_t *error_flag) {
  hipLaunchKernelGGL(BoxGridKernel, dim3(1), dim3(64), 0, ctx->stream, root_ref, leaf_box, inner_box, grid);
  if (n_inner)
    hipLaunchKernelGGL(PackBoxesKernel, dim3(4096), dim3(256), 0, ctx->stream, reinterpret_cast<uint4 *>(inner),
                       n_inner, leaf_box, inner_box, grid, error_flag);
  return hipGetLastError();
}

// Derives ctx->ix_quads from the resident binary records (on upload; replicas receive the array like the others).
// Leaves index.quads null -- and the mapping path on the exact walk -- for an index beyond the scan's 64 M elements.
hipError_t BuildQuads(smhip_ctx *ctx) {
  IndexView &ix = ctx->index;
  ix.quads = nullptr;
  ix.n_quads = 0;
  const uint64_t n = ix.n_inner;
  if (n == 0 || n > (64ull << 20)) return hipSuccess;
  DevBuf head, flag, ids, changed;
  hipError_t e = hipSuccess;
  auto done = [&](hipError_t r) {
    (void)hipStreamSynchronize(ctx->stream);
    head.Free();
    flag.Free();
    ids.Free();
    changed.Free();
    return r;
  };
  if ((e = head.Reserve(n * 4)) != hipSuccess || (e = flag.Reserve(n * 4)) != hipSuccess ||
      (e = ids.Reserve((n + 1) * 4)) != hipSuccess || (e = changed.Reserve(4)) != hipSuccess)
    return done(e);
  if ((e = hipMemsetAsync(head.p, 0, n * 4, ctx->stream)) != hipSuccess) return done(e);
  const uint32_t one = 1u;
  if ((e = hipMemcpyAsync(head.p, &one, 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return done(e);
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 8192);
  uint32_t any = 1;
  for (uint32_t round = 1; round <= 4u * (uint32_t)kStackDepth && any; ++round) {  // two tree levels per round
    any = 0;
    if ((e = hipMemsetAsync(changed.p, 0, 4, ctx->stream)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(QuadMarkKernel, dim3(blocks), dim3(256), 0, ctx->stream, ix.inner, n, head.as<uint32_t>(), round,
                       changed.as<uint32_t>());
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = hipMemcpyAsync(&any, changed.p, 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return done(e);
  }
  // A tree deeper than the rounds above (nanoflann splits a span in the middle: depth is not bounded by log N) is left
  // without quad records -- every head below an unmarked level would be missing -- and is walked in the exact mode.
  if (any) return done(hipSuccess);
  hipLaunchKernelGGL(QuadFlagKernel, dim3(blocks), dim3(256), 0, ctx->stream, head.as<uint32_t>(), n, flag.as<uint32_t>());
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  if ((e = LaunchScanU32(ctx, flag.as<uint32_t>(), (uint32_t)n, ids.as<uint32_t>())) != hipSuccess) return done(e);
  uint32_t n_quads = 0;
  if ((e = hipMemcpyAsync(&n_quads, ids.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
    return done(e);
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return done(e);
  if ((e = ctx->ix_quads.Reserve((uint64_t)n_quads * 128)) != hipSuccess) return done(e);
  hipLaunchKernelGGL(QuadFillKernel, dim3(blocks), dim3(256), 0, ctx->stream, ix.inner, n, flag.as<uint32_t>(),
                     ids.as<uint32_t>(), ctx->ix_quads.as<uint4>());
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  ix.quads = ctx->ix_quads.as<uint4>();
  ix.n_quads = n_quads;
  return done(hipSuccess);
}

hipError_t LaunchQueryExpand(smhip_ctx *ctx, const smhip_params &p, const float *feat, uint32_t feat_stride,
                             const uint32_t *q_off, uint32_t n_chunks, void *qrec) {
  hipLaunchKernelGGL(QueryExpandKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, feat, feat_stride, q_off,
                     p.step_size, reinterpret_cast<QueryRec *>(qrec));
  return hipGetLastError();
}

namespace {
// Visiting order for the walker (see file header).  n_max bounds the query count.
hipError_t BuildQueryOrder(smhip_ctx *ctx, const QueryRec *qrec, const uint32_t *n_queries_dev,
                           uint32_t n_queries_host, uint64_t n_max, const uint32_t **order_out,
                           const uint32_t **range_end_out, uint32_t *range_stride_out) {
  uint32_t bits = 1;
  while ((1ull << bits) < ctx->index.n_leaves) ++bits;
  const int env_cls = ctx->knobs.classes, env_bits = ctx->knobs.bucket_bits;
  const uint32_t classes = (env_cls > 0 && env_cls <= (int)kCostClasses) ? (uint32_t)env_cls : kCostClasses;
  const uint32_t want_bits = env_bits > 0 ? (uint32_t)env_bits : kLeafBucketBits;
  const uint32_t shift = bits > want_bits ? bits - want_bits : 0;
  const uint32_t per_range = ((uint32_t)(ctx->index.n_leaves >> shift) + 8) / 8;  // leaf buckets per XCD range
  const uint32_t n_buckets = 8 * classes * per_range + 1;
  hipError_t e;
  if ((e = ctx->q_keys.Reserve(n_max * 4)) != hipSuccess) return e;
  if ((e = ctx->q_order.Reserve(n_max * 32)) != hipSuccess) return e;
  if ((e = ctx->q_hist.Reserve(((size_t)n_buckets + 1) * 8)) != hipSuccess) return e;
  uint32_t *hist = ctx->q_hist.as<uint32_t>(), *cursor = hist + n_buckets + 1;
  if ((e = hipMemsetAsync(hist, 0, (size_t)n_buckets * 4, ctx->stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(QueryKeyKernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->index, qrec, n_queries_dev,
                     n_queries_host, shift, per_range, classes, ctx->q_keys.as<uint32_t>(), hist);
  if ((e = LaunchScanU32(ctx, hist, n_buckets - 1, cursor)) != hipSuccess) return e;
  hipLaunchKernelGGL(QueryScatterKernel, dim3(2048), dim3(256), 0, ctx->stream, n_queries_dev, n_queries_host,
                     ctx->q_keys.as<uint32_t>(), cursor, qrec, ctx->q_order.as<QueryRec>());
  *order_out = ctx->q_order.as<uint32_t>();
  // after the scatter cursor[k] = end of bucket k, so range r of the order ends at
  // cursor[(r + 1) * classes * per_range - 1]
  *range_end_out = cursor;
  *range_stride_out = classes * per_range;
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchSeedLookup(smhip_ctx *ctx, const smhip_params &p, const void *qrec, const uint32_t *q_off,
                            uint32_t n_chunks, uint64_t n_queries_max, HitSlots region, uint32_t *q_count,
                            uint32_t *q_nodes, uint32_t *q_points, uint32_t *counters, Hit *pool,
                            uint32_t *ovf_tab, uint64_t pool_cap) {
  Sink sink{region, p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_count, q_nodes, q_points,
            nullptr, counters, pool, ovf_tab, pool_cap, 0u, 0u};
  const QueryRec *qr = reinterpret_cast<const QueryRec *>(qrec);
  const uint32_t *order = nullptr, *range_end = nullptr;
  uint32_t range_stride = 0;
  hipError_t e;
  {
    ScopedTimer t(ctx, T_ORDER);
    e = BuildQueryOrder(ctx, qr, q_off + n_chunks, 0, n_queries_max, &order, &range_end, &range_stride);
  }
  if (e != hipSuccess) return e;
  ScopedTimer t(ctx, T_SEED);  // the search itself (the set walk, and the second search of the queries beyond the cap)
  ctx->last_walk_mode = p.reference_walk ? kWalkCensus : kWalkExact;
  if (p.reference_walk)
    return LaunchWalk(ctx, p, kWalkCensus, reinterpret_cast<const QueryRec *>(order), q_off + n_chunks, 0, sink, range_end,
                      range_stride);
  // Large references put many queries beyond the cap (0.6 % at 250 Mb, each a traversal of tens of thousands of
  // points that the second search would run again, one lane each, after everything else has finished): there
  // every query is searched in the reference's order at once, as before the set walk existed.  smhip_map_run has
  // resolved SMHIP_SEARCH_AUTO by now (api.hip ResolveParams: the census of the index's own points, and the share of
  // such queries in the batches seen so far); an index without quad records is always walked in the exact mode.
  if (!ctx->index.quads || p.search_mode == SMHIP_SEARCH_EXACT)
    return LaunchWalk(ctx, p, kWalkExact, reinterpret_cast<const QueryRec *>(order), q_off + n_chunks, 0, sink, range_end,
                      range_stride);
  ctx->last_walk_mode = kWalkSet;
  const bool ranked = (uint32_t)p.num_nearest_points <= kSelMaxKeep && !ctx->knobs.second_search;
  sink.store_all = ranked ? 1u : 0u;
  e = LaunchWalk(ctx, p, kWalkSet, reinterpret_cast<const QueryRec *>(order), q_off + n_chunks, 0, sink, range_end,
                 range_stride, 0, n_chunks <= kShareMaxChunks);
  sink.store_all = 0u;
  if (e != hipSuccess) return e;
  // Queries with more hits than num_nearest_points: which 5000 become anchors depends on the reference's traversal
  // order (spatial_index.cc:371-372), so they are searched again in that order; their slots, overflow segments
  // and counts are simply written a second time.  (None on a 12 Mb reference; 0.6 % of the queries on a 250 Mb one.)
  // The list takes the room of the visiting order, which the set walk has consumed; its length stays on the
  // device: counters[7]; the range cursors [16..23] start again.
  QueryRec *list = reinterpret_cast<QueryRec *>(const_cast<uint32_t *>(order));
  if ((e = hipMemsetAsync(counters + 16, 0, 8 * sizeof(uint32_t), ctx->stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(counters + 6, 0, 2 * sizeof(uint32_t), ctx->stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(OverCapCollectKernel, dim3(1024), dim3(256), 0, ctx->stream, qr, q_off + n_chunks, q_count,
                     (uint32_t)p.num_nearest_points, list, counters + 7);
  const uint32_t *second_n = counters + 7;
  if (ranked) {
    // ... or, rather: the set walk's hits are RANKED in the reference's order and the first num_nearest_points kept
    // (OverCapSelectKernel); only the queries of a tree too deep for the ranking's keys (counters[6] of them, list2)
    // are searched again
    if (!ctx->slot_leaf_valid) {  // block of eight point slots -> leaf, once per resident index
      const uint64_t blocks = (ctx->leaf_pts_bytes / 32 + 7) / 8 + 8;
      if ((e = ctx->ix_slot_leaf.Reserve(blocks * sizeof(uint32_t))) != hipSuccess) return e;
      hipLaunchKernelGGL(SlotLeafKernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->index.leaf_tab, ctx->index.n_leaves,
                         ctx->ix_slot_leaf.as<uint32_t>());
      ctx->slot_leaf_valid = true;
    }
    if ((e = ctx->ovf_list.Reserve(n_queries_max * sizeof(QueryRec))) != hipSuccess) return e;
    QueryRec *list2 = ctx->ovf_list.as<QueryRec>();
    // (how many workgroups: the queries beyond the cap are counted on the device, the grid is chosen here.  An index whose
    // census holds no such query -- every 12 Mb-sized reference -- gets a token grid: these launches, empty then, and the
    // one below otherwise wait for room behind the other contexts' kernels for nothing.  Speed only: each workgroup
    // strides over whatever the list holds.)
    const bool census_over_cap = ctx->census_hits.empty() || ctx->census_hits.back() > (uint32_t)p.num_nearest_points;
    hipLaunchKernelGGL(OverCapSelectKernel, dim3(census_over_cap ? 2048 : 128), dim3(kSelThreads), 0, ctx->stream, ctx->index,
                       ctx->ix_slot_leaf.as<uint32_t>(), list, counters + 7, sink, list2, counters + 6);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    list = list2;
    second_n = counters + 6;
  }
  sink.stop_at_cap = 1u;
  // (ranked: only the queries of a tree too deep for the ranking's keys are left -- none on any index built so far)
  return LaunchWalk(ctx, p, kWalkExact, list, second_n, 0, sink, nullptr, 0, ranked ? 64u : 0u);
}

// Census of the index: `samples` of its own points (the first point of evenly spaced leaves) as queries, hit counts
// only (one slot per query; hits beyond it are counted, not stored), sorted into ctx->census_hits.
namespace {
__global__ void CensusQueryKernel(IndexView ix, uint32_t samples, QueryRec *__restrict__ qrec) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= samples) return;
  const uint32_t leaf = (uint32_t)(((uint64_t)i * ix.n_leaves) / samples);
  const uint32_t w = ix.leaf_tab[leaf], n = w >> 27;
  const size_t slot = (size_t)(w & 0x7FFFFFFu) * 8u + (n ? i % n : 0u);
  const float *pt = ix.leaf_pts + slot * 8u;  // 32-byte slots: v0..v5, position
  QueryRec r;
#pragma unroll
  for (int d = 0; d < 6; ++d) r.q[d] = pt[d];
  r.chunk = 0;
  r.k = i;
  qrec[i] = r;
}
}  // namespace
hipError_t IndexCensus(smhip_ctx *ctx, const smhip_params &p, uint32_t samples) {
  ctx->census_hits.clear();
  samples = (uint32_t)std::min<uint64_t>(samples, ctx->index.n_leaves);
  if (samples == 0) return hipSuccess;
  DevBuf d_qrec, d_region, d_tab, d_cnt;
  hipError_t e = hipSuccess;
  auto done = [&](hipError_t r) {
    (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf *b : {&d_qrec, &d_region, &d_tab, &d_cnt}) b->Free();
    return r;
  };
  if ((e = d_qrec.Reserve((size_t)samples * 32)) != hipSuccess || (e = d_region.Reserve((size_t)samples * sizeof(Hit))) != hipSuccess ||
      (e = d_tab.Reserve(sizeof(void *))) != hipSuccess || (e = d_cnt.Reserve((size_t)samples * 4)) != hipSuccess ||
      (e = ctx->counters.Reserve(40 * sizeof(uint32_t))) != hipSuccess)
    return done(e);
  void *h_tab[1] = {d_region.p};
  if ((e = hipMemcpyAsync(d_tab.p, h_tab, sizeof h_tab, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return done(e);
  if ((e = hipMemsetAsync(ctx->counters.p, 0, 32 * sizeof(uint32_t), ctx->stream)) != hipSuccess) return done(e);
  if ((e = hipMemsetAsync(d_cnt.p, 0, (size_t)samples * 4, ctx->stream)) != hipSuccess) return done(e);
  QueryRec *qr = d_qrec.as<QueryRec>();
  hipLaunchKernelGGL(CensusQueryKernel, dim3((samples + 255) / 256), dim3(256), 0, ctx->stream, ctx->index, samples, qr);
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  const uint32_t *order = nullptr, *range_end = nullptr;
  uint32_t range_stride = 0;
  if ((e = BuildQueryOrder(ctx, qr, nullptr, samples, samples, &order, &range_end, &range_stride)) != hipSuccess) return done(e);
  Sink sink{HitSlots{d_tab.as<Hit *>(), 31u, 1u}, 1u, 1u, d_cnt.as<uint32_t>(), nullptr, nullptr, nullptr,
            ctx->counters.as<uint32_t>(), nullptr, nullptr, 0, 0u, 0u};
  const int saved_mode = ctx->last_walk_mode;
  e = LaunchWalk(ctx, p, ctx->index.quads ? kWalkSet : kWalkExact, reinterpret_cast<const QueryRec *>(order), nullptr, samples,
                 sink, range_end, range_stride);
  ctx->last_walk_mode = saved_mode;
  if (e != hipSuccess) return done(e);
  ctx->census_hits.resize(samples);
  if ((e = hipMemcpyAsync(ctx->census_hits.data(), d_cnt.p, (size_t)samples * 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
    return done(e);
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return done(e);
  std::sort(ctx->census_hits.begin(), ctx->census_hits.end());
  return done(hipSuccess);
}

hipError_t LaunchSeedLookupRaw(smhip_ctx *ctx, const smhip_params &p, const float *queries, uint32_t n,
                               uint32_t cap, HitSlots region, uint32_t *q_count, uint32_t *q_nodes,
                               uint32_t *q_points, uint32_t *counters, void *qrec) {
  QueryRec *qr = reinterpret_cast<QueryRec *>(qrec);
  hipLaunchKernelGGL(QueryExpandRawKernel, dim3(256), dim3(256), 0, ctx->stream, queries, n, qr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const uint32_t *order = nullptr, *range_end = nullptr;
  uint32_t range_stride = 0;
  if ((e = BuildQueryOrder(ctx, qr, nullptr, n, n, &order, &range_end, &range_stride)) != hipSuccess) return e;
  // the test entry point reports the first `cap` hits of every query (no pool, no 5000 cut)
  Sink sink{region, cap, cap, q_count, q_nodes, q_points, nullptr, counters, nullptr, nullptr, 0, 0u, 0u};
  ScopedTimer t(ctx, T_SEED);
  // the per-stage entry point reports hits in the reference's traversal order, or -- SMHIP_SEARCH_SET -- as the
  // mapping path's order-free walk over quad records finds them (same set; an index without quads: exact walk)
  const int mode = p.reference_walk ? kWalkCensus : (p.search_mode == SMHIP_SEARCH_SET && ctx->index.quads ? kWalkSet : kWalkExact);
  ctx->last_walk_mode = mode;
  return LaunchWalk(ctx, p, mode, reinterpret_cast<const QueryRec *>(order), nullptr, n, sink, range_end, range_stride, 0,
                    n <= kShareMaxChunks * 256u);
}
}  // namespace smhip

#ifdef SMHIP_PHASE_PROF
extern "C" int smhip_debug_walk_prof(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(smhip::g_walk_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    z[14] = ~0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(smhip::g_walk_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
