// seed_lookup.hip -- batched exact radius search over the flattened k-d tree (gfx950).
//
// Replaces, per seed, nanoflann's radiusSearch -> findNeighbors -> computeInitialDistances
// -> searchLevel -> L2_Adaptor::evalMetric -> RadiusResultSet::addPoint
// (nanoflann.hpp:1278-1301, 1221-1242, 1006-1023, 1346-1410, 383-408, 249-253) as called
// from SpatialIndex::GenerateChains (spatial_index.cc:351-409).
//
// Data layout in HBM (built once per index from the pre-order smhip_node array + point cloud):
//   inner    : 32-byte records of the INNER nodes only, in pre-order:
//              {feat, divlow, divhigh, child1 ref, child2 ref, pad}.  A child ref is either an
//              inner index or kLeafTag | leaf ordinal, so reaching a leaf costs no node read.
//   leaf_recs: one 64-byte (one sector) record per leaf, in tree order: {left, n, boff, pad,
//              lo[6], hi[6]} where lo/hi is the TIGHT bounding box of the leaf's points.
//   leaf_pts : one 128-byte-aligned block of 16-byte slots per leaf at slot offset `boff`: n "first
//              halves" {v0,v1,v2,v3}, then (from slot boff + roundup(n, 8)) n "second halves"
//              {v4, v5, uint64 Point.position}.  Eight first halves are exactly one 128-byte line.
//              This replaces <= 20 scattered (vind -> 6 strided floats) gathers per leaf
//              (sigmap_adaptor.h:94-97), and a hit needs no second lookup for its position.
//   qrec     : one 32-byte record per query: float q[6]; uint32 chunk; uint32 k
//
// The recursion of searchLevel is replaced by an explicit stack of pending "other child" visits
// (child ref, mindistsq, dists[6] = 32 bytes per entry).  That reproduces nanoflann's visiting
// order and pruning arithmetic (`mindistsq + cut_dist - dists[idx]`, same operation order)
// exactly, so hits come out in the reference's traversal order; the k-th hit of a query is stored
// at slot k, and k >= num_nearest_points (5000, spatial_index.cc:371-372) is dropped exactly like
// the reference.
//
// What is NOT read although nanoflann reads it (results are identical, argument in place):
//  * a visited leaf whose tight box is at squared distance >= radius from the query is skipped:
//    the box distance is accumulated with nanoflann's own association order, and every f32
//    operation involved is monotone, so it is a lower bound of the f32 distance nanoflann would
//    compute for any point of the leaf -- no point of that leaf can pass `dist < radius`
//    (measured: 86 % of the visited leaves, 83 % of the points);
// The per-chunk counters still report the reference's work (nodes visited, points it tests).
//
// Mapping: ONE LANE PER QUERY, a state machine in which every loop iteration issues one batch of
// 16-byte loads per lane (an inner node, a leaf record = one sector, or four points of a leaf that
// passed the box test: one sector of first halves + one of second halves) and then retires it, so the instruction stream
// of a wave is amortised over 64 queries.  The two newest stack entries of a lane live in LDS,
// older ones spill to HBM ([lane][64] x 32 B).  Queries are visited in the order of the leaf that
// contains them (QueryKey/QueryScatter: a stackless descent + a counting sort that permutes the
// query records), each XCD taking
// one contiguous eighth of that order, so the lanes of a wave and the waves of an XCD walk the
// same part of the tree and share its cache lines; results are indexed by query id, so the
// visiting order never shows in the output.
#include <stdlib.h>

#include "common.h"

namespace smhip {
namespace {

constexpr int kSeedThreads = 256;

struct QueryRec {  // 32 B
  float q[6];
  uint32_t chunk, k;
};

struct Sink {
  Hit *region;          // [n_queries][fast_cap] (fast pass) or the overflow pool
  uint32_t fast_cap;    // slots per query in the fast pass
  uint32_t nnp;         // num_nearest_points
  uint32_t *q_count;    // total hits per query
  uint32_t *q_nodes;    // optional per-query stats
  uint32_t *q_points;
  smhip_chunk_result *results;  // optional per-chunk accumulation
  uint32_t *counters;   // [0] stack overflow flag, [1] #overflow queries, [2] pool top, [3] pool full
  Hit *pool;            // overflow pool: hits beyond the fast slots, in geometric segments
  uint32_t *ovf_tab;    // [n_queries][kOvfLevels] pool offset of each segment
  uint64_t pool_cap;
};

// nanoflann.hpp:1006-1023 computeInitialDistances (low test, then high test, per dimension)
#define SM_INIT_DIM(k, dk)                                    \
  {                                                           \
    const float lo = ix.bbox[2 * k], hi = ix.bbox[2 * k + 1]; \
    if (q##k < lo) {                                          \
      dk = (q##k - lo) * (q##k - lo);                         \
      mind += dk;                                             \
    }                                                         \
    if (q##k > hi) {                                          \
      dk = (q##k - hi) * (q##k - hi);                         \
      mind += dk;                                             \
    }                                                         \
  }

enum : uint32_t { M_FETCH = 0, M_WALK = 1, M_LEAF = 2, M_SCAN = 3 };

// Hit number `ord` of query `qi` (0-based, in traversal order): the first fast_cap hits go to the
// query's own slots; later ones go to segments of the shared pool whose sizes double
// (fast_cap, 2*fast_cap, ...), each allocated with one atomic when its first hit arrives, so a
// query with many hits never has to be searched twice.
__device__ __forceinline__ void EmitHit(const Sink &sink, uint32_t qi, uint32_t ord, const Hit &h,
                                        uint32_t &ovf_base) {
  if (ord < sink.fast_cap) {
    sink.region[(size_t)qi * sink.fast_cap + ord] = h;
    return;
  }
  const uint32_t k = ord - sink.fast_cap;
  const uint32_t level = 31u - (uint32_t)__clz((int)(k / sink.fast_cap + 1));
  const uint32_t first = sink.fast_cap * ((1u << level) - 1u);
  if (k == first) {  // first hit of this segment: allocate it
    const uint32_t size = sink.fast_cap << level;
    const uint32_t base = atomicAdd(&sink.counters[2], size);
    ovf_base = base;
    if (level == 0) atomicAdd(&sink.counters[1], 1u);
    if ((uint64_t)base + size > sink.pool_cap || level >= (uint32_t)kOvfLevels) {
      atomicOr(&sink.counters[3], 1u);  // reported as SMHIP_ST_HIT_POOL_FULL
      ovf_base = 0xFFFFFFFFu;
    }
    if (level < (uint32_t)kOvfLevels) sink.ovf_tab[(size_t)qi * kOvfLevels + level] = ovf_base;
  }
  if (ovf_base != 0xFFFFFFFFu) sink.pool[(size_t)ovf_base + (k - first)] = h;
}

__global__ __launch_bounds__(kSeedThreads, 6) void SeedLaneKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                                                               const uint32_t *__restrict__ n_queries_dev,
                                                               uint32_t n_queries_host, Sink sink, float radius,
                                                               float4 *__restrict__ stack_mem,
                                                               const uint32_t *__restrict__ order, int xcd_ranges) {
  const uint32_t gl = blockIdx.x * kSeedThreads + threadIdx.x;
  uint32_t n_lanes = gridDim.x * kSeedThreads;
  float4 *stk = stack_mem + (size_t)gl * (kStackDepth * 2);
  uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  const uint4 *pts4 = reinterpret_cast<const uint4 *>(ix.leaf_pts);
  uint32_t w = gl;
  if (xcd_ranges) {
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD: speed only,
    // never correctness).  Give each XCD one contiguous eighth of the visiting order so that its
    // private L2 only has to hold that part of the tree.
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t per = (total + 7) / 8;
    const uint32_t begin = min(total, xcd * per);
    total = min(total, begin + per);
    n_lanes = (gridDim.x >> 3) * kSeedThreads;
    w = begin + slot * kSeedThreads + threadIdx.x;
  }
  uint32_t mode = M_FETCH;
  float q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
  float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
  float mind = 0;
  uint32_t node = 0, sp = 0, hits = 0, nn = 0, np = 0, j = 0, n = 0, lbase = 0, lleft = 0, qi = 0;
  // top-of-stack cache: two 32-byte entries per lane in LDS, [slot][half][thread]
  __shared__ float4 tos[2][2][kSeedThreads];
  uint32_t tcnt = 0, thead = 0;
  uint32_t ovf_base = 0;  // pool offset of the current overflow segment of this lane's query
  for (;;) {
    if (mode == M_FETCH) {
      if (w >= total) break;
      // the fast pass reads records already permuted into visiting order (one round trip);
      // the record's last word is the query id
      const uint32_t ri = w;
      w += n_lanes;
      const float4 a = *reinterpret_cast<const float4 *>(&qrec[ri]);
      const float4 b = *(reinterpret_cast<const float4 *>(&qrec[ri]) + 1);
      qi = __float_as_uint(b.w);
      q0 = a.x; q1 = a.y; q2 = a.z; q3 = a.w; q4 = b.x; q5 = b.y;
      d0 = d1 = d2 = d3 = d4 = d5 = 0.0f;
      mind = 0.0f;
      SM_INIT_DIM(0, d0) SM_INIT_DIM(1, d1) SM_INIT_DIM(2, d2) SM_INIT_DIM(3, d3) SM_INIT_DIM(4, d4)
      SM_INIT_DIM(5, d5)
      node = ix.root_ref; sp = 0; tcnt = 0; hits = 0; nn = 0; np = 0;
      mode = (node & kLeafTag) ? M_LEAF : M_WALK;
    }
    // ---- one batch of 16-byte loads per lane per iteration: a 64-byte window at `wa`, the same
    // four load instructions whatever the lane's mode (node: 32 of them used; leaf record: all;
    // four first halves: all; second half: 16).  The address is formed with selects and made
    // opaque, and all results are pinned right after the loads, so that the compiler keeps ONE
    // memory round trip per iteration instead of sinking a load + wait into every mode's branch.
    uint64_t wa;
    {
      const uint64_t a_walk = (uint64_t)ix.inner + (uint64_t)node * 32;
      const uint64_t a_leaf = (uint64_t)ix.leaf_recs + (uint64_t)(node & ~kLeafTag) * 64;
      const uint64_t a_pts = (uint64_t)pts4 + (uint64_t)(lbase + j) * 16;
      wa = mode == M_WALK ? a_walk : (mode == M_LEAF ? a_leaf : a_pts);
      uint32_t lo32 = (uint32_t)wa, hi32 = (uint32_t)(wa >> 32);
      asm volatile("" : "+v"(lo32), "+v"(hi32));
      wa = ((uint64_t)hi32 << 32) | lo32;
    }
    const uint4 *addr = reinterpret_cast<const uint4 *>(wa);
    uint4 r = addr[0];
    uint4 rr[3];
    rr[0] = addr[1];
    rr[1] = addr[2];
    rr[2] = addr[3];
    // scanning lanes also fetch the four second halves {v4, v5, position} of the same points
    uint4 rs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rs[t] = r;
    if (mode == M_SCAN) {
      const uint4 *sh = addr + ((n + 7u) & ~7u);
#pragma unroll
      for (int t = 0; t < 4; ++t) rs[t] = sh[t];
    }
    asm volatile("" : "+v"(rs[0].x), "+v"(rs[0].y), "+v"(rs[0].z), "+v"(rs[0].w), "+v"(rs[1].x), "+v"(rs[1].y),
                 "+v"(rs[1].z), "+v"(rs[1].w));
    asm volatile("" : "+v"(rs[2].x), "+v"(rs[2].y), "+v"(rs[2].z), "+v"(rs[2].w), "+v"(rs[3].x), "+v"(rs[3].y),
                 "+v"(rs[3].z), "+v"(rs[3].w));
    asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w), "+v"(rr[0].x), "+v"(rr[0].y), "+v"(rr[0].z),
                 "+v"(rr[0].w));
    asm volatile("" : "+v"(rr[1].x), "+v"(rr[1].y), "+v"(rr[1].z), "+v"(rr[1].w), "+v"(rr[2].x), "+v"(rr[2].y),
                 "+v"(rr[2].z), "+v"(rr[2].w));
    bool leaf_end = false;
    if (mode == M_WALK) {  // nanoflann.hpp:1371-1408; r = {feat, divlow, divhigh, c1}, rr[0].x = c2
      ++nn;
      const int idx = (int)r.x;
      const float val = idx == 0 ? q0 : idx == 1 ? q1 : idx == 2 ? q2 : idx == 3 ? q3 : idx == 4 ? q4 : q5;
      const float divlow = __uint_as_float(r.y), divhigh = __uint_as_float(r.z);
      const float diff1 = val - divlow, diff2 = val - divhigh;
      uint32_t next, other;
      float cut;
      if ((diff1 + diff2) < 0) {
        next = r.w;
        other = rr[0].x;
        cut = (val - divhigh) * (val - divhigh);
      } else {
        next = rr[0].x;
        other = r.w;
        cut = (val - divlow) * (val - divlow);
      }
      const float dst = idx == 0 ? d0 : idx == 1 ? d1 : idx == 2 ? d2 : idx == 3 ? d3 : idx == 4 ? d4 : d5;
      const float nm = mind + cut - dst;
      if (nm <= radius) {  // mindistsq * epsError <= worstDist(), epsError == 1
        if (sp + tcnt < (uint32_t)kStackDepth) {
          const uint32_t slot = thead ^ 1u;
          if (tcnt == 2) {  // spill the older cached entry
            stk[2 * sp] = tos[slot][0][threadIdx.x];
            stk[2 * sp + 1] = tos[slot][1][threadIdx.x];
            ++sp;
          } else {
            ++tcnt;
          }
          tos[slot][0][threadIdx.x] = make_float4(__uint_as_float(other), nm, idx == 0 ? cut : d0, idx == 1 ? cut : d1);
          tos[slot][1][threadIdx.x] = make_float4(idx == 2 ? cut : d2, idx == 3 ? cut : d3, idx == 4 ? cut : d4,
                                                  idx == 5 ? cut : d5);
          thead = slot;
        } else {
          atomicOr(&sink.counters[0], 1u);
        }
      }
      node = next;
      mode = (node & kLeafTag) ? M_LEAF : M_WALK;
    } else if (mode == M_LEAF) {
      // a leaf visit of searchLevel (nanoflann.hpp:1350-1369); r = {left, n, boff, -},
      // rr[0..2] = lo[0..5], hi[0..5]
      ++nn;
      lleft = r.x;
      n = r.y;
      lbase = r.z;
      np += n;  // the reference evaluates every point of a visited leaf
      // squared distance from the query to the leaf's tight box, accumulated like evalMetric
      // (nanoflann.hpp:391-406): a lower bound of every point's f32 distance (monotone ops)
      const float l0 = __uint_as_float(rr[0].x), l1 = __uint_as_float(rr[0].y), l2 = __uint_as_float(rr[0].z),
                  l3 = __uint_as_float(rr[0].w), l4 = __uint_as_float(rr[1].x), l5 = __uint_as_float(rr[1].y);
      const float h0 = __uint_as_float(rr[1].z), h1 = __uint_as_float(rr[1].w), h2 = __uint_as_float(rr[2].x),
                  h3 = __uint_as_float(rr[2].y), h4 = __uint_as_float(rr[2].z), h5 = __uint_as_float(rr[2].w);
#define SM_BOX(qq, ll, hh) ((qq) > (hh) ? (qq) - (hh) : ((qq) < (ll) ? (qq) - (ll) : 0.0f))
      const float e0 = SM_BOX(q0, l0, h0), e1 = SM_BOX(q1, l1, h1), e2 = SM_BOX(q2, l2, h2);
      const float e3 = SM_BOX(q3, l3, h3), e4 = SM_BOX(q4, l4, h4), e5 = SM_BOX(q5, l5, h5);
#undef SM_BOX
      float lb = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      lb += e4 * e4;
      lb += e5 * e5;
      j = 0;
      if (lb < radius && n > 0) {
        mode = M_SCAN;
      } else {
        leaf_end = true;  // no point of this leaf can satisfy dist < radius
      }
    } else {  // M_SCAN: nanoflann.hpp:391-406 for up to four points of the leaf
      const uint32_t left_in_leaf = n - j;
#define SM_POINT(rv, sv, t)                                                                   \
  if (left_in_leaf > (uint32_t)t) {                                                           \
    const float f0 = q0 - __uint_as_float(rv.x), f1 = q1 - __uint_as_float(rv.y);            \
    const float f2 = q2 - __uint_as_float(rv.z), f3 = q3 - __uint_as_float(rv.w);            \
    float res = f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3; /* 0 + s == s exactly */               \
    const float f4 = q4 - __uint_as_float(sv.x);                                              \
    res += f4 * f4;                                                                           \
    const float f5 = q5 - __uint_as_float(sv.y);                                              \
    res += f5 * f5;                                                                           \
    if (res < radius) { /* RadiusResultSet::addPoint */                                       \
      const uint32_t ord = hits++;                                                            \
      if (ord < sink.nnp) {                                                                   \
        Hit h;                                                                                \
        h.position = (uint64_t)sv.z | ((uint64_t)sv.w << 32);                                 \
        h.dist = res;                                                                         \
        h.slot = lleft + j + t;                                                               \
        EmitHit(sink, qi, ord, h, ovf_base);                                                  \
      }                                                                                       \
    }                                                                                         \
  }
      SM_POINT(r, rs[0], 0) SM_POINT(rr[0], rs[1], 1) SM_POINT(rr[1], rs[2], 2) SM_POINT(rr[2], rs[3], 3)
#undef SM_POINT
      j += 4;
      leaf_end = j >= n;
    }
    if (leaf_end) {
      if (sp + tcnt == 0) {  // query complete
        // per-query counters (plain stores; the per-chunk sums are formed by AnchorCountKernel)
        sink.q_count[qi] = hits;
        if (sink.q_nodes) sink.q_nodes[qi] = nn;
        if (sink.q_points) sink.q_points[qi] = np;
        mode = M_FETCH;
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
        mode = (node & kLeafTag) ? M_LEAF : M_WALK;
      }
    }
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

// One thread per leaf: writes the leaf's point block (layout in the file header) and the tight
// bounding box into its record.  v[d] = values[vind[i] + d] (sigmap_adaptor.h:94-97: k-d point i is
// the window values[i..i+dim)), position = Point.position of the window's first element.
__global__ void LeafGatherKernel(const float *__restrict__ values, const uint64_t *__restrict__ positions,
                                 const uint32_t *__restrict__ vind, uint4 *__restrict__ leaf_recs,
                                 uint64_t n_leaves, float *__restrict__ leaf_pts) {
  float4 *o = reinterpret_cast<float4 *>(leaf_pts);
  for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_leaves;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 hd = leaf_recs[k * 4];
    const uint32_t left = hd.x, n = hd.y, boff = hd.z, rn = (n + 7u) & ~7u;
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
      o[(size_t)boff + t] = make_float4(p[0], p[1], p[2], p[3]);
      o[(size_t)boff + rn + t] = make_float4(p[4], p[5], __uint_as_float((uint32_t)pos),
                                             __uint_as_float((uint32_t)(pos >> 32)));
    }
    float4 *rec = reinterpret_cast<float4 *>(leaf_recs + k * 4);
    rec[1] = make_float4(lo[0], lo[1], lo[2], lo[3]);
    rec[2] = make_float4(lo[4], lo[5], hi[0], hi[1]);
    rec[3] = make_float4(hi[2], hi[3], hi[4], hi[5]);
  }
}

// ---- visiting order of the queries: key = ordinal (tree order) of the leaf that contains the
// query point (a stackless best-child descent), then a counting sort on key >> shift.
__global__ void QueryKeyKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                               const uint32_t *__restrict__ n_queries_dev, uint32_t n_queries_host,
                               uint32_t shift, uint32_t *__restrict__ keys, uint32_t *__restrict__ hist) {
  const uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const QueryRec r = qrec[q];
    uint32_t node = ix.root_ref;
    while (!(node & kLeafTag)) {
      const uint4 rec = ix.inner[(size_t)node * 2];
      const uint32_t c2 = reinterpret_cast<const uint32_t *>(ix.inner + (size_t)node * 2)[4];
      const float val = r.q[rec.x];
      const float diff1 = val - __uint_as_float(rec.y), diff2 = val - __uint_as_float(rec.z);
      node = (diff1 + diff2) < 0 ? rec.w : c2;
    }
    const uint32_t key = (node & ~kLeafTag) >> shift;
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

constexpr int kLaneBlocksPerCU = 6;  // 6 x 256 threads = 24 waves per CU (<= 80 allocated VGPRs)

hipError_t LaunchWalk(smhip_ctx *ctx, uint32_t blocks_per_cu_param, const QueryRec *qrec,
                      const uint32_t *n_queries_dev, uint32_t n_queries_host, const Sink &sink, float radius,
                      const uint32_t *order) {
  static int env_blocks = -1, xcd_ranges = 1;
  if (env_blocks < 0) {
    const char *e1 = getenv("SMHIP_LANE_BLOCKS_PER_CU"), *e2 = getenv("SMHIP_XCD_RANGES");
    env_blocks = e1 ? atoi(e1) : 0;
    if (e2) xcd_ranges = atoi(e2);
  }
  int blocks_per_cu = blocks_per_cu_param ? (int)blocks_per_cu_param : (env_blocks ? env_blocks : kLaneBlocksPerCU);
  if (blocks_per_cu < 1 || blocks_per_cu > 8) blocks_per_cu = kLaneBlocksPerCU;
  const uint32_t blocks = 256 * blocks_per_cu;
  const size_t stack_bytes = (size_t)blocks * kSeedThreads * kStackDepth * 2 * sizeof(float4);
  hipError_t e = ctx->walk_stack.Reserve(stack_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(SeedLaneKernel, dim3(blocks), dim3(kSeedThreads), 0, ctx->stream, ctx->index,
                     qrec, n_queries_dev, n_queries_host, sink, radius, ctx->walk_stack.as<float4>(), order,
                     order ? xcd_ranges : 0);
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchLeafGather(smhip_ctx *ctx, int dim, const float *values, const uint64_t *positions,
                            const uint32_t *vind, void *leaf_recs, uint64_t n_leaves, float *leaf_pts) {
  if (dim != 6) return hipErrorInvalidValue;
  hipLaunchKernelGGL(LeafGatherKernel, dim3(4096), dim3(256), 0, ctx->stream, values, positions, vind,
                     reinterpret_cast<uint4 *>(leaf_recs), n_leaves, leaf_pts);
  return hipGetLastError();
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
                           uint32_t n_queries_host, uint64_t n_max, const uint32_t **order_out) {
  uint32_t bits = 1;
  while ((1ull << bits) < ctx->index.n_leaves) ++bits;
  const uint32_t shift = bits > 17 ? bits - 17 : 0;
  const uint32_t n_buckets = (uint32_t)(ctx->index.n_leaves >> shift) + 1;
  hipError_t e;
  if ((e = ctx->q_keys.Reserve(n_max * 4)) != hipSuccess) return e;
  if ((e = ctx->q_order.Reserve(n_max * 32)) != hipSuccess) return e;
  if ((e = ctx->q_hist.Reserve(((size_t)n_buckets + 1) * 8)) != hipSuccess) return e;
  uint32_t *hist = ctx->q_hist.as<uint32_t>(), *cursor = hist + n_buckets + 1;
  if ((e = hipMemsetAsync(hist, 0, (size_t)n_buckets * 4, ctx->stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(QueryKeyKernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->index, qrec, n_queries_dev,
                     n_queries_host, shift, ctx->q_keys.as<uint32_t>(), hist);
  if ((e = LaunchScanU32(ctx, hist, n_buckets - 1, cursor)) != hipSuccess) return e;
  hipLaunchKernelGGL(QueryScatterKernel, dim3(2048), dim3(256), 0, ctx->stream, n_queries_dev, n_queries_host,
                     ctx->q_keys.as<uint32_t>(), cursor, qrec, ctx->q_order.as<QueryRec>());
  *order_out = ctx->q_order.as<uint32_t>();
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchSeedLookup(smhip_ctx *ctx, const smhip_params &p, const void *qrec, const uint32_t *q_off,
                            uint32_t n_chunks, uint64_t n_queries_max, Hit *region, uint32_t *q_count,
                            uint32_t *q_nodes, uint32_t *q_points, uint32_t *counters, Hit *pool,
                            uint32_t *ovf_tab, uint64_t pool_cap) {
  Sink sink{region, p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_count, q_nodes, q_points,
            nullptr, counters, pool, ovf_tab, pool_cap};
  const QueryRec *qr = reinterpret_cast<const QueryRec *>(qrec);
  const uint32_t *order = nullptr;
  hipError_t e = BuildQueryOrder(ctx, qr, q_off + n_chunks, 0, n_queries_max, &order);
  if (e != hipSuccess) return e;
  return LaunchWalk(ctx, p.walk_blocks_per_cu, reinterpret_cast<const QueryRec *>(order), q_off + n_chunks, 0, sink,
                    p.search_radius, order);
}

hipError_t LaunchSeedLookupRaw(smhip_ctx *ctx, const smhip_params &p, const float *queries, uint32_t n,
                               uint32_t cap, Hit *region, uint32_t *q_count, uint32_t *q_nodes,
                               uint32_t *q_points, uint32_t *counters, void *qrec) {
  QueryRec *qr = reinterpret_cast<QueryRec *>(qrec);
  hipLaunchKernelGGL(QueryExpandRawKernel, dim3(256), dim3(256), 0, ctx->stream, queries, n, qr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const uint32_t *order = nullptr;
  if ((e = BuildQueryOrder(ctx, qr, nullptr, n, n, &order)) != hipSuccess) return e;
  // the test entry point reports the first `cap` hits of every query (no pool, no 5000 cut)
  Sink sink{region, cap, cap, q_count, q_nodes, q_points, nullptr, counters, nullptr, nullptr, 0};
  return LaunchWalk(ctx, p.walk_blocks_per_cu, reinterpret_cast<const QueryRec *>(order), nullptr, n, sink,
                    p.search_radius, order);
}
}  // namespace smhip
