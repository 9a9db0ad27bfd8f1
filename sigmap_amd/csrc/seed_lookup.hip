// seed_lookup.hip -- batched exact radius search over the flattened k-d tree (gfx950).
//
// Replaces, per seed, nanoflann's radiusSearch -> findNeighbors -> computeInitialDistances
// -> searchLevel -> L2_Adaptor::evalMetric -> RadiusResultSet::addPoint
// (nanoflann.hpp:1278-1301, 1221-1242, 1006-1023, 1346-1410, 383-408, 249-253) as called
// from SpatialIndex::GenerateChains (spatial_index.cc:351-409).
//
// Data layout in HBM (built once per index by LeafGatherKernel):
//   nodes    : 16-byte pre-order records (smhip_node), child1 = self + 1
//   leaf_pts : one 128-byte-aligned block of 16-byte slots per leaf, in TREE order (the order of
//              nanoflann's vind).  A leaf [left,right) with n points, block offset `boff` (kept
//              in the leaf node's 4th word): slots [boff, boff+n) hold the "first halves"
//              {v0,v1,v2,v3}, slots [boff+R, boff+R+n) with R = roundup(n, 8) the "second halves"
//              {v4, v5, uint64 Point.position}.  Eight first halves are exactly one 128-byte
//              line, so every line fetched for a leaf is used whole; this replaces <= 20
//              scattered (vind -> 6 strided floats) gathers, and a hit needs no second lookup for
//              its reference position.
//   qrec     : one 32-byte record per query: float q[6]; uint32 chunk; uint32 k
//
// The recursion of searchLevel is replaced by an explicit stack of pending "other child"
// visits (node, mindistsq, dists[6] = 32 bytes per entry).  That reproduces nanoflann's visiting
// order exactly, so hits come out in the reference's traversal order; the k-th hit of a query
// is stored at slot k, and k >= num_nearest_points (5000, spatial_index.cc:371-372) is dropped
// exactly like the reference.
//
// Two walkers share that logic:
//  * SeedLaneKernel (lanes_per_query = 1, default): ONE LANE PER QUERY, written as a state
//    machine in which every loop iteration issues one batch of 16-byte loads per lane (a node
//    record, the first halves of up to eight leaf points = one 128-byte line, or one second half)
//    and then retires it.  A point whose first four squared differences already reach the radius is rejected
//    without reading its second half (the partial sum is a lower bound of nanoflann's result:
//    f32 addition of non-negative terms is monotone).  The two newest stack entries of a lane
//    live in LDS; older ones spill to HBM ([lane][64 entries] x 32 B), so the instruction
//    stream of a wave is amortised over 64 queries and most push/pop pairs never leave the CU.  Queries are
//    visited in the order of the leaf that contains them (QueryOrder kernels: a stackless
//    descent + a counting sort), so the lanes of a wave and the waves of a CU walk the same
//    part of the tree and share its cache lines; results are indexed by query id, so the
//    visiting order never shows in the output.
//  * SeedLookupKernel<G> (lanes_per_query = 8/16/32/64): a sub-group of G lanes walks one query
//    with the stack in LDS and fans out over the points of a leaf (coalesced leaf reads).
//    Measured slower on MI355X (issue-bound: one traversal step per G lanes); kept for A/B.
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
  const uint32_t *ovf_list;
  const uint32_t *q_ovf_off;
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

// ------------------------------------------------------------------ lane-per-query walker
enum : uint32_t { M_FETCH = 0, M_WALK = 1, M_SCAN = 2, M_SCAN2 = 3 };

template <bool kOverflow>
__global__ __launch_bounds__(kSeedThreads) void SeedLaneKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                                                               const uint32_t *__restrict__ n_queries_dev,
                                                               uint32_t n_queries_host, Sink sink, float radius,
                                                               float4 *__restrict__ stack_mem,
                                                               const uint32_t *__restrict__ order, int xcd_ranges) {
  const uint32_t gl = blockIdx.x * kSeedThreads + threadIdx.x;
  uint32_t n_lanes = gridDim.x * kSeedThreads;
  float4 *stk = stack_mem + (size_t)gl * (kStackDepth * 2);
  uint32_t total = kOverflow ? sink.counters[1] : (n_queries_dev ? *n_queries_dev : n_queries_host);
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
  float pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t node = 0, sp = 0, hits = 0, nn = 0, np = 0, j = 0, n = 0, lbase = 0, lleft = 0, cmask = 0, qi = 0,
           chunk = 0;
  // top-of-stack cache: two 32-byte entries per lane in LDS, [slot][half][thread]
  __shared__ float4 tos[2][2][kSeedThreads];
  uint32_t tcnt = 0, thead = 0;
  Hit *out = nullptr;
  uint32_t out_cap = 0;
  for (;;) {
    if (mode == M_FETCH) {
      if (w >= total) break;
      qi = kOverflow ? sink.ovf_list[w] : (order ? order[w] : w);
      w += n_lanes;
      const float4 a = *reinterpret_cast<const float4 *>(&qrec[qi]);
      const float4 b = *(reinterpret_cast<const float4 *>(&qrec[qi]) + 1);
      q0 = a.x; q1 = a.y; q2 = a.z; q3 = a.w; q4 = b.x; q5 = b.y;
      chunk = __float_as_uint(b.z);
      d0 = d1 = d2 = d3 = d4 = d5 = 0.0f;
      mind = 0.0f;
      SM_INIT_DIM(0, d0) SM_INIT_DIM(1, d1) SM_INIT_DIM(2, d2) SM_INIT_DIM(3, d3) SM_INIT_DIM(4, d4)
      SM_INIT_DIM(5, d5)
      node = 0; sp = 0; tcnt = 0; hits = 0; nn = 0; np = 0;
      if (kOverflow) {
        out = sink.region + sink.q_ovf_off[qi];
        out_cap = sink.nnp;
      } else {
        out = sink.region + (size_t)qi * sink.fast_cap;
        out_cap = sink.fast_cap;
      }
      mode = M_WALK;
    }
    // ---- one batch of 16-byte loads per lane per iteration
    const uint4 *addr = mode == M_WALK ? reinterpret_cast<const uint4 *>(ix.nodes + node)
                                       : pts4 + lbase + (mode == M_SCAN2 ? ((n + 7u) & ~7u) + j + (uint32_t)(__ffs(cmask) - 1) : j);
    const uint4 r = addr[0];
    uint4 rr[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) rr[t] = r;
    if (mode == M_SCAN) {  // first halves of points j .. j+7: one aligned 128-byte line
#pragma unroll
      for (int t = 0; t < 7; ++t) rr[t] = addr[t + 1];
    }
    bool leaf_end = false;
    if (mode == M_WALK) {
      ++nn;
      if ((int32_t)r.x < 0) {  // leaf: start scanning its points
        lleft = r.y;
        n = r.z - r.y;
        lbase = r.w;
        j = 0;
        mode = M_SCAN;
        leaf_end = n == 0;
      } else {  // nanoflann.hpp:1371-1408
        const int idx = (int)r.x;
        const float val = idx == 0 ? q0 : idx == 1 ? q1 : idx == 2 ? q2 : idx == 3 ? q3 : idx == 4 ? q4 : q5;
        const float divlow = __uint_as_float(r.y), divhigh = __uint_as_float(r.z);
        const float diff1 = val - divlow, diff2 = val - divhigh;
        uint32_t next, other;
        float cut;
        if ((diff1 + diff2) < 0) {
          next = node + 1;
          other = r.w;
          cut = (val - divhigh) * (val - divhigh);
        } else {
          next = r.w;
          other = node + 1;
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
      }
    } else if (mode == M_SCAN) {
      // nanoflann.hpp:391-401: first group of four differences, for up to eight points
      const uint32_t left_in_leaf = n - j;
      np += min(8u, left_in_leaf);
      cmask = 0;
#define SM_PART(rv, t)                                                                       \
  {                                                                                          \
    const float f0 = q0 - __uint_as_float(rv.x), f1 = q1 - __uint_as_float(rv.y);           \
    const float f2 = q2 - __uint_as_float(rv.z), f3 = q3 - __uint_as_float(rv.w);           \
    const float s4 = f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3; /* 0 + s == s exactly */         \
    pr[t] = s4;                                                                              \
    /* result >= partial sum, so partial >= radius cannot be a hit */                        \
    cmask |= (s4 < radius && left_in_leaf > (uint32_t)t) ? (1u << t) : 0u;                   \
  }
      SM_PART(r, 0) SM_PART(rr[0], 1) SM_PART(rr[1], 2) SM_PART(rr[2], 3) SM_PART(rr[3], 4) SM_PART(rr[4], 5)
      SM_PART(rr[5], 6) SM_PART(rr[6], 7)
#undef SM_PART
      if (cmask) {
        mode = M_SCAN2;
      } else {
        j += 8;
        leaf_end = j >= n;
      }
    } else {  // M_SCAN2, nanoflann.hpp:403-406: two-element tail; r = {v4, v5, position}
      const uint32_t t = (uint32_t)(__ffs(cmask) - 1);
      const float partial = t == 0 ? pr[0] : t == 1 ? pr[1] : t == 2 ? pr[2] : t == 3 ? pr[3]
                          : t == 4 ? pr[4] : t == 5 ? pr[5] : t == 6 ? pr[6] : pr[7];
      const float f4 = q4 - __uint_as_float(r.x);
      float res = partial + f4 * f4;
      const float f5 = q5 - __uint_as_float(r.y);
      res += f5 * f5;
      if (res < radius) {  // RadiusResultSet::addPoint
        const uint32_t ord = hits++;
        if (ord < sink.nnp && ord < out_cap) {
          Hit h;
          h.position = (uint64_t)r.z | ((uint64_t)r.w << 32);
          h.dist = res;
          h.slot = lleft + j + t;
          out[ord] = h;
        }
      }
      cmask &= cmask - 1;
      if (cmask == 0) {
        j += 8;
        mode = M_SCAN;
        leaf_end = j >= n;
      }
    }
    if (leaf_end) {
      if (sp + tcnt == 0) {  // query complete
        if (!kOverflow) {
          sink.q_count[qi] = hits;
          if (sink.q_nodes) sink.q_nodes[qi] = nn;
          if (sink.q_points) sink.q_points[qi] = np;
          if (sink.results) {
            atomicAdd((unsigned long long *)&sink.results[chunk].nodes_visited, (unsigned long long)nn);
            atomicAdd((unsigned long long *)&sink.results[chunk].points_tested, (unsigned long long)np);
            atomicAdd((unsigned long long *)&sink.results[chunk].hits_total, (unsigned long long)hits);
          }
        }
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
        mode = M_WALK;
      }
    }
  }
}

// ------------------------------------------------------------------ G-lanes-per-query walker
template <int G, bool kOverflow>
__global__ __launch_bounds__(kSeedThreads) void SeedLookupKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                                                                 const uint32_t *__restrict__ n_queries_dev,
                                                                 uint32_t n_queries_host, Sink sink, float radius) {
  extern __shared__ __align__(16) float4 stack_mem[];
  constexpr int kGroupsPerBlock = kSeedThreads / G;
  const uint32_t tid = threadIdx.x;
  const uint32_t group_in_block = tid / G;
  const uint32_t lig = tid % G;
  const uint32_t group_shift = (tid & 63) / G * G;
  float4 *stack = stack_mem + (size_t)group_in_block * kStackDepth * 2;
  const uint32_t n_groups_total = gridDim.x * kGroupsPerBlock;
  const uint32_t total = kOverflow ? sink.counters[1] : (n_queries_dev ? *n_queries_dev : n_queries_host);
  for (uint32_t w = blockIdx.x * kGroupsPerBlock + group_in_block; w < total; w += n_groups_total) {
    const uint32_t qi = kOverflow ? sink.ovf_list[w] : w;
    const QueryRec qr = qrec[qi];
    const float q0 = qr.q[0], q1 = qr.q[1], q2 = qr.q[2], q3 = qr.q[3], q4 = qr.q[4], q5 = qr.q[5];
    const uint32_t chunk = qr.chunk;
    Hit *out;
    uint32_t out_cap;
    if (kOverflow) {
      out = sink.region + sink.q_ovf_off[qi];
      out_cap = sink.nnp;
    } else {
      out = sink.region + (size_t)qi * sink.fast_cap;
      out_cap = sink.fast_cap;
    }
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
    float mind = 0.0f;
    SM_INIT_DIM(0, d0) SM_INIT_DIM(1, d1) SM_INIT_DIM(2, d2) SM_INIT_DIM(3, d3) SM_INIT_DIM(4, d4)
    SM_INIT_DIM(5, d5)
    uint32_t hit_count = 0, n_nodes = 0, n_points = 0;
    uint32_t sp = 0, node = 0;
    bool alive = ix.n_nodes > 0;
    uint4 rec = make_uint4(0, 0, 0, 0);
    if (alive) rec = *reinterpret_cast<const uint4 *>(&ix.nodes[0]);
    while (alive) {
      ++n_nodes;
      const uint4 cur = rec;
      const bool is_leaf = (int32_t)cur.x < 0;
      uint32_t next = 0;
      if (is_leaf) {
        if (sp == 0) {
          alive = false;
        } else {
          --sp;
          const float4 a = stack[2 * sp], b = stack[2 * sp + 1];
          next = __float_as_uint(a.x);
          mind = a.y;
          d0 = a.z; d1 = a.w; d2 = b.x; d3 = b.y; d4 = b.z; d5 = b.w;
        }
      } else {
        const int idx = (int)cur.x;
        const float val = idx == 0 ? q0 : idx == 1 ? q1 : idx == 2 ? q2 : idx == 3 ? q3 : idx == 4 ? q4 : q5;
        const float divlow = __uint_as_float(cur.y), divhigh = __uint_as_float(cur.z);
        const float diff1 = val - divlow, diff2 = val - divhigh;
        uint32_t other;
        float cut;
        if ((diff1 + diff2) < 0) {
          next = node + 1;
          other = cur.w;
          cut = (val - divhigh) * (val - divhigh);
        } else {
          next = cur.w;
          other = node + 1;
          cut = (val - divlow) * (val - divlow);
        }
        const float dst = idx == 0 ? d0 : idx == 1 ? d1 : idx == 2 ? d2 : idx == 3 ? d3 : idx == 4 ? d4 : d5;
        const float nm = mind + cut - dst;
        if (nm <= radius) {
          if (sp < (uint32_t)kStackDepth) {
            if (lig == 0) {
              stack[2 * sp] = make_float4(__uint_as_float(other), nm, idx == 0 ? cut : d0, idx == 1 ? cut : d1);
              stack[2 * sp + 1] = make_float4(idx == 2 ? cut : d2, idx == 3 ? cut : d3, idx == 4 ? cut : d4,
                                              idx == 5 ? cut : d5);
            }
            ++sp;
          } else if (lig == 0) {
            atomicOr(&sink.counters[0], 1u);
          }
        }
      }
      node = next;
      if (alive) rec = *reinterpret_cast<const uint4 *>(&ix.nodes[node]);  // in flight during the leaf scan
      if (is_leaf) {
        const uint32_t left = cur.y, right = cur.z;
        for (uint32_t base = left; base < right; base += G) {
          const uint32_t i = base + lig;
          const bool valid = i < right;
          float dist = 0.0f;
          uint64_t pos = 0;
          if (valid) {
            const float4 *p = reinterpret_cast<const float4 *>(ix.leaf_pts) + cur.w;
            const float4 pa = p[i - left], pb = p[((right - left + 7u) & ~7u) + (i - left)];
            const float f0 = q0 - pa.x, f1 = q1 - pa.y, f2 = q2 - pa.z, f3 = q3 - pa.w;
            float r = 0.0f;
            r += f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;
            const float f4 = q4 - pb.x;
            r += f4 * f4;
            const float f5 = q5 - pb.y;
            r += f5 * f5;
            dist = r;
            pos = (uint64_t)__float_as_uint(pb.z) | ((uint64_t)__float_as_uint(pb.w) << 32);
          }
          const bool hit = valid && dist < radius;
          const uint64_t wave_mask = __ballot(hit);
          const uint64_t gmask = G == 64 ? wave_mask : ((wave_mask >> group_shift) & ((1ull << (G & 63)) - 1));
          const uint32_t rank = __popcll(gmask & ((1ull << lig) - 1));
          const uint32_t ord = hit_count + rank;
          if (hit && ord < sink.nnp && ord < out_cap) {
            Hit h;
            h.position = pos;
            h.dist = dist;
            h.slot = i;
            out[ord] = h;
          }
          hit_count += __popcll(gmask);
          n_points += min((uint32_t)G, right - base);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
    }
    if (lig == 0 && !kOverflow) {
      sink.q_count[qi] = hit_count;
      if (sink.q_nodes) sink.q_nodes[qi] = n_nodes;
      if (sink.q_points) sink.q_points[qi] = n_points;
      if (sink.results) {
        atomicAdd((unsigned long long *)&sink.results[chunk].nodes_visited, (unsigned long long)n_nodes);
        atomicAdd((unsigned long long *)&sink.results[chunk].points_tested, (unsigned long long)n_points);
        atomicAdd((unsigned long long *)&sink.results[chunk].hits_total, (unsigned long long)hit_count);
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

// Queries whose hit count exceeded the fast slots get an exact-size segment in the pool.
__global__ void OverflowPrepareKernel(const uint32_t *__restrict__ n_queries_dev,
                                      const uint32_t *__restrict__ q_count, uint32_t fast_cap, uint32_t nnp,
                                      uint32_t *__restrict__ q_ovf_off, uint32_t *__restrict__ ovf_list,
                                      uint32_t *__restrict__ counters, uint64_t pool_cap,
                                      uint64_t max_queries) {
  const uint32_t total = *n_queries_dev;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const uint32_t c = q_count[q];
    if (c > fast_cap) {
      const uint32_t need = min(c, nnp);
      const uint32_t off = atomicAdd(&counters[2], need);
      if ((uint64_t)off + need > pool_cap) {
        atomicOr(&counters[3], 1u);  // pool exhausted: reported as SMHIP_ST_HIT_POOL_FULL
        q_ovf_off[q] = 0xFFFFFFFFu;
        continue;
      }
      q_ovf_off[q] = off;
      const uint32_t k = atomicAdd(&counters[1], 1u);
      if (k < max_queries) ovf_list[k] = q;
    }
  }
}

// Per-leaf 128-byte-aligned blocks of 16-byte slots (layout in the file header): v[d] = values[vind[i] + d]
// (sigmap_adaptor.h:94-97: k-d point i is the window values[i..i+dim)), position = Point.position
// of the window's first element.  One thread per tree node; leaves write their block.
__global__ void LeafGatherKernel(const float *__restrict__ values, const uint64_t *__restrict__ positions,
                                 const uint32_t *__restrict__ vind, const smhip_node *__restrict__ nodes,
                                 uint64_t n_nodes, float *__restrict__ leaf_pts) {
  float4 *o = reinterpret_cast<float4 *>(leaf_pts);
  for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_nodes;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const smhip_node nd = nodes[k];
    if (nd.feat >= 0) continue;
    const uint32_t left = nd.a, n = nd.b - nd.a, boff = nd.child2, rn = (n + 7u) & ~7u;
    for (uint32_t t = 0; t < n; ++t) {
      const uint32_t v = vind[left + t];
      const uint64_t pos = positions[v];
      float4 a, b;
      a.x = values[(uint64_t)v + 0]; a.y = values[(uint64_t)v + 1]; a.z = values[(uint64_t)v + 2];
      a.w = values[(uint64_t)v + 3]; b.x = values[(uint64_t)v + 4]; b.y = values[(uint64_t)v + 5];
      b.z = __uint_as_float((uint32_t)pos);
      b.w = __uint_as_float((uint32_t)(pos >> 32));
      o[(size_t)boff + t] = a;
      o[(size_t)boff + rn + t] = b;
    }
  }
}

// ---- visiting order of the queries: key = tree-order index of the leaf that contains the
// query point (a stackless best-child descent), then a counting sort on key >> shift.
__global__ void QueryKeyKernel(IndexView ix, const QueryRec *__restrict__ qrec,
                               const uint32_t *__restrict__ n_queries_dev, uint32_t n_queries_host,
                               uint32_t shift, uint32_t *__restrict__ keys, uint32_t *__restrict__ hist) {
  const uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const QueryRec r = qrec[q];
    uint32_t node = 0;
    uint4 rec = *reinterpret_cast<const uint4 *>(&ix.nodes[0]);
    while ((int32_t)rec.x >= 0) {
      const float val = r.q[rec.x];
      const float diff1 = val - __uint_as_float(rec.y), diff2 = val - __uint_as_float(rec.z);
      node = (diff1 + diff2) < 0 ? node + 1 : rec.w;
      rec = *reinterpret_cast<const uint4 *>(&ix.nodes[node]);
    }
    const uint32_t key = rec.y >> shift;
    keys[q] = key;
    atomicAdd(&hist[key], 1u);
  }
}

__global__ void QueryScatterKernel(const uint32_t *__restrict__ n_queries_dev, uint32_t n_queries_host,
                                   const uint32_t *__restrict__ keys, uint32_t *__restrict__ cursor,
                                   uint32_t *__restrict__ order) {
  const uint32_t total = n_queries_dev ? *n_queries_dev : n_queries_host;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x)
    order[atomicAdd(&cursor[keys[q]], 1u)] = q;
}

constexpr int kLaneBlocksPerCU = 5;  // 5 x 256 threads = 20 waves per CU (what 92 VGPRs admit)

template <bool kOverflow>
hipError_t LaunchWalk(smhip_ctx *ctx, int G, const QueryRec *qrec, const uint32_t *n_queries_dev,
                      uint32_t n_queries_host, const Sink &sink, float radius, const uint32_t *order) {
  if (G == 1) {
    static int blocks_per_cu = 0, xcd_ranges = 1;
    if (blocks_per_cu == 0) {
      const char *e1 = getenv("SMHIP_LANE_BLOCKS_PER_CU"), *e2 = getenv("SMHIP_XCD_RANGES");
      blocks_per_cu = e1 ? atoi(e1) : kLaneBlocksPerCU;
      if (blocks_per_cu < 1 || blocks_per_cu > 8) blocks_per_cu = kLaneBlocksPerCU;
      if (e2) xcd_ranges = atoi(e2);
    }
    const uint32_t blocks = 256 * blocks_per_cu;
    const size_t stack_bytes = (size_t)blocks * kSeedThreads * kStackDepth * 2 * sizeof(float4);
    hipError_t e = ctx->walk_stack.Reserve(stack_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((SeedLaneKernel<kOverflow>), dim3(blocks), dim3(kSeedThreads), 0, ctx->stream, ctx->index,
                       qrec, n_queries_dev, n_queries_host, sink, radius, ctx->walk_stack.as<float4>(), order,
                       (order && !kOverflow) ? xcd_ranges : 0);
    return hipGetLastError();
  }
  const int blocks_per_cu = G >= 32 ? 8 : (G == 16 ? 5 : 2);
  const dim3 grid(256 * blocks_per_cu), block(kSeedThreads);
  const size_t lds = (size_t)(kSeedThreads / G) * kStackDepth * 2 * sizeof(float4);
#define SM_LAUNCH(GG)                                                                                  \
  {                                                                                                    \
    static bool attr = false;                                                                          \
    if (!attr && lds > 48 * 1024) {                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(SeedLookupKernel<GG, kOverflow>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
      if (e != hipSuccess) return e;                                                                   \
      attr = true;                                                                                     \
    }                                                                                                  \
    hipLaunchKernelGGL((SeedLookupKernel<GG, kOverflow>), grid, block, lds, ctx->stream, ctx->index,   \
                       qrec, n_queries_dev, n_queries_host, sink, radius);                             \
  }
  switch (G) {
    case 8: SM_LAUNCH(8) break;
    case 16: SM_LAUNCH(16) break;
    case 32: SM_LAUNCH(32) break;
    case 64: SM_LAUNCH(64) break;
    default: return hipErrorInvalidValue;
  }
#undef SM_LAUNCH
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchLeafGather(smhip_ctx *ctx, int dim, const float *values, const uint64_t *positions,
                            const uint32_t *vind, const smhip_node *nodes, uint64_t n_nodes, float *leaf_pts) {
  if (dim != 6) return hipErrorInvalidValue;
  hipLaunchKernelGGL(LeafGatherKernel, dim3(4096), dim3(256), 0, ctx->stream, values, positions, vind, nodes,
                     n_nodes, leaf_pts);
  return hipGetLastError();
}

hipError_t LaunchQueryExpand(smhip_ctx *ctx, const smhip_params &p, const float *feat, uint32_t feat_stride,
                             const uint32_t *q_off, uint32_t n_chunks, void *qrec) {
  hipLaunchKernelGGL(QueryExpandKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, feat, feat_stride, q_off,
                     p.step_size, reinterpret_cast<QueryRec *>(qrec));
  return hipGetLastError();
}

namespace {
// Visiting order for the lane walker (see file header).  n_max bounds the query count.
hipError_t BuildQueryOrder(smhip_ctx *ctx, const QueryRec *qrec, const uint32_t *n_queries_dev,
                           uint32_t n_queries_host, uint64_t n_max, const uint32_t **order_out) {
  uint32_t bits = 1;
  while ((1ull << bits) < ctx->index.m_size) ++bits;
  const uint32_t shift = bits > 17 ? bits - 17 : 0;
  const uint32_t n_buckets = (uint32_t)(ctx->index.m_size >> shift) + 1;
  hipError_t e;
  if ((e = ctx->q_keys.Reserve(n_max * 4)) != hipSuccess) return e;
  if ((e = ctx->q_order.Reserve(n_max * 4)) != hipSuccess) return e;
  if ((e = ctx->q_hist.Reserve(((size_t)n_buckets + 1) * 8)) != hipSuccess) return e;
  uint32_t *hist = ctx->q_hist.as<uint32_t>(), *cursor = hist + n_buckets + 1;
  if ((e = hipMemsetAsync(hist, 0, (size_t)n_buckets * 4, ctx->stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(QueryKeyKernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->index, qrec, n_queries_dev,
                     n_queries_host, shift, ctx->q_keys.as<uint32_t>(), hist);
  if ((e = LaunchScanU32(ctx, hist, n_buckets - 1, cursor)) != hipSuccess) return e;
  hipLaunchKernelGGL(QueryScatterKernel, dim3(2048), dim3(256), 0, ctx->stream, n_queries_dev, n_queries_host,
                     ctx->q_keys.as<uint32_t>(), cursor, ctx->q_order.as<uint32_t>());
  *order_out = ctx->q_order.as<uint32_t>();
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchSeedLookup(smhip_ctx *ctx, const smhip_params &p, const void *qrec, const uint32_t *q_off,
                            uint32_t n_chunks, uint64_t n_queries_max, Hit *region, uint32_t *q_count,
                            smhip_chunk_result *results, uint32_t *counters) {
  Sink sink{region, p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_count, nullptr, nullptr,
            results, counters, nullptr, nullptr};
  const QueryRec *qr = reinterpret_cast<const QueryRec *>(qrec);
  const uint32_t *order = nullptr;
  if (p.lanes_per_query == 1) {
    hipError_t e = BuildQueryOrder(ctx, qr, q_off + n_chunks, 0, n_queries_max, &order);
    if (e != hipSuccess) return e;
  }
  return LaunchWalk<false>(ctx, (int)p.lanes_per_query, qr, q_off + n_chunks, 0, sink, p.search_radius, order);
}

hipError_t LaunchSeedLookupRaw(smhip_ctx *ctx, const smhip_params &p, const float *queries, uint32_t n,
                               uint32_t cap, Hit *region, uint32_t *q_count, uint32_t *q_nodes,
                               uint32_t *q_points, uint32_t *counters, void *qrec) {
  QueryRec *qr = reinterpret_cast<QueryRec *>(qrec);
  hipLaunchKernelGGL(QueryExpandRawKernel, dim3(256), dim3(256), 0, ctx->stream, queries, n, qr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const uint32_t *order = nullptr;
  if (p.lanes_per_query == 1 && (e = BuildQueryOrder(ctx, qr, nullptr, n, n, &order)) != hipSuccess) return e;
  // the test entry point reports every hit up to `cap`: no 5000 truncation here
  Sink sink{region, cap, 0xFFFFFFFFu, q_count, q_nodes, q_points, nullptr, counters, nullptr, nullptr};
  return LaunchWalk<false>(ctx, (int)p.lanes_per_query, qr, nullptr, n, sink, p.search_radius, order);
}

hipError_t LaunchSeedOverflow(smhip_ctx *ctx, const smhip_params &p, const void *qrec, const uint32_t *q_off,
                              uint32_t n_chunks, uint32_t *q_count, uint32_t *q_ovf_off, uint32_t *ovf_list,
                              Hit *ovf_pool, uint64_t pool_cap, smhip_chunk_result *results, uint32_t *counters,
                              uint64_t max_queries) {
  hipLaunchKernelGGL(OverflowPrepareKernel, dim3(1024), dim3(256), 0, ctx->stream, q_off + n_chunks, q_count,
                     p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_ovf_off, ovf_list, counters,
                     pool_cap, max_queries);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  Sink sink{ovf_pool, p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_count, nullptr, nullptr,
            results, counters, ovf_list, q_ovf_off};
  return LaunchWalk<true>(ctx, (int)p.lanes_per_query, reinterpret_cast<const QueryRec *>(qrec), nullptr, 0, sink,
                          p.search_radius, nullptr);
}
}  // namespace smhip
