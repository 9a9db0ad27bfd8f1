// seed_lookup.hip -- batched exact radius search over the flattened k-d tree (gfx950).
//
// Replaces, per seed, nanoflann's radiusSearch -> findNeighbors -> computeInitialDistances
// -> searchLevel -> L2_Adaptor::evalMetric -> RadiusResultSet::addPoint
// (nanoflann.hpp:1278-1301, 1221-1242, 1006-1023, 1346-1410, 383-408, 249-253) as called
// from SpatialIndex::GenerateChains (spatial_index.cc:351-409).
//
// Mapping to the hardware: a sub-group of G lanes (G = 8/16/32/64) of a 64-wide wavefront
// walks one query.  All G lanes carry the same traversal state, so a node fetch is one
// 16-byte request; at a leaf the lanes fan out over the leaf's points, which are stored
// leaf-major ([m_size][6] floats in tree order) so one leaf is one contiguous <= 480-byte
// read instead of <= 20 scattered (vind -> 6 strided floats) gathers.  The recursion of
// searchLevel is replaced by an explicit stack of pending "other child" visits held in LDS
// (node, mindistsq, dists[6] = 32 bytes per entry), which reproduces nanoflann's visiting
// order exactly, so hits come out in the reference's traversal order; the k-th hit of a
// query (k = running count + rank inside the leaf) is stored at slot k, and k >= 5000
// (num_nearest_points, spatial_index.cc:371-372) is dropped exactly like the reference.
// The next node record is requested before the current leaf is scanned, so the pointer
// chase and the leaf read overlap.
#include "common.h"

namespace smhip {
namespace {

constexpr int kSeedThreads = 256;

struct QuerySource {
  // mode A: queries are windows of the per-chunk feature arrays
  const float *feat;
  uint32_t feat_stride;
  const uint32_t *q_off;  // [n_chunks + 1]
  uint32_t n_chunks;
  int step;
  // mode B: explicit n x 6 array
  const float *raw;
  uint32_t n_raw;
};

struct Sink {
  Hit *region;          // [n_queries][fast_cap]   (fast pass) or overflow pool
  uint32_t fast_cap;    // slots per query in the fast pass
  uint32_t nnp;         // num_nearest_points
  uint32_t *q_count;    // total hits per query
  uint32_t *q_nodes;    // optional per-query stats
  uint32_t *q_points;
  smhip_chunk_result *results;  // optional per-chunk accumulation
  uint32_t *counters;   // [0] = stack overflow flag
  // overflow pass
  const uint32_t *ovf_list;
  const uint32_t *q_ovf_off;
};

__device__ __forceinline__ uint32_t FindChunk(const uint32_t *q_off, uint32_t n_chunks, uint32_t q) {
  // largest c with q_off[c] <= q
  uint32_t lo = 0, hi = n_chunks;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (q_off[mid] <= q) lo = mid; else hi = mid;
  }
  return lo;
}

// One query, walked by the G lanes that share `lig`'s group.  kOverflow selects the sink.
template <int G, bool kOverflow>
__device__ __forceinline__ void WalkQuery(const IndexView &ix, const float q[6], float radius,
                                          float4 *stack, uint32_t lig, uint32_t group_shift,
                                          Hit *out, uint32_t out_cap, uint32_t nnp,
                                          uint32_t &hit_count, uint32_t &n_nodes, uint32_t &n_points,
                                          uint32_t *counters) {
  // nanoflann.hpp:1006-1023 computeInitialDistances (low test, then high test, per dim)
  float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
  float mind = 0.0f;
#define SM_INIT(k, dk)                                  \
  {                                                     \
    const float lo = ix.bbox[2 * k], hi = ix.bbox[2 * k + 1]; \
    if (q[k] < lo) {                                    \
      dk = (q[k] - lo) * (q[k] - lo);                   \
      mind += dk;                                       \
    }                                                   \
    if (q[k] > hi) {                                    \
      dk = (q[k] - hi) * (q[k] - hi);                   \
      mind += dk;                                       \
    }                                                   \
  }
  SM_INIT(0, d0) SM_INIT(1, d1) SM_INIT(2, d2) SM_INIT(3, d3) SM_INIT(4, d4) SM_INIT(5, d5)
#undef SM_INIT
  uint32_t sp = 0;
  uint32_t node = 0;
  bool alive = ix.n_nodes > 0;
  uint4 rec = make_uint4(0, 0, 0, 0);
  if (alive) rec = *reinterpret_cast<const uint4 *>(&ix.nodes[0]);
  while (alive) {
    ++n_nodes;
    const uint4 cur = rec;
    const bool is_leaf = (int32_t)cur.x < 0;
    uint32_t next = 0;
    if (is_leaf) {
      // return from searchLevel: resume the innermost pending "other child" visit
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
      // nanoflann.hpp:1371-1408
      const int idx = (int)cur.x;
      const float val = idx == 0 ? q[0] : idx == 1 ? q[1] : idx == 2 ? q[2] : idx == 3 ? q[3] : idx == 4 ? q[4] : q[5];
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
      if (nm <= radius) {  // mindistsq * epsError <= worstDist(), epsError == 1
        if (sp < (uint32_t)kStackDepth) {
          if (lig == 0) {
            stack[2 * sp] = make_float4(__uint_as_float(other), nm, idx == 0 ? cut : d0, idx == 1 ? cut : d1);
            stack[2 * sp + 1] = make_float4(idx == 2 ? cut : d2, idx == 3 ? cut : d3, idx == 4 ? cut : d4,
                                            idx == 5 ? cut : d5);
          }
          ++sp;
        } else if (lig == 0) {
          atomicOr(&counters[0], 1u);
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
        if (valid) {
          const float2 *p = reinterpret_cast<const float2 *>(ix.leaf_pts + (size_t)i * 6);
          const float2 p01 = p[0], p23 = p[1], p45 = p[2];
          // nanoflann.hpp:391-406: one group of four, then the two-element tail
          const float f0 = q[0] - p01.x, f1 = q[1] - p01.y, f2 = q[2] - p23.x, f3 = q[3] - p23.y;
          float r = 0.0f;
          r += f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;
          const float f4 = q[4] - p45.x;
          r += f4 * f4;
          const float f5 = q[5] - p45.y;
          r += f5 * f5;
          dist = r;
        }
        const bool hit = valid && dist < radius;
        const uint64_t wave_mask = __ballot(hit);
        const uint64_t gmask = G == 64 ? wave_mask : ((wave_mask >> group_shift) & ((1ull << (G & 63)) - 1));
        const uint32_t rank = __popcll(gmask & ((1ull << lig) - 1));
        const uint32_t ord = hit_count + rank;
        if (hit && ord < nnp && ord < out_cap) {
          Hit h;
          h.position = ix.leaf_pos[i];
          h.dist = dist;
          h.slot = i;
          out[ord] = h;
        }
        hit_count += __popcll(gmask);
        n_points += min((uint32_t)G, right - base);
      }
      // the stack is read by every lane of the group and written by lane 0
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
}

template <int G, bool kOverflow>
__global__ __launch_bounds__(kSeedThreads) void SeedLookupKernel(IndexView ix, QuerySource src, Sink sink,
                                                                 float radius, uint32_t total_work_override) {
  extern __shared__ __align__(16) float4 stack_mem[];
  constexpr int kGroupsPerBlock = kSeedThreads / G;
  const uint32_t tid = threadIdx.x;
  const uint32_t group_in_block = tid / G;
  const uint32_t lig = tid % G;
  const uint32_t group_shift = (tid & 63) / G * G;
  float4 *stack = stack_mem + (size_t)group_in_block * kStackDepth * 2;
  const uint32_t n_groups_total = gridDim.x * kGroupsPerBlock;
  uint32_t total;
  if (kOverflow) {
    total = sink.counters[1];  // number of overflowing queries
  } else {
    total = src.raw ? src.n_raw : src.q_off[src.n_chunks];
  }
  (void)total_work_override;
  for (uint32_t w = blockIdx.x * kGroupsPerBlock + group_in_block; w < total; w += n_groups_total) {
    const uint32_t qi = kOverflow ? sink.ovf_list[w] : w;
    float q[6];
    uint32_t chunk = 0;
    if (src.raw) {
      const float *p = src.raw + (size_t)qi * 6;
#pragma unroll
      for (int d = 0; d < 6; ++d) q[d] = p[d];
    } else {
      chunk = FindChunk(src.q_off, src.n_chunks, qi);
      const uint32_t k = qi - src.q_off[chunk];
      const float *p = src.feat + (size_t)chunk * src.feat_stride + (size_t)(k + 1) * src.step;
#pragma unroll
      for (int d = 0; d < 6; ++d) q[d] = p[d];
    }
    uint32_t hits = 0, nn = 0, np = 0;
    Hit *out;
    uint32_t out_cap;
    if (kOverflow) {
      out = sink.region + sink.q_ovf_off[qi];
      out_cap = sink.nnp;
    } else {
      out = sink.region + (size_t)qi * sink.fast_cap;
      out_cap = sink.fast_cap;
    }
    WalkQuery<G, kOverflow>(ix, q, radius, stack, lig, group_shift, out, out_cap, sink.nnp, hits, nn, np,
                            sink.counters);
    if (lig == 0 && !kOverflow) {
      sink.q_count[qi] = hits;
      if (sink.q_nodes) sink.q_nodes[qi] = nn;
      if (sink.q_points) sink.q_points[qi] = np;
      if (sink.results) {
        atomicAdd((unsigned long long *)&sink.results[chunk].nodes_visited, (unsigned long long)nn);
        atomicAdd((unsigned long long *)&sink.results[chunk].points_tested, (unsigned long long)np);
        atomicAdd((unsigned long long *)&sink.results[chunk].hits_total, (unsigned long long)hits);
      }
    }
  }
}

// Queries whose hit count exceeded the fast slots get an exact-size segment in the pool.
__global__ void OverflowPrepareKernel(const uint32_t *__restrict__ q_off, uint32_t n_chunks,
                                      const uint32_t *__restrict__ q_count, uint32_t fast_cap, uint32_t nnp,
                                      uint32_t *__restrict__ q_ovf_off, uint32_t *__restrict__ ovf_list,
                                      uint32_t *__restrict__ counters, uint64_t pool_cap,
                                      uint64_t max_queries) {
  const uint32_t total = q_off[n_chunks];
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

// leaf-major copies of the point cloud: leaf_pts[i][d] = values[vind[i] + d],
// leaf_pos[i] = positions[vind[i]]  (sigmap_adaptor.h:94-97: k-d point i = values[i..i+dim))
__global__ void LeafGatherKernel(const float *__restrict__ values, const uint64_t *__restrict__ positions,
                                 const uint32_t *__restrict__ vind, uint64_t m_size,
                                 float *__restrict__ leaf_pts, uint64_t *__restrict__ leaf_pos) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < m_size;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t v = vind[i];
#pragma unroll
    for (int d = 0; d < 6; ++d) leaf_pts[i * 6 + d] = values[(uint64_t)v + d];
    leaf_pos[i] = positions[v];
  }
}

template <bool kOverflow>
hipError_t LaunchWalk(smhip_ctx *ctx, int G, const QuerySource &src, const Sink &sink, float radius) {
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
                       src, sink, radius, 0u);                                                         \
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
                            const uint32_t *vind, uint64_t m_size, float *leaf_pts, uint64_t *leaf_pos) {
  if (dim != 6) return hipErrorInvalidValue;
  hipLaunchKernelGGL(LeafGatherKernel, dim3(2048), dim3(256), 0, ctx->stream, values, positions, vind,
                     m_size, leaf_pts, leaf_pos);
  return hipGetLastError();
}

hipError_t LaunchSeedLookup(smhip_ctx *ctx, const smhip_params &p, const float *feat, uint32_t feat_stride,
                            const uint32_t *q_off, uint32_t n_chunks, Hit *region, uint32_t *q_count,
                            smhip_chunk_result *results, uint32_t *counters) {
  QuerySource src{feat, feat_stride, q_off, n_chunks, p.step_size, nullptr, 0};
  Sink sink{region, p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_count, nullptr, nullptr,
            results, counters, nullptr, nullptr};
  return LaunchWalk<false>(ctx, (int)p.lanes_per_query, src, sink, p.search_radius);
}

hipError_t LaunchSeedLookupRaw(smhip_ctx *ctx, const smhip_params &p, const float *queries, uint32_t n,
                               uint32_t cap, Hit *region, uint32_t *q_count, uint32_t *q_nodes,
                               uint32_t *q_points, uint32_t *counters) {
  QuerySource src{nullptr, 0, nullptr, 0, p.step_size, queries, n};
  // the test entry point reports every hit up to `cap`: no 5000 truncation here
  Sink sink{region, cap, 0xFFFFFFFFu, q_count, q_nodes, q_points, nullptr, counters, nullptr, nullptr};
  return LaunchWalk<false>(ctx, (int)p.lanes_per_query, src, sink, p.search_radius);
}

hipError_t LaunchSeedOverflow(smhip_ctx *ctx, const smhip_params &p, const float *feat, uint32_t feat_stride,
                              const uint32_t *q_off, uint32_t n_chunks, uint32_t *q_count,
                              uint32_t *q_ovf_off, uint32_t *ovf_list, Hit *ovf_pool, uint64_t pool_cap,
                              smhip_chunk_result *results, uint32_t *counters, uint64_t max_queries) {
  hipLaunchKernelGGL(OverflowPrepareKernel, dim3(1024), dim3(256), 0, ctx->stream, q_off, n_chunks, q_count,
                     p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_ovf_off, ovf_list, counters,
                     pool_cap, max_queries);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  QuerySource src{feat, feat_stride, q_off, n_chunks, p.step_size, nullptr, 0};
  Sink sink{ovf_pool, p.hits_per_query_fast, (uint32_t)p.num_nearest_points, q_count, nullptr, nullptr,
            results, counters, ovf_list, q_ovf_off};
  return LaunchWalk<true>(ctx, (int)p.lanes_per_query, src, sink, p.search_radius);
}
}  // namespace smhip
