// chain.hip -- chaining DP, traceback, primary-chain selection and MAPQ; one wavefront
// per chunk.
//
// Replaces spatial_index.cc:419-576 (+ TracebackChains :165-220, GeneratePrimaryChains
// :222-253, ComputeMAPQ :255-274) and the chain-0 statistics of sigmap.cc:714-729.
//
// The DP of one (reference, strand) bucket is sequential (score[i] depends on earlier
// scores and on the order-dependent num_skips early exit, spatial_index.cc:461-540), but
// buckets are independent, so the 64 lanes of the wave take buckets round-robin.  The only
// cross-bucket coupling in the reference is the running `max_chaining_score`
// (spatial_index.cc:420,542-549): it never influences scores, only which anchors qualify
// as chain ends, so it is re-applied after the DP as (prefix max over earlier buckets) +
// (running max inside the bucket), which is the same sequence of values.
#include <float.h>

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

__global__ __launch_bounds__(64) void ChainKernel(ChainArgs a, float radius, int dim, int max_gap_length,
                                                  int max_target_gap_length, int chaining_band_length,
                                                  int max_num_skips, int min_num_anchors, int num_best_chains,
                                                  float min_chaining_score) {
  __shared__ uint32_t s_nchains;
  const uint32_t c = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint64_t base = a.anchor_off[c];
  const uint32_t n = (uint32_t)(a.anchor_off[c + 1] - base);
  const uint32_t G = a.n_groups[c];
  if (lane == 0) {
    s_nchains = 0;
    a.out_anchor_cnt[c] = 0;
  }
  __syncthreads();
  if (n == 0 || G == 0) return;
  const uint32_t *tpos = a.s_tpos + base, *qpos = a.s_qpos + base, *grp = a.s_grp + base;
  const float *dist = a.s_dist + base;
  const uint32_t *gstart = a.group_start + base;
  float *score = a.score + base;
  uint32_t *pred = a.pred + base;
  uint8_t *used = a.used + base;
  // dead sort scratch (2n keys per chunk) holds the per-bucket maxima and their prefix
  float *gmax = reinterpret_cast<float *>(a.scratch + 2 * base);
  float *pmax = gmax + n;

  // ---- phase A: DP per bucket (spatial_index.cc:436-541)
  for (uint32_t g = lane; g < G; g += 64) {
    const uint32_t s = gstart[g], e = g + 1 < G ? gstart[g + 1] : n;
    float gm = 0.0f;
    for (uint32_t ai = s; ai < e; ++ai) {
      const uint32_t anchor_index = ai - s;
      const float coef = (float)(1 - 0.2 * (double)dist[ai] / (double)radius);
      float best = coef * (float)dim;
      uint32_t best_pred = anchor_index;
      const int32_t ct = (int32_t)tpos[ai], cq = (int32_t)qpos[ai];
      int32_t start_idx = 0;
      if (anchor_index > (uint32_t)chaining_band_length) start_idx = (int32_t)anchor_index - chaining_band_length;
      int32_t num_skips = 0;
      for (int32_t pj = (int32_t)anchor_index - 1; pj >= start_idx; --pj) {
        const int32_t pt = (int32_t)tpos[s + pj], pq = (int32_t)qpos[s + pj];
        if (pq == cq) continue;
        if (pt == ct) continue;
        if (pt + max_target_gap_length < ct) break;
        const int32_t td = ct - pt;
        const int32_t qd = cq - pq;
        float cur = 0;
        if (qd < 0) continue;
        {
          const int32_t m = min(min(td, qd), dim);
          const float matching = (float)m * coef;
          const int gap_length = abs(td - qd);
          const float gap_scale = td > 0 ? (float)qd / (float)td : 1.0f;
          if (gap_length < max_gap_length && gap_scale < 5.0f && gap_scale > 0.75f) cur = score[s + pj] + matching;
        }
        if (cur > best) {
          best = cur;
          best_pred = (uint32_t)pj;
          --num_skips;
        } else {
          ++num_skips;
          if (num_skips > max_num_skips) break;
        }
      }
      score[ai] = best;
      pred[ai] = best_pred;
      used[ai] = 0;
      gm = fmaxf(gm, best);
    }
    gmax[g] = gm;
  }
  __syncthreads();

  // ---- phase B: max_chaining_score on entry to each bucket = max over all earlier buckets
  // (order ref0+, ref0-, ref1+, ...; initial value 0, spatial_index.cc:420): exclusive
  // prefix max by wave shuffles, 64 buckets at a time.
  {
    float carry = 0.0f;
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
      const uint32_t g = g0 + lane;
      float v = g < G ? gmax[g] : 0.0f;
      for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d, 64);
        if ((int)lane >= d) v = fmaxf(v, o);
      }
      float ex = __shfl_up(v, 1, 64);
      if (lane == 0) ex = 0.0f;
      if (g < G) pmax[g] = fmaxf(carry, ex);
      carry = fmaxf(carry, __shfl(v, 63, 64));
    }
  }
  __syncthreads();

  // ---- phase C: chain ends, traceback
  for (uint32_t g = lane; g < G; g += 64) {
    const uint32_t s = gstart[g], e = g + 1 < G ? gstart[g + 1] : n;
    float run_max = pmax[g];  // max_chaining_score on entry to this bucket
    Cand top[3];
    uint32_t ntop = 0;
    uint32_t ncand = 0;
    for (uint32_t ai = s; ai < e; ++ai) {
      const float sc = score[ai];
      if (sc > run_max) run_max = sc;
      if (sc >= min_chaining_score && sc > run_max / 2) {  // spatial_index.cc:545-549
        ++ncand;
        Cand cnd{sc, ai - s};
        // keep the best three in sorted order (num_best_chains == 3)
        uint32_t pos = ntop;
        while (pos > 0 && CandBefore(cnd, top[pos - 1])) --pos;
        if (pos < 3) {
          for (uint32_t k = (ntop < 3 ? ntop : 2); k > pos; --k) top[k] = top[k - 1];
          top[pos] = cnd;
          if (ntop < 3) ++ntop;
        }
      }
    }
    (void)ncand;
    const float max_after = run_max;  // max_chaining_score after the bucket
    const uint32_t grp_id = grp[s];
    const uint32_t ref = grp_id >> 1;
    const uint32_t direction = (grp_id & 1) == 0 ? 1u : 0u;  // strand 0 = Positive
    const uint32_t lim = min(ntop, (uint32_t)min(num_best_chains, 3));
    for (uint32_t k = 0; k < lim; ++k) {
      // TracebackChains, spatial_index.cc:165-220
      const uint32_t end_idx = top[k].idx;
      if (!used[s + end_idx]) {
        bool stop_at_used = false;
        uint32_t cur = end_idx;
        if (pred[s + cur] != cur && used[s + pred[s + cur]]) stop_at_used = true;
        used[s + cur] = 1;
        uint32_t num = 1;
        while (pred[s + cur] != cur && !used[s + pred[s + cur]]) {
          cur = pred[s + cur];
          if (pred[s + cur] != cur && used[s + pred[s + cur]]) stop_at_used = true;
          used[s + cur] = 1;
          ++num;
        }
        if (num >= (uint32_t)min_num_anchors) {
          float adjusted = score[s + end_idx];
          if (stop_at_used) adjusted -= score[s + pred[s + cur]];
          const uint32_t slot = atomicAdd(&s_nchains, 1u);
          if (slot < (uint32_t)kChainCap) {
            ChainTmp ch;
            ch.score = adjusted;
            ch.ref = ref;
            ch.start = tpos[s + cur];
            ch.end = tpos[s + end_idx];
            ch.n_anchors = num;
            ch.direction = direction;
            ch.end_anchor = s + end_idx;
            a.chain_tmp[(size_t)c * kChainCap + slot] = ch;
          }
        }
      }
      if (score[s + end_idx] < max_after / 2) break;  // spatial_index.cc:563-567
    }
  }
  __syncthreads();

  // ---- phase D: GeneratePrimaryChains + ComputeMAPQ + record (one lane)
  if (lane != 0) return;
  smhip_chunk_result &res = a.results[c];
  uint32_t nch = s_nchains;
  if (nch > (uint32_t)kChainCap) {
    res.status |= SMHIP_ST_CHAIN_CAP;
    nch = kChainCap;
  }
  if (nch == 0) return;
  ChainTmp *ch = a.chain_tmp + (size_t)c * kChainCap;
  for (uint32_t i = 1; i < nch; ++i) {  // insertion sort, descending by operator>
    const ChainTmp x = ch[i];
    uint32_t j = i;
    while (j > 0 && ChainGreater(x, ch[j - 1])) {
      ch[j] = ch[j - 1];
      --j;
    }
    ch[j] = x;
  }
  // spatial_index.cc:222-253: primaries are compacted to the front of ch[]
  uint32_t np = 1;
  for (uint32_t ci = 1; ci < nch; ++ci) {
    bool is_primary = true;
    if (ch[ci].score < ch[np - 1].score / 3) break;
    for (uint32_t pi = 0; pi < np; ++pi) {
      if (ch[ci].ref == ch[pi].ref) {
        if (max(ch[ci].start, ch[pi].start) > min(ch[ci].end, ch[pi].end)) {
        } else {
          is_primary = false;
          break;
        }
      }
    }
    if (is_primary) {
      const ChainTmp x = ch[ci];
      ch[np++] = x;
    }
  }
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
  // anchors of every primary chain, end -> start (the order TracebackChains pushes them),
  // for the next chunk's carry-over; chain 0 statistics as sigmap.cc:714-729 accumulates them
  uint32_t w = 0;
  smhip_carried_anchor *out = a.out_anchors ? a.out_anchors + base : nullptr;
  for (uint32_t i = 0; i < np; ++i) {
    // pred[] holds bucket-local indices: recover the bucket start of this chain's anchors
    const uint32_t g_id = grp[ch[i].end_anchor];
    uint32_t lo = 0, hi = G;  // last group start <= end_anchor
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (gstart[mid] <= ch[i].end_anchor) lo = mid; else hi = mid;
    }
    const uint32_t s = gstart[lo];
    uint32_t cur = ch[i].end_anchor - s;
    float ad = 0, at = 0, aq = 0;
    for (uint32_t k = 0; k < ch[i].n_anchors; ++k) {
      const uint32_t ai = s + cur;
      if (out) {
        smhip_carried_anchor o;
        o.ref_strand = g_id;
        o.target_position = tpos[ai];
        o.query_position = qpos[ai];
        o.distance = dist[ai];
        out[w] = o;
      }
      ++w;
      if (i == 0) {
        ad += dist[ai];
        if (k == 0) res.c0_first_qpos = qpos[ai];
        if (k + 1 == ch[i].n_anchors) res.c0_last_qpos = qpos[ai];
        if (k + 1 < ch[i].n_anchors) {
          const uint32_t nx = s + pred[ai];
          at += (float)(uint32_t)(tpos[ai] - tpos[nx]);
          aq += (float)(uint32_t)(qpos[ai] - qpos[nx]);
        }
      }
      cur = pred[ai];
    }
    if (i == 0) {
      res.c0_sum_dist = ad;
      res.c0_sum_tgap = at;
      res.c0_sum_qgap = aq;
    }
  }
  a.out_anchor_cnt[c] = w;
}
}  // namespace

hipError_t LaunchChain(smhip_ctx *ctx, const smhip_params &p, const ChainArgs &a, uint32_t n_chunks) {
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(ChainKernel, dim3(n_chunks), dim3(64), 0, ctx->stream, a, p.search_radius, kMaxDim,
                     p.max_gap_length, p.max_target_gap_length, p.chaining_band_length, p.max_num_skips,
                     p.min_num_anchors, p.num_best_chains, p.min_chaining_score);
  return hipGetLastError();
}
}  // namespace smhip
