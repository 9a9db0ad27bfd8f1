// sdtw.hip -- banded subsequence DTW, anti-diagonal wavefront sweep with the three live
// diagonals in LDS; one wavefront per (target window, query) problem.
//
// Recurrence and boundary of Sigmap::sDTW (sigmap.cc:1259-1291): rows = target (free start:
// column 0 is 0 in every row), columns = query, cost |t - q|,
//   D[t][q] = cost + min(D[t-1][q-1], D[t-1][q], D[t][q-1]),
// answer = min over rows of D[t][Q], first (smallest t) minimum, t reported 1-based.
// The band restricts row t to columns [lo[t], hi[t]] (1-based, inclusive); cells outside are
// FLT_MAX, which is also what fast_dtw.cc's windowed DTW() does for cells not in its window
// (fast_dtw.cc:96-98,131-133).  With lo = 1, hi = Q the result equals the unbanded sDTW.
//
// Cells on one anti-diagonal (t + q = d) depend only on diagonals d-1 and d-2, so the 64
// lanes sweep a diagonal in parallel (no MFMA: min-plus DP, not a contraction).
#include <float.h>

#include <stdlib.h>

#include "common.h"

namespace smhip {
namespace {
constexpr int kSdtwMaxT = 8192;

__global__ __launch_bounds__(64) void SdtwKernel(const float *__restrict__ target,
                                                 const uint64_t *__restrict__ t_off,
                                                 const float *__restrict__ query,
                                                 const uint64_t *__restrict__ q_off,
                                                 const uint32_t *__restrict__ lo,
                                                 const uint32_t *__restrict__ hi, float *__restrict__ dist,
                                                 int64_t *__restrict__ end) {
  extern __shared__ float sm[];
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  const uint64_t tb = t_off[p], qb = q_off[p];
  const uint32_t T = (uint32_t)(t_off[p + 1] - tb), Q = (uint32_t)(q_off[p + 1] - qb);
  const uint32_t stride = T + 1;
  float *d0 = sm, *d1 = sm + stride, *d2 = sm + 2 * stride;  // diagonals d-2, d-1, d (indexed by row t)
  const float *tg = target + tb, *qr = query + qb;
  const uint32_t *plo = lo + tb, *phi = hi + tb;
  float best = FLT_MAX;
  uint32_t best_t = 0xFFFFFFFFu;
  if (T == 0 || Q == 0 || T > (uint32_t)kSdtwMaxT) {
    if (lane == 0) {
      dist[p] = FLT_MAX;
      end[p] = -1;
    }
    return;
  }
  // diagonal index d = t + q.  d = 0: cell (0,0) = 0.  d = 1: (0,1) = FLT_MAX, (1,0) = 0.
  for (uint32_t t = lane; t <= T; t += 64) {
    d0[t] = t == 0 ? 0.0f : FLT_MAX;                  // diagonal 0 holds only (0,0)
    d1[t] = t == 1 ? 0.0f : FLT_MAX;                  // (1,0) = 0, (0,1) = FLT_MAX
    d2[t] = FLT_MAX;
  }
  __syncthreads();
  for (uint32_t d = 2; d <= T + Q; ++d) {
    // rows on this diagonal: t in [max(0, d-Q), min(T, d)], q = d - t
    const uint32_t t_lo = d > Q ? d - Q : 0, t_hi = d < T ? d : T;
    for (uint32_t t = t_lo + lane; t <= t_hi; t += 64) {
      const uint32_t q = d - t;
      float v;
      if (q == 0) {
        v = 0.0f;  // current_row[0] = 0
      } else if (t == 0) {
        v = FLT_MAX;  // previous_row initial state
      } else if (q < plo[t - 1] || q > phi[t - 1]) {
        v = FLT_MAX;
      } else {
        const float cost = fabsf(tg[t - 1] - qr[q - 1]);
        const float a = d0[t - 1];  // (t-1, q-1)
        const float b = d1[t - 1];  // (t-1, q)
        const float c = d1[t];      // (t, q-1)
        float m = a;                // std::min({a, b, c})
        if (b < m) m = b;
        if (c < m) m = c;
        v = cost + m;
      }
      d2[t] = v;
      if (q == Q && t >= 1) {
        if (v < best || (v == best && t < best_t)) {
          best = v;
          best_t = t;
        }
      }
    }
    __syncthreads();
    float *tmp = d0;
    d0 = d1;
    d1 = d2;
    d2 = tmp;
  }
  // wave reduction: minimum value, then smallest row
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_down(best, off, 64);
    const uint32_t ot = __shfl_down(best_t, off, 64);
    if (ob < best || (ob == best && ot < best_t)) {
      best = ob;
      best_t = ot;
    }
  }
  if (lane == 0) {
    // the reference only records rows whose value is < FLT_MAX (strict < against the initial max)
    if (best < FLT_MAX) {
      dist[p] = best;
      end[p] = (int64_t)best_t;
    } else {
      dist[p] = FLT_MAX;
      end[p] = -1;
    }
  }
}

__device__ __forceinline__ int64_t FloorDiv(int64_t a, int64_t b) {  // b > 0
  int64_t q = a / b;
  if ((a % b) != 0 && a < 0) --q;
  return q;
}

// ---- D5: fast_dtw.cc::DTW() (fast_dtw.cc:74-140) over per-row column windows, with the path.
// Same boundary as above (previous_row = {0, FLT_MAX...}, current_row[0] = 0: fast_dtw.cc:94-99) but
// FOUR candidates per cell, in this order (fast_dtw.cc:113):
//   { D[t-1][q-1] + cost, D[t][q-1] + cost, D[t][q-1] + 2, D[t-1][q] + 2 }     (skip costs 2, :76-77)
// the FIRST minimum wins (std::min_element) and its index 0..3 is the cell's path code (:127-129).
// A row can end the alignment iff its last window cell is the last query column; first row with the
// smallest value (strict <, :131-134).  The codes of the in-window cells go to global memory (one byte
// per cell, row r at cell_off[r]); after the sweep lane 0 walks back like generate_path (:51-72) and
// writes the codes from the end cell to the first cell that lies in query column 0.
// t_off[p] = first ROW of problem p in lo / hi / cell_off; the target samples start at tgt_off[p]
// (nullptr: at t_off[p]); the query at q_off[p] with q_len[p] values (nullptr: q_off[p+1] - q_off[p]).
template <typename CellOffT>
__global__ __launch_bounds__(64) void DtwPathKernel(const float *__restrict__ target, const uint64_t *__restrict__ t_off,
                                                    const uint64_t *__restrict__ tgt_off,
                                                    const float *__restrict__ query, const uint64_t *__restrict__ q_off,
                                                    const uint32_t *__restrict__ q_len,
                                                    const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi,
                                                    const CellOffT *__restrict__ cell_off, uint8_t *__restrict__ codes,
                                                    float *__restrict__ dist, int64_t *__restrict__ end,
                                                    uint8_t *__restrict__ path, const uint64_t *__restrict__ path_off,
                                                    uint32_t *__restrict__ path_len, int64_t *__restrict__ start_t) {
  extern __shared__ float sm[];
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  const uint64_t tb = t_off[p], qb = q_off[p];
  const uint32_t T = (uint32_t)(t_off[p + 1] - tb), Q = q_len ? q_len[p] : (uint32_t)(q_off[p + 1] - qb);
  const uint32_t stride = T + 1;
  float *d0 = sm, *d1 = sm + stride, *d2 = sm + 2 * stride;  // diagonals d-2, d-1, d (indexed by row t)
  const float *tg = target + (tgt_off ? tgt_off[p] : tb), *qr = query + qb;
  const uint32_t *plo = lo + tb, *phi = hi + tb;
  const CellOffT *coff = cell_off + tb;
  float best = FLT_MAX;
  uint32_t best_t = 0xFFFFFFFFu;
  if (lane == 0) {
    dist[p] = FLT_MAX;
    end[p] = -1;
    path_len[p] = 0;
    start_t[p] = -1;
  }
  if (T == 0 || Q == 0 || T > (uint32_t)kSdtwMaxT) return;
  for (uint32_t t = lane; t <= T; t += 64) {
    d0[t] = t == 0 ? 0.0f : FLT_MAX;
    d1[t] = t == 1 ? 0.0f : FLT_MAX;
    d2[t] = FLT_MAX;
  }
  __syncthreads();
  for (uint32_t d = 2; d <= T + Q; ++d) {
    const uint32_t t_lo = d > Q ? d - Q : 0, t_hi = d < T ? d : T;
    for (uint32_t t = t_lo + lane; t <= t_hi; t += 64) {
      const uint32_t q = d - t;
      float v;
      if (q == 0) {
        v = 0.0f;  // current_row[0] = 0
      } else if (t == 0) {
        v = FLT_MAX;  // previous_row initial state
      } else {
        const uint32_t a = max(plo[t - 1], 1u), b = min(phi[t - 1], Q);
        if (q < a || q > b) {
          v = FLT_MAX;
        } else {
          const float cost = fabsf(tg[t - 1] - qr[q - 1]);
          const float diag = d0[t - 1], up = d1[t - 1], left = d1[t];
          const float c0 = diag + cost, c1 = left + cost, c2 = left + 2.0f, c3 = up + 2.0f;
          v = c0;
          uint32_t arg = 0;
          if (c1 < v) { v = c1; arg = 1; }
          if (c2 < v) { v = c2; arg = 2; }
          if (c3 < v) { v = c3; arg = 3; }
          codes[coff[t - 1] + (q - a)] = (uint8_t)arg;
          if (q == Q && (v < best || (v == best && t < best_t))) {  // b == Q: the row's last cell
            best = v;
            best_t = t;
          }
        }
      }
      d2[t] = v;
    }
    __syncthreads();
    float *tmp = d0;
    d0 = d1;
    d1 = d2;
    d2 = tmp;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_down(best, off, 64);
    const uint32_t ot = __shfl_down(best_t, off, 64);
    if (ob < best || (ob == best && ot < best_t)) {
      best = ob;
      best_t = ot;
    }
  }
  __threadfence_block();  // the codes written by the other lanes
  __syncthreads();
  if (lane != 0 || !(best < FLT_MAX)) return;
  dist[p] = best;
  end[p] = (int64_t)best_t - 1;  // window[end][0].target_coord: 0-based (fast_dtw.cc:138)
  // generate_path
  uint8_t *out = path + path_off[p];
  const uint64_t cap = path_off[p + 1] - path_off[p];
  uint32_t t = best_t, q = Q, n = 0;
  bool ok = true;
  while (q != 1) {
    const uint32_t a = max(plo[t - 1], 1u);
    const uint8_t c = codes[coff[t - 1] + (q - a)];
    if (n < cap) out[n] = c;
    ++n;
    if (c == 0) { --t; --q; }
    else if (c == 3) { --t; }
    else { --q; }
    if (t == 0) { ok = false; break; }
    const uint32_t na = max(plo[t - 1], 1u), nb = min(phi[t - 1], Q);
    if (q < na || q > nb) { ok = false; break; }  // (a finite cell's predecessor is always in the window)
  }
  if (ok) {
    const uint32_t a = max(plo[t - 1], 1u);
    if (n < cap) out[n] = codes[coff[t - 1] + (q - a)];
    ++n;
    if (n <= cap) {
      path_len[p] = n;  // end -> start order; the host reverses (std::reverse, fast_dtw.cc:71)
      start_t[p] = (int64_t)t - 1;
    }
  }
}

// Refinement of chain 0 of every chunk (contract: include/sigmap_hip.h smhip_refine_run).
// Same sweep as SdtwKernel; the band is the linear diagonal through the chain's end anchors.
__global__ __launch_bounds__(64) void SdtwChainKernel(const smhip_chunk_result *__restrict__ results,
                                                      const float *__restrict__ feat, uint32_t feat_stride,
                                                      const uint32_t *__restrict__ qstart,
                                                      const float *__restrict__ ref_sig,
                                                      const uint64_t *__restrict__ ref_off, uint32_t n_refs,
                                                      uint32_t band, uint32_t pad, int dim,
                                                      float *__restrict__ dist, int64_t *__restrict__ end_pos,
                                                      uint32_t max_rows, int second_pass, uint32_t ring) {
  extern __shared__ float sm[];
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  // pass 1 runs with LDS for max_rows rows and leaves longer windows (end_pos = -2) to pass 2
  if (second_pass && end_pos[c] != -2) return;
  const smhip_chunk_result &res = results[c];
  bool ok = res.n_chains >= 1 && res.chains[0].ref < n_refs;
  uint32_t T = 0, Q = res.n_features;
  int64_t t0 = 0, start = 0, den = 1, qa = 0, qb = 0;
  const float *tg = nullptr;
  if (ok) {
    const smhip_chain ch = res.chains[0];
    const uint32_t strand = ch.direction == 1 ? 0u : 1u;
    const uint64_t so = ref_off[2 * ch.ref + strand], se = ref_off[2 * ch.ref + strand + 1];
    const int64_t len = (int64_t)(se - so);
    start = ch.start;
    const int64_t endp = ch.end;
    t0 = start - (int64_t)pad < 0 ? 0 : start - (int64_t)pad;
    int64_t t1 = endp + dim + (int64_t)pad;
    if (t1 > len) t1 = len;
    const uint32_t qs = qstart ? qstart[c] : 0;
    qa = (int64_t)res.c0_last_qpos - qs;
    qb = (int64_t)res.c0_first_qpos - qs;
    if (qa < 0) qa = 0;
    if (qb < 0) qb = 0;
    den = endp - start > 0 ? endp - start : 1;
    if (t1 <= t0 || t1 - t0 > (int64_t)kSdtwMaxT || Q == 0) {
      ok = false;
    } else if (t1 - t0 > (int64_t)max_rows) {
      if (lane == 0) {
        dist[c] = FLT_MAX;
        end_pos[c] = -2;
      }
      return;
    } else {
      T = (uint32_t)(t1 - t0);
      tg = ref_sig + so + t0;
    }
  }
  if (!ok) {
    if (lane == 0) {
      dist[c] = FLT_MAX;
      end_pos[c] = -1;
    }
    return;
  }
  const float *qr = feat + (size_t)c * feat_stride;
  const uint32_t stride = T + 2;
  // The three live diagonals are rings of `ring` (a power of two >= band + 16) cells indexed by
  // t mod ring: a diagonal only ever touches the rows [tb - 3, tb + band + 6] (below), and every
  // cell a later diagonal reads was written by the diagonal it belongs to, so rows a ring apart
  // never meet.  That keeps LDS at ~6 KB per wave whatever the window length (occupancy is what
  // this latency-bound sweep lives on).  Cells with t >= 1, q >= 1 only.
  const uint32_t rmask = ring - 1;
  float *d0 = sm, *d1 = sm + ring, *d2 = sm + 2 * ring;
  int32_t *cen = reinterpret_cast<int32_t *>(sm + 3 * ring);  // band centre of every row
  const int64_t half = band / 2;
  float best = FLT_MAX;
  uint32_t best_t = 0xFFFFFFFFu;
  for (uint32_t t = lane; t < ring; t += 64) {
    d0[t] = FLT_MAX;
    d1[t] = FLT_MAX;
    d2[t] = FLT_MAX;
  }
  for (uint32_t t = lane; t < stride; t += 64) {
    const int64_t ta = t0 + (int64_t)t - 1;
    cen[t] = (int32_t)(1 + qa + FloorDiv((ta - start) * (qb - qa), den));
  }
  __syncthreads();
  // g(t) = t + centre(t) is strictly increasing, and cell (t, q) is inside the band only if
  // |t + q - g(t)| <= half, so on diagonal d = t + q the band is the short row interval that
  // starts at tb = first t with g(t) + half >= d; tb only moves forward as d grows.  Each
  // diagonal writes that interval plus a margin of FLT_MAX cells, which is all the next two
  // diagonals can read.  Row 0 and column 0 are constants (FLT_MAX / 0) and are never stored.
  // (all quantities below fit 32 bits: rows and columns are < 2^14)
  // tb is found without a dependent LDS read per diagonal: lane l keeps g(tb_base + l) + half in a
  // register (INT_MAX past the last row), the qualifying rows are a suffix of the wave (g grows),
  // and the registers are re-based when tb has moved half a wave.
  const int32_t half32 = (int32_t)half, Q32 = (int32_t)Q;
  uint32_t tb = 1, tb_base = 1, te_hi = 0xFFFFFFFFu;
  auto g_of = [&](uint32_t row) { return row <= T ? (int32_t)row + cen[row] + half32 : 0x7FFFFFFF; };
  int32_t g_lane = g_of(tb_base + lane);
  // The interval of diagonal d + 1 is computed while diagonal d is swept, and the operands of its
  // first-pass cell that do not depend on diagonal d (signal values, band centre) are loaded then:
  // the sweep of a diagonal waits for LDS only.
  auto interval = [&](uint32_t d, uint32_t &w_lo, uint32_t &w_hi) {
    for (;;) {
      const uint64_t mk = __ballot(g_lane >= (int32_t)d);
      const uint32_t adv = mk ? (uint32_t)__builtin_ctzll(mk) : 64u;
      tb = tb_base + adv;
      // rows whose band has begun on this diagonal (g - half <= d): a prefix of the wave; when it
      // ends inside the wave its last row bounds the interval from above
      const uint64_t bk = __ballot(g_lane <= (int32_t)d + 2 * half32);
      te_hi = (bk >> 63) ? 0xFFFFFFFFu : tb_base + (uint32_t)__popcll(bk) + 2u;  // last begun row + 3
      if (adv < 32u || tb > T) break;
      tb_base = tb;
      g_lane = g_of(tb_base + lane);
    }
    const uint32_t t_min = d > Q ? d - Q : 1, t_max = d - 1 < T ? d - 1 : T;  // q in [1, Q], t in [1, T]
    w_lo = max(t_min, tb > 3 ? tb - 3 : 1u);
    w_hi = min(t_max, min(te_hi, tb + band + 6));
  };
  struct Ops {
    float tv, qv;
    int32_t centre;
  };
  auto fetch = [&](uint32_t d, uint32_t t) {  // clamped: always addressable
    Ops o;
    const uint32_t tc = min(t, T);
    o.tv = tg[tc - 1];
    o.qv = qr[min(max((int32_t)(d - t) - 1, 0), Q32 - 1)];
    o.centre = cen[tc];
    return o;
  };
  auto cell = [&](uint32_t d, uint32_t t, const Ops &o) {
    const int32_t q = (int32_t)(d - t);
    float v = FLT_MAX;
    // the three ring cells are read unconditionally (always addressable), so that they are issued
    // together instead of one after the other behind the boundary tests
    const float ra = d0[(t - 1) & rmask], rb = d1[(t - 1) & rmask], rc = d1[t & rmask];
    const int32_t lo = max(o.centre - half32, 1), hi = min(o.centre + half32, Q32);
    if (q >= lo && q <= hi) {
      const float cost = fabsf(o.tv - o.qv);
      // (t-1, q-1), (t-1, q), (t, q-1) with the boundary constants of sigmap.cc:1263-1271
      const float a = q == 1 ? 0.0f : (t == 1 ? FLT_MAX : ra);
      const float b = t == 1 ? FLT_MAX : rb;
      const float cc = q == 1 ? 0.0f : rc;
      float m = a;
      if (b < m) m = b;
      if (cc < m) m = cc;
      v = cost + m;
    }
    d2[t & rmask] = v;
    if (q == Q32) {
      if (v < best || (v == best && t < best_t)) {
        best = v;
        best_t = t;
      }
    }
  };
  uint32_t w_lo, w_hi, n_lo = 0, n_hi = 0;
  interval(2, w_lo, w_hi);
  Ops cur = fetch(2, w_lo + lane), nxt = cur;
  for (uint32_t d = 2; d <= T + Q; ++d) {
    if (d < T + Q) {
      interval(d + 1, n_lo, n_hi);
      nxt = fetch(d + 1, n_lo + lane);
    }
    if (w_lo <= w_hi) {
      const uint32_t t = w_lo + lane;
      if (t <= w_hi) cell(d, t, cur);
      for (uint32_t t2 = t + 64; t2 <= w_hi; t2 += 64) cell(d, t2, fetch(d, t2));  // rare second pass
    }
    __syncthreads();
    float *tmp = d0;
    d0 = d1;
    d1 = d2;
    d2 = tmp;
    w_lo = n_lo;
    w_hi = n_hi;
    cur = nxt;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_down(best, off, 64);
    const uint32_t ot = __shfl_down(best_t, off, 64);
    if (ob < best || (ob == best && ot < best_t)) {
      best = ob;
      best_t = ot;
    }
  }
  if (lane == 0) {
    if (best < FLT_MAX) {
      dist[c] = best;
      end_pos[c] = t0 + (int64_t)best_t;
    } else {
      dist[c] = FLT_MAX;
      end_pos[c] = -1;
    }
  }
}

// Register version of the same sweep for narrow bands: lane l owns the rows t = l (mod 64), one at a
// time, keeps its row's two newest cells in registers and receives the neighbouring row's newest
// cell with one wave rotate per diagonal -- no LDS traffic for the DP itself (the query and the band
// centres are staged in LDS once).  It needs the rows that are live at the same time to span
// fewer than 62 rows, which is checked up front on the band centres (g(t + 62) - g(t) > band + 3:
// row t + 62 begins after row t has ended); problems that fail the check, have more than
// `max_rows` rows or more than kSysQ query values are left to the ring kernel (end_pos = -2).
// Same cells, same operations, same tie rule: results are bit-identical to the ring kernel's.
constexpr uint32_t kSysQ = 640;  // query values staged in LDS (a 4000-sample chunk has ~450 events)

__global__ __launch_bounds__(64) void SdtwChainSystolicKernel(const smhip_chunk_result *__restrict__ results,
                                                              const float *__restrict__ feat, uint32_t feat_stride,
                                                              const uint32_t *__restrict__ qstart,
                                                              const float *__restrict__ ref_sig,
                                                              const uint64_t *__restrict__ ref_off, uint32_t n_refs,
                                                              uint32_t band, uint32_t pad, int dim,
                                                              float *__restrict__ dist, int64_t *__restrict__ end_pos,
                                                              uint32_t max_rows) {
  extern __shared__ float sm[];
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  const smhip_chunk_result &res = results[c];
  bool ok = res.n_chains >= 1 && res.chains[0].ref < n_refs;
  uint32_t T = 0, Q = res.n_features;
  int64_t t0 = 0, start = 0, den = 1, qa = 0, qb = 0;
  const float *tg = nullptr;
  bool defer = false;
  if (ok) {
    const smhip_chain ch = res.chains[0];
    const uint32_t strand = ch.direction == 1 ? 0u : 1u;
    const uint64_t so = ref_off[2 * ch.ref + strand], se = ref_off[2 * ch.ref + strand + 1];
    const int64_t len = (int64_t)(se - so);
    start = ch.start;
    const int64_t endp = ch.end;
    t0 = start - (int64_t)pad < 0 ? 0 : start - (int64_t)pad;
    int64_t t1 = endp + dim + (int64_t)pad;
    if (t1 > len) t1 = len;
    const uint32_t qs0 = qstart ? qstart[c] : 0;
    qa = (int64_t)res.c0_last_qpos - qs0;
    qb = (int64_t)res.c0_first_qpos - qs0;
    if (qa < 0) qa = 0;
    if (qb < 0) qb = 0;
    den = endp - start > 0 ? endp - start : 1;
    if (t1 <= t0 || t1 - t0 > (int64_t)kSdtwMaxT || Q == 0) {
      ok = false;
    } else if (t1 - t0 > (int64_t)max_rows || Q > kSysQ) {
      defer = true;
    } else {
      T = (uint32_t)(t1 - t0);
      tg = ref_sig + so + t0;
    }
  }
  if (!ok || defer) {
    if (lane == 0) {
      dist[c] = FLT_MAX;
      end_pos[c] = ok ? -2 : -1;
    }
    return;
  }
  // 4.6 KB of LDS per wave: what this sweep lives on is the number of waves a SIMD can interleave
  float *qs = sm;                                           // [kSysQ] the query
  int16_t *cen = reinterpret_cast<int16_t *>(sm + kSysQ);   // [T + 2] band centre of every row (< 2^15)
  const float *qr = feat + (size_t)c * feat_stride;
  for (uint32_t q = lane; q < Q; q += 64) qs[q] = qr[q];
  bool out_of_range = false;
  for (uint32_t t = lane; t < T + 2; t += 64) {
    const int64_t ta = t0 + (int64_t)t - 1;
    const int64_t cv = 1 + qa + FloorDiv((ta - start) * (qb - qa), den);
    if (cv < -32768 || cv > 32767) out_of_range = true;
    cen[t] = (int16_t)cv;
  }
  __syncthreads();
  const int32_t half32 = (int32_t)(band / 2), Q32 = (int32_t)Q;
  {
    bool wide = out_of_range;  // (a centre that does not fit 16 bits: left to the ring kernel as well)
    for (uint32_t t = 1 + lane; t + 62 <= T; t += 64)
      if ((int32_t)(t + 62) + (int32_t)cen[t + 62] - ((int32_t)t + (int32_t)cen[t]) <= 2 * half32 + 3) wide = true;
    if (__ballot(wide)) {
      if (lane == 0) {
        dist[c] = FLT_MAX;
        end_pos[c] = -2;
      }
      return;
    }
  }
  // this lane's current row and its constants
  uint32_t t = lane == 0 ? 64u : lane;
  float tv = 0.0f;
  int32_t lo = 1, hi = 0, gend = 0x7FFFFFFF;  // a row past T: never in the band, never left
  uint32_t d_last = 0;                        // the diagonal on which this row reaches the last query column (none: 0)
  auto load_row = [&]() {
    if (t <= T) {
      const int32_t centre = cen[t];
      tv = tg[t - 1];
      lo = max(centre - half32, 1);
      hi = min(centre + half32, Q32);
      gend = (int32_t)t + centre + half32;  // last diagonal with a cell of this row inside the band
      d_last = t + Q;
    } else {
      lo = 1;
      hi = 0;
      gend = 0x7FFFFFFF;
      d_last = 0;
    }
  };
  load_row();
  float v1 = FLT_MAX;      // this row's cell on the previous diagonal: (t, q - 1)
  float a_prev = FLT_MAX;  // the row above on the diagonal before that: (t - 1, q - 1)
  float best = FLT_MAX;
  uint32_t best_t = 0xFFFFFFFFu;
  for (uint32_t d = 2; d <= T + Q; ++d) {
    // (t - 1, q): the row above on the previous diagonal, from the neighbouring lane (lane 0 <- 63)
    const float b_in =
        __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0x13C /* wave_ror:1 */, 0xF, 0xF, false));
    const int32_t q = (int32_t)d - (int32_t)t;
    int32_t qi;  // q - 1 clamped to the query (any valid address outside the band): one v_med3_i32
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(qi) : "v"(q - 1), "v"(Q32 - 1));
    const float qv = qs[qi];
    // boundary constants of sigmap.cc:1263-1271
    const float a = q == 1 ? 0.0f : (t == 1 ? FLT_MAX : a_prev);
    const float b = t == 1 ? FLT_MAX : b_in;
    const float cc = q == 1 ? 0.0f : v1;
    // std::min({a, b, c}) of non-negative values without NaN: one v_min3_f32 (the compare / select chain was four)
    const float m = __builtin_fminf(__builtin_fminf(a, b), cc);
    const float v = (q >= lo && q <= hi) ? fabsf(tv - qv) + m : FLT_MAX;
    if (d == d_last) {  // q == Q in a row of the window
      if (v < best || (v == best && t < best_t)) {
        best = v;
        best_t = t;
      }
    }
    a_prev = b_in;
    v1 = v;
    // The row is left one diagonal after its last in-band cell, so that the row below has read that
    // cell; the new row (64 further) has not begun, so FLT_MAX is its cell as well as the old row's.
    if (gend < (int32_t)d) {
      t += 64;
      load_row();
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_down(best, off, 64);
    const uint32_t ot = __shfl_down(best_t, off, 64);
    if (ob < best || (ob == best && ot < best_t)) {
      best = ob;
      best_t = ot;
    }
  }
  if (lane == 0) {
    if (best < FLT_MAX) {
      dist[c] = best;
      end_pos[c] = t0 + (int64_t)best_t;
    } else {
      dist[c] = FLT_MAX;
      end_pos[c] = -1;
    }
  }
}
}  // namespace

hipError_t InitSdtwAttributes() {  // per device (smhip_init)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(SdtwKernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(3 * (size_t)(kSdtwMaxT + 1) * sizeof(float)));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(DtwPathKernel<uint64_t>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * (size_t)(kSdtwMaxT + 1) * sizeof(float)));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(SdtwChainKernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t LaunchSdtw(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                      const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, uint32_t n, float *dist,
                      int64_t *end, uint32_t max_t) {
  if (n == 0) return hipSuccess;
  if (max_t > (uint32_t)kSdtwMaxT) max_t = kSdtwMaxT;  // longer problems report FLT_MAX / -1
  const size_t lds = 3 * (size_t)(max_t + 1) * sizeof(float);
  hipLaunchKernelGGL(SdtwKernel, dim3(n), dim3(64), lds, ctx->stream, target, t_off, query, q_off, lo, hi,
                     dist, end);
  return hipGetLastError();
}
hipError_t LaunchDtwPath(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                         const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, const uint64_t *cell_off,
                         uint8_t *codes, uint32_t n, float *dist, int64_t *end, uint8_t *path, const uint64_t *path_off,
                         uint32_t *path_len, int64_t *start_t, uint32_t max_t) {
  if (n == 0) return hipSuccess;
  if (max_t > (uint32_t)kSdtwMaxT) max_t = kSdtwMaxT;  // longer problems report FLT_MAX / -1
  const size_t lds = 3 * (size_t)(max_t + 1) * sizeof(float);
  hipLaunchKernelGGL(DtwPathKernel<uint64_t>, dim3(n), dim3(64), lds, ctx->stream, target, t_off, nullptr, query, q_off,
                     nullptr, lo, hi, cell_off, codes, dist, end, path, path_off, path_len, start_t);
  return hipGetLastError();
}

// ---- the same for chain 0 of every chunk of the last smhip_map_run: window and band exactly as
// smhip_refine_run defines them (include/sigmap_hip.h).  Pass 1 counts the rows of every chunk's
// window, pass 2 writes the per-row column intervals and their widths.
__device__ __forceinline__ bool ChainWindow(const smhip_chunk_result &res, const uint32_t *qstart, uint32_t c,
                                            const uint64_t *ref_off, uint32_t n_refs, uint32_t pad, int dim,
                                            int64_t &t0, int64_t &start, int64_t &den, int64_t &qa, int64_t &qb,
                                            uint32_t &T, uint64_t &sig_off) {
  T = 0;
  if (!(res.n_chains >= 1 && res.chains[0].ref < n_refs) || res.n_features == 0) return false;
  const smhip_chain ch = res.chains[0];
  const uint32_t strand = ch.direction == 1 ? 0u : 1u;
  const uint64_t so = ref_off[2 * ch.ref + strand], se = ref_off[2 * ch.ref + strand + 1];
  const int64_t len = (int64_t)(se - so);
  start = ch.start;
  const int64_t endp = ch.end;
  t0 = start - (int64_t)pad < 0 ? 0 : start - (int64_t)pad;
  int64_t t1 = endp + dim + (int64_t)pad;
  if (t1 > len) t1 = len;
  const uint32_t qs = qstart ? qstart[c] : 0;
  qa = (int64_t)res.c0_last_qpos - qs;
  qb = (int64_t)res.c0_first_qpos - qs;
  if (qa < 0) qa = 0;
  if (qb < 0) qb = 0;
  den = endp - start > 0 ? endp - start : 1;
  if (t1 <= t0 || t1 - t0 > (int64_t)kSdtwMaxT) return false;
  T = (uint32_t)(t1 - t0);
  sig_off = so + (uint64_t)t0;
  return true;
}

__global__ void ChainWindowRowsKernel(const smhip_chunk_result *__restrict__ results, const uint32_t *__restrict__ qstart,
                                      uint32_t n_chunks, const uint64_t *__restrict__ ref_off, uint32_t n_refs,
                                      uint32_t pad, int dim, uint32_t *__restrict__ rows, uint32_t *__restrict__ path_cap) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  int64_t t0, start, den, qa, qb;
  uint32_t T;
  uint64_t so;
  const bool ok = ChainWindow(results[c], qstart, c, ref_off, n_refs, pad, dim, t0, start, den, qa, qb, T, so);
  rows[c] = ok ? T : 0u;
  path_cap[c] = ok ? T + results[c].n_features : 0u;
}

__global__ __launch_bounds__(64) void ChainWindowFillKernel(const smhip_chunk_result *__restrict__ results,
                                                            const uint32_t *__restrict__ qstart, uint32_t feat_stride,
                                                            const uint64_t *__restrict__ ref_off, uint32_t n_refs,
                                                            uint32_t band, uint32_t pad, int dim,
                                                            const uint64_t *__restrict__ t_off, uint64_t *__restrict__ tgt_off,
                                                            uint64_t *__restrict__ q_off, uint32_t *__restrict__ q_len,
                                                            uint32_t *__restrict__ lo, uint32_t *__restrict__ hi,
                                                            uint32_t *__restrict__ width) {
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  int64_t t0, start, den, qa, qb;
  uint32_t T;
  uint64_t so = 0;
  const bool ok = ChainWindow(results[c], qstart, c, ref_off, n_refs, pad, dim, t0, start, den, qa, qb, T, so);
  const uint32_t Q = results[c].n_features;
  if (lane == 0) {
    tgt_off[c] = so;
    q_off[c] = (uint64_t)c * feat_stride;
    q_len[c] = ok ? Q : 0u;
  }
  if (!ok) return;
  const int64_t half = band / 2;
  const uint64_t rb = t_off[c];
  for (uint32_t t = 1 + lane; t <= T; t += 64) {  // row t (1-based): absolute target index ta
    const int64_t ta = t0 + (int64_t)t - 1;
    const int64_t cen = 1 + qa + FloorDiv((ta - start) * (qb - qa), den);
    int64_t a = cen - half, b = cen + half;
    if (a < 1) a = 1;
    if (b > (int64_t)Q) b = Q;
    lo[rb + t - 1] = (uint32_t)a;
    hi[rb + t - 1] = b < a ? (uint32_t)(a - 1) : (uint32_t)b;  // (empty row: hi = lo - 1)
    width[rb + t - 1] = b < a ? 0u : (uint32_t)(b - a + 1);
  }
}

hipError_t LaunchChainWindowRows(smhip_ctx *ctx, const smhip_chunk_result *results, const uint32_t *qstart,
                                 uint32_t n_chunks, const uint64_t *ref_off, uint32_t n_refs, uint32_t pad, uint32_t *rows,
                                 uint32_t *path_cap) {
  hipLaunchKernelGGL(ChainWindowRowsKernel, dim3((n_chunks + 255) / 256), dim3(256), 0, ctx->stream, results, qstart,
                     n_chunks, ref_off, n_refs, pad, kMaxDim, rows, path_cap);
  return hipGetLastError();
}

hipError_t LaunchChainWindowFill(smhip_ctx *ctx, const smhip_chunk_result *results, const uint32_t *qstart,
                                 uint32_t feat_stride, uint32_t n_chunks, const uint64_t *ref_off, uint32_t n_refs,
                                 uint32_t band, uint32_t pad, const uint64_t *t_off, uint64_t *tgt_off, uint64_t *q_off,
                                 uint32_t *q_len, uint32_t *lo, uint32_t *hi, uint32_t *width) {
  hipLaunchKernelGGL(ChainWindowFillKernel, dim3(n_chunks), dim3(64), 0, ctx->stream, results, qstart, feat_stride,
                     ref_off, n_refs, band, pad, kMaxDim, t_off, tgt_off, q_off, q_len, lo, hi, width);
  return hipGetLastError();
}

hipError_t LaunchDtwPathChains(smhip_ctx *ctx, const float *ref_sig, const uint64_t *t_off, const uint64_t *tgt_off,
                               const float *feat, const uint64_t *q_off, const uint32_t *q_len, const uint32_t *lo,
                               const uint32_t *hi, const uint64_t *cell_off, uint8_t *codes, uint32_t n, float *dist,
                               int64_t *end, uint8_t *path, const uint64_t *path_off, uint32_t *path_len,
                               int64_t *start_t) {
  if (n == 0) return hipSuccess;
  const size_t lds = 3 * (size_t)(kSdtwMaxT + 1) * sizeof(float);
  hipLaunchKernelGGL(DtwPathKernel<uint64_t>, dim3(n), dim3(64), lds, ctx->stream, ref_sig, t_off, tgt_off, feat, q_off,
                     q_len, lo, hi, cell_off, codes, dist, end, path, path_off, path_len, start_t);
  return hipGetLastError();
}

hipError_t LaunchSdtwChains(smhip_ctx *ctx, const smhip_chunk_result *results, const float *feat,
                            uint32_t feat_stride, const uint32_t *qstart, uint32_t n_chunks, const float *ref_sig,
                            const uint64_t *ref_off, uint32_t n_refs, uint32_t band, uint32_t pad, float *dist,
                            int64_t *end_pos) {
  if (n_chunks == 0) return hipSuccess;
  // rows of a window = chain span + dim + 2*pad (a 4000-sample chunk spans ~600 events).  The sweep
  // is a chain of LDS round trips per diagonal, so it lives on occupancy: with the diagonals kept
  // in rings, pass 1 needs (3 * ring + 1023) * 4 B = 6.6 KB at band 64 (24 single-wave workgroups
  // per CU); windows longer than 1021 rows (end_pos = -2) are redone by pass 2 with room for 8192.
  // (Whole diagonals in LDS, 16 KB per workgroup: 3.0 ms; 9.7 KB: 2.1 ms.)
  uint32_t ring = 64;
  while (ring < band + 16) ring <<= 1;
  const uint32_t fast_rows = 1021;
  const size_t lds_fast = (3 * (size_t)ring + fast_rows + 2) * sizeof(float);
  const size_t lds_full = (3 * (size_t)ring + kSdtwMaxT + 2) * sizeof(float);
  if (lds_full > 160 * 1024) return hipErrorInvalidValue;  // band beyond what the LDS rings hold
  if (ctx->knobs.sdtw_ring) {  // (experiments: skip the register kernel)
    hipLaunchKernelGGL(SdtwChainKernel, dim3(n_chunks), dim3(64), lds_fast, ctx->stream, results, feat, feat_stride,
                       qstart, ref_sig, ref_off, n_refs, band, pad, kMaxDim, dist, end_pos, fast_rows, 0, ring);
  } else {
    // pass 1: register sweep (narrow, short problems); what it defers (end_pos = -2) is swept by the
    // ring kernel, first with room for 1021 rows (high occupancy), then, below, with room for 8192
    const size_t lds_sys = (size_t)kSysQ * sizeof(float) + ((size_t)fast_rows + 2) * sizeof(int16_t);
    hipLaunchKernelGGL(SdtwChainSystolicKernel, dim3(n_chunks), dim3(64), lds_sys, ctx->stream, results, feat,
                       feat_stride, qstart, ref_sig, ref_off, n_refs, band, pad, kMaxDim, dist, end_pos, fast_rows);
    hipLaunchKernelGGL(SdtwChainKernel, dim3(n_chunks), dim3(64), lds_fast, ctx->stream, results, feat, feat_stride,
                       qstart, ref_sig, ref_off, n_refs, band, pad, kMaxDim, dist, end_pos, fast_rows, 1, ring);
  }
  hipLaunchKernelGGL(SdtwChainKernel, dim3(n_chunks), dim3(64), lds_full, ctx->stream, results, feat, feat_stride,
                     qstart, ref_sig, ref_off, n_refs, band, pad, kMaxDim, dist, end_pos, (uint32_t)kSdtwMaxT, 1, ring);
  return hipGetLastError();
}
}  // namespace smhip
