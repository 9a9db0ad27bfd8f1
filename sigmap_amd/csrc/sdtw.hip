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
}  // namespace

hipError_t LaunchSdtw(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                      const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, uint32_t n, float *dist,
                      int64_t *end, uint32_t max_t) {
  if (n == 0) return hipSuccess;
  if (max_t > (uint32_t)kSdtwMaxT) max_t = kSdtwMaxT;  // longer problems report FLT_MAX / -1
  const size_t lds = 3 * (size_t)(max_t + 1) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(SdtwKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(3 * (size_t)(kSdtwMaxT + 1) * sizeof(float)));
    if (e != hipSuccess) return e;
    attr = true;
  }
  hipLaunchKernelGGL(SdtwKernel, dim3(n), dim3(64), lds, ctx->stream, target, t_off, query, q_off, lo, hi,
                     dist, end);
  return hipGetLastError();
}
}  // namespace smhip
