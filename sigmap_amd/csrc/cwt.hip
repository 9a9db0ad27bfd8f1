// cwt.hip -- the CWT feature path of the reference, per chunk:
//   D1 MAD normalisation   Sigmap::GenerateMADNormalizedSignal, sigmap.cc:1100-1129
//   D2 one-scale DoG CWT   Sigmap::GenerateCWTSignal, sigmap.cc:1157-1174 -> cwt.cc:1633-1841
//   D3 z-score + peaks     GenerateZscoreNormalizedSignal sigmap.cc:1131-1155, GeneratePeaks :1176-1196
// (GenerateFeatureSignalUsingCWT, sigmap.cc:1085-1098).  Dormant in the reference's command line
// (CWTAlign, sigmap.cc:932, is disabled at :1499-1504); built so that the path is there when a
// caller turns it on.
//
// D1 and D3 are exact: the "median" is the n/2-th order statistic (std::nth_element), found by a
// full bitonic sort of the chunk in LDS; the z-score sums are sequential f64 and the peak picker is
// a sequential scan, one lane per chunk.  D2 is the same LINEAR MAP as the reference's
// IFFT(D .* FFT(x - mean, npad) / npad) but evaluated as the convolution with its impulse response
// h[j] = (1/npad) sum_k D[k] cos(2 pi k j / npad) (tabulated on the host in double, |j| <= 9 s,
// where it is < 1e-17 of its peak for scales >= 4) with f64 accumulation: the reference computes
// it with an f32 mixed-radix FFT, so agreement is to that FFT's rounding (measured 2e-7 of the
// signal's maximum against the real cwt.cc), not bit for bit, and a peak that hinges on the last
// bit can differ.  Direct convolution instead of an FFT: 153 taps x 4000 samples is 0.6 MFLOP per
// chunk, the whole batch a fraction of a millisecond of f64 FMAs out of LDS.
#include <float.h>

#include "common.h"

namespace smhip {
namespace {
constexpr int kCwtThreads = 256;
constexpr uint32_t kCwtPad = 4096;  // chunk capacity (kMaxSamples) rounded to a power of two

__device__ __forceinline__ void BitonicLds(float *v, uint32_t tid) {  // ascending, kCwtPad entries
  for (uint32_t k = 2; k <= kCwtPad; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < kCwtPad / 2; t += kCwtThreads) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const float a = v[i], b = v[l];
        if ((a > b) == ((i & k) == 0)) {
          v[i] = b;
          v[l] = a;
        }
      }
      __syncthreads();
    }
  }
}

// D1 + D2: one workgroup per chunk.  cwt_out row stride kCwtPad.
__global__ __launch_bounds__(kCwtThreads) void CwtKernel(const float *__restrict__ samples,
                                                         const uint64_t *__restrict__ cbegin,
                                                         const uint32_t *__restrict__ clen, const double *__restrict__ h,
                                                         uint32_t L, float *__restrict__ cwt_out) {
  __shared__ float s_sort[kCwtPad];
  __shared__ float s_x[kCwtPad];
  __shared__ float s_stat[2];
  extern __shared__ double s_h[];  // [L + 1]
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  uint32_t n = clen[c];
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  const float *x = samples + cbegin[c];
  float *out = cwt_out + (size_t)c * kCwtPad;
  if (n < 3) return;
  for (uint32_t i = tid; i < kCwtPad; i += kCwtThreads) {
    const float v = i < n ? x[i] : FLT_MAX;
    s_x[i] = v;
    s_sort[i] = v;
  }
  for (uint32_t j = tid; j <= L; j += kCwtThreads) s_h[j] = h[j];
  __syncthreads();
  BitonicLds(s_sort, tid);
  const float median = s_sort[n / 2];
  __syncthreads();
  for (uint32_t i = tid; i < kCwtPad; i += kCwtThreads) s_sort[i] = i < n ? fabsf(s_x[i] - median) : FLT_MAX;
  __syncthreads();
  BitonicLds(s_sort, tid);
  const float mad = s_sort[n / 2];
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kCwtThreads) s_x[i] = (s_x[i] - median) / mad;
  __syncthreads();
  if (tid == 0) {  // cwt_type is float: the mean is a sequential f32 sum (cwt.cc:1786-1790)
    float m = 0.0f;
    for (uint32_t i = 0; i < n; ++i) m += s_x[i];
    s_stat[0] = m / (float)n;
  }
  __syncthreads();
  const float ymean = s_stat[0];
  for (uint32_t i = tid; i < n; i += kCwtThreads) {
    double acc = 0.0;
    const uint32_t lo = i > L ? i - L : 0, hi = min(n - 1, i + L);
    for (uint32_t m = lo; m <= hi; ++m) acc += (double)(s_x[m] - ymean) * s_h[m > i ? m - i : i - m];
    out[i] = (float)acc;
  }
}

// D3: one lane per chunk: z-score (sequential f64 sums, sigmap.cc:1131-1155: returns SD), then the
// alternating peak / valley picker with hysteresis SD / 4 (sigmap.cc:1176-1196).
__global__ __launch_bounds__(64) void CwtPeaksKernel(const uint32_t *__restrict__ clen, uint32_t n_chunks,
                                                     const float *__restrict__ cwt, uint32_t feat_stride,
                                                     float *__restrict__ feat, uint32_t *__restrict__ pos,
                                                     uint32_t *__restrict__ n_feat) {
  const uint32_t c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_chunks) return;
  uint32_t n = clen[c];
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  if (n < 3) {
    n_feat[c] = 0;
    return;
  }
  const float *x = cwt + (size_t)c * kCwtPad;
  double mean = 0;
  for (uint32_t i = 0; i < n; ++i) mean += x[i];
  mean /= n;
  double SD = 0;
  for (uint32_t i = 0; i < n; ++i) SD += (x[i] - mean) * (x[i] - mean);
  SD /= (n - 1);
  SD = sqrt(SD);
  const float selective = (float)SD / 4;
  float *f = feat + (size_t)c * feat_stride;
  uint32_t *p = pos + (size_t)c * feat_stride;
  uint32_t np = 0;
  float zm = (float)((x[0] - mean) / SD), z0 = (float)((x[1] - mean) / SD);
  float previous_valley = zm, previous_peak = zm;
  for (uint32_t i = 1; i + 1 < n; ++i) {
    const float zp = (float)((x[i + 1] - mean) / SD);
    if (z0 > zm && z0 >= zp && z0 >= previous_valley + selective) {
      if (np < feat_stride) {
        f[np] = z0;
        p[np] = i;
      }
      ++np;
      previous_peak = z0;
    } else if (z0 < zm && z0 <= zp && z0 <= previous_peak - selective) {
      if (np < feat_stride) {
        f[np] = z0;
        p[np] = i;
      }
      ++np;
      previous_valley = z0;
    }
    zm = z0;
    z0 = zp;
  }
  n_feat[c] = np;
}
}  // namespace

hipError_t LaunchCwtFeatures(smhip_ctx *ctx, const float *samples, const uint64_t *cbegin, const uint32_t *clen,
                             uint32_t n_chunks,
                             const double *h, uint32_t L, float *cwt_rows, uint32_t feat_stride, float *feat,
                             uint32_t *pos, uint32_t *n_feat) {
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(CwtKernel, dim3(n_chunks), dim3(kCwtThreads), (L + 1) * sizeof(double), ctx->stream, samples,
                     cbegin, clen, h, L, cwt_rows);
  hipLaunchKernelGGL(CwtPeaksKernel, dim3((n_chunks + 63) / 64), dim3(64), 0, ctx->stream, clen, n_chunks, cwt_rows,
                     feat_stride, feat, pos, n_feat);
  return hipGetLastError();
}
}  // namespace smhip
