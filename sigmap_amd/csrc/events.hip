// events.hip -- per-chunk event detection, z-score and compression on gfx950.
//
// Replaces Sigmap::GenerateEvents (sigmap.cc:1048-1083) = DetectEvents (event.h:226-267)
// + GenerateZscoreNormalizedSignal (sigmap.cc:1131-1155) + the 0.1 compression
// (sigmap.cc:1073-1079).  One 256-thread workgroup per chunk; the whole chunk lives in LDS
// (4 float arrays of kMaxSamples + the peak list, ~74 KB => 2 workgroups per CU).
//
// Bit-exactness rules (SURVEY.md section 7 "hard parts"): the two prefix sums are
// sequential f32 chains (event.h:58-68), so they run in index order on two lanes of one
// wave (lane 0: sum x, lane 1: sum x*x); the t-statistic mixes f32 with f64 fabs/sqrt/div
// (event.h:109); the z-score sums are sequential f64.  Everything else is parallel.
#include <float.h>

#include "common.h"

namespace smhip {
namespace {
constexpr int kLdsStride = kMaxSamples + 4;  // floats per LDS array (16-byte multiple)
constexpr int kEvThreads = 256;

// event.h:89-110 for one i in [w, n-w]
__device__ __forceinline__ float TStatAt(const float *ps, const float *pss, uint32_t i, uint32_t w) {
  const float wf = (float)w;
  float sum1 = ps[i];
  float sumsq1 = pss[i];
  if (i > w) {
    sum1 -= ps[i - w];
    sumsq1 -= pss[i - w];
  }
  const float sum2 = ps[i + w] - ps[i];
  const float sumsq2 = pss[i + w] - pss[i];
  const float mean1 = sum1 / wf;
  const float mean2 = sum2 / wf;
  float combined_var = sumsq1 / wf - mean1 * mean1 + sumsq2 / wf - mean2 * mean2;
  combined_var = fmaxf(combined_var, FLT_MIN);
  const float delta_mean = mean2 - mean1;
  return (float)(fabs((double)delta_mean) / sqrt((double)(combined_var / wf)));
}

struct Detector {  // event.h:45-56
  float threshold;
  uint32_t window_length;
  uint64_t masked_to;
  int peak_pos;
  float peak_value;
  bool valid_peak;
};

__global__ __launch_bounds__(kEvThreads) void EventsKernel(
    const float *__restrict__ samples, const uint32_t *__restrict__ off, uint32_t feat_stride,
    float *__restrict__ feat, float *__restrict__ stdv, smhip_chunk_result *__restrict__ results,
    uint32_t *__restrict__ nq, uint32_t w1, uint32_t w2, float thr1, float thr2, float peak_height,
    double compress_delta, uint32_t min_num_events, int dim, int step) {
  extern __shared__ __align__(16) float lds[];
  float *ps = lds;                    // prefix sums, later z-scores
  float *pss = ps + kLdsStride;       // prefix sums of squares, later kept features
  float *A = pss + kLdsStride;        // x, then t-stat (w1), then event means, then kept stdv
  float *B = A + kLdsStride;          // x*x, then t-stat (w2), then event stdv
  uint16_t *peaks = reinterpret_cast<uint16_t *>(B + kLdsStride);
  __shared__ uint32_t s_npeaks, s_nfeat;
  __shared__ double s_mean, s_sd;

  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint32_t begin = off[c];
  uint32_t n = off[c + 1] - begin;
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;  // caller validated; defensive clamp
  const float *x = samples + begin;

  if (tid == 0) {  // the record is zeroed here; later kernels accumulate into it
    smhip_chunk_result z;
    memset(&z, 0, sizeof z);
    results[c] = z;
  }
  for (uint32_t i = tid; i < n; i += kEvThreads) {
    const float v = x[i];
    A[i] = v;
    B[i] = v * v;
  }
  __syncthreads();

  // event.h:58-68: ps[i+1] = ps[i] + x[i]; pss[i+1] = pss[i] + x[i]*x[i], in index order.
  if (tid < 2) {
    const float *src = tid == 0 ? A : B;
    float *dst = tid == 0 ? ps : pss;
    float acc = 0.0f;
    dst[0] = 0.0f;
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(src + i);
      const float a0 = acc + v.x;
      const float a1 = a0 + v.y;
      const float a2 = a1 + v.z;
      const float a3 = a2 + v.w;
      dst[i + 1] = a0;
      dst[i + 2] = a1;
      dst[i + 3] = a2;
      dst[i + 4] = a3;
      acc = a3;
    }
    for (; i < n; ++i) {
      acc = acc + src[i];
      dst[i + 1] = acc;
    }
  }
  __syncthreads();

  // event.h:70-115 for both windows; entries [0,w) and (n-w, n] are zero.
  for (uint32_t i = tid; i <= n; i += kEvThreads) {
    float t1 = 0.0f, t2 = 0.0f;
    if (!(n < 2 * w1 || w1 < 2) && i >= w1 && i <= n - w1) t1 = TStatAt(ps, pss, i, w1);
    if (!(n < 2 * w2 || w2 < 2) && i >= w2 && i <= n - w2) t2 = TStatAt(ps, pss, i, w2);
    A[i] = t1;
    B[i] = t2;
  }
  __syncthreads();

  // event.h:117-182: sequential two-detector state machine (short detector first).
  if (tid == 0) {
    Detector d[2] = {{thr1, w1, 0, -1, FLT_MAX, false}, {thr2, w2, 0, -1, FLT_MAX, false}};
    uint32_t np = 0;
    for (uint32_t i = 0; i < n; ++i) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        Detector &det = d[k];
        if (det.masked_to >= i) continue;
        const float cur = k == 0 ? A[i] : B[i];
        if (det.peak_pos == -1) {
          if (cur < det.peak_value) {
            det.peak_value = cur;
          } else if (cur - det.peak_value > peak_height) {
            det.peak_value = cur;
            det.peak_pos = (int)i;
          }
        } else {
          if (cur > det.peak_value) {
            det.peak_value = cur;
            det.peak_pos = (int)i;
          }
          if (k == 0) {
            if (det.peak_value > det.threshold) {
              d[1].masked_to = (uint64_t)det.peak_pos + det.window_length;
              d[1].peak_pos = -1;
              d[1].peak_value = FLT_MAX;
              d[1].valid_peak = false;
            }
          }
          if (det.peak_value - cur > peak_height && det.peak_value > det.threshold) det.valid_peak = true;
          if (det.valid_peak && ((uint64_t)i - (uint64_t)det.peak_pos) > det.window_length / 2) {
            peaks[np++] = (uint16_t)det.peak_pos;
            det.peak_pos = -1;
            det.peak_value = cur;
            det.valid_peak = false;
          }
        }
      }
    }
    s_npeaks = np;
  }
  __syncthreads();

  // event.h:200-224: num_events = 1 + #{i>=1 : 0 < peaks[i] < n}.  Every emitted peak is a
  // sample index in [1, n-1] (index 0 is never examined: masked_to >= 0), so that count is
  // n_peaks.  With fewer than 2 peaks the reference reads out of bounds: the chunk is
  // flagged and yields no features.
  const uint32_t E = s_npeaks >= 2 ? s_npeaks : 0;
  for (uint32_t e = tid; e < E; e += kEvThreads) {
    const uint64_t start = e > 0 ? peaks[e - 1] : 0;
    const uint64_t end = e < E - 1 ? peaks[e] : n;
    const uint64_t length = end - start;  // size_t arithmetic, wraps like the reference
    const float mean = (ps[end] - ps[start]) / (float)length;
    const float deltasqr = pss[end] - pss[start];
    const float var = deltasqr / (float)length - mean * mean;
    A[e] = mean;
    B[e] = sqrtf(fmaxf(var, 0.0f));
  }
  __syncthreads();

  // sigmap.cc:1135-1146: sequential double sums
  if (tid == 0 && E > 0) {
    double mean = 0;
    for (uint32_t i = 0; i < E; ++i) mean += (double)A[i];
    mean /= (double)E;
    double SD = 0;
    for (uint32_t i = 0; i < E; ++i) SD += ((double)A[i] - mean) * ((double)A[i] - mean);
    SD /= (double)(uint64_t)(E - 1);
    SD = sqrt(SD);
    s_mean = mean;
    s_sd = SD;
  }
  __syncthreads();
  {
    const double mean = s_mean, SD = s_sd;
    for (uint32_t i = tid; i < E; i += kEvThreads) ps[i] = (float)(((double)A[i] - mean) / SD);
  }
  __syncthreads();

  // sigmap.cc:1073-1079: keep i iff i == 0 or |z[i] - last kept| > 0.1 (float abs vs double 0.1)
  if (tid == 0) {
    uint32_t k = 0;
    float last = 0.0f;
    for (uint32_t i = 0; i < E; ++i) {
      const float z = ps[i];
      if (i == 0 || (double)fabsf(z - last) > compress_delta) {
        pss[k] = z;
        A[k] = B[i];  // k <= i, and A[i] (event mean) is dead once z is computed
        last = z;
        ++k;
      }
    }
    s_nfeat = k;
  }
  __syncthreads();
  const uint32_t F = s_nfeat;
  for (uint32_t i = tid; i < F && i < feat_stride; i += kEvThreads) {
    feat[(size_t)c * feat_stride + i] = pss[i];
    if (stdv) stdv[(size_t)c * feat_stride + i] = A[i];
  }
  if (tid == 0) {
    uint32_t Q = 0;
    // sigmap.cc:660 (> 50 features) and spatial_index.cc:326-356,403-408: seeds at
    // step, 2*step, ... <= nwin-1 (the "reliable seed" sort is a no-op)
    if (F > min_num_events && F >= (uint32_t)dim) {
      const uint32_t nwin = F - (uint32_t)dim + 1;
      Q = (nwin - 1) / (uint32_t)step;
    }
    results[c].status = s_npeaks < 2 ? SMHIP_ST_FEW_PEAKS : 0;
    results[c].n_events = E;
    results[c].n_features = F;
    results[c].n_queries = Q;
    nq[c] = Q;
  }
}

// Exclusive scan of n uint32 values by one workgroup (n is a chunk count: small).
template <typename OutT>
__global__ __launch_bounds__(1024) void ScanKernel(const uint32_t *__restrict__ in, uint32_t n,
                                                   OutT *__restrict__ out) {
  __shared__ uint64_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t b = tid * per, e = min(n, b + per);
  uint64_t s = 0;
  for (uint32_t i = b; i < e; ++i) s += in[i];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint64_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint64_t run = tid ? part[tid - 1] : 0;
  for (uint32_t i = b; i < e; ++i) {
    out[i] = (OutT)run;
    run += in[i];
  }
  if (tid == 1023) out[n] = (OutT)part[1023];
}
}  // namespace

hipError_t LaunchEvents(smhip_ctx *ctx, const smhip_params &p, const float *samples, const uint32_t *off,
                        uint32_t n_chunks, uint32_t feat_stride, float *feat, float *stdv,
                        smhip_chunk_result *results, uint32_t *nq) {
  if (n_chunks == 0) return hipSuccess;
  const size_t lds = 4 * (size_t)kLdsStride * sizeof(float) + (size_t)kMaxSamples * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(EventsKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(EventsKernel, dim3(n_chunks), dim3(kEvThreads), lds, ctx->stream, samples, off,
                     feat_stride, feat, stdv, results, nq, p.window_length1, p.window_length2,
                     p.threshold1, p.threshold2, p.peak_height, p.compress_delta, p.min_num_events,
                     kMaxDim, p.step_size);
  return hipGetLastError();
}

hipError_t LaunchScanU32(smhip_ctx *ctx, const uint32_t *in, uint32_t n, uint32_t *out_excl) {
  hipLaunchKernelGGL(ScanKernel<uint32_t>, dim3(1), dim3(1024), 0, ctx->stream, in, n, out_excl);
  return hipGetLastError();
}

hipError_t LaunchScanU32ToU64(smhip_ctx *ctx, const uint32_t *in, uint32_t n, uint64_t *out_excl) {
  hipLaunchKernelGGL(ScanKernel<uint64_t>, dim3(1), dim3(1024), 0, ctx->stream, in, n, out_excl);
  return hipGetLastError();
}
}  // namespace smhip
