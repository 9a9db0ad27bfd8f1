// events.hip -- per-chunk event detection, z-score and compression on gfx950.
//
// Replaces Sigmap::GenerateEvents (sigmap.cc:1048-1083) = DetectEvents (event.h:226-267)
// + GenerateZscoreNormalizedSignal (sigmap.cc:1131-1155) + the 0.1 compression
// (sigmap.cc:1073-1079).
//
// Bit-exactness rules (SURVEY.md section 7 "hard parts"): the two prefix sums are sequential
// f32 chains (event.h:58-68), the peak picker is a sequential state machine (event.h:117-182),
// the z-score sums are sequential f64 (sigmap.cc:1135-1146) and the compression carries the
// last kept value; the t-statistic mixes f32 with f64 fabs/sqrt/div (event.h:109).
//
// Mapping: the order-dependent stages run ONE LANE PER CHUNK (64 chunks per wavefront; blocks of
// values fetched ahead of the dependent chain, four per load instruction) -- the peak picker one
// lane per SEGMENT of a chunk, speculatively, then checked --, the order-free stages (t-statistics,
// event means) run one thread per element.  Five small kernels over HBM-resident scratch rows:
//   E1 PrefixKernel (lane/chunk) -> E2 TStatKernel (thread/sample, LDS-staged prefix sums)
//   -> E3 PeakKernel (lane/segment, a branch-free state machine) -> E4 EventMeanKernel
//   (thread/event) -> E5 FeatureKernel (lane/chunk).
#include <float.h>


#include "common.h"

namespace smhip {
namespace {
constexpr int kRowStride = kMaxSamples + 4;  // floats per chunk row of the scratch arrays
constexpr uint32_t kNarrowLaneChunks = 4096;  // batches up to this size run the lane-per-chunk kernels in waves of 16 lanes
#ifndef SMHIP_PREFIX_PRE
#define SMHIP_PREFIX_PRE 64
#endif
#ifndef SMHIP_FEAT_BLK
#define SMHIP_FEAT_BLK 16
#endif
constexpr int kBlk = SMHIP_FEAT_BLK;         // values fetched ahead of the sequential chains
// ... of the peak picker: a lane walks its own segment of a t-statistic row, so its 4-byte reads pull a whole
// 128-byte line each; 32 samples = the line is consumed while it is still in the L1 (8: 357 us and 6.5 x the rows'
// bytes from the fabric per launch; 16: 310 us; 32: 297 us)
constexpr int kPeakBlk = 32;

// x / w for a window length w: five of these per t-statistic were half of the kernel's instructions (an IEEE f32
// division is ten).  q = x * RN(1 / w), corrected once with the exact remainder (two FMAs), is the correctly rounded
// quotient -- checked EXHAUSTIVELY for every w in 1 .. 64 and every float with 2^-100 <= |x| <= 2^100
// (tools/micro/div_by_window_check.c: the remainder must not underflow, hence the range); anything else divides.
constexpr uint32_t kFastDivMaxW = 64;
__device__ __forceinline__ float DivW(float x, float wf, float rw) {
  const float q = x * rw;
  return __fmaf_rn(__fmaf_rn(-q, wf, x), rw, q);
}
__device__ __forceinline__ bool DivWRange(float lo_abs, float hi_abs) { return lo_abs >= 0x1p-100f && hi_abs <= 0x1p100f; }

// event.h:89-110 for one i in [w, n-w]
__device__ __forceinline__ float TStatAt(const float *ps, const float *pss, uint32_t i, uint32_t w, float rw, bool fast_w) {
  const float wf = (float)w;
  float sum1 = ps[i];
  float sumsq1 = pss[i];
  if (i > w) {
    sum1 -= ps[i - w];
    sumsq1 -= pss[i - w];
  }
  const float sum2 = ps[i + w] - ps[i];
  const float sumsq2 = pss[i + w] - pss[i];
  const float a1 = fabsf(sum1), a2 = fabsf(sum2), a3 = fabsf(sumsq1), a4 = fabsf(sumsq2);
  float combined_var, delta_mean;
  if (fast_w && DivWRange(fminf(fminf(a1, a2), fminf(a3, a4)), fmaxf(fmaxf(a1, a2), fmaxf(a3, a4)))) {
    const float mean1 = DivW(sum1, wf, rw), mean2 = DivW(sum2, wf, rw);
    combined_var = DivW(sumsq1, wf, rw) - mean1 * mean1 + DivW(sumsq2, wf, rw) - mean2 * mean2;
    delta_mean = mean2 - mean1;
  } else {
    const float mean1 = sum1 / wf;
    const float mean2 = sum2 / wf;
    combined_var = sumsq1 / wf - mean1 * mean1 + sumsq2 / wf - mean2 * mean2;
    delta_mean = mean2 - mean1;
  }
  combined_var = fmaxf(combined_var, FLT_MIN);
  const float cvw = (fast_w && DivWRange(combined_var, combined_var)) ? DivW(combined_var, wf, rw) : combined_var / wf;
  return (float)(fabs((double)delta_mean) / sqrt((double)cvw));
}

struct Detector {  // event.h:45-56
  float threshold;
  uint32_t window_length;
  uint64_t masked_to;
  int peak_pos;
  float peak_value;
  bool valid_peak;
};

struct __attribute__((packed, aligned(4))) F4u {  // four floats at any 4-byte boundary
  float x, y, z, w;
};

// E1 -- event.h:58-68.  ps[i+1] = ps[i] + x[i]; pss[i+1] = pss[i] + x[i]*x[i]: a lane per chunk; FOUR samples per load
// and four sums per store (the sums BEFORE each sample: ps[i .. i + 3] is a 16-byte aligned piece of the row), so the
// addresser sees a quarter of the instructions of the sample-by-sample form (245 us; below).
__global__ __launch_bounds__(64) void PrefixKernel(const float *__restrict__ samples,
                                                   const uint64_t *__restrict__ cbegin,
                                                   const uint32_t *__restrict__ clen, uint32_t n_chunks,
                                                   float *__restrict__ ps_all, float *__restrict__ pss_all) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;  // (blockDim: 64, or 16 for a small batch -- LaunchEvents)
  if (c >= n_chunks) return;
  const uint64_t begin = cbegin[c];
  uint32_t n = clen[c];
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  const float *x = samples + begin;
  float *ps = ps_all + (size_t)c * kRowStride, *pss = pss_all + (size_t)c * kRowStride;
  float a = 0.0f, b = 0.0f;
  uint32_t i = 0;
  // blocks of kPre samples, the next block's loads in flight while the current one is added up
  constexpr int kPre = SMHIP_PREFIX_PRE, kV = kPre / 4;
  F4u cur[kV], nxt[kV];
  if (kPre <= n) {
#pragma unroll
    for (int k = 0; k < kV; ++k) cur[k] = *reinterpret_cast<const F4u *>(x + 4 * k);
  }
  for (; i + kPre <= n; i += kPre) {
    const bool more = i + 2 * kPre <= n;
    if (more) {
#pragma unroll
      for (int k = 0; k < kV; ++k) nxt[k] = *reinterpret_cast<const F4u *>(x + i + kPre + 4 * k);
    }
#pragma unroll
    for (int k = 0; k < kV; ++k) {
      const F4u v = cur[k];
      float4 oa, ob;
      oa.x = a; ob.x = b; a = a + v.x; b = b + v.x * v.x;
      oa.y = a; ob.y = b; a = a + v.y; b = b + v.y * v.y;
      oa.z = a; ob.z = b; a = a + v.z; b = b + v.z * v.z;
      oa.w = a; ob.w = b; a = a + v.w; b = b + v.w * v.w;
      *reinterpret_cast<float4 *>(ps + i + 4 * k) = oa;
      *reinterpret_cast<float4 *>(pss + i + 4 * k) = ob;
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < kV; ++k) cur[k] = nxt[k];
    }
  }
  for (; i < n; ++i) {
    const float v = x[i];
    ps[i] = a;
    pss[i] = b;
    a = a + v;
    b = b + v * v;
  }
  ps[n] = a;
  pss[n] = b;
}

// E2 -- event.h:70-115 for both windows, one thread per sample index (entries [0,w) and
// (n-w, n] are zero).  The chunk's prefix sums are staged in LDS.
__global__ __launch_bounds__(256) void TStatKernel(const uint32_t *__restrict__ clen,
                                                   const float *__restrict__ ps_all,
                                                   const float *__restrict__ pss_all, float *__restrict__ t1_all,
                                                   float *__restrict__ t2_all, uint32_t w1, uint32_t w2) {
  __shared__ float ps[kRowStride], pss[kRowStride];
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  uint32_t n = clen[c];
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  const float *gps = ps_all + (size_t)c * kRowStride, *gpss = pss_all + (size_t)c * kRowStride;
  for (uint32_t i = tid; i <= n; i += 256) {
    ps[i] = gps[i];
    pss[i] = gpss[i];
  }
  __syncthreads();
  float *t1 = t1_all + (size_t)c * kRowStride, *t2 = t2_all + (size_t)c * kRowStride;
  const bool fast1 = w1 >= 1 && w1 <= kFastDivMaxW, fast2 = w2 >= 1 && w2 <= kFastDivMaxW;
  const float rw1 = 1.0f / (float)(w1 ? w1 : 1u), rw2 = 1.0f / (float)(w2 ? w2 : 1u);
  for (uint32_t i = tid; i <= n; i += 256) {
    float a = 0.0f, b = 0.0f;
    if (!(n < 2 * w1 || w1 < 2) && i >= w1 && i <= n - w1) a = TStatAt(ps, pss, i, w1, rw1, fast1);
    if (!(n < 2 * w2 || w2 < 2) && i >= w2 && i <= n - w2) b = TStatAt(ps, pss, i, w2, rw2, fast2);
    t1[i] = a;
    t2[i] = b;
  }
}

// One step of one detector, event.h:128-180, WITHOUT BRANCHES: every condition of the state machine is evaluated and
// the new state is a select -- a wave runs 64 of these machines in step, and as nested branches every step cost it a
// dozen jumps around a few instructions each (the picker ran at a third of its instruction rate).  Returns whether
// the step emits a peak, and its position in `pos_out`.
__device__ __forceinline__ bool DetectorStep(Detector *d, int k, float cur, uint32_t i, float peak_height, uint32_t &pos_out) {
  Detector &det = d[k];
  const bool active = !(det.masked_to >= (uint64_t)i);
  const bool none = det.peak_pos == -1;
  // no peak candidate yet: follow the signal down; a rise of more than peak_height makes one
  const bool lower = cur < det.peak_value;
  const bool rise = !lower && cur - det.peak_value > peak_height;
  const float a_value = (lower || rise) ? cur : det.peak_value;
  const int a_pos = rise ? (int)i : -1;
  // a candidate: follow the signal up
  const bool higher = cur > det.peak_value;
  const float b_value = higher ? cur : det.peak_value;
  const int b_pos = higher ? (int)i : det.peak_pos;
  const bool over = b_value > det.threshold;
  if (k == 0) {  // a short-window peak above its threshold masks the long-window detector
    const bool m = active && !none && over;
    d[1].masked_to = m ? (uint64_t)(int64_t)b_pos + det.window_length : d[1].masked_to;
    d[1].peak_pos = m ? -1 : d[1].peak_pos;
    d[1].peak_value = m ? FLT_MAX : d[1].peak_value;
    d[1].valid_peak = m ? false : d[1].valid_peak;
  }
  const bool valid = det.valid_peak || (b_value - cur > peak_height && over);
  const bool emit_b = valid && ((uint64_t)i - (uint64_t)(int64_t)b_pos) > (uint64_t)(det.window_length / 2);
  const bool emit = active && !none && emit_b;
  pos_out = (uint32_t)b_pos;
  const float n_value = none ? a_value : (emit_b ? cur : b_value);
  const int n_pos = none ? a_pos : (emit_b ? -1 : b_pos);
  const bool n_valid = none ? det.valid_peak : (emit_b ? false : valid);
  det.peak_value = active ? n_value : det.peak_value;
  det.peak_pos = active ? n_pos : det.peak_pos;
  det.valid_peak = active ? n_valid : det.valid_peak;
  return emit;
}

// E3 -- event.h:117-182: sequential in i, short detector first.  A lane per chunk would leave the
// chip idle (10 000 chunks = 157 waves, each a chain of 8000 dependent detector steps), so a chunk
// is cut into kPeakSegs segments run by different lanes of one wave, speculatively: a segment
// starts kPeakWarm samples early from the initial state, which almost always reaches the state
// the sequential scan has at the segment boundary (both detectors re-synchronise at the next peak).
// The speculation is then CHECKED, exactly: a segment whose predecessor's end state differs from
// the state it started from is re-run from the true state, round after round until nothing
// changes (at most kPeakSegs - 1 rounds = the sequential order), so the peak list is the one the
// reference's loop produces whatever the signal looks like.
constexpr int kPeakSegs = 16;
constexpr uint32_t kPeakWarm = 96;

struct PeakState {  // the varying part of the two detectors (det 0 is never masked)
  int p0, p1;
  float v0, v1;
  uint32_t ok0, ok1;
  uint64_t m1;
};
__device__ __forceinline__ PeakState SaveState(const Detector *d) {
  return PeakState{d[0].peak_pos, d[1].peak_pos, d[0].peak_value, d[1].peak_value, d[0].valid_peak ? 1u : 0u,
                   d[1].valid_peak ? 1u : 0u, d[1].masked_to};
}
__device__ __forceinline__ void LoadState(Detector *d, const PeakState &s) {
  d[0].peak_pos = s.p0; d[1].peak_pos = s.p1;
  d[0].peak_value = s.v0; d[1].peak_value = s.v1;
  d[0].valid_peak = s.ok0 != 0; d[1].valid_peak = s.ok1 != 0;
  d[0].masked_to = 0; d[1].masked_to = s.m1;
}
__device__ __forceinline__ bool SameState(const PeakState &a, const PeakState &b) {
  return a.p0 == b.p0 && a.p1 == b.p1 && __float_as_uint(a.v0) == __float_as_uint(b.v0) &&
         __float_as_uint(a.v1) == __float_as_uint(b.v1) && a.ok0 == b.ok0 && a.ok1 == b.ok1 && a.m1 == b.m1;
}
__device__ __forceinline__ PeakState ShflUpState(const PeakState &s, int delta) {
  PeakState r;
  r.p0 = __shfl_up(s.p0, delta, 64); r.p1 = __shfl_up(s.p1, delta, 64);
  r.v0 = __shfl_up(s.v0, delta, 64); r.v1 = __shfl_up(s.v1, delta, 64);
  r.ok0 = __shfl_up(s.ok0, delta, 64); r.ok1 = __shfl_up(s.ok1, delta, 64);
  const uint32_t lo = __shfl_up((uint32_t)s.m1, delta, 64), hi = __shfl_up((uint32_t)(s.m1 >> 32), delta, 64);
  r.m1 = ((uint64_t)hi << 32) | lo;
  return r;
}

// steps [ib, ie) of both detectors; kEmit: peaks go to the lane's list -- its first kPeakLds - 1 entries are a column
// in LDS (entry k at out[k * 64]; written at EVERY step, with the position of whichever candidate there is, and kept
// only when np moves on; the last LDS entry takes the writes of a longer list), later ones to far[k] in memory (rare:
// a segment of 250 steps averages 30 peaks)
constexpr uint32_t kPeakLds = 64;
template <bool kEmit>
__device__ __forceinline__ void RunSteps(Detector *d, const float *__restrict__ t1, const float *__restrict__ t2,
                                         uint32_t ib, uint32_t ie, float peak_height, uint16_t *out, uint16_t *far,
                                         uint32_t &np) {
  auto push = [&](bool e, uint32_t pos) {
    out[min(np, kPeakLds - 1u) * 64u] = (uint16_t)pos;
    if (e && np >= kPeakLds - 1u) far[np] = (uint16_t)pos;
    np += e ? 1u : 0u;
  };
  auto both = [&](float a, float b, uint32_t i) {
    uint32_t pos;
    const bool e0 = DetectorStep(d, 0, a, i, peak_height, pos);
    if (kEmit) push(e0, pos);
    const bool e1 = DetectorStep(d, 1, b, i, peak_height, pos);
    if (kEmit) push(e1, pos);
  };
  uint32_t i = ib;
  for (; i + kPeakBlk <= ie; i += kPeakBlk) {
    float a[kPeakBlk], b[kPeakBlk];
#pragma unroll
    for (int k = 0; k < kPeakBlk; k += 4) {  // (four values per load instruction: a quarter of the addresser's work)
      const F4u va = *reinterpret_cast<const F4u *>(t1 + i + k), vb = *reinterpret_cast<const F4u *>(t2 + i + k);
      a[k] = va.x; a[k + 1] = va.y; a[k + 2] = va.z; a[k + 3] = va.w;
      b[k] = vb.x; b[k + 1] = vb.y; b[k + 2] = vb.z; b[k + 3] = vb.w;
    }
#pragma unroll
    for (int k = 0; k < kPeakBlk; ++k) both(a[k], b[k], i + k);
  }
  for (; i < ie; ++i) both(t1[i], t2[i], i);
}

__global__ __launch_bounds__(64) void PeakKernel(const uint32_t *__restrict__ clen, uint32_t n_chunks,
                                                 const float *__restrict__ t1_all, const float *__restrict__ t2_all,
                                                 uint16_t *__restrict__ peaks_all, uint32_t *__restrict__ n_peaks,
                                                 uint16_t *__restrict__ seg_peaks_all, uint32_t w1, uint32_t w2,
                                                 float thr1, float thr2, float peak_height) {
  constexpr int kChunksPerWave = 64 / kPeakSegs;
  const uint32_t lane = threadIdx.x, cl = lane % kChunksPerWave, seg = lane / kChunksPerWave;
  const uint32_t c = blockIdx.x * kChunksPerWave + cl;
  const bool live = c < n_chunks;
  uint32_t n = 0;
  if (live) {
    n = clen[c];
    if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  }
  const size_t row = (size_t)(live ? c : 0) * kRowStride;
  const float *t1 = t1_all + row, *t2 = t2_all + row;
  const uint32_t S = (n + kPeakSegs - 1) / kPeakSegs;
  const uint32_t beg = min(n, seg * S), end = min(n, beg + S);
  __shared__ uint16_t s_pk[kPeakLds * 64];  // the lanes' peak lists: entry k of lane l at [k * 64 + l]
  uint16_t *mine = s_pk + lane;
  uint16_t *far = seg_peaks_all + row + beg;  // ... beyond kPeakLds - 1 entries (a segment of S steps emits at most S peaks)
  Detector d[2] = {{thr1, w1, 0, -1, FLT_MAX, false}, {thr2, w2, 0, -1, FLT_MAX, false}};
  uint32_t np = 0;
  // speculative run: warm-up from the initial state, then the segment
  RunSteps<false>(d, t1, t2, beg > kPeakWarm ? beg - kPeakWarm : 0u, beg, peak_height, nullptr, nullptr, np);
  PeakState start = SaveState(d);
  RunSteps<true>(d, t1, t2, beg, end, peak_height, mine, far, np);
  PeakState fin = SaveState(d);
  // check against the predecessor segment's end state; re-run what started from a wrong state
  for (int round = 0; round < kPeakSegs; ++round) {
    const PeakState prev = ShflUpState(fin, kChunksPerWave);
    const bool redo = seg > 0 && !SameState(prev, start);
    if (!__ballot(redo)) break;
    if (redo) {
      start = prev;
      LoadState(d, prev);
      np = 0;
      RunSteps<true>(d, t1, t2, beg, end, peak_height, mine, far, np);
      fin = SaveState(d);
    }
  }
  // concatenate the segments' peaks in order
  uint32_t before = 0;  // peaks of the earlier segments of this chunk
  for (int k = 1; k < kPeakSegs; ++k) {
    const uint32_t o = __shfl_up(np, k * kChunksPerWave, 64);
    if ((int)seg >= k) before += o;
  }
  if (live) {
    uint16_t *peaks = peaks_all + row;
    for (uint32_t k = 0; k < np; ++k) peaks[before + k] = k < kPeakLds - 1u ? mine[k * 64u] : far[k];
    if (seg == kPeakSegs - 1) n_peaks[c] = before + np;
  }
}

// E4 -- event.h:184-224: event e spans [peaks[e-1], peaks[e]) (first from 0, last to n; the
// last peak is dropped).  num_events = 1 + #{i>=1 : 0 < peaks[i] < n}; every emitted peak is a
// sample index in [1, n-1] (index 0 is never examined: masked_to >= 0), so that is n_peaks.
// With fewer than 2 peaks the reference reads out of bounds: the chunk yields no events.
__global__ __launch_bounds__(256) void EventMeanKernel(const uint32_t *__restrict__ clen,
                                                       const float *__restrict__ ps_all,
                                                       const float *__restrict__ pss_all,
                                                       const uint16_t *__restrict__ peaks_all,
                                                       const uint32_t *__restrict__ n_peaks,
                                                       float *__restrict__ ev_mean, float *__restrict__ ev_stdv,
                                                       uint32_t feat_stride) {
  const uint32_t c = blockIdx.x;
  uint32_t n = clen[c];
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  const uint32_t E = n_peaks[c] >= 2 ? n_peaks[c] : 0;
  const float *ps = ps_all + (size_t)c * kRowStride, *pss = pss_all + (size_t)c * kRowStride;
  const uint16_t *peaks = peaks_all + (size_t)c * kRowStride;
  for (uint32_t e = threadIdx.x; e < E && e < feat_stride; e += 256) {
    const uint64_t start = e > 0 ? peaks[e - 1] : 0;
    const uint64_t end = e < E - 1 ? peaks[e] : n;
    const uint64_t length = end - start;  // size_t arithmetic, wraps like the reference
    const float mean = (ps[end] - ps[start]) / (float)length;
    const float deltasqr = pss[end] - pss[start];
    const float var = deltasqr / (float)length - mean * mean;
    ev_mean[(size_t)c * feat_stride + e] = mean;
    ev_stdv[(size_t)c * feat_stride + e] = sqrtf(fmaxf(var, 0.0f));
  }
}

// The spans EventMeanKernel averages over, written out for the per-stage tests (event.h:200-224:
// Event.start, Event.length).
__global__ __launch_bounds__(256) void EventSpanKernel(const uint32_t *__restrict__ clen,
                                                       const uint16_t *__restrict__ peaks_all,
                                                       const uint32_t *__restrict__ n_peaks, uint32_t feat_stride,
                                                       uint32_t *__restrict__ ev_start, uint32_t *__restrict__ ev_length) {
  const uint32_t c = blockIdx.x;
  uint32_t n = clen[c];
  if (n > (uint32_t)kMaxSamples) n = kMaxSamples;
  const uint32_t E = n_peaks[c] >= 2 ? n_peaks[c] : 0;
  const uint16_t *peaks = peaks_all + (size_t)c * kRowStride;
  for (uint32_t e = threadIdx.x; e < E && e < feat_stride; e += 256) {
    const uint32_t start = e > 0 ? peaks[e - 1] : 0;
    const uint32_t end = e < E - 1 ? peaks[e] : n;
    ev_start[(size_t)c * feat_stride + e] = start;
    ev_length[(size_t)c * feat_stride + e] = end - start;
  }
}

// E5 -- sigmap.cc:1131-1155 (z-score: sequential f64 sums) + sigmap.cc:1073-1079 (keep i iff
// i == 0 or |z[i] - last kept| > 0.1, float abs against the double constant).  Also derives the
// number of seeds of the chunk and initialises its result record.
__global__ __launch_bounds__(64) void FeatureKernel(uint32_t n_chunks, const uint32_t *__restrict__ n_peaks,
                                                    const float *__restrict__ ev_mean,
                                                    const float *__restrict__ ev_stdv, uint32_t feat_stride,
                                                    float *__restrict__ feat, float *__restrict__ stdv,
                                                    smhip_chunk_result *__restrict__ results,
                                                    uint32_t *__restrict__ nq, double compress_delta,
                                                    uint32_t min_num_events, int dim, int step) {
  const uint32_t c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_chunks) return;
  const uint32_t npk = n_peaks[c];
  uint32_t E = npk >= 2 ? npk : 0;
  if (E > feat_stride) E = feat_stride;
  const float *m = ev_mean + (size_t)c * feat_stride, *sd = ev_stdv + (size_t)c * feat_stride;
  float *fo = feat + (size_t)c * feat_stride;
  float *so = stdv ? stdv + (size_t)c * feat_stride : nullptr;
  uint32_t F = 0;
  if (E > 0) {
    double mean = 0;
    uint32_t i = 0;
    for (; i + kBlk <= E; i += kBlk) {
      float v[kBlk];
#pragma unroll
      for (int k = 0; k < kBlk; k += 4) {
        const F4u t = *reinterpret_cast<const F4u *>(m + i + k);
        v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < kBlk; ++k) mean += (double)v[k];
    }
    for (; i < E; ++i) mean += (double)m[i];
    mean /= (double)E;
    double SD = 0;
    for (i = 0; i + kBlk <= E; i += kBlk) {
      float v[kBlk];
#pragma unroll
      for (int k = 0; k < kBlk; k += 4) {
        const F4u t = *reinterpret_cast<const F4u *>(m + i + k);
        v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < kBlk; ++k) SD += ((double)v[k] - mean) * ((double)v[k] - mean);
    }
    for (; i < E; ++i) SD += ((double)m[i] - mean) * ((double)m[i] - mean);
    SD /= (double)(uint64_t)(E - 1);
    SD = sqrt(SD);
    float last = 0.0f;
    auto keep = [&](uint32_t at, float mv, float sv) {
      const float z = (float)(((double)mv - mean) / SD);
      if (at == 0 || (double)fabsf(z - last) > compress_delta) {
        fo[F] = z;
        if (so) so[F] = sv;
        last = z;
        ++F;
      }
    };
    // (blocks of values fetched together, like the two passes above: one memory round trip per block, not per event)
    for (i = 0; i + kBlk <= E; i += kBlk) {
      float v[kBlk], w[kBlk];
#pragma unroll
      for (int k = 0; k < kBlk; k += 4) {
        const F4u t = *reinterpret_cast<const F4u *>(m + i + k);
        v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
        F4u u = {0.0f, 0.0f, 0.0f, 0.0f};
        if (so) u = *reinterpret_cast<const F4u *>(sd + i + k);
        w[k] = u.x; w[k + 1] = u.y; w[k + 2] = u.z; w[k + 3] = u.w;
      }
#pragma unroll
      for (int k = 0; k < kBlk; ++k) keep(i + k, v[k], w[k]);
    }
    for (; i < E; ++i) keep(i, m[i], so ? sd[i] : 0.0f);
  }
  uint32_t Q = 0;
  // sigmap.cc:660 (> 50 features) and spatial_index.cc:326-356,403-408: seeds at
  // step, 2*step, ... <= nwin-1 (the "reliable seed" sort is a no-op)
  if (F > min_num_events && F >= (uint32_t)dim) {
    const uint32_t nwin = F - (uint32_t)dim + 1;
    Q = (nwin - 1) / (uint32_t)step;
  }
  smhip_chunk_result z;
  memset(&z, 0, sizeof z);  // the record is zeroed here; later kernels accumulate into it
  z.status = npk < 2 ? SMHIP_ST_FEW_PEAKS : 0;
  z.n_events = E;
  z.n_features = F;
  z.n_queries = Q;
  results[c] = z;
  nq[c] = Q;
}

// Exclusive scan of n uint32 values by one workgroup (n is a chunk count: small).
template <typename OutT, typename InT = uint32_t>
__global__ __launch_bounds__(1024) void ScanKernel(const InT *__restrict__ in, uint32_t n,
                                                   OutT *__restrict__ out) {
  __shared__ uint64_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t b = min(n, tid * per), e = min(n, b + per);
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
// Large inputs (the walker's query-order histogram: ~0.5 M bins): three coalesced launches --
// per-block sums, scan of the sums, per-block rescan with the block's offset.
constexpr uint32_t kScanTile = 4096;  // elements per 256-thread block
template <typename SumT>
__global__ __launch_bounds__(256) void ScanReduceKernel(const uint32_t *__restrict__ in, uint32_t n,
                                                        SumT *__restrict__ sums) {
  __shared__ SumT red[256];
  const uint32_t base = blockIdx.x * kScanTile, tid = threadIdx.x;
  SumT s = 0;
  for (uint32_t i = base + tid; i < min(n, base + kScanTile); i += 256) s += in[i];
  red[tid] = s;
  __syncthreads();
  for (uint32_t d = 128; d > 0; d >>= 1) {
    if (tid < d) red[tid] += red[tid + d];
    __syncthreads();
  }
  if (tid == 0) sums[blockIdx.x] = red[0];
}

// (OutT = uint64_t: 64-bit offsets; the partial sums inside 256 consecutive inputs stay 32-bit)
template <typename OutT>
__global__ __launch_bounds__(256) void ScanApplyKernel(const uint32_t *__restrict__ in, uint32_t n,
                                                       const OutT *__restrict__ block_excl,
                                                       OutT *__restrict__ out) {
  __shared__ uint32_t part[256];
  const uint32_t base = blockIdx.x * kScanTile, tid = threadIdx.x;
  OutT carry = block_excl[blockIdx.x];
  for (uint32_t t0 = base; t0 < min(n, base + kScanTile); t0 += 256) {  // 256 elements at a time, coalesced
    const uint32_t i = t0 + tid;
    const uint32_t v = i < n ? in[i] : 0;
    part[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
      const uint32_t o = tid >= d ? part[tid - d] : 0;
      __syncthreads();
      part[tid] += o;
      __syncthreads();
    }
    if (i < n) out[i] = carry + (OutT)(part[tid] - v);
    carry += (OutT)part[255];  // (256 inputs: below 2^32 for every caller -- per-chunk anchor counts, window widths)
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) out[n] = carry;
}
}  // namespace

hipError_t LaunchEvents(smhip_ctx *ctx, const smhip_params &p, const float *samples, const uint64_t *cbegin,
                        const uint32_t *clen, uint32_t n_chunks, uint32_t feat_stride, float *feat, float *stdv,
                        smhip_chunk_result *results, uint32_t *nq) {
  if (n_chunks == 0) return hipSuccess;
  const size_t row = (size_t)n_chunks * kRowStride;
  const size_t evn = (size_t)n_chunks * feat_stride;
  // scratch rows: ps, pss, t1, t2 (f32), event mean / stdv (f32), n_peaks (u32), peaks (u16)
  hipError_t e = ctx->ev_scratch.Reserve((row * 4 + evn * 2) * sizeof(float) + (size_t)n_chunks * 4 +
                                         2 * row * sizeof(uint16_t) + 256);
  if (e != hipSuccess) return e;
  float *ps = ctx->ev_scratch.as<float>(), *pss = ps + row, *t1 = pss + row, *t2 = t1 + row;
  float *ev_mean = t2 + row, *ev_stdv = ev_mean + evn;
  uint32_t *n_peaks = reinterpret_cast<uint32_t *>(ev_stdv + evn);
  uint16_t *peaks = reinterpret_cast<uint16_t *>(n_peaks + n_chunks), *seg_peaks = peaks + row;
  ctx->ev_rows = smhip_ctx::EvRows{ps, pss, t1, t2, ev_mean, ev_stdv, n_peaks, peaks, n_chunks, feat_stride,
                                   (uint32_t)kRowStride};
  // The prefix sums run a lane per chunk: every lane streams its own row, so a memory instruction of a full wave touches
  // 64 different lines; a SMALL batch -- a streaming round of a few hundred chunks: two or three such waves on a chip of
  // 256 CUs -- is cut into waves of 16 lanes on four times as many CUs: 124 -> 84 us at 64 chunks, 134 -> 92 at 1 200,
  // 150 -> 105 at 4 096.  (The z-score kernel, a lane per chunk too, does not change: 80 us of dependent f64 sums.)
  const uint32_t lanes_per_wave = n_chunks <= kNarrowLaneChunks ? 16u : 64u;
  const uint32_t lane_blocks = (n_chunks + lanes_per_wave - 1) / lanes_per_wave;
  hipLaunchKernelGGL(PrefixKernel, dim3(lane_blocks), dim3(lanes_per_wave), 0, ctx->stream, samples, cbegin, clen, n_chunks, ps, pss);
  hipLaunchKernelGGL(TStatKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, clen, ps, pss, t1, t2,
                     p.window_length1, p.window_length2);
  hipLaunchKernelGGL(PeakKernel, dim3((n_chunks * kPeakSegs + 63) / 64), dim3(64), 0, ctx->stream, clen, n_chunks, t1,
                     t2, peaks, n_peaks, seg_peaks,
                     p.window_length1, p.window_length2, p.threshold1, p.threshold2, p.peak_height);
  hipLaunchKernelGGL(EventMeanKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, clen, ps, pss, peaks, n_peaks,
                     ev_mean, ev_stdv, feat_stride);
  hipLaunchKernelGGL(FeatureKernel, dim3((n_chunks + 63) / 64), dim3(64), 0, ctx->stream, n_chunks, n_peaks, ev_mean, ev_stdv,
                     feat_stride, feat, stdv, results, nq, p.compress_delta, p.min_num_events, kMaxDim, p.step_size);
  return hipGetLastError();
}

// Features handed in by the caller (smhip_features_upload: the inputs of SpatialIndex::GenerateChains): chunk c =
// src[cbegin[c] .. + clen[c]) goes to its row of the feature array, and its record and query count are written as
// FeatureKernel writes them -- seeds at step, 2 * step, ... <= nwin - 1 (spatial_index.cc:326-356,403-408).
// GenerateChains itself has no minimum number of features (that is StreamingMap's test, sigmap.cc:660), and is
// undefined for fewer than `dim` of them (size_t underflow at :330): such a chunk issues no query here.
__global__ void FeaturesInKernel(const float *__restrict__ src, const uint64_t *__restrict__ cbegin,
                                 const uint32_t *__restrict__ clen, uint32_t feat_stride, int dim, int step,
                                 float *__restrict__ feat, smhip_chunk_result *__restrict__ results,
                                 uint32_t *__restrict__ nq) {
  const uint32_t c = blockIdx.x;
  const uint32_t F = clen[c];
  const float *in = src + cbegin[c];
  float *out = feat + (size_t)c * feat_stride;
  for (uint32_t i = threadIdx.x; i < F; i += blockDim.x) out[i] = in[i];
  if (threadIdx.x == 0) {
    uint32_t Q = 0;
    if (F >= (uint32_t)dim) Q = (F - (uint32_t)dim) / (uint32_t)step;  // (nwin - 1) / step
    smhip_chunk_result z;
    memset(&z, 0, sizeof z);
    z.n_features = F;
    z.n_queries = Q;
    results[c] = z;
    nq[c] = Q;
  }
}

hipError_t LaunchFeaturesIn(smhip_ctx *ctx, const smhip_params &p, const float *src, const uint64_t *cbegin,
                            const uint32_t *clen, uint32_t n_chunks, uint32_t feat_stride, float *feat,
                            smhip_chunk_result *results, uint32_t *nq) {
  ctx->ev_rows = smhip_ctx::EvRows{};  // (no event-detection intermediates for smhip_events_raw)
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(FeaturesInKernel, dim3(n_chunks), dim3(256), 0, ctx->stream, src, cbegin, clen, feat_stride,
                     kMaxDim, p.step_size, feat, results, nq);
  return hipGetLastError();
}

// A batch that is mapped a second time (a speculative batch that outgrew a buffer, api.hip FinishRun) starts from the
// features of its first pass, not from the read slots again -- the caller may have replaced the reads by then
// (smhip_reads_upload_raw of the next batch's reads into the same slots, queued behind this batch's kernels).  Only
// the front stage writes the feature rows and these four fields of a record; everything else is zeroed like
// FeatureKernel zeroes it, and the per-chunk query counts (emptied by the guard kernel) come back from the records.
__global__ void RestoreFrontKernel(uint32_t n_chunks, smhip_chunk_result *__restrict__ results, uint32_t *__restrict__ nq) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const smhip_chunk_result r = results[c];
  smhip_chunk_result z;
  memset(&z, 0, sizeof z);
  z.status = r.status & SMHIP_ST_FEW_PEAKS;
  z.n_events = r.n_events;
  z.n_features = r.n_features;
  z.n_queries = r.n_queries;
  results[c] = z;
  nq[c] = r.n_queries;
}

hipError_t LaunchRestoreFront(smhip_ctx *ctx, uint32_t n_chunks, smhip_chunk_result *results, uint32_t *nq) {
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(RestoreFrontKernel, dim3((n_chunks + 255) / 256), dim3(256), 0, ctx->stream, n_chunks, results, nq);
  return hipGetLastError();
}

hipError_t LaunchEventSpans(smhip_ctx *ctx, const uint32_t *clen, uint32_t *ev_start, uint32_t *ev_length) {
  const auto &r = ctx->ev_rows;
  if (r.n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(EventSpanKernel, dim3(r.n_chunks), dim3(256), 0, ctx->stream, clen, r.peaks, r.n_peaks,
                     r.feat_stride, ev_start, ev_length);
  return hipGetLastError();
}

hipError_t LaunchScanU32(smhip_ctx *ctx, const uint32_t *in, uint32_t n, uint32_t *out_excl) {
  if (n <= 16384) {
    hipLaunchKernelGGL(ScanKernel<uint32_t>, dim3(1), dim3(1024), 0, ctx->stream, in, n, out_excl);
    return hipGetLastError();
  }
  const uint32_t blocks = (n + kScanTile - 1) / kScanTile;
  hipError_t e = ctx->scan_tmp.Reserve(((size_t)blocks * 2 + 2) * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  uint32_t *sums = ctx->scan_tmp.as<uint32_t>(), *excl = sums + blocks;
  hipLaunchKernelGGL(ScanReduceKernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream, in, n, sums);
  if (blocks <= 16384) {
    hipLaunchKernelGGL(ScanKernel<uint32_t>, dim3(1), dim3(1024), 0, ctx->stream, sums, blocks, excl);
  } else {
    return hipErrorInvalidValue;  // > 64 M elements: not a size this path produces
  }
  hipLaunchKernelGGL(ScanApplyKernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream, in, n, excl, out_excl);
  return hipGetLastError();
}

hipError_t LaunchScanU32ToU64(smhip_ctx *ctx, const uint32_t *in, uint32_t n, uint64_t *out_excl) {
  if (n <= 16384) {
    hipLaunchKernelGGL(ScanKernel<uint64_t>, dim3(1), dim3(1024), 0, ctx->stream, in, n, out_excl);
    return hipGetLastError();
  }
  // per-block sums (64-bit), their scan, per-block rescan with a 64-bit offset
  const uint32_t blocks = (n + kScanTile - 1) / kScanTile;
  hipError_t e = ctx->scan_tmp.Reserve(((size_t)blocks * 2 + 2) * sizeof(uint64_t));
  if (e != hipSuccess) return e;
  uint64_t *sums = ctx->scan_tmp.as<uint64_t>(), *excl = sums + blocks;
  hipLaunchKernelGGL(ScanReduceKernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream, in, n, sums);
  hipLaunchKernelGGL((ScanKernel<uint64_t, uint64_t>), dim3(1), dim3(1024), 0, ctx->stream, sums, blocks, excl);
  hipLaunchKernelGGL(ScanApplyKernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream, in, n, excl, out_excl);
  return hipGetLastError();
}
}  // namespace smhip

// ---- smhip_debug_div_window_sweep: the t-statistics' x / w (DivW inside its range, the IEEE division outside: exactly
// what TStatAt does) against x / w, on the device, for every float whose bit pattern lies in [lo_bits, hi_bits)
namespace smhip {
namespace {
__global__ void DivWindowSweepKernel(uint32_t lo_bits, uint64_t count, uint32_t w, unsigned long long *mismatches) {
  const float wf = (float)w, rw = 1.0f / wf;
  const bool fast_w = w >= 1 && w <= kFastDivMaxW;
  unsigned long long bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float(lo_bits + (uint32_t)i);
    const float ax = fabsf(x);
    const float got = (fast_w && DivWRange(ax, ax)) ? DivW(x, wf, rw) : x / wf;
    const float want = x / wf;
    const bool same = __float_as_uint(got) == __float_as_uint(want) || (got != got && want != want);
    bad += same ? 0ull : 1ull;
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace
}  // namespace smhip
extern "C" int smhip_debug_div_window_sweep(smhip_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint32_t w,
                                            uint64_t *mismatches) {
  if (!ctx || !mismatches || hi_bits < lo_bits || w == 0) return SMHIP_EINVAL;
  unsigned long long *dev = nullptr;
  if (hipMalloc(&dev, sizeof *dev) != hipSuccess) return SMHIP_ENOMEM;
  int rc = SMHIP_OK;
  if (hipMemsetAsync(dev, 0, sizeof *dev, ctx->stream) != hipSuccess) rc = SMHIP_EHIP;
  if (rc == SMHIP_OK && hi_bits > lo_bits)
    hipLaunchKernelGGL(smhip::DivWindowSweepKernel, dim3(4096), dim3(256), 0, ctx->stream, lo_bits,
                       (uint64_t)hi_bits - lo_bits, w, dev);
  unsigned long long host = 0;
  if (rc == SMHIP_OK && (hipMemcpyAsync(&host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipStreamSynchronize(ctx->stream) != hipSuccess))
    rc = SMHIP_EHIP;
  (void)hipFree(dev);
  *mismatches = host;
  return rc;
}
