/*
 * sigmap_oracle.c -- CPU restatement of the Sigmap streaming-mapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sigmap_oracle.h).  Build: oracle/Makefile
 *   gcc -std=c99 -O2 -ffp-contract=off -fPIC -shared
 *
 * All citations are file:line under /root/reference/src.
 */
#define _POSIX_C_SOURCE 200809L
#include "sigmap_oracle.h"

#include <assert.h>
#include <float.h>
#include <malloc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ events */

/* event.h:58-68: sequential f32 prefix sums (order-sensitive). */
/* signal_batch.cc:139-151 / :195-207 */
size_t so_add_signal(const int16_t *raw, size_t n, float digitisation, float range, float offset, float *out) {
  size_t valid_signal_length = 0;
  const float scale = range / digitisation;              /* :140 / :196 (float / float) */
  for (size_t i = 0; i < n; ++i) {
    const float v = (float)raw[i];                       /* :137-138: file.read() into vector<float> */
    if ((v + offset) * scale > 30 && (v + offset) * scale < 200) {  /* :142-143 / :198-199 */
      out[valid_signal_length] = (v + offset) * scale;   /* :144 / :200 */
      ++valid_signal_length;
    }
  }
  return valid_signal_length;                            /* :148-151: the tail is erased */
}

void so_prefix_sums(const float *x, size_t n, float *ps, float *pss) {
  ps[0] = 0.0f;
  pss[0] = 0.0f;
  for (size_t i = 0; i < n; ++i) {
    ps[i + 1] = ps[i] + x[i];
    pss[i + 1] = pss[i] + x[i] * x[i];
  }
}

/* event.h:70-115.  `sum / window_length` is float / size_t -> float division by
 * (float)w; fabs/sqrt bind to the double overloads (SURVEY.md section 7 "hard parts"),
 * result narrowed to float by the vector<float>::emplace_back. */
void so_tstat(const float *ps, const float *pss, size_t n, size_t w, float *t) {
  const float eta = FLT_MIN;
  if (n < 2 * w || w < 2) { /* event.h:78-83 */
    for (size_t i = 0; i < n; ++i) t[i] = 0.0f;
    t[n] = 0.0f;
    return;
  }
  size_t k = 0;
  for (size_t i = 0; i < w; ++i) t[k++] = 0.0f; /* event.h:85-87 */
  const float wf = (float)w;
  for (size_t i = w; i <= n - w; ++i) { /* event.h:89-110 */
    float sum1 = ps[i];
    float sumsq1 = pss[i];
    if (i > w) {
      sum1 -= ps[i - w];
      sumsq1 -= pss[i - w];
    }
    float sum2 = ps[i + w] - ps[i];
    float sumsq2 = pss[i + w] - pss[i];
    float mean1 = sum1 / wf;
    float mean2 = sum2 / wf;
    float combined_var = sumsq1 / wf - mean1 * mean1 + sumsq2 / wf - mean2 * mean2;
    combined_var = fmaxf(combined_var, eta);
    const float delta_mean = mean2 - mean1;
    t[k++] = (float)(fabs((double)delta_mean) / sqrt((double)(combined_var / wf)));
  }
  for (size_t i = 0; i < w; ++i) t[k++] = 0.0f; /* event.h:112-114 */
  /* k == n + 1 */
}

typedef struct {
  const float *sig;
  float threshold;
  size_t window_length;
  size_t masked_to;
  int peak_pos;
  float peak_value;
  int valid_peak;
} so_detector; /* event.h:45-56 */

/* event.h:117-182 */
size_t so_peaks(const float *t1, const float *t2, size_t n, size_t w1, size_t w2,
                float thr1, float thr2, float peak_height, uint64_t *peaks) {
  const int DEF_PEAK_POS = -1;
  const float DEF_PEAK_VAL = FLT_MAX;
  so_detector d[2] = {{t1, thr1, w1, 0, -1, FLT_MAX, 0}, {t2, thr2, w2, 0, -1, FLT_MAX, 0}};
  so_detector *sd = &d[0], *ld = &d[1];
  size_t np = 0;
  for (size_t i = 0; i < n; i++) {
    for (size_t k = 0; k < 2; k++) {
      so_detector *det = &d[k];
      if (det->masked_to >= i) continue; /* event.h:130-132 */
      float cur = det->sig[i];
      if (det->peak_pos == DEF_PEAK_POS) {
        if (cur < det->peak_value) { /* event.h:136-138 */
          det->peak_value = cur;
        } else if (cur - det->peak_value > peak_height) { /* event.h:139-146 */
          det->peak_value = cur;
          det->peak_pos = (int)i;
        }
      } else {
        if (cur > det->peak_value) { /* event.h:149-153 */
          det->peak_value = cur;
          det->peak_pos = (int)i;
        }
        if (det == sd) { /* event.h:155-164 */
          if (det->peak_value > det->threshold) {
            /* int + size_t -> size_t */
            ld->masked_to = (size_t)det->peak_pos + det->window_length;
            ld->peak_pos = DEF_PEAK_POS;
            ld->peak_value = DEF_PEAK_VAL;
            ld->valid_peak = 0;
          }
        }
        if (det->peak_value - cur > peak_height && det->peak_value > det->threshold) {
          det->valid_peak = 1; /* event.h:166-169 */
        }
        /* event.h:171-178: (i - peak_pos) is size_t arithmetic */
        if (det->valid_peak && (i - (size_t)det->peak_pos) > det->window_length / 2) {
          peaks[np++] = (uint64_t)det->peak_pos;
          det->peak_pos = DEF_PEAK_POS;
          det->peak_value = cur;
          det->valid_peak = 0;
        }
      }
    }
  }
  return np;
}

/* event.h:184-198.  event.length is size_t; the divisions convert it to float. */
static so_event so_create_event(uint64_t start, uint64_t end, const float *ps,
                                const float *pss) {
  so_event e;
  e.start = start;
  e.length = end - start;
  e.mean = (ps[end] - ps[start]) / (float)e.length;
  float deltasqr = pss[end] - pss[start];
  float var = deltasqr / (float)e.length - e.mean * e.mean;
  e.stdv = sqrtf(fmaxf(var, 0.0f));
  return e;
}

/* event.h:200-224 */
size_t so_create_events(const uint64_t *peaks, size_t n_peaks, const float *ps,
                        const float *pss, size_t n, so_event *events) {
  if (n_peaks < 2) return 0; /* reference indexes peaks[-1] / peaks[0] out of bounds */
  size_t num_events = 1;
  for (size_t i = 1; i < n_peaks; ++i) {
    if (peaks[i] > 0 && peaks[i] < n) num_events++;
  }
  if (num_events < 2) return 0;
  size_t k = 0;
  events[k++] = so_create_event(0, peaks[0], ps, pss);
  for (size_t pi = 1; pi < num_events - 1; pi++) {
    events[k++] = so_create_event(peaks[pi - 1], peaks[pi], ps, pss);
  }
  events[k++] = so_create_event(peaks[num_events - 2], n, ps, pss);
  return k;
}

/* event.h:226-267 with event_detection_defaults (event.h:31-37) */
size_t so_detect_events(const float *x, size_t n, so_event *events, uint64_t *peaks_out,
                        size_t *n_peaks_out) {
  float *ps = (float *)malloc((n + 1) * sizeof(float));
  float *pss = (float *)malloc((n + 1) * sizeof(float));
  float *t1 = (float *)malloc((n + 1) * sizeof(float));
  float *t2 = (float *)malloc((n + 1) * sizeof(float));
  uint64_t *peaks = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
  so_prefix_sums(x, n, ps, pss);
  so_tstat(ps, pss, n, 3, t1);
  so_tstat(ps, pss, n, 6, t2);
  size_t np = so_peaks(t1, t2, n, 3, 6, 4.30265f, 2.57058f, 1.0f, peaks);
  size_t ne = so_create_events(peaks, np, ps, pss, n, events);
  if (peaks_out) memcpy(peaks_out, peaks, np * sizeof(uint64_t));
  if (n_peaks_out) *n_peaks_out = np;
  free(ps);
  free(pss);
  free(t1);
  free(t2);
  free(peaks);
  return ne;
}

/* sigmap.cc:1131-1155: sums in double, in index order; output narrowed to float. */
void so_zscore(const float *x, size_t n, float *out) {
  double mean = 0;
  for (size_t i = 0; i < n; ++i) mean += x[i];
  mean /= n;
  double SD = 0;
  for (size_t i = 0; i < n; ++i) SD += (x[i] - mean) * (x[i] - mean);
  SD /= (n - 1);
  SD = sqrt(SD);
  for (size_t i = 0; i < n; ++i) out[i] = (float)((x[i] - mean) / SD);
}

/* sigmap.cc:1048-1083 */
size_t so_generate_events(const float *x, size_t n, float *feat, float *stdv) {
  so_event *ev = (so_event *)malloc((n + 1) * sizeof(so_event));
  size_t ne = so_detect_events(x, n, ev, NULL, NULL);
  float *means = (float *)malloc((ne + 1) * sizeof(float));
  float *z = (float *)malloc((ne + 1) * sizeof(float));
  for (size_t i = 0; i < ne; ++i) means[i] = ev[i].mean;
  if (ne > 0) so_zscore(means, ne, z);
  size_t k = 0;
  for (size_t i = 0; i < ne; ++i) {
    /* sigmap.cc:1075: float abs(), compared against the double literal 0.1 */
    if (i == 0 || ((double)fabsf(z[i] - feat[k - 1]) > 0.1)) {
      feat[k] = z[i];
      if (stdv) stdv[k] = ev[i].stdv;
      ++k;
    }
  }
  free(ev);
  free(means);
  free(z);
  return k;
}

/* ------------------------------------------------------- sequence -> signal */

/* utils.h:73-87 */
static const uint8_t so_c2u[256] = {
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

/* utils.h:98-116 */
static uint64_t so_seed_from_sequence(const char *seq, size_t seq_len, uint32_t start,
                                      uint32_t seed_len) {
  uint64_t mask = (((uint64_t)1) << (2 * seed_len)) - 1;
  uint64_t seed = 0;
  for (uint32_t i = 0; i < seed_len; ++i) {
    if ((size_t)start + i < seq_len) {
      uint8_t b = so_c2u[(uint8_t)seq[i + start]];
      if (b < 4)
        seed = ((seed << 2) | b) & mask;
      else
        seed = (seed << 2) & mask;
    } else {
      seed = (seed << 2) & mask;
    }
  }
  return seed;
}

/* pore_model.cc:57-81.  Quirks kept: the `sequence_length` argument handed to
 * GenerateSeedFromSequence is start+signal_length, and the rolling update reads
 * sequence[position + k] (skipping base start+k; last read is the NUL terminator). */
size_t so_level_means(const float *level_mean, int k, const char *seq, uint32_t start,
                      uint32_t end, float *out) {
  int32_t signal_length = (int32_t)(end - start - (uint32_t)k + 1);
  if (signal_length <= 0) return 0;
  uint32_t mask = ((uint32_t)1 << (2 * k)) - 1;
  uint32_t h = (uint32_t)so_seed_from_sequence(seq, (size_t)start + (size_t)signal_length,
                                               start, (uint32_t)k);
  size_t n = 0;
  out[n++] = level_mean[h];
  for (uint32_t pos = start + 1; pos < end - (uint32_t)k + 1; ++pos) {
    uint8_t b = so_c2u[(uint8_t)seq[pos + (uint32_t)k]];
    if (b < 4)
      h = ((h << 2) | b) & mask;
    else
      h = (h << 2) & mask;
    out[n++] = level_mean[h];
  }
  return n;
}

/* sequence_batch.h:71-83 */
void so_reverse_complement(const char *seq, size_t len, char *out) {
  static const char u2c[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
  for (size_t i = 0; i < len; ++i) {
    out[i] = u2c[((uint8_t)3) ^ so_c2u[(uint8_t)seq[len - i - 1]]];
  }
  out[len] = 0;
}

/* --- tiny open-addressing u64 -> u64 counter (stands in for khash k64) --- */
typedef struct {
  uint64_t *keys, *vals;
  uint8_t *used;
  size_t cap, n;
} so_hist;
static uint64_t so_mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
static void so_hist_init(so_hist *h, size_t cap) {
  h->cap = cap;
  h->n = 0;
  h->keys = (uint64_t *)malloc(cap * 8);
  h->vals = (uint64_t *)calloc(cap, 8);
  h->used = (uint8_t *)calloc(cap, 1);
}
static void so_hist_free(so_hist *h) {
  free(h->keys);
  free(h->vals);
  free(h->used);
}
static uint64_t *so_hist_slot(so_hist *h, uint64_t key, int create);
static void so_hist_grow(so_hist *h) {
  so_hist o = *h;
  so_hist_init(h, o.cap * 2);
  for (size_t i = 0; i < o.cap; ++i)
    if (o.used[i]) *so_hist_slot(h, o.keys[i], 1) = o.vals[i];
  so_hist_free(&o);
}
static uint64_t *so_hist_slot(so_hist *h, uint64_t key, int create) {
  if (create && (h->n + 1) * 10 > h->cap * 7) so_hist_grow(h);
  size_t i = so_mix(key) & (h->cap - 1);
  while (h->used[i]) {
    if (h->keys[i] == key) return &h->vals[i];
    i = (i + 1) & (h->cap - 1);
  }
  if (!create) return NULL;
  h->used[i] = 1;
  h->keys[i] = key;
  h->vals[i] = 0;
  h->n++;
  return &h->vals[i];
}

/* One strand of the masking pass, sigmap.cc:80-130 (positive) == :131-181 (negative). */
static void so_mask_one(const char *seq, uint32_t len, int kmer_size, float frequency,
                        so_hist *hist, uint64_t num_kmers, uint8_t *masked) {
  uint64_t shift = 2 * (uint64_t)(kmer_size - 1);
  uint64_t mask = (((uint64_t)1) << (2 * kmer_size)) - 1;
  uint64_t s[2] = {0, 0};
  int unamb = 0;
  for (uint32_t pos = 0; pos < len; ++pos) {
    uint8_t b = so_c2u[(uint8_t)seq[pos]];
    if (b < 4) {
      s[0] = ((s[0] << 2) | b) & mask;
      s[1] = (s[1] >> 2) | (((uint64_t)(3 ^ b)) << shift);
      ++unamb;
      if (unamb >= kmer_size) {
        uint64_t strand = s[0] < s[1] ? 0 : 1;
        uint64_t *v = so_hist_slot(hist, s[strand], 0);
        assert(v);
        uint64_t freq = *v;
        masked[pos + 1 - (uint32_t)kmer_size] = ((freq / (float)num_kmers) > frequency) ? 1 : 0;
      }
    } else {
      unamb = 0;
      s[0] = 0;
      s[1] = 0;
      if (pos >= (uint32_t)kmer_size - 1) masked[pos + 1 - (uint32_t)kmer_size] = 1;
    }
  }
}

/* sigmap.cc:19-185 */
void so_masked_positions(int kmer_size, float frequency, uint32_t n_seqs,
                         const char *const *seqs, const char *const *neg_seqs,
                         const uint32_t *lens, uint8_t **pos_masked, uint8_t **neg_masked) {
  so_hist hist;
  so_hist_init(&hist, 1 << 16);
  uint64_t shift = 2 * (uint64_t)(kmer_size - 1);
  uint64_t mask = (((uint64_t)1) << (2 * kmer_size)) - 1;
  uint64_t num_kmers = 0;
  for (uint32_t si = 0; si < n_seqs; ++si) { /* sigmap.cc:31-76: counts + strand only */
    uint64_t s[2] = {0, 0};
    int unamb = 0;
    for (uint32_t pos = 0; pos < lens[si]; ++pos) {
      uint8_t b = so_c2u[(uint8_t)seqs[si][pos]];
      if (b < 4) {
        s[0] = ((s[0] << 2) | b) & mask;
        s[1] = (s[1] >> 2) | (((uint64_t)(3 ^ b)) << shift);
        ++unamb;
        if (unamb >= kmer_size) {
          uint64_t strand = s[0] < s[1] ? 0 : 1;
          *so_hist_slot(&hist, s[strand], 1) += 1;
          ++num_kmers;
        }
      } else {
        unamb = 0;
        s[0] = 0;
        s[1] = 0;
      }
    }
  }
  for (uint32_t si = 0; si < n_seqs; ++si) {
    size_t m = (size_t)lens[si] - (size_t)kmer_size + 1;
    memset(pos_masked[si], 0, m);
    memset(neg_masked[si], 0, m);
    so_mask_one(seqs[si], lens[si], kmer_size, frequency, &hist, num_kmers, pos_masked[si]);
    so_mask_one(neg_seqs[si], lens[si], kmer_size, frequency, &hist, num_kmers, neg_masked[si]);
  }
  so_hist_free(&hist);
}

/* spatial_index.cc:33-58.  std::fabs(float) > 0.01 (double literal). */
size_t so_point_cloud_one_direction(int positive, uint32_t signal_index,
                                    const uint8_t *is_masked, const float *signal,
                                    size_t signal_length, int dim, uint64_t *positions,
                                    float *values, size_t count) {
  if (signal_length >= (uint32_t)dim) {
    for (size_t sp = 0; sp < signal_length - (size_t)dim + 1; sp += 1) {
      if (!is_masked[sp]) {
        if (sp == 0 || count == 0 ||
            (sp > 0 && (double)fabsf(signal[sp] - values[count - 1]) > 0.01)) {
          uint64_t strand = positive ? 0 : 1;
          uint64_t position = ((((uint64_t)signal_index) << 32 | (uint32_t)sp) << 1) | strand;
          positions[count] = position;
          values[count] = signal[sp];
          ++count;
        }
      }
    }
  }
  return count;
}

/* ------------------------------------------------------------------ k-d tree */

#define SO_PT(t, idx, d) ((t)->values[(idx) + (uint64_t)(d)]) /* sigmap_adaptor.h:94-97 */

static int64_t so_new_node(so_kdtree *t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (so_node *)realloc(t->nodes, t->cap_nodes * sizeof(so_node));
  }
  memset(&t->nodes[t->n_nodes], 0, sizeof(so_node));
  return (int64_t)t->n_nodes++;
}

/* nanoflann.hpp:837-849 */
static void so_minmax(const so_kdtree *t, const uint64_t *ind, uint64_t count, int el,
                      float *mn, float *mx) {
  *mn = SO_PT(t, ind[0], el);
  *mx = SO_PT(t, ind[0], el);
  for (uint64_t i = 1; i < count; ++i) {
    float v = SO_PT(t, ind[i], el);
    if (v < *mn) *mn = v;
    if (v > *mx) *mx = v;
  }
}

/* nanoflann.hpp:960-1004 (IndexType = size_t, so `right` is unsigned) */
static void so_plane_split(const so_kdtree *t, uint64_t *ind, uint64_t count, int cutfeat,
                           float cutval, uint64_t *lim1, uint64_t *lim2) {
  uint64_t left = 0;
  uint64_t right = count - 1;
  for (;;) {
    while (left <= right && SO_PT(t, ind[left], cutfeat) < cutval) ++left;
    while (right && left <= right && SO_PT(t, ind[right], cutfeat) >= cutval) --right;
    if (left > right || !right) break;
    uint64_t tmp = ind[left];
    ind[left] = ind[right];
    ind[right] = tmp;
    ++left;
    --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && SO_PT(t, ind[left], cutfeat) <= cutval) ++left;
    while (right && left <= right && SO_PT(t, ind[right], cutfeat) > cutval) --right;
    if (left > right || !right) break;
    uint64_t tmp = ind[left];
    ind[left] = ind[right];
    ind[right] = tmp;
    ++left;
    --right;
  }
  *lim2 = left;
}

/* nanoflann.hpp:906-949 */
static void so_middle_split(const so_kdtree *t, uint64_t *ind, uint64_t count, uint64_t *index,
                            int *cutfeat, float *cutval, const float *lo, const float *hi) {
  const float EPS = 0.00001f;
  const int dim = t->dim;
  float max_span = hi[0] - lo[0];
  for (int i = 1; i < dim; ++i) {
    float span = hi[i] - lo[i];
    if (span > max_span) max_span = span;
  }
  float max_spread = -1;
  *cutfeat = 0;
  for (int i = 0; i < dim; ++i) {
    float span = hi[i] - lo[i];
    if (span > (1 - EPS) * max_span) {
      float mn, mx;
      so_minmax(t, ind, count, i, &mn, &mx);
      float spread = mx - mn;
      if (spread > max_spread) {
        *cutfeat = i;
        max_spread = spread;
      }
    }
  }
  float split_val = (lo[*cutfeat] + hi[*cutfeat]) / 2;
  float mn, mx;
  so_minmax(t, ind, count, *cutfeat, &mn, &mx);
  if (split_val < mn)
    *cutval = mn;
  else if (split_val > mx)
    *cutval = mx;
  else
    *cutval = split_val;
  uint64_t lim1, lim2;
  so_plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
  if (lim1 > count / 2)
    *index = lim1;
  else if (lim2 < count / 2)
    *index = lim2;
  else
    *index = count / 2;
}

/* nanoflann.hpp:858-904.  bbox (lo/hi) is in-out. */
static int64_t so_divide_tree(so_kdtree *t, uint64_t left, uint64_t right, float *lo, float *hi) {
  const int dim = t->dim;
  int64_t ni = so_new_node(t);
  if ((right - left) <= t->leaf_max) {
    t->nodes[ni].is_leaf = 1;
    t->nodes[ni].left = left;
    t->nodes[ni].right = right;
    t->nodes[ni].child1 = t->nodes[ni].child2 = -1;
    for (int i = 0; i < dim; ++i) {
      lo[i] = SO_PT(t, t->vind[left], i);
      hi[i] = SO_PT(t, t->vind[left], i);
    }
    for (uint64_t k = left + 1; k < right; ++k) {
      for (int i = 0; i < dim; ++i) {
        float v = SO_PT(t, t->vind[k], i);
        if (lo[i] > v) lo[i] = v;
        if (hi[i] < v) hi[i] = v;
      }
    }
  } else {
    uint64_t idx;
    int cutfeat;
    float cutval;
    so_middle_split(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, lo, hi);
    float llo[16], lhi[16], rlo[16], rhi[16];
    memcpy(llo, lo, sizeof(float) * (size_t)dim);
    memcpy(lhi, hi, sizeof(float) * (size_t)dim);
    lhi[cutfeat] = cutval;
    int64_t c1 = so_divide_tree(t, left, left + idx, llo, lhi);
    memcpy(rlo, lo, sizeof(float) * (size_t)dim);
    memcpy(rhi, hi, sizeof(float) * (size_t)dim);
    rlo[cutfeat] = cutval;
    int64_t c2 = so_divide_tree(t, left + idx, right, rlo, rhi);
    so_node *nd = &t->nodes[ni]; /* re-fetch: nodes may have been realloc'd */
    nd->is_leaf = 0;
    nd->divfeat = cutfeat;
    nd->child1 = c1;
    nd->child2 = c2;
    nd->divlow = lhi[cutfeat];
    nd->divhigh = rlo[cutfeat];
    for (int i = 0; i < dim; ++i) {
      lo[i] = llo[i] < rlo[i] ? llo[i] : rlo[i];   /* std::min(left, right) */
      hi[i] = lhi[i] < rhi[i] ? rhi[i] : lhi[i];   /* std::max(left, right) */
    }
  }
  return ni;
}

/* nanoflann.hpp:1196-1210 buildIndex + :1318-1344 computeBoundingBox + :1308-1316 init_vind */
so_kdtree *so_kdtree_build(const float *values, uint64_t n_values, int dim, int leaf_max) {
  if (dim > 16 || n_values < (uint64_t)dim) return NULL;
  so_kdtree *t = (so_kdtree *)calloc(1, sizeof(so_kdtree));
  t->dim = dim;
  t->leaf_max = (uint64_t)leaf_max;
  t->values = values;
  t->n_values = n_values;
  t->m_size = n_values - (uint64_t)dim + 1; /* sigmap_adaptor.h:89-91 */
  t->vind = (uint64_t *)malloc(t->m_size * sizeof(uint64_t));
  for (uint64_t i = 0; i < t->m_size; ++i) t->vind[i] = i;
  t->bbox_low = (float *)malloc(sizeof(float) * (size_t)dim);
  t->bbox_high = (float *)malloc(sizeof(float) * (size_t)dim);
  for (int i = 0; i < dim; ++i) t->bbox_low[i] = t->bbox_high[i] = SO_PT(t, 0, i);
  for (uint64_t k = 1; k < t->m_size; ++k) {
    for (int i = 0; i < dim; ++i) {
      float v = SO_PT(t, k, i);
      if (v < t->bbox_low[i]) t->bbox_low[i] = v;
      if (v > t->bbox_high[i]) t->bbox_high[i] = v;
    }
  }
  so_divide_tree(t, 0, t->m_size, t->bbox_low, t->bbox_high);
  return t;
}

void so_kdtree_free(so_kdtree *t) {
  if (!t) return;
  free(t->vind);
  free(t->nodes);
  free(t->bbox_low);
  free(t->bbox_high);
  free(t);
}

/* On-disk node, nanoflann.hpp:755-768: 16-byte union + two 8-byte pointers. */
typedef struct {
  union {
    struct {
      uint64_t left, right;
    } lr;
    struct {
      int32_t divfeat;
      float divlow, divhigh;
    } sub;
  } u;
  uint64_t child1, child2;
} so_disk_node;

/* nanoflann.hpp:1051-1058 saveIndex_, :1025-1033 save_tree (pre-order) */
int so_kdtree_save(const so_kdtree *t, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  uint64_t m = t->m_size;
  int32_t dim = t->dim;
  fwrite(&m, 8, 1, f);
  fwrite(&dim, 4, 1, f);
  uint64_t cnt = (uint64_t)dim; /* save_value(vector<Interval>) */
  fwrite(&cnt, 8, 1, f);
  for (int i = 0; i < dim; ++i) {
    fwrite(&t->bbox_low[i], 4, 1, f);
    fwrite(&t->bbox_high[i], 4, 1, f);
  }
  uint64_t lm = t->leaf_max;
  fwrite(&lm, 8, 1, f);
  fwrite(&m, 8, 1, f); /* save_value(vector<size_t> vind) */
  fwrite(t->vind, 8, m, f);
  for (uint64_t i = 0; i < t->n_nodes; ++i) {
    so_disk_node d;
    memset(&d, 0, sizeof(d));
    if (t->nodes[i].is_leaf) {
      d.u.lr.left = t->nodes[i].left;
      d.u.lr.right = t->nodes[i].right;
    } else {
      d.u.sub.divfeat = t->nodes[i].divfeat;
      d.u.sub.divlow = t->nodes[i].divlow;
      d.u.sub.divhigh = t->nodes[i].divhigh;
      d.child1 = 1; /* only NULL / non-NULL is meaningful on reload */
      d.child2 = 1;
    }
    fwrite(&d, sizeof(d), 1, f);
  }
  fclose(f);
  return 0;
}

static int64_t so_load_tree(so_kdtree *t, FILE *f) { /* nanoflann.hpp:1035-1044 */
  so_disk_node d;
  if (fread(&d, sizeof(d), 1, f) != 1) return -2;
  int64_t ni = so_new_node(t);
  if (d.child1 == 0 && d.child2 == 0) {
    t->nodes[ni].is_leaf = 1;
    t->nodes[ni].left = d.u.lr.left;
    t->nodes[ni].right = d.u.lr.right;
    t->nodes[ni].child1 = t->nodes[ni].child2 = -1;
  } else {
    int64_t c1 = -1, c2 = -1;
    if (d.child1 != 0) c1 = so_load_tree(t, f);
    if (d.child2 != 0) c2 = so_load_tree(t, f);
    so_node *nd = &t->nodes[ni];
    nd->is_leaf = 0;
    nd->divfeat = d.u.sub.divfeat;
    nd->divlow = d.u.sub.divlow;
    nd->divhigh = d.u.sub.divhigh;
    nd->child1 = c1;
    nd->child2 = c2;
  }
  return ni;
}

/* nanoflann.hpp:1065-1072 loadIndex_ */
so_kdtree *so_kdtree_load(const char *path, const float *values, uint64_t n_values) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  so_kdtree *t = (so_kdtree *)calloc(1, sizeof(so_kdtree));
  uint64_t m, cnt, lm, vn;
  int32_t dim;
  if (fread(&m, 8, 1, f) != 1 || fread(&dim, 4, 1, f) != 1 || fread(&cnt, 8, 1, f) != 1) goto bad;
  t->m_size = m;
  t->dim = dim;
  t->values = values;
  t->n_values = n_values;
  t->bbox_low = (float *)malloc(sizeof(float) * cnt);
  t->bbox_high = (float *)malloc(sizeof(float) * cnt);
  for (uint64_t i = 0; i < cnt; ++i) {
    if (fread(&t->bbox_low[i], 4, 1, f) != 1 || fread(&t->bbox_high[i], 4, 1, f) != 1) goto bad;
  }
  if (fread(&lm, 8, 1, f) != 1 || fread(&vn, 8, 1, f) != 1) goto bad;
  t->leaf_max = lm;
  t->vind = (uint64_t *)malloc(vn * 8);
  if (fread(t->vind, 8, vn, f) != vn) goto bad;
  if (so_load_tree(t, f) < 0) goto bad;
  fclose(f);
  return t;
bad:
  fclose(f);
  so_kdtree_free(t);
  return NULL;
}

/* The tree of loadIndex_ (nanoflann.hpp:1035-1072) from the SAME content as a .si file, handed over
 * in memory instead of through the file: vind as uint32, the pre-order node stream as 16-byte records
 * {int32 feat (-1 = leaf); uint32 a, b (leaf: left, right; inner: divlow, divhigh bits); uint32
 * child2 (inner: index of the second child; child1 = self + 1)}, root box {low0, high0, low1, ...}.
 * (A 250 Mb index is 14 GB of .pt/.si on disk and 7 minutes to write; the tests that need the oracle on
 * it hand the arrays over directly.) */
so_kdtree *so_kdtree_from_preorder(const float *values, uint64_t n_values, int dim, uint64_t leaf_max,
                                   const uint32_t *vind, uint64_t m_size, const void *nodes16, uint64_t n_nodes,
                                   const float *bbox) {
  if (dim > 16 || n_values < (uint64_t)dim || m_size != n_values - (uint64_t)dim + 1 || n_nodes == 0) return NULL;
  const struct {
    int32_t feat;
    uint32_t a, b, child2;
  } *in = nodes16;
  so_kdtree *t = (so_kdtree *)calloc(1, sizeof(so_kdtree));
  t->dim = dim;
  t->leaf_max = leaf_max;
  t->values = values;
  t->n_values = n_values;
  t->m_size = m_size;
  t->bbox_low = (float *)malloc(sizeof(float) * (size_t)dim);
  t->bbox_high = (float *)malloc(sizeof(float) * (size_t)dim);
  for (int d = 0; d < dim; ++d) {
    t->bbox_low[d] = bbox[2 * d];
    t->bbox_high[d] = bbox[2 * d + 1];
  }
  t->vind = (uint64_t *)malloc(m_size * sizeof(uint64_t));
  for (uint64_t i = 0; i < m_size; ++i) t->vind[i] = vind[i];
  t->nodes = (so_node *)calloc(n_nodes, sizeof(so_node));
  t->n_nodes = t->cap_nodes = n_nodes;
  for (uint64_t i = 0; i < n_nodes; ++i) {
    so_node *nd = &t->nodes[i];
    if (in[i].feat < 0) {
      nd->is_leaf = 1;
      nd->left = in[i].a;
      nd->right = in[i].b;
      nd->child1 = nd->child2 = -1;
    } else {
      nd->is_leaf = 0;
      nd->divfeat = in[i].feat;
      memcpy(&nd->divlow, &in[i].a, 4);
      memcpy(&nd->divhigh, &in[i].b, 4);
      nd->child1 = (int64_t)i + 1;
      nd->child2 = (int64_t)in[i].child2;
    }
  }
  return t;
}

/* spatial_index.cc:105-123: Point = {uint64 position; float value; 4 B pad} */
int so_points_save(const char *path, int dim, int max_leaf, const uint64_t *positions,
                   const float *values, uint64_t n) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  int32_t d = dim, ml = max_leaf;
  fwrite(&d, 4, 1, f);
  fwrite(&ml, 4, 1, f);
  fwrite(&n, 8, 1, f);
  for (uint64_t i = 0; i < n; ++i) {
    struct {
      uint64_t position;
      float value;
      uint32_t pad;
    } p = {positions[i], values[i], 0};
    fwrite(&p, 16, 1, f);
  }
  fclose(f);
  return 0;
}

/* spatial_index.cc:132-146 */
int so_points_load(const char *path, int *dim, int *max_leaf, uint64_t **positions,
                   float **values, uint64_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  int32_t d, ml;
  uint64_t cnt;
  if (fread(&d, 4, 1, f) != 1 || fread(&ml, 4, 1, f) != 1 || fread(&cnt, 8, 1, f) != 1) {
    fclose(f);
    return -2;
  }
  *dim = d;
  *max_leaf = ml;
  *n = cnt;
  *positions = (uint64_t *)malloc(cnt * 8 + 8);
  *values = (float *)malloc(cnt * 4 + 4);
  for (uint64_t i = 0; i < cnt; ++i) {
    struct {
      uint64_t position;
      float value;
      uint32_t pad;
    } p;
    if (fread(&p, 16, 1, f) != 1) {
      fclose(f);
      return -3;
    }
    (*positions)[i] = p.position;
    (*values)[i] = p.value;
  }
  fclose(f);
  return 0;
}

/* nanoflann.hpp:383-408 L2_Adaptor::evalMetric, called with worst_dist = -1 (:1357-1358).
 * General in `size`: groups of 4 then a tail. */
static float so_eval_metric(const so_kdtree *t, const float *a, uint64_t b_idx, int size) {
  float result = 0.0f;
  int d = 0;
  const float *last = a + size;
  const float *lastgroup = last - 3;
  while (a < lastgroup) {
    const float diff0 = a[0] - SO_PT(t, b_idx, d);
    const float diff1 = a[1] - SO_PT(t, b_idx, d + 1);
    const float diff2 = a[2] - SO_PT(t, b_idx, d + 2);
    const float diff3 = a[3] - SO_PT(t, b_idx, d + 3);
    d += 4;
    result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
    a += 4;
  }
  while (a < last) {
    const float diff0 = *a++ - SO_PT(t, b_idx, d);
    d++;
    result += diff0 * diff0;
  }
  return result;
}

typedef struct {
  const so_kdtree *t;
  const float *vec;
  float radius;
  uint64_t *idx;
  float *dist;
  uint64_t cap, n;
  so_search_stats *st;
} so_search;

/* nanoflann.hpp:1346-1410 searchLevel, RadiusResultSet (:219-272: worstDist()==radius,
 * addPoint keeps dist < radius and always returns true) */
static void so_search_level(so_search *s, int64_t ni, float mindistsq, float *dists) {
  const so_kdtree *t = s->t;
  const so_node *node = &t->nodes[ni];
  if (s->st) s->st->nodes++;
  if (node->is_leaf) {
    if (s->st) s->st->leaves++;
    float worst = s->radius;
    for (uint64_t i = node->left; i < node->right; ++i) {
      const uint64_t index = t->vind[i];
      float dist = so_eval_metric(t, s->vec, index, t->dim);
      if (s->st) s->st->points++;
      if (dist < worst) {
        if (dist < s->radius) { /* addPoint */
          if (s->n < s->cap) {
            s->idx[s->n] = t->vind[i];
            s->dist[s->n] = dist;
          }
          s->n++;
        }
      }
    }
    return;
  }
  int idx = node->divfeat;
  float val = s->vec[idx];
  float diff1 = val - node->divlow;
  float diff2 = val - node->divhigh;
  int64_t best, other;
  float cut_dist;
  if ((diff1 + diff2) < 0) {
    best = node->child1;
    other = node->child2;
    cut_dist = (val - node->divhigh) * (val - node->divhigh); /* accum_dist :410-413 */
  } else {
    best = node->child2;
    other = node->child1;
    cut_dist = (val - node->divlow) * (val - node->divlow);
  }
  so_search_level(s, best, mindistsq, dists);
  float dst = dists[idx];
  mindistsq = mindistsq + cut_dist - dst;
  dists[idx] = cut_dist;
  const float epsError = 1.0f; /* 1 + searchParams.eps, eps = 0 */
  if (mindistsq * epsError <= s->radius) {
    so_search_level(s, other, mindistsq, dists);
  }
  dists[idx] = dst;
}

uint64_t so_radius_search(const so_kdtree *t, const float *q, float radius, uint64_t *idx,
                          float *dist, uint64_t cap, so_search_stats *st) {
  so_search s = {t, q, radius, idx, dist, cap, 0, st};
  float dists[16];
  for (int i = 0; i < t->dim; ++i) dists[i] = 0.0f;
  float distsq = 0.0f; /* nanoflann.hpp:1006-1023 computeInitialDistances */
  for (int i = 0; i < t->dim; ++i) {
    if (q[i] < t->bbox_low[i]) {
      dists[i] = (q[i] - t->bbox_low[i]) * (q[i] - t->bbox_low[i]);
      distsq += dists[i];
    }
    if (q[i] > t->bbox_high[i]) {
      dists[i] = (q[i] - t->bbox_high[i]) * (q[i] - t->bbox_high[i]);
      distsq += dists[i];
    }
  }
  if (t->m_size == 0 || t->n_nodes == 0) return 0;
  so_search_level(&s, 0, distsq, dists);
  return s.n;
}

static int so_depth_rec(const so_kdtree *t, int64_t ni) {
  if (t->nodes[ni].is_leaf) return 1;
  int a = so_depth_rec(t, t->nodes[ni].child1), b = so_depth_rec(t, t->nodes[ni].child2);
  return 1 + (a > b ? a : b);
}
int so_kdtree_depth(const so_kdtree *t) { return t->n_nodes ? so_depth_rec(t, 0) : 0; }

/* ------------------------------------------------------------------ chaining */

typedef struct {
  so_anchor *v;
  size_t n, cap;
} so_avec;
static void so_avec_push(so_avec *a, so_anchor x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 64;
    a->v = (so_anchor *)realloc(a->v, a->cap * sizeof(so_anchor));
  }
  a->v[a->n++] = x;
}

void so_chains_clear(so_chains *c) {
  for (size_t i = 0; i < c->n; ++i) free(c->v[i].anchors);
  free(c->v);
  c->v = NULL;
  c->n = c->cap = 0;
}
static void so_chains_push(so_chains *c, so_chain x) {
  if (c->n == c->cap) {
    c->cap = c->cap ? c->cap * 2 : 16;
    c->v = (so_chain *)realloc(c->v, c->cap * sizeof(so_chain));
  }
  c->v[c->n++] = x;
}

/* spatial_index.h:22-25 SignalAnchor::operator< */
static int so_anchor_cmp(const void *pa, const void *pb) {
  const so_anchor *a = (const so_anchor *)pa, *b = (const so_anchor *)pb;
  if (a->target_position != b->target_position)
    return a->target_position < b->target_position ? -1 : 1;
  if (a->query_position != b->query_position)
    return a->query_position < b->query_position ? -1 : 1;
  if (a->distance < b->distance) return -1;
  if (a->distance > b->distance) return 1;
  return 0;
}

/* spatial_index.cc:326-356, 403-408.  Every sort key is FLT_MAX (+= of small diffs onto
 * FLT_MAX), so compare1 orders by position ascending; then the step filter. */
size_t so_seed_positions(size_t n_query, int dim, int step, uint32_t *pos) {
  if (n_query < (size_t)dim) return 0;
  size_t nwin = n_query - (size_t)dim + 1;
  uint32_t previous_position = 0;
  uint32_t num_positions = 0;
  size_t k = 0;
  for (uint32_t pi = 0; pi < nwin; ++pi) {
    uint32_t position = pi;
    if (position < previous_position + (uint32_t)step &&
        position + (uint32_t)step > previous_position)
      continue;
    pos[k++] = position;
    ++num_positions;
    if (num_positions >= nwin / (size_t)step) break;
    previous_position = position;
  }
  return k;
}

typedef struct {
  float score;
  size_t index;
} so_end;
/* spatial_index.cc:11-20 compare: score desc, then index desc */
static int so_end_cmp(const void *pa, const void *pb) {
  const so_end *a = (const so_end *)pa, *b = (const so_end *)pb;
  if (a->score > b->score) return -1;
  if (a->score == b->score) {
    if (a->index > b->index) return -1;
    if (a->index < b->index) return 1;
    return 0;
  }
  return 1;
}

/* spatial_index.cc:165-220 */
static void so_traceback(int min_num_anchors, int direction, size_t end_idx, uint32_t ref,
                         const float *scores, const size_t *pred, const so_anchor *anchors,
                         uint8_t *used, so_chains *chains) {
  if (used[end_idx]) return;
  so_avec av = {0, 0, 0};
  int stop_at_used = 0;
  size_t cur = end_idx;
  so_avec_push(&av, anchors[cur]);
  if (pred[cur] != cur && used[pred[cur]]) stop_at_used = 1;
  used[cur] = 1;
  uint32_t num = 1;
  while (pred[cur] != cur && !used[pred[cur]]) {
    cur = pred[cur];
    so_avec_push(&av, anchors[cur]);
    if (pred[cur] != cur && used[pred[cur]]) stop_at_used = 1;
    used[cur] = 1;
    ++num;
  }
  if (num >= (uint32_t)min_num_anchors) {
    float adjusted = scores[end_idx];
    if (stop_at_used) adjusted -= scores[pred[cur]];
    so_chain c;
    c.score = adjusted;
    c.ref = ref;
    c.start = anchors[cur].target_position;
    c.end = anchors[end_idx].target_position;
    c.n_anchors = num;
    c.mapq = 0;
    c.direction = direction;
    c.anchors = av.v;
    so_chains_push(chains, c);
  } else {
    free(av.v);
  }
}

/* spatial_index.h:37-44 operator>, as a "descending" comparator for qsort */
static int so_chain_cmp_desc(const void *pa, const void *pb) {
  const so_chain *a = (const so_chain *)pa, *b = (const so_chain *)pb;
  if (a->score != b->score) return a->score > b->score ? -1 : 1;
  if (a->n_anchors != b->n_anchors) return a->n_anchors > b->n_anchors ? -1 : 1;
  if (a->direction != b->direction) return a->direction > b->direction ? -1 : 1;
  if (a->ref != b->ref) return a->ref > b->ref ? -1 : 1;
  if (a->start != b->start) return a->start > b->start ? -1 : 1;
  if (a->end != b->end) return a->end > b->end ? -1 : 1;
  return 0;
}

/* spatial_index.cc:222-253 */
static void so_primary_chains(so_chains *chains) {
  qsort(chains->v, chains->n, sizeof(so_chain), so_chain_cmp_desc);
  so_chains prim = {0, 0, 0};
  uint8_t *keep = (uint8_t *)calloc(chains->n, 1);
  so_chains_push(&prim, chains->v[0]);
  keep[0] = 1;
  for (uint32_t ci = 1; ci < chains->n; ++ci) {
    int is_primary = 1;
    if (chains->v[ci].score < prim.v[prim.n - 1].score / 3) {
      break;
    } else {
      for (uint32_t pi = 0; pi < prim.n; ++pi) {
        if (chains->v[ci].ref == prim.v[pi].ref) {
          uint32_t mx = chains->v[ci].start > prim.v[pi].start ? chains->v[ci].start : prim.v[pi].start;
          uint32_t mn = chains->v[ci].end < prim.v[pi].end ? chains->v[ci].end : prim.v[pi].end;
          if (mx > mn) {
          } else {
            is_primary = 0;
            break;
          }
        }
      }
    }
    if (is_primary) {
      so_chains_push(&prim, chains->v[ci]);
      keep[ci] = 1;
    }
  }
  for (size_t i = 0; i < chains->n; ++i)
    if (!keep[i]) free(chains->v[i].anchors);
  free(keep);
  free(chains->v);
  *chains = prim;
}

/* spatial_index.cc:255-274 */
static void so_compute_mapq(so_chains *chains) {
  if (chains->n == 1) {
    chains->v[0].mapq = 60;
    return;
  }
  int mapq = 40 * (1 - chains->v[1].score / chains->v[0].score);
  if (mapq > 60) mapq = 60;
  if (mapq < 0) mapq = 0;
  chains->v[0].mapq = (uint8_t)mapq;
}

/* spatial_index.cc:419-576 on sorted per-(strand,ref) anchor groups */
void so_chain_sorted_anchors(const so_anchor *const *groups, const size_t *counts,
                             size_t n_refs, int dim, float radius, so_chains *chains) {
  const int max_gap_length = 2000, max_target_gap_length = 5000, chaining_band_length = 5000,
            max_num_skips = 25, min_num_anchors = 2, num_best_chains = 3;
  const float min_chaining_score = 10;
  float max_chaining_score = 0;
  for (size_t ref = 0; ref < n_refs; ++ref) {
    for (int di = 0; di < 2; ++di) {
      const so_anchor *A = groups[(size_t)di * n_refs + ref];
      size_t n = counts[(size_t)di * n_refs + ref];
      float *scores = (float *)malloc((n + 1) * sizeof(float));
      size_t *pred = (size_t *)malloc((n + 1) * sizeof(size_t));
      uint8_t *used = (uint8_t *)calloc(n + 1, 1);
      so_end *ends = (so_end *)malloc((n + 1) * sizeof(so_end));
      size_t n_ends = 0;
      for (size_t ai = 0; ai < n; ++ai) {
        /* spatial_index.cc:438-444: double expression narrowed to float */
        float distance_coefficient = (float)(1 - 0.2 * A[ai].distance / radius);
        scores[ai] = distance_coefficient * dim; /* float * int */
        pred[ai] = ai;
        int32_t ct = (int32_t)A[ai].target_position;
        int32_t cq = (int32_t)A[ai].query_position;
        int32_t start_ai = 0;
        if (ai > (size_t)chaining_band_length) start_ai = (int32_t)(ai - (size_t)chaining_band_length);
        int32_t pj = (int32_t)ai - 1;
        int32_t num_skips = 0;
        for (; pj >= start_ai; --pj) {
          int32_t pt = (int32_t)A[pj].target_position;
          int32_t pq = (int32_t)A[pj].query_position;
          if (pq == cq) continue;
          if (pt == ct) continue;
          if (pt + max_target_gap_length < ct) break;
          int32_t td = ct - pt;
          int32_t qd = cq - pq;
          float cur = 0;
          if (qd < 0) {
            continue;
          } else {
            int32_t m = td < qd ? td : qd;
            if (dim < m) m = dim;
            float matching = m * distance_coefficient;
            int gap_length = abs(td - qd);
            float gap_scale = td > 0 ? (float)qd / td : 1;
            if (gap_length < max_gap_length && gap_scale < 5 && gap_scale > 0.75) {
              cur = scores[pj] + matching;
            }
          }
          if (cur > scores[ai]) {
            scores[ai] = cur;
            pred[ai] = (size_t)pj;
            --num_skips;
          } else {
            ++num_skips;
            if (num_skips > max_num_skips) break;
          }
        }
        if (scores[ai] > max_chaining_score) max_chaining_score = scores[ai];
        if (scores[ai] >= min_chaining_score && scores[ai] > max_chaining_score / 2) {
          ends[n_ends].score = scores[ai];
          ends[n_ends].index = ai;
          n_ends++;
        }
      }
      qsort(ends, n_ends, sizeof(so_end), so_end_cmp);
      for (size_t k = 0; k < n_ends && k < (size_t)num_best_chains; ++k) {
        so_traceback(min_num_anchors, di == 0 ? 1 : 0, ends[k].index, (uint32_t)ref, scores,
                     pred, A, used, chains);
        if (scores[ends[k].index] < max_chaining_score / 2) break;
      }
      free(scores);
      free(pred);
      free(used);
      free(ends);
    }
  }
  if (chains->n > 0) {
    so_primary_chains(chains);
    so_compute_mapq(chains);
  }
}

/* spatial_index.cc:276-577 */
void so_generate_chains(const so_kdtree *t, const uint64_t *positions, const float *query,
                        size_t n_query, uint32_t query_start_offset, int step, float radius,
                        size_t n_refs, so_chains *chains, so_chain_stats *st) {
  so_generate_chains_ex(t, positions, query, n_query, query_start_offset, step, radius, n_refs, chains, st,
                        5000 /* num_nearest_points, spatial_index.cc:290 */);
}

void so_generate_chains_ex(const so_kdtree *t, const uint64_t *positions, const float *query,
                           size_t n_query, uint32_t query_start_offset, int step, float radius,
                           size_t n_refs, so_chains *chains, so_chain_stats *st,
                           uint64_t num_nearest_points) {
  const int dim = t->dim;
  so_avec *groups = (so_avec *)calloc(2 * n_refs, sizeof(so_avec)); /* [strand][ref] */
  /* carry anchors of previous chains, spatial_index.cc:301-322 */
  so_chains prev = *chains;
  chains->v = NULL;
  chains->n = chains->cap = 0;
  for (size_t ci = 0; ci < prev.n; ++ci) {
    int strand = prev.v[ci].direction == 1 ? 0 : 1;
    for (uint32_t ai = 0; ai < prev.v[ci].n_anchors; ++ai)
      so_avec_push(&groups[(size_t)strand * n_refs + prev.v[ci].ref], prev.v[ci].anchors[ai]);
  }
  so_chains_clear(&prev);
  uint32_t *seeds = (uint32_t *)malloc((n_query + 1) * sizeof(uint32_t));
  size_t n_seeds = so_seed_positions(n_query, dim, step, seeds);
  uint64_t cap = 1 << 16;
  uint64_t *hidx = (uint64_t *)malloc(cap * 8);
  float *hdist = (float *)malloc(cap * 4);
  for (size_t si = 0; si < n_seeds; ++si) {
    uint32_t position = seeds[si];
    uint64_t nh = so_radius_search(t, query + position, radius, hidx, hdist, cap,
                                   st ? &st->search : NULL);
    if (nh > cap) { /* keep every hit the reference would see (only 5000 are used) */
      cap = nh;
      hidx = (uint64_t *)realloc(hidx, cap * 8);
      hdist = (float *)realloc(hdist, cap * 4);
      so_search_stats dummy = {0, 0, 0};
      so_radius_search(t, query + position, radius, hidx, hdist, cap, &dummy);
    }
    if (st) {
      st->queries++;
      st->hits += nh;
      st->anchors += nh < num_nearest_points ? nh : num_nearest_points;
    }
    for (uint64_t ai = 0; ai < nh && ai < num_nearest_points; ai++) { /* :371-402 */
      uint64_t p = positions[hidx[ai]];
      uint32_t tref = (uint32_t)(p >> 33), tpos = (uint32_t)(p >> 1);
      int strand = (p & 1) == 0 ? 0 : 1;
      so_anchor a = {tpos, position + query_start_offset, hdist[ai]};
      so_avec_push(&groups[(size_t)strand * n_refs + tref], a);
    }
  }
  free(seeds);
  free(hidx);
  free(hdist);
  const so_anchor **gp = (const so_anchor **)malloc(2 * n_refs * sizeof(void *));
  size_t *gc = (size_t *)malloc(2 * n_refs * sizeof(size_t));
  for (size_t g = 0; g < 2 * n_refs; ++g) { /* :411-417 */
    if (groups[g].n) qsort(groups[g].v, groups[g].n, sizeof(so_anchor), so_anchor_cmp);
    gp[g] = groups[g].v;
    gc[g] = groups[g].n;
  }
  so_chain_sorted_anchors(gp, gc, n_refs, dim, radius, chains);
  for (size_t g = 0; g < 2 * n_refs; ++g) free(groups[g].v);
  free(groups);
  free(gp);
  free(gc);
}

/* ------------------------------------------------------------ streaming map */

void so_default_params(so_map_params *p) { /* sigmap.cc:1380-1419 */
  p->search_radius = 0.08f;
  p->step_size = 2;
  p->max_num_chunks = 30;
  p->stop_min_anchors = 10;
  p->output_min_anchors = 10;
  p->stop_ratio = 1.4f;
  p->output_ratio = 1.2f;
  p->stop_mean_ratio = 5;
  p->output_mean_ratio = 5;
}

/* sigmap.cc:631-866 for one read */
void so_map_read(const so_kdtree *t, const uint64_t *positions, const float *signal_pa,
                 size_t signal_length, size_t n_refs, const uint32_t *ref_lens,
                 const so_map_params *p, so_mapping *out) {
  const uint32_t bp_per_sec = 450, sample_rate = 4000, chunk_size = 4000;
  memset(out, 0, sizeof(*out));
  size_t num_chunks = signal_length / chunk_size;
  so_chains chains = {0, 0, 0};
  uint32_t num_events = 0;
  uint32_t chunk_index = 0;
  float *feat = (float *)malloc((chunk_size + 1) * sizeof(float));
  for (chunk_index = 0; chunk_index < num_chunks && chunk_index < (uint32_t)p->max_num_chunks;
       ++chunk_index) {
    size_t s0 = (size_t)chunk_size * chunk_index;
    size_t s1 = (size_t)chunk_size * (chunk_index + 1);
    if (s1 > signal_length) s1 = signal_length;
    size_t nf = so_generate_events(signal_pa + s0, s1 - s0, feat, NULL);
    if (nf > 50) {
      so_generate_chains(t, positions, feat, nf, num_events, p->step_size, p->search_radius,
                         n_refs, &chains, &out->stats);
      num_events += (uint32_t)nf;
      if (chains.n >= 2) {
        if (chains.v[0].score / chains.v[1].score >= p->stop_ratio) break;
        float mean = 0;
        for (uint32_t ci = 0; ci < chains.n; ++ci) mean += chains.v[ci].score;
        mean /= chains.n;
        if (chains.v[0].score >= p->stop_mean_ratio * mean) break;
      } else if (chains.n == 1 && chains.v[0].n_anchors >= (uint32_t)p->stop_min_anchors) {
        break;
      }
    }
  }
  if (chunk_index > 0 && (chunk_index == num_chunks || chunk_index == (uint32_t)p->max_num_chunks))
    --chunk_index;
  float read_position_scale = ((float)(chunk_index + 1) * chunk_size / num_events) /
                              ((float)sample_rate / bp_per_sec);
  float mean_chain_score = 0;
  for (uint32_t ci = 0; ci < chains.n; ++ci) mean_chain_score += chains.v[ci].score;
  mean_chain_score /= chains.n;
  out->read_length = (uint32_t)signal_length;
  out->chunks_used = chunk_index + 1;
  out->n_chains = (uint32_t)chains.n;
  if (chains.n >= 1) { /* tag values, sigmap.cc:714-745 == :830-859 */
    const so_chain *c0 = &chains.v[0];
    float at = 0, aq = 0, ad = 0;
    for (size_t ai = 0; ai < c0->n_anchors; ++ai) {
      ad += c0->anchors[ai].distance;
      if (ai < (size_t)c0->n_anchors - 1) {
        at += c0->anchors[ai].target_position - c0->anchors[ai + 1].target_position;
        aq += c0->anchors[ai].query_position - c0->anchors[ai + 1].query_position;
      }
    }
    ad /= c0->n_anchors;
    at /= c0->n_anchors;
    aq /= c0->n_anchors;
    out->cm = c0->n_anchors;
    out->s1 = c0->score;
    out->s2 = chains.n > 1 ? chains.v[1].score : 0;
    out->sm = mean_chain_score;
    out->ad = ad;
    out->at = at;
    out->aq = aq;
  }
  if ((chains.n >= 2 && (chains.v[0].score / chains.v[1].score >= p->output_ratio ||
                         chains.v[0].score >= p->output_mean_ratio * mean_chain_score)) ||
      (chains.n == 1 && chains.v[0].n_anchors >= (uint32_t)p->output_min_anchors)) {
    const so_chain *c0 = &chains.v[0];
    out->mapped = 1;
    out->ref = c0->ref;
    out->read_start = (uint32_t)(read_position_scale * c0->anchors[c0->n_anchors - 1].query_position);
    out->read_end = (uint32_t)(read_position_scale * c0->anchors[0].query_position);
    out->frag_start = c0->direction == 1 ? c0->start : (uint32_t)(ref_lens[c0->ref] + 1 - c0->end);
    out->frag_length = (uint32_t)(c0->end - c0->start + 1);
    out->mapq = c0->mapq;
    out->direction = c0->direction == 1 ? 1 : 0;
  } else {
    out->mapped = 0;
    out->ref = 0;
    out->mapq = 61;
  }
  free(feat);
  so_chains_clear(&chains);
}

/* sigmap.cc:731-745 tags + output_tools.h:200-210 (unmapped), :336-354 (mapped).
 * std::to_string(float) == "%f".  The wall-clock mt:f: tag is omitted. */
int so_format_paf(const so_mapping *m, const char *read_name, const char *ref_name,
                  uint32_t ref_len, char *buf, size_t cap) {
  char tags[512];
  int k = snprintf(tags, sizeof tags, "ci:i:%u\tsl:i:%u", m->chunks_used, m->read_length);
  if (m->n_chains >= 1) {
    snprintf(tags + k, sizeof tags - (size_t)k,
             "\tcm:i:%u\tnc:i:%u\ts1:f:%f\ts2:f:%f\tsm:f:%f\tad:f:%f\tat:f:%f\taq:f:%f", m->cm,
             m->n_chains, m->s1, m->s2, m->sm, m->ad, m->at, m->aq);
  }
  if (m->mapped && m->mapq <= 60) {
    return snprintf(buf, cap, "%s\t%u\t%u\t%u\t%s\t%s\t%u\t%u\t%u\t%u\t%u\t%u\t%s\n", read_name,
                    m->read_length, m->read_start, m->read_end, m->direction ? "+" : "-",
                    ref_name, ref_len, m->frag_start, m->frag_start + m->frag_length,
                    m->read_length, m->frag_length, (unsigned)m->mapq, tags);
  }
  return snprintf(buf, cap, "%s\t%u\t*\t*\t*\t*\t*\t*\t*\t*\t*\t%u\t%s\n", read_name,
                  m->read_length, (unsigned)m->mapq, tags);
}

/* ------------------------------------------------------------------ sDTW */

static float so_min3(float a, float b, float c) { /* std::min({a,b,c}) */
  float m = a;
  if (b < m) m = b;
  if (c < m) m = c;
  return m;
}

/* sigmap.cc:1259-1291 */
float so_sdtw(const float *target, size_t tlen, const float *query, size_t qlen, int64_t *end) {
  float best = FLT_MAX;
  *end = -1;
  float *prev = (float *)malloc((qlen + 1) * sizeof(float));
  float *cur = (float *)malloc((qlen + 1) * sizeof(float));
  for (size_t q = 0; q <= qlen; ++q) prev[q] = FLT_MAX;
  prev[0] = 0;
  for (size_t tp = 1; tp <= tlen; ++tp) {
    cur[0] = 0;
    for (size_t qp = 1; qp <= qlen; ++qp) {
      float cost = fabsf(target[tp - 1] - query[qp - 1]);
      cur[qp] = cost + so_min3(prev[qp - 1], prev[qp], cur[qp - 1]);
    }
    if (cur[qlen] < best) {
      best = cur[qlen];
      *end = (int64_t)tp;
    }
    float *tmp = cur;
    cur = prev;
    prev = tmp;
  }
  free(prev);
  free(cur);
  return best;
}

/* Same recurrence restricted to per-row column windows [lo[t], hi[t]] (1-based, inclusive);
 * cells outside the window read as FLT_MAX (FLT_MAX + cost stays FLT_MAX in f32 for the
 * small costs involved; we clamp explicitly to keep it so). */
float so_sdtw_banded(const float *target, size_t tlen, const float *query, size_t qlen,
                     const uint32_t *lo, const uint32_t *hi, int64_t *end) {
  float best = FLT_MAX;
  *end = -1;
  float *prev = (float *)malloc((qlen + 2) * sizeof(float));
  float *cur = (float *)malloc((qlen + 2) * sizeof(float));
  for (size_t q = 0; q <= qlen; ++q) prev[q] = FLT_MAX;
  prev[0] = 0;
  for (size_t tp = 1; tp <= tlen; ++tp) {
    for (size_t q = 0; q <= qlen; ++q) cur[q] = FLT_MAX;
    cur[0] = 0;
    uint32_t a = lo[tp - 1] < 1 ? 1 : lo[tp - 1];
    uint32_t b = hi[tp - 1] > qlen ? (uint32_t)qlen : hi[tp - 1];
    for (size_t qp = a; qp <= b; ++qp) {
      float cost = fabsf(target[tp - 1] - query[qp - 1]);
      float m = so_min3(prev[qp - 1], prev[qp], cur[qp - 1]);
      cur[qp] = (m == FLT_MAX) ? FLT_MAX : cost + m;
    }
    if (cur[qlen] < best) {
      best = cur[qlen];
      *end = (int64_t)tp;
    }
    float *tmp = cur;
    cur = prev;
    prev = tmp;
  }
  free(prev);
  free(cur);
  return best;
}


/* The reference's read loop (sigmap.cc:618-632: one OpenMP task per read over a shared read-only index):
 * n reads, read i = signal_pa[offsets[i] .. offsets[i+1]), on n_threads threads.  Used by bench.py's
 * cpu_baseline leg and nothing else.  Compiled with -fopenmp; without it the loop runs serially. */
void so_map_reads(const so_kdtree *t, const uint64_t *positions, const float *signal_pa, const uint64_t *offsets,
                  size_t n, size_t n_refs, const uint32_t *ref_lens, const so_map_params *p, so_mapping *out,
                  int n_threads) {
  long i;
  /* Per-read scratch is malloc'd and freed inside so_map_read (the reference keeps its scratch vectors
   * per thread, sigmap.cc:625-627).  Keep freed memory in the threads' arenas instead of handing it back
   * to the kernel after every read: with dozens of threads the mmap / munmap / page-fault traffic on the
   * process's address space otherwise serialises them (measured: 32 threads faster than 256). */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 64 << 20);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (i = 0; i < (long)n; ++i)
    so_map_read(t, positions, signal_pa + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), n_refs, ref_lens, p,
                &out[i]);
}

/* ===================== windowed DTW with path (D5) =====================
 * fast_dtw.cc::DTW(), fast_dtw.cc:74-140, over a window whose row i holds the query columns
 * lo[i]..hi[i] (1-based, inclusive, clipped to [1, qlen]) -- the shape of the rows expand_window()
 * produces (fast_dtw.cc:11-49: sorted by target, then query) when the window is a band.
 * Rows: previous_row = {0, FLT_MAX, ...} before row 0 and current_row[0] = 0 in every row (free start
 * on the target, :94-99); four candidates in this order (:113)
 *     { prev[q-1] + cost, cur[q-1] + cost, cur[q-1] + skip_query_cost(2), prev[q] + skip_target_cost(2) }
 * with std::min_element's FIRST minimum, whose index 0..3 is the cell's path code (:127-129); a row
 * is a candidate end iff its LAST cell is the last query column and its value is < the best so far
 * (:131-134).  Cells outside the window stay FLT_MAX (the row buffers are re-filled with it, :136).
 * generate_path (:51-72) walks back from the end row's last cell until it reaches query column 0
 * (shifts by code: {-1,-1},{0,-1},{0,-1},{-1,0} in (target, query)), includes that cell, and reverses.
 * Returns the distance; *end_target = 0-based target coordinate of the end row (:138), -1 if no row
 * ends in the last column with a finite value (the reference would index window[-1] there);
 * path_codes[0..*path_len) = the codes from the path's first cell to its last (cap tlen + qlen);
 * *start_target = 0-based target coordinate of the first cell.
 * PARITY UNPINNED: fast_dtw.cc includes signal_batch.h -> <slow5/slow5.h>, absent from this image. */
float so_dtw_window(const float *target, size_t tlen, const float *query, size_t qlen, const uint32_t *lo,
                    const uint32_t *hi, int64_t *end_target, uint8_t *path_codes, size_t *path_len,
                    int64_t *start_target) {
  float best = FLT_MAX;
  int64_t end_row = -1;
  *end_target = -1;
  if (path_len) *path_len = 0;
  if (start_target) *start_target = -1;
  if (tlen == 0 || qlen == 0) return best;
  size_t *row_off = (size_t *)malloc((tlen + 1) * sizeof(size_t));
  uint32_t *ra = (uint32_t *)malloc(tlen * sizeof(uint32_t)), *rb = (uint32_t *)malloc(tlen * sizeof(uint32_t));
  size_t cells = 0;
  for (size_t i = 0; i < tlen; ++i) {
    ra[i] = lo[i] < 1 ? 1 : lo[i];
    rb[i] = hi[i] > qlen ? (uint32_t)qlen : hi[i];
    row_off[i] = cells;
    if (rb[i] >= ra[i]) cells += rb[i] - ra[i] + 1;
  }
  row_off[tlen] = cells;
  uint8_t *code = (uint8_t *)malloc(cells ? cells : 1);
  float *prev = (float *)malloc((qlen + 1) * sizeof(float));
  float *cur = (float *)malloc((qlen + 1) * sizeof(float));
  for (size_t q = 0; q <= qlen; ++q) prev[q] = cur[q] = FLT_MAX;
  prev[0] = 0;
  for (size_t i = 0; i < tlen; ++i) {
    cur[0] = 0;
    if (rb[i] >= ra[i]) {
      for (size_t qp = ra[i]; qp <= rb[i]; ++qp) {
        const float cost = fabsf(target[i] - query[qp - 1]);
        const float cand[4] = {prev[qp - 1] + cost, cur[qp - 1] + cost, cur[qp - 1] + 2.0f, prev[qp] + 2.0f};
        int arg = 0;
        for (int k = 1; k < 4; ++k)
          if (cand[k] < cand[arg]) arg = k; /* std::min_element: first of the smallest */
        cur[qp] = cand[arg];
        code[row_off[i] + (qp - ra[i])] = (uint8_t)arg;
      }
      if (rb[i] == qlen && cur[qlen] < best) { /* query_position of the row's last cell == query_length */
        best = cur[qlen];
        end_row = (int64_t)i;
      }
    }
    float *tmp = cur;
    cur = prev;
    prev = tmp;
    for (size_t q = 0; q <= qlen; ++q) cur[q] = FLT_MAX;
  }
  if (end_row >= 0) {
    /* generate_path, fast_dtw.cc:51-72 */
    size_t n = 0, t = (size_t)end_row, q = qlen; /* q 1-based: query_coord = q - 1 */
    int ok = 1;
    while (q != 1) {
      const uint8_t c = code[row_off[t] + (q - ra[t])];
      if (path_codes) path_codes[n] = c;
      ++n;
      if (c == 0) {
        if (t == 0) { ok = 0; break; }
        --t; --q;
      } else if (c == 3) {
        if (t == 0) { ok = 0; break; }
        --t;
      } else {
        --q;
      }
      if (!(rb[t] >= ra[t] && q >= ra[t] && q <= rb[t])) { ok = 0; break; } /* the reference's map lookup would miss */
    }
    if (ok) {
      if (path_codes) path_codes[n] = code[row_off[t] + (q - ra[t])];
      ++n;
      if (path_codes) /* std::reverse */
        for (size_t a = 0, b = n - 1; a < b; ++a, --b) {
          const uint8_t x = path_codes[a];
          path_codes[a] = path_codes[b];
          path_codes[b] = x;
        }
      if (path_len) *path_len = n;
      if (start_target) *start_target = (int64_t)t;
    }
    *end_target = end_row; /* window[end][0].target_coord */
  }
  free(row_off); free(ra); free(rb); free(code); free(prev); free(cur);
  return best;
}

/* print_alignment, fast_dtw.cc:160-209: per-base CIGAR "(..)(..)" from the path codes (0 one-to-one,
 * 1 one base to several signals, 2 skip one signal, 3 skip one base; letters "MMID" indexed by the
 * running flag).  Quirks kept: the run in progress when the path ends is never written, and a final
 * code 0 / 3 closes the previous base without opening its own.  Returns the length written (without
 * NUL), or the length needed when it exceeds cap. */
size_t so_print_alignment(const uint8_t *codes, size_t n, char *out, size_t cap) {
  static const char flags[] = "MMID";
  size_t len = 0;
#define SO_PUT(ch) do { if (len + 1 < cap) out[len] = (ch); ++len; } while (0)
  if (n == 0) {
    if (cap) out[0] = 0;
    return 0;
  }
  size_t same = 1;
  int last;
  SO_PUT('(');
  if (codes[0] == 3) {
    SO_PUT('1'); SO_PUT('D'); SO_PUT(')'); SO_PUT('(');
    last = 3;
  } else {
    last = codes[0] == 0 ? 1 : 2;
  }
  char num[24];
  for (size_t i = 1; i < n; ++i) {
    const int flag = codes[i];
    if (flag == 1 || flag == 2) {
      if (last == flag) {
        ++same;
      } else {
        const int k = snprintf(num, sizeof num, "%zu", same);
        for (int j = 0; j < k; ++j) SO_PUT(num[j]);
        SO_PUT(flags[last]);
        same = 1;
        last = flag;
      }
    } else {
      const int k = snprintf(num, sizeof num, "%zu", same);
      for (int j = 0; j < k; ++j) SO_PUT(num[j]);
      SO_PUT(flags[last]);
      if (flag == 0) last = 1;
      else if (flag == 3) last = 3;
      if (i != n - 1) {
        SO_PUT(')'); SO_PUT('(');
        same = 1;
      }
    }
  }
  SO_PUT(')');
#undef SO_PUT
  if (cap) out[len < cap ? len : cap - 1] = 0;
  return len;
}

/* ===================== CWT feature path (D1-D3) ===================== */
static int so_cmp_float(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

/* sigmap.cc:1100-1129 (std::nth_element at n/2 == the value at index n/2 of the sorted array) */
float so_mad_normalize(const float *x, size_t n, float *out) {
  float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
  memcpy(tmp, x, n * sizeof(float));
  qsort(tmp, n, sizeof(float), so_cmp_float);
  const float median = tmp[n / 2];
  for (size_t i = 0; i < n; ++i) tmp[i] = fabsf(x[i] - median); /* taken on the unsorted values: same multiset */
  qsort(tmp, n, sizeof(float), so_cmp_float);
  const float mad = tmp[n / 2];
  for (size_t i = 0; i < n; ++i) out[i] = (x[i] - median) / mad;
  free(tmp);
  return mad;
}

/* cwt.cc:1633-1709 (npad), :1762-1841 (cwavelet), :1595-1630 (DoG daughter, m = 2: sign +1, real) */
void so_cwt_dog(const float *x, size_t n, float scale0, float *out) {
  if (n == 0) return;
  const float t1 = 0.499999f + logf((float)n) / logf(2.0f);
  const int ibase2 = 1 + (int)t1;
  const size_t npad = (size_t)1 << ibase2;
  const double pi = 4.0 * atan(1.0);
  const double s = (double)scale0, dt = 1.0;
  /* sqrt(2 pi s / dt) * sqrt(1 / Gamma(m + 1/2)), Gamma(2.5) = 3 sqrt(pi) / 4 */
  const double norm = sqrt(2.0 * pi * s / dt) * sqrt(1.0 / (0.75 * sqrt(pi)));
  /* daughter on the wavenumber grid kwave[k] = k * 2 pi / (npad dt) (symmetric in k) */
  double *D = (double *)malloc((npad / 2 + 1) * sizeof(double));
  for (size_t k = 0; k <= npad / 2; ++k) {
    const double sk = s * ((double)k * 2.0 * pi / ((double)npad * dt));
    D[k] = norm * sk * sk * exp(-0.5 * sk * sk);
  }
  /* impulse response.  For s >= 4 the spectrum is ~0 at the Nyquist wavenumber and h decays like
   * exp(-(j/s)^2/2): beyond |j| = 9 s it is below 1e-17 of its peak.  For smaller scales the
   * sampled spectrum is cut at Nyquist with a kink and h has 1/j^2 tails: keep every lag. */
  size_t L = s >= 4.0 ? (size_t)(9.0 * s) + 4 : npad / 2 - 1;
  if (L > npad / 2 - 1) L = npad / 2 - 1;
  if (L > n - 1) L = n - 1;
  double *h = (double *)malloc((L + 1) * sizeof(double));
  for (size_t j = 0; j <= L; ++j) {
    double acc = D[0] + ((npad / 2) * j % 2 ? -D[npad / 2] : D[npad / 2]);
    for (size_t k = 1; k < npad / 2; ++k) acc += 2.0 * D[k] * cos(2.0 * pi * (double)((k * j) % npad) / (double)npad);
    h[j] = acc / (double)npad;
  }
  float ymean = 0.0f; /* cwt_type is float: sequential f32 sum (cwt.cc:1786-1790) */
  for (size_t i = 0; i < n; ++i) ymean += x[i];
  ymean /= (float)n;
  for (size_t i = 0; i < n; ++i) {
    double acc = 0.0;
    const size_t lo = i > L ? i - L : 0, hi = i + L < n ? i + L : n - 1;
    for (size_t m = lo; m <= hi; ++m) acc += (double)(x[m] - ymean) * h[m > i ? m - i : i - m];
    out[i] = (float)acc;
  }
  free(D);
  free(h);
}

/* sigmap.cc:1176-1196 */
size_t so_generate_peaks(const float *x, size_t n, float selective, float *peaks, uint32_t *positions) {
  size_t np = 0;
  if (n < 3) return 0;
  float previous_valley = x[0], previous_peak = x[0];
  for (size_t i = 1; i + 1 < n; ++i) {
    if (x[i] > x[i - 1] && x[i] >= x[i + 1] && x[i] >= previous_valley + selective) {
      peaks[np] = x[i];
      positions[np++] = (uint32_t)i;
      previous_peak = x[i];
    } else if (x[i] < x[i - 1] && x[i] <= x[i + 1] && x[i] <= previous_peak - selective) {
      peaks[np] = x[i];
      positions[np++] = (uint32_t)i;
      previous_valley = x[i];
    }
  }
  return np;
}

/* sigmap.cc:1085-1098 */
size_t so_cwt_features(const float *x, size_t n, float scale0, float *feat, uint32_t *positions, float *cwt_out) {
  if (n < 3) return 0;
  float *a = (float *)malloc(n * sizeof(float)), *b = (float *)malloc(n * sizeof(float));
  so_mad_normalize(x, n, a);
  so_cwt_dog(a, n, scale0, b);
  if (cwt_out) memcpy(cwt_out, b, n * sizeof(float));
  /* GenerateZscoreNormalizedSignal returns SD (sigmap.cc:1131-1155); selective = SD / 4 */
  double mean = 0;
  for (size_t i = 0; i < n; ++i) mean += b[i];
  mean /= n;
  double SD = 0;
  for (size_t i = 0; i < n; ++i) SD += (b[i] - mean) * (b[i] - mean);
  SD /= (n - 1);
  SD = sqrt(SD);
  for (size_t i = 0; i < n; ++i) a[i] = (float)((b[i] - mean) / SD);
  const float sd_f = (float)SD;
  const size_t np = so_generate_peaks(a, n, sd_f / 4, feat, positions);
  free(a);
  free(b);
  return np;
}
