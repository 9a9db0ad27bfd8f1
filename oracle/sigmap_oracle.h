/*
 * sigmap_oracle.h -- CPU restatement of the Sigmap streaming-mapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (sigmap_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Plain C99, compiled with `gcc -O2 -ffp-contract=off` (the
 * reference's coordinates depend on FMA contraction, so the pin is "no contraction").
 *
 * Parity pin status (see oracle/README.md for the full statement):
 *   - so_detect_events (A2-A6), so_kdtree_build / so_radius_search (A10, N1, T9),
 *     so_level_means (pore_model.cc), so_format_paf (A18: the PAF row text, output_tools.h) and
 *     so_reverse_complement (sequence_batch.h) are pinned against the REAL reference code,
 *     compiled from /root/reference/src by oracle/Makefile into oracle/_ref/.
 *   - so_add_signal (A1), so_zscore / so_generate_events tail (A7-A8), so_generate_chains (A9, A11-A16),
 *     so_map_read (A17: the decisions and the fields of the row) and so_sdtw (D4) restate translation units
 *     (signal_batch.cc, sigmap.cc, spatial_index.cc, fast_dtw.cc) that cannot be compiled in this image
 *     without a stand-in for <slow5/slow5.h>: PARITY UNPINNED for those functions.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#ifndef SIGMAP_ORACLE_H_
#define SIGMAP_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: SignalBatch::AddSignal, signal_batch.cc:139-151 (FAST5; the samples were read into a float
 * vector first, :137-138) and :195-207 (BLOW5; int16 promoted in the expression): the same loop.
 * `scale = range / digitisation` in float; a sample survives iff 30 < (raw + offset) * scale < 200 (both
 * strict); survivors are compacted in order.  out capacity n.  Returns the number kept.
 * PARITY UNPINNED (signal_batch.cc includes <slow5/slow5.h> through signal_batch.h:6). ---- */
size_t so_add_signal(const int16_t *raw, size_t n, float digitisation, float range, float offset, float *out);

/* ---- event.h:16-21 ---- */
typedef struct {
  uint64_t start;
  uint64_t length;
  float mean;
  float stdv;
} so_event;

/* event.h:58-68 */
void so_prefix_sums(const float *x, size_t n, float *ps /*n+1*/, float *pss /*n+1*/);
/* event.h:70-115; writes n+1 values (the reference emits w + (n-2w+1) + w entries) */
void so_tstat(const float *ps, const float *pss, size_t n, size_t w, float *t /*n+1*/);
/* event.h:117-182; returns number of peaks */
size_t so_peaks(const float *t1, const float *t2, size_t n, size_t w1, size_t w2,
                float thr1, float thr2, float peak_height, uint64_t *peaks /*n*/);
/* event.h:184-224; returns number of events (0 when n_peaks < 2: reference has UB there) */
size_t so_create_events(const uint64_t *peaks, size_t n_peaks, const float *ps,
                        const float *pss, size_t n, so_event *events /*n_peaks*/);
/* event.h:226-267 with event_detection_defaults (event.h:31-37).  Returns n events. */
size_t so_detect_events(const float *x, size_t n, so_event *events /*cap n*/,
                        uint64_t *peaks_out /*cap n, may be NULL*/, size_t *n_peaks_out);

/* sigmap.cc:1131-1155 */
void so_zscore(const float *x, size_t n, float *out);
/* sigmap.cc:1048-1083.  feat/stdv capacity n.  Returns number of kept features. */
size_t so_generate_events(const float *x, size_t n, float *feat, float *stdv);

/* ---- pore_model.cc:57-81 (level means incl. the skipped-base quirk) ---- */
/* level_mean table indexed by 2-bit k-mer hash (4^k entries). seq must be NUL terminated
 * (the reference reads sequence[end_position]).  Returns signal length written. */
size_t so_level_means(const float *level_mean, int k, const char *seq, uint32_t start,
                      uint32_t end, float *out);
/* sequence_batch.h:71-83 */
void so_reverse_complement(const char *seq, size_t len, char *out /*len+1*/);
/* sigmap.cc:19-185: masks[0]=positive, masks[1]=negative, each len-kmer+1 bytes/contig */
void so_masked_positions(int kmer_size, float frequency, uint32_t n_seqs,
                         const char *const *seqs, const char *const *neg_seqs,
                         const uint32_t *lens, uint8_t **pos_masked, uint8_t **neg_masked);
/* spatial_index.cc:33-58: appends to positions/values (capacity len), returns new count */
size_t so_point_cloud_one_direction(int positive, uint32_t signal_index,
                                    const uint8_t *is_masked, const float *signal,
                                    size_t signal_length, int dim, uint64_t *positions,
                                    float *values, size_t count);

/* ---- k-d tree: nanoflann.hpp (1.3.2 as vendored) ---- */
typedef struct {
  int is_leaf;
  uint64_t left, right;   /* leaf: [left,right) into vind */
  int divfeat;            /* inner */
  float divlow, divhigh;  /* inner */
  int64_t child1, child2; /* inner: indices into nodes[] (pre-order: child1 == self+1) */
} so_node;

typedef struct {
  int dim;
  uint64_t leaf_max;
  uint64_t m_size;    /* number of k-d points = n_values - dim + 1 */
  const float *values; /* borrowed: 1-D value array; point i = values[i..i+dim) */
  uint64_t n_values;
  float *bbox_low, *bbox_high; /* root bbox, dim each */
  uint64_t *vind;              /* m_size */
  so_node *nodes;              /* pre-order */
  uint64_t n_nodes, cap_nodes;
} so_kdtree;

/* nanoflann.hpp:1196-1210 (buildIndex), :837-1004 (divideTree/middleSplit_/planeSplit) */
so_kdtree *so_kdtree_build(const float *values, uint64_t n_values, int dim, int leaf_max);
void so_kdtree_free(so_kdtree *t);
/* nanoflann.hpp:1051-1058, 1025-1033 (.si layout) */
int so_kdtree_save(const so_kdtree *t, const char *path);
so_kdtree *so_kdtree_load(const char *path, const float *values, uint64_t n_values);
/* the same tree as so_kdtree_load builds, from the .si content handed over in memory (see .c) */
so_kdtree *so_kdtree_from_preorder(const float *values, uint64_t n_values, int dim, uint64_t leaf_max,
                                   const uint32_t *vind, uint64_t m_size, const void *nodes16, uint64_t n_nodes,
                                   const float *bbox);
/* spatial_index.cc:105-123 / :132-146 (.pt layout) */
int so_points_save(const char *path, int dim, int max_leaf, const uint64_t *positions,
                   const float *values, uint64_t n);
int so_points_load(const char *path, int *dim, int *max_leaf, uint64_t **positions,
                   float **values, uint64_t *n);

typedef struct {
  uint64_t nodes, leaves, points;
} so_search_stats;
/* nanoflann.hpp:1278-1301 -> 1221-1242 -> 1006-1023 -> 1346-1410 -> 383-408 -> 249-253.
 * Hits in traversal order; returns total hits (may exceed cap; only cap are stored). */
uint64_t so_radius_search(const so_kdtree *t, const float *q, float radius,
                          uint64_t *idx, float *dist, uint64_t cap, so_search_stats *st);
int so_kdtree_depth(const so_kdtree *t);

/* ---- chaining: spatial_index.{h,cc} ---- */
typedef struct {
  uint32_t target_position, query_position;
  float distance;
} so_anchor; /* spatial_index.h:18-26 */

typedef struct {
  float score;
  uint32_t ref, start, end, n_anchors;
  uint8_t mapq;
  int direction; /* 1 = Positive, 0 = Negative (spatial_index.h:13-16) */
  so_anchor *anchors; /* end -> start order, malloc'd */
} so_chain; /* spatial_index.h:28-45 */

typedef struct {
  so_chain *v;
  size_t n, cap;
} so_chains;

void so_chains_clear(so_chains *c);

typedef struct {
  uint64_t queries, hits, anchors; /* anchors = sum min(hits, 5000) */
  so_search_stats search;
} so_chain_stats;

/* spatial_index.cc:276-577.  chains: in = previous chunk's chains, out = new chains. */
void so_generate_chains(const so_kdtree *t, const uint64_t *positions, const float *query,
                        size_t n_query, uint32_t query_start_offset, int step, float radius,
                        size_t n_refs, so_chains *chains, so_chain_stats *st);

/* Same with the reference's hard-coded num_nearest_points (5000) made a parameter, so the
 * "first N hits in traversal order" truncation can be exercised on small inputs. */
void so_generate_chains_ex(const so_kdtree *t, const uint64_t *positions, const float *query,
                           size_t n_query, uint32_t query_start_offset, int step, float radius,
                           size_t n_refs, so_chains *chains, so_chain_stats *st,
                           uint64_t num_nearest_points);

/* Lower-level pieces exposed for per-stage parity tests. */
/* seed positions (spatial_index.cc:326-356,403-408); returns count */
size_t so_seed_positions(size_t n_query, int dim, int step, uint32_t *pos /*cap n_query*/);
/* chaining DP + traceback + primary + mapq on caller-provided, already sorted anchors,
 * grouped [strand][ref] (spatial_index.cc:419-576). counts[strand*n_refs+ref]. */
void so_chain_sorted_anchors(const so_anchor *const *groups, const size_t *counts,
                             size_t n_refs, int dim, float radius, so_chains *chains);

/* ---- streaming map per read: sigmap.cc:631-866 + output_tools.h:200-210,336-354 ---- */
typedef struct {
  float search_radius;      /* 0.08 */
  int step_size;            /* 2 */
  int max_num_chunks;       /* 30 */
  int stop_min_anchors;     /* 10 */
  int output_min_anchors;   /* 10 */
  float stop_ratio;         /* 1.4 */
  float output_ratio;       /* 1.2 */
  float stop_mean_ratio;    /* 5 */
  float output_mean_ratio;  /* 5 */
} so_map_params;
void so_default_params(so_map_params *p);

typedef struct {
  int mapped;            /* 1 = mapped row, 0 = unmapped (mapq 61) */
  uint32_t ref;          /* bucket index the reference files the row under */
  uint32_t read_length, read_start, read_end, frag_start, frag_length;
  uint8_t mapq, direction;
  uint32_t chunks_used;  /* ci:i */
  uint32_t n_chains;
  /* tag values (valid iff n_chains >= 1) */
  uint32_t cm;
  float s1, s2, sm, ad, at, aq;
  so_chain_stats stats;
} so_mapping;

void so_map_read(const so_kdtree *t, const uint64_t *positions, const float *signal_pa,
                 size_t signal_length, size_t n_refs, const uint32_t *ref_lens,
                 const so_map_params *p, so_mapping *out);
/* n reads on n_threads OpenMP threads (the reference's taskloop, sigmap.cc:618-632); bench.py's CPU baseline */
void so_map_reads(const so_kdtree *t, const uint64_t *positions, const float *signal_pa, const uint64_t *offsets,
                  size_t n, size_t n_refs, const uint32_t *ref_lens, const so_map_params *p, so_mapping *out,
                  int n_threads);
/* PAF text without the leading "mt:f:" tag.  Returns bytes written (excluding NUL). */
int so_format_paf(const so_mapping *m, const char *read_name, const char *ref_name,
                  uint32_t ref_len, char *buf, size_t cap);

/* ---- sDTW ---- */
/* sigmap.cc:1259-1291 (unbanded, pointer overload). end is 1-based target row. */
float so_sdtw(const float *target, size_t tlen, const float *query, size_t qlen,
              int64_t *end);
/* Banded restriction of the same recurrence (the build's definition):
 * cell (t,q) (1-based) is inside the band iff |(t - t0) - q * slope| <= band/2 ...
 * implemented as: row t covers q in [lo(t), hi(t)] with lo/hi supplied per row.
 * Cells outside are FLT_MAX.  With lo=1,hi=qlen for all rows this equals so_sdtw. */
float so_sdtw_banded(const float *target, size_t tlen, const float *query, size_t qlen,
                     const uint32_t *lo, const uint32_t *hi, int64_t *end);
/* D5: fast_dtw.cc::DTW() (fast_dtw.cc:74-140) over per-row column windows + generate_path (:51-72);
 * print_alignment (:160-209).  See sigmap_oracle.c. */
float so_dtw_window(const float *target, size_t tlen, const float *query, size_t qlen, const uint32_t *lo,
                    const uint32_t *hi, int64_t *end_target, uint8_t *path_codes, size_t *path_len,
                    int64_t *start_target);
size_t so_print_alignment(const uint8_t *codes, size_t n, char *out, size_t cap);

/* ---- CWT feature path (dormant in the reference's CLI: CWTAlign, sigmap.cc:932) ---- */
/* sigmap.cc:1100-1129 GenerateMADNormalizedSignal: the n/2-th order statistic as "median", same
 * for the absolute deviations; out = (x - median) / MAD.  Returns MAD. */
float so_mad_normalize(const float *x, size_t n, float *out);
/* sigmap.cc:1157-1174 GenerateCWTSignal -> cwt.cc cwt_init/setCWTScales/cwt/cwavelet with the DoG
 * (m = 2) daughter of cwt.cc:1595-1630, one scale.  The reference evaluates
 *   out = IFFT( D .* FFT(x - mean(x), npad) / npad ),  D[k] = norm (s k)^2 exp(-(s k)^2 / 2)
 * with an f32 mixed-radix FFT; this restatement evaluates the same linear map as the convolution
 * with h[j] = (1/npad) sum_k D[k] cos(2 pi k j / npad) in double, so it agrees with the real cwt.cc
 * (oracle/_ref ref_cwt_dog) to the f32 FFT's rounding, not bit for bit: "tolerance parity". */
void so_cwt_dog(const float *x, size_t n, float scale0, float *out);
/* sigmap.cc:1176-1196 GeneratePeaks.  Returns the number of peaks / valleys written. */
size_t so_generate_peaks(const float *x, size_t n, float selective, float *peaks, uint32_t *positions);
/* sigmap.cc:1085-1098 GenerateFeatureSignalUsingCWT: MAD -> CWT(scale0) -> z-score -> peaks with
 * selective = SD / 4.  cwt_out (optional, n floats) receives the CWT signal. */
size_t so_cwt_features(const float *x, size_t n, float scale0, float *feat, uint32_t *positions, float *cwt_out);

#ifdef __cplusplus
}
#endif
#endif
