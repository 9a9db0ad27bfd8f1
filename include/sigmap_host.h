/*
 * sigmap_host.h -- C ABI over the host-side (CPU, no HIP) plumbing around the hot path:
 * index construction and the reference's `.pt` / `.si` files, ADC->pA conversion, and the
 * per-read PAF decision.  Implemented in C++ in sigmap_amd/host/ (the mirrors of the
 * reference's SpatialIndex / PoreModel / SequenceBatch / SignalBatch / Sigmap classes);
 * this header exists so that tests, bench.py and other-language hosts can drive them.
 * Citations: file:line under /root/reference/src.
 */
#ifndef SIGMAP_HOST_H_
#define SIGMAP_HOST_H_

#include <stddef.h>
#include <stdint.h>

#include "sigmap_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smhost_index smhost_index;

/* Sigmap::ConstructIndex, sigmap.cc:999-1046: reverse complements, high-frequency
 * (dim+k-1)-mer mask (freq > 2e-4), pore-model expected signal, per-contig per-strand
 * z-score, point cloud (+ strands of all contigs, then - strands), k-d tree.
 * level_means: 4^k table indexed by 2-bit k-mer hash.  seqs need not be NUL terminated. */
smhost_index *smhost_index_build(const float *level_means, int kmer_size, const char *const *seqs,
                                 const uint32_t *seq_lens, uint32_t n_seqs, int dim, int max_leaf,
                                 int num_threads);
/* SpatialIndex::Construct tail only (spatial_index.cc:95-97): tree over a given point cloud */
smhost_index *smhost_index_from_points(const uint64_t *positions, const float *values,
                                       uint64_t n_points, int dim, int max_leaf, int num_threads);
/* SpatialIndex::Load / Save, spatial_index.cc:103-163 (PREFIX.pt + PREFIX.si) */
smhost_index *smhost_index_load(const char *prefix);
int smhost_index_save(const smhost_index *ix, const char *prefix);
void smhost_index_free(smhost_index *ix);

typedef struct {
  int32_t dim, max_leaf, depth, pad;
  uint64_t n_points, m_size, n_nodes;
  const float *values;        /* n_points */
  const uint64_t *positions;  /* n_points */
  const uint32_t *vind;       /* m_size */
  const smhip_node *nodes;    /* n_nodes, pre-order */
  const float *bbox;          /* 2*dim: low0, high0, ... */
} smhost_index_view;
int smhost_index_get(const smhost_index *ix, smhost_index_view *out);

/* Z-scored expected reference signal of one contig strand as the index build computes it
 * (sigmap.cc:1019-1036); used as the sDTW target.  Returns length written (len-k+1). */
uint64_t smhost_reference_signal(const float *level_means, int kmer_size, const char *seq,
                                 uint32_t seq_len, int negative_strand, float *out);
/* PoreModel::GetLevelMeansAt, pore_model.cc:57-81 */
uint64_t smhost_level_means(const float *level_means, int kmer_size, const char *seq,
                            uint32_t seq_len, float *out);

/* SignalBatch::AddSignal, signal_batch.cc:139-151 / :195-207: pA = (raw+offset)*(range/
 * digitisation), keep iff 30 < pA < 200.  Returns number kept. */
uint32_t smhost_raw_to_pa(const int16_t *raw, uint32_t n_raw, float digitisation, float range,
                          float offset, float *out);

/* `sigmap -m` sizes its chunk rounds with this before the first batch (--batch-chunks is a request, the loaded index
 * and the device's free memory decide): chunks one of `contexts` device contexts may have in flight, from the fast hit
 * slots and the mean hits per query of the index census (smhip_index_census).  host/mapping_logic.h. */
uint64_t smhost_chunk_budget(uint64_t free_bytes, uint32_t keep_samples, uint32_t fast_slots, double hits_per_query,
                             uint32_t contexts);

/* Device contexts of a GPU that `sigmap -m` gives work by default (host/mapping_logic.h ContextsForReads; the loop it
   feeds replaces sigmap.cc:618-632): one when the GPU's reads fit one context's read slots, else as many as they fill. */
uint64_t smhost_contexts_for_reads(uint64_t block_reads, uint64_t lane_budget, uint64_t max_contexts);

/* Raw mode of the same function (the CLI's load phase): the read stays as ADC counts; what the reference's AddSignal
 * yields as signal_values.size() -- the FILTERED length (sigmap.cc:643: num_chunks = length / chunk_size) -- is counted,
 * and bounds[c] = the raw position at which the filtered array has c * chunk_size samples, c = 0 .. min(length /
 * chunk_size, max_chunks): chunk c of the filtered array (sigmap.cc:647-652) = the survivors of raw[bounds[c] ..
 * bounds[c + 1]).  Returns the number of bounds written (<= cap). */
uint32_t smhost_chunk_bounds(const int16_t *raw, uint32_t n_raw, float digitisation, float range, float offset,
                             uint32_t chunk_size, uint32_t max_chunks, uint32_t *bounds, uint32_t cap,
                             uint64_t *filtered_length);

/* SignalBatch::LoadAllReadSignals, signal_batch.cc:24-67: every FAST5 file (single- or multi-read,
 * signal_batch.cc:69-151), every BLOW5 file (signal_batch.cc:153-210; the binary SLOW5 layout parsed
 * directly -- record compression none / zlib / zstd, no signal compression) and every .sgr file under
 * `path` (sub-directories included; or the one file), already converted to pA and filtered.  FAST5 needs a build with HDF5
 * (smhost_have_fast5() == 1); without it a .fast5 file terminates the process like the reference's
 * ExitWithMessage. */
typedef struct smhost_signals smhost_signals;
int smhost_have_fast5(void);
smhost_signals *smhost_signals_load(const char *path);
void smhost_signals_free(smhost_signals *s);
uint64_t smhost_signals_count(const smhost_signals *s);
/* read i: id (NUL terminated), pA values, their count, and the channel's conversion attributes */
int smhost_signals_get(const smhost_signals *s, uint64_t i, const char **id, const float **pa, uint64_t *n,
                       float *digitisation, float *range, float *offset);
/* Test / generator support: one FAST5 file holding n_reads reads of raw ADC samples; multi_read = 0
 * writes the single-read layout (/Raw/Reads/Read_k + /UniqueGlobalKey/channel_id), 1 the
 * multi-read layout (/read_<id>/{Raw,channel_id}); attributes are written as doubles / strings
 * like MinKNOW does.  Returns 0 on success. */
int smhost_write_fast5(const char *path, int multi_read, uint32_t n_reads, const char *const *ids,
                       const int16_t *const *raws, const uint32_t *raw_lens, float digitisation,
                       float range, float offset);

/* Test / generator support: one BLOW5 file (file version 0.2.0, one read group, primary fields only).
 * record_press: 0 none, 1 zlib, 2 zstd (-1 is returned when this build has no zstd:
 * smhost_have_zstd() == 0); signal_press: 0 none, 1 svb-zd (StreamVByte over zig-zag deltas; slow5tools' default
 * is zlib + svb-zd).  The calibration constants are doubles in the file, like slow5_rec_t's. */
int smhost_write_blow5(const char *path, int record_press, int signal_press, uint32_t n_reads, const char *const *ids,
                       const int16_t *const *raws, const uint32_t *raw_lens, double digitisation, double range,
                       double offset);
int smhost_have_zstd(void);

typedef struct {
  float search_radius;
  int32_t step_size, max_num_chunks, stop_min_anchors, output_min_anchors;
  float stop_ratio, output_ratio, stop_mean_ratio, output_mean_ratio;
} smhost_map_params; /* CLI defaults, sigmap.cc:1380-1419 */
void smhost_default_map_params(smhost_map_params *p);

/* print_alignment, fast_dtw.cc:160-209: the per-base CIGAR "(..)(..)" of a DTW path given as its
 * codes (smhip_dtw_path).  Returns the length of the CIGAR (written NUL-terminated, truncated to cap). */
int64_t smhost_print_alignment(const uint8_t *codes, uint64_t n, char *out, uint64_t cap);

/* Read sharding of `sigmap -m --gpus N`: part `rank` of `world` maps the
 * contiguous block [*lo, *hi) of n_items reads (read i -> part i * world / n_items); rows are merged by
 * (reference index, read index), the reference's -t 1 order (sigmap.cc:201-213,223-238). */
void smhost_shard_range(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi);

/* sigmap.cc:667-687: 1 = stop feeding chunks of this read */
int smhost_should_stop(const smhip_chunk_result *r, const smhost_map_params *p);
/* sigmap.cc:690-865 + output_tools.h:200-210,336-354: PAF line (without the wall-clock
 * mt:f: tag) for a read whose last processed chunk produced *r.  ref_bucket receives the
 * reference index the row is filed under.  Returns bytes written (excluding NUL). */
int smhost_format_paf(const smhip_chunk_result *r, const char *read_name, uint32_t read_length,
                      uint32_t chunk_index, uint32_t num_events, const char *const *ref_names,
                      const uint32_t *ref_lengths, uint32_t n_refs, const smhost_map_params *p,
                      char *buf, size_t cap, uint32_t *ref_bucket);

/* The reference's per-chunk class methods, as the host library's mirrors implement them on the device
 * (sigmap_amd/host/device_methods.cc), behind a C face for the ctypes tests:
 *   SpatialIndex::GenerateChains(query_signal, stdvs, query_start_offset, step, radius, num_target_signals, chains)
 *     (spatial_index.h:87-92): prev_* = the chains handed in (their anchors chain after chain, n_anchors each),
 *     out_* = the chains handed back; returns the number of chains, -2 when a capacity is too small (*n_out_anchors
 *     then holds the anchors needed).  smhost_index_bind_device uploads the index to the given smhip_ctx (NULL: a
 *     context of the index's own on GPU 0).
 *   Sigmap::GenerateEvents(start, end, signal, feature_signal, feature_signal_stdvs) (sigmap.h:76): returns the
 *     number of features; feat receives them, stdvs receives `stdvs_already` placeholder values (-1) followed by
 *     the appended deviations (the method clears feature_signal but APPENDS to the deviations, sigmap.cc:1065-1079).
 * Without a usable GPU these end the process with a message, like the reference's ExitWithMessage. */
int smhost_index_bind_device(smhost_index *ix, void *smhip_context);
int64_t smhost_generate_chains(smhost_index *ix, const float *query, uint32_t n_query, uint32_t query_start_offset, int step,
                               float radius, uint32_t n_refs, const smhip_chain *prev_chains, uint32_t n_prev,
                               const smhip_carried_anchor *prev_anchors, smhip_chain *out_chains, uint32_t chain_cap,
                               smhip_carried_anchor *out_anchors, uint64_t anchor_cap, uint64_t *n_out_anchors);
int64_t smhost_generate_events(void *smhip_context, const float *signal_pa, uint64_t n, uint64_t start, uint64_t end, float *feat,
                               float *stdvs, uint32_t cap, uint32_t stdvs_already);

/* One PAF row from the fields of the reference's PAFMapping record (output_tools.h:17-26): mapq <= 60 prints
 * PAFOutputTools<PAFMapping>::AppendMapping's row (output_tools.h:336-354), anything else AppendUnmappedRead's
 * (:200-210) -- OutputMappingsInVector's choice (sigmap.cc:205-213) at the CLI's threshold 0.  mapq / direction
 * are taken modulo the record's 6-bit / 1-bit fields.  Returns bytes written (excluding NUL), -1 if cap is
 * too small.  (The per-record form of smhost_format_paf: the row printer alone, pinned against the real
 * output_tools.h in tests/test_oracle_golden.py.) */
int smhost_format_paf_record(const char *read_name, uint32_t read_length, uint32_t read_start, uint32_t read_end,
                             uint32_t frag_start, uint32_t frag_length, uint32_t mapq, uint32_t direction,
                             const char *tags, const char *ref_name, uint32_t ref_length, char *buf, size_t cap);

/* The reference genome as the driver loads it (sigmap.cc:559-568): SequenceBatch::InitializeLoading +
 * LoadAllSequences (sequence_batch.cc:8-15,82-122; FASTA / FASTQ, plain or gzip) + PrepareNegativeSequenceAt
 * per contig (sequence_batch.h:66-78).  Pointers stay valid until smhost_sequences_free. */
typedef struct smhost_sequences smhost_sequences;
smhost_sequences *smhost_sequences_load(const char *path);
void smhost_sequences_free(smhost_sequences *s);
uint32_t smhost_sequences_count(const smhost_sequences *s);
uint64_t smhost_sequences_num_bases(const smhost_sequences *s);
int smhost_sequences_get(const smhost_sequences *s, uint32_t i, const char **name, const char **bases,
                         uint32_t *length, const char **negative);

#ifdef __cplusplus
}
#endif
#endif
