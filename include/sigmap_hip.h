/*
 * sigmap_hip.h -- C ABI of the MI355X (gfx950) device side of the Sigmap hot path.
 *
 * The reference (haowenz/sigmap) has no FFI / plugin layer: the seam is the C++ class API
 * its driver calls.  Each entry point below names the reference interface it replaces
 * (file:line relative to /root/reference/src).  Callers are the host-side mirrors of the
 * reference classes in sigmap_amd/host/ (SpatialIndex, Sigmap), the `sigmap` CLI, and the
 * ctypes binding in sigmap_amd/_abi.py -- never end-user code.
 *
 * Conventions: plain C, POD structs, plain pointers + sizes.  Return 0 on success,
 * negative SMHIP_E* on failure (never exit()/abort(); the reference's assert/exit(-1),
 * utils.h:67, become error codes + smhip_last_error()).  One context per GPU, one host
 * thread per context at a time; different contexts are independent.  All device memory
 * is context-owned unless attached with smhip_index_attach().
 */
#ifndef SIGMAP_HIP_H_
#define SIGMAP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMHIP_OK 0
#define SMHIP_EINVAL (-1)   /* bad argument */
#define SMHIP_EHIP (-2)     /* HIP runtime error (message in smhip_last_error) */
#define SMHIP_ENOMEM (-3)   /* device allocation failed */
#define SMHIP_ESTATE (-4)   /* call order violated (no index, no batch ...) */
#define SMHIP_ECAP (-5)     /* a device-side capacity was exceeded (see status bits) */

typedef struct smhip_ctx smhip_ctx;

/* Flattened nanoflann node (replaces the 32-byte pointer node, nanoflann.hpp:755-768, as
 * dumped pre-order by save_tree, nanoflann.hpp:1025-1033).  child1 is always self+1.
 *   inner: feat = split dimension (>=0), a = divlow bits, b = divhigh bits, child2 = index
 *   leaf : feat = -1, a = left, b = right (range into the tree-ordered point list) */
typedef struct {
  int32_t feat;
  uint32_t a;
  uint32_t b;
  uint32_t child2;
} smhip_node;

/* Every constant the reference hard-codes on this path, surfaced once. */
typedef struct {
  /* event detector, event.h:31-37 */
  uint32_t window_length1, window_length2;
  float threshold1, threshold2, peak_height;
  double compress_delta;      /* sigmap.cc:1075 (0.1) */
  uint32_t min_num_events;    /* sigmap.cc:660: chains only if features > 50 */
  /* seeding, sigmap.cc:1381-1388 */
  float search_radius;        /* 0.08, compared against SQUARED L2 */
  int32_t step_size;          /* 2 */
  /* chaining, spatial_index.cc:284-291 */
  int32_t max_gap_length;         /* 2000 */
  int32_t max_target_gap_length;  /* 5000 */
  int32_t chaining_band_length;   /* 5000 */
  int32_t max_num_skips;          /* 25 */
  int32_t min_num_anchors;        /* 2 */
  int32_t num_best_chains;        /* 3 */
  int32_t num_nearest_points;     /* 5000 */
  float min_chaining_score;       /* 10 */
  /* device-side capacities (not reference semantics; exceeding one is reported, never
   * silently truncated) */
  uint32_t hits_per_query_fast;   /* per-query slots before the overflow pool; 0 (default) = automatic: the power of
                                     two in [512, 2048] that holds three quarters of the queries of the index census
                                     (smhip_index_census) entirely */
  uint32_t walk_blocks_per_cu;    /* resident 256-thread workgroups per CU of the tree walker
                                     (0 = default 4; speed only) */
  uint32_t reference_walk;        /* 0: skip subtrees whose tight box is out of range (same hits,
                                     same order).  1: visit exactly nanoflann's node set, so that
                                     nodes_visited / points_tested count the reference's work
                                     (the census behind bench.py's algorithmic bytes) */
  uint32_t search_mode;           /* how smhip_map_run and smhip_seed_lookup walk the tree (same hits, same anchors,
                                     same PAF in every mode; speed only):
                                       SMHIP_SEARCH_AUTO  the order-free set walk over quad records; of a query with
                                                          more than num_nearest_points hits the first num_nearest_points
                                                          IN THE REFERENCE'S TRAVERSAL ORDER are kept
                                                          (spatial_index.cc:371-372) by ranking the hits found
                                                          (root-to-leaf path strings; seed_lookup.hip).  Where the
                                                          ranking does not apply (num_nearest_points > 8192) such
                                                          queries are searched a second time in the reference's order,
                                                          and an index or a context with more than
                                                          set_walk_max_over_cap of them searches everything in that
                                                          order at once;
                                       SMHIP_SEARCH_EXACT always the reference's traversal order (binary records);
                                       SMHIP_SEARCH_SET   always the set walk.
                                     smhip_seed_lookup: AUTO = EXACT (hits reported in nanoflann's order); SET reports
                                     every query's hits in the set walk's own order (the same SET of hits).
                                     An index without quad records (no inner node) is always walked in EXACT mode. */
  float set_walk_max_over_cap;    /* AUTO without the ranking: share of queries beyond the hit cap above which the
                                     set walk is not used (0.0005) */
} smhip_params;
#define SMHIP_SEARCH_AUTO 0u
#define SMHIP_SEARCH_EXACT 1u
#define SMHIP_SEARCH_SET 2u

void smhip_default_params(smhip_params *p);

/* ---- lifecycle ---- */
int smhip_init(int device_ordinal, smhip_ctx **out);
void smhip_destroy(smhip_ctx *ctx);
const char *smhip_last_error(smhip_ctx *ctx);
/* HIP stream all kernels of this context are launched on (hipStream_t as void*). */
void *smhip_stream(smhip_ctx *ctx);

/* ---- index: replaces SpatialIndex::Load, spatial_index.cc:132-163 ----
 * values/positions: the `.pt` point cloud (Point{position,value}, sigmap_adaptor.h:7-17)
 * split into two arrays; vind + nodes + bbox: the `.si` tree (nanoflann.hpp:1051-1058).
 * root_bbox = {low0, high0, low1, high1, ...}. */
int smhip_index_upload(smhip_ctx *ctx, int dim, const float *values, const uint64_t *positions,
                       uint64_t n_points, const uint32_t *vind, uint64_t m_size,
                       const smhip_node *nodes, uint64_t n_nodes, const float *root_bbox);

/* Device-resident index arrays, for replication to peer GPUs by one RCCL broadcast per
 * array.  The caller owns the transport (torch.distributed / RCCL).
 * These are the walker's own structures, derived from the smhip_node array at upload. */
typedef struct {
  int32_t dim;
  uint32_t root_ref;                           /* 0, or 0x80000000 when the tree is a single leaf */
  uint64_t n_nodes, n_leaves, m_size;          /* n_nodes = INNER nodes (one record each) */
  void *nodes;      uint64_t nodes_bytes;      /* 64 B per inner node, pre-order among the inner nodes:
                                                  {divlow, divhigh, ref1 | feat << 28, ref2}, then the
                                                  tight boxes of both children as 24 x u16 on the grid
                                                  in bbox[12..23]; ref = inner index, or 0x80000000 |
                                                  leaf ordinal */
  void *leaf_tab;   uint64_t leaf_tab_bytes;   /* uint32 per leaf: (boff / 8) | n << 27 */
  void *leaf_pts;   uint64_t leaf_pts_bytes;   /* per-leaf 256-byte-aligned blocks at 32-byte slot `boff`:
                                                  n x {v0..v5, uint64 position} */
  void *leaf_vind;  uint64_t leaf_vind_bytes;  /* uint32[slots]: original point index of the
                                                  point in slot boff + t */
  void *bbox;       uint64_t bbox_bytes;       /* float[4*dim]: root box {low0, high0, ...}, then
                                                  base[dim], step[dim] of the box grid (one step for
                                                  all dimensions) */
  void *quads;      uint64_t quads_bytes;      /* 128 B per quad record (the order-free search's node array: the
                                                  boxes and references of a node's four grandchildren, derived from
                                                  `nodes` at upload); null / 0 for an index without them */
  uint64_t n_quads;
} smhip_index_dev;
int smhip_index_export(smhip_ctx *ctx, smhip_index_dev *out);
/* Borrow externally owned device arrays (e.g. torch tensors filled by broadcast). */
int smhip_index_attach(smhip_ctx *ctx, const smhip_index_dev *in);
/* Copy exported array `array_id` (0 nodes, 1 leaf_tab, 2 leaf_pts, 3 leaf_vind, 4 bbox, 5 quads) into memory
 * the caller owns (device memory of the same GPU, or host memory) -- e.g. a torch tensor that torch.distributed then broadcasts
 * (the copy runs on this library's HIP runtime and stream, not through a second runtime handle). */
int smhip_index_export_copy(smhip_ctx *ctx, int array_id, void *dst_device, uint64_t dst_bytes);
/* Replicate the index of ctxs[root] onto the other contexts, ONE context per GPU of this process:
 * one RCCL broadcast per walker array over xGMI (RCCL is used directly,
 * ncclCommInitAll + ncclBroadcast inside one group call).  The reference has no counterpart (it is
 * a single-process OpenMP program, sigmap.cc:618-632); reads then shard over the GPUs with no
 * further communication.  n_gpus == 1 is a no-op.  Errors are reported on ctxs[root]. */
int smhip_index_bcast(smhip_ctx *const *ctxs, int n_gpus, int root);
/* The same RCCL call sequence on ONE GPU: a one-rank communicator (ncclCommInitAll(1)) and one grouped
 * ncclBroadcast per walker array, out of place from src's arrays into buffers dst then owns (with a single
 * rank RCCL copies the root's send buffer into its receive buffer).  Both contexts must be on the same GPU.
 * A self-test of the replication path for boxes with one GPU (smhip_index_bcast with n_gpus == 1 never
 * touches RCCL); dst afterwards maps from its own copy, independent of src.  No reference counterpart. */
int smhip_index_bcast_loopback(smhip_ctx *src, smhip_ctx *dst);
/* Census of the resident index: the hit counts (radius p->search_radius, no cap) of 4096 of its own points used
 * as queries -- reads are noisy copies of such windows --, ascending.  Taken once per index and radius (the first
 * smhip_map_run does it when a parameter is left automatic); decides SMHIP_SEARCH_AUTO and the automatic
 * hits_per_query_fast.  hits (nullable): up to cap counts; *n_out = size of the sample.  No reference counterpart. */
int smhip_index_census(smhip_ctx *ctx, const smhip_params *p, uint32_t *hits, uint32_t cap, uint32_t *n_out);
/* The parameters of the last smhip_map_run with the automatic choices resolved (hits_per_query_fast, search_mode). */
int smhip_resolved_params(smhip_ctx *ctx, smhip_params *out);
/* Development knobs (speed experiments and test hooks; results never depend on them).  A context reads SMHIP_<NAME>
 * from the environment ONCE, in smhip_init, and only when SMHIP_DEV=1 is set there as well; this call sets a knob of
 * a live context and is refused (SMHIP_ESTATE) for a context created without SMHIP_DEV=1.  Names (lower case):
 * lane_blocks_per_cu, xcd_ranges, scan_threshold, classes, bucket_bits, handout, sdtw_ring, sync_run, pool_min,
 * spec_tight, float_gap, sort_lines, second_search, classify_dense, classify_resolve. */
int smhip_dev_knob(smhip_ctx *ctx, const char *name, long long value);
/* GPUs visible to this process (0 when there is none / no HIP runtime). */
int smhip_device_count(void);

/* ---- one batch of signal chunks: replaces the body of the taskloop in
 * Sigmap::StreamingMap (sigmap.cc:647-666): GenerateEvents + GenerateChains per chunk ---- */

#define SMHIP_TOPK 4
typedef struct {
  float score;
  uint32_t ref, start, end, n_anchors;
  uint32_t mapq;       /* only chains[0] carries a MAPQ (spatial_index.cc:255-274) */
  uint32_t direction;  /* 1 = Positive (+), 0 = Negative (-), spatial_index.h:13-16 */
  uint32_t pad;
} smhip_chain; /* SignalAnchorChain, spatial_index.h:28-45, without the anchor vector */

#define SMHIP_ST_FEW_PEAKS 1u      /* < 2 peaks: reference is out of bounds (event.h:213-223); chunk yields nothing */
#define SMHIP_ST_HIT_POOL_FULL 2u  /* overflow hit pool exhausted */
#define SMHIP_ST_STACK_FULL 4u     /* tree-walk stack deeper than the device stack */
#define SMHIP_ST_CHAIN_CAP 8u      /* more candidate chains than the per-chunk scratch */
#define SMHIP_ST_SORT_CAP 16u      /* more anchors than the sort key can index */

typedef struct {
  uint32_t status;       /* SMHIP_ST_* bits */
  uint32_t n_events;     /* raw events (event.h:226) */
  uint32_t n_features;   /* after z-score + compression (sigmap.cc:1073-1079) */
  uint32_t n_queries;    /* radius searches issued (spatial_index.cc:351) */
  uint32_t n_anchors;    /* sum over queries of min(hits, num_nearest_points) + carried */
  uint32_t n_chains;     /* primary chains (spatial_index.cc:571-576) */
  float sum_score;       /* chain scores added in order, float (sigmap.cc:702-705) */
  uint32_t pad0;
  uint64_t nodes_visited;  /* tree nodes touched by this chunk's queries */
  uint64_t points_tested;  /* leaf points distance-tested */
  uint64_t hits_total;     /* hits before the 5000 cap */
  smhip_chain chains[SMHIP_TOPK];
  /* chain[0] anchor statistics in the reference's accumulation order (sigmap.cc:714-729) */
  float c0_sum_dist, c0_sum_tgap, c0_sum_qgap;
  uint32_t c0_first_qpos; /* anchors[0].query_position (chain end) */
  uint32_t c0_last_qpos;  /* anchors.back().query_position (chain start) */
  uint32_t pad1;
} smhip_chunk_result;

typedef struct {
  uint32_t ref_strand;  /* (ref << 1) | strand, strand 0 = + */
  uint32_t target_position, query_position;
  float distance;
} smhip_carried_anchor; /* SignalAnchor + its bucket, spatial_index.cc:303-322 */

/* Stage inputs into HBM.  samples_pa: pA values (after SignalBatch::AddSignal's conversion
 * and (30,200) filter, signal_batch.cc:139-151); chunk c = samples[offsets[c]..offsets[c+1]).
 * query_start_offsets (nullable = zeros): events consumed by earlier chunks of the same read
 * (sigmap.cc:663).  carried (nullable): anchors of the previous chunk's chains. */
int smhip_batch_upload(smhip_ctx *ctx, const float *samples_pa, const uint32_t *sample_offsets,
                       uint32_t n_chunks, const uint32_t *query_start_offsets,
                       const smhip_carried_anchor *carried, const uint32_t *carried_offsets);
/* The same batch interface for chunks that are FEATURE signals already (the inputs of SpatialIndex::GenerateChains,
 * spatial_index.h:87-92, batched): chunk c = features[feat_offsets[c] .. feat_offsets[c + 1]) (at most 2736 each)
 * takes the place of the events a smhip_batch_upload chunk would yield; smhip_map_run then starts at the seeds
 * (spatial_index.cc:326), and everything after it (download, chain anchors, sDTW refinement against the chunk's
 * features) works as for a batch of samples.  GenerateChains has no minimum number of features (min_num_events is
 * StreamingMap's test, sigmap.cc:660) and is undefined below `dim` of them (spatial_index.cc:330): such a chunk
 * issues no query and chains its carried anchors only. */
int smhip_features_upload(smhip_ctx *ctx, const float *features, const uint32_t *feat_offsets, uint32_t n_chunks,
                          const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                          const uint32_t *carried_offsets);
/* SpatialIndex::GenerateChains(query_signal, stdvs, query_start_offset, step, radius, num_target_signals, chains),
 * spatial_index.cc:276-577, for n_chunks queries at once: smhip_features_upload + smhip_map_run + smhip_map_sync +
 * smhip_map_download(results).  (step = p->step_size, radius = p->search_radius; `carried` = the anchors of the
 * chains handed in, `results` + smhip_map_download's chain anchors = the chains handed back.  The stdvs argument of
 * the reference is unused by it.) */
int smhip_generate_chains(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, const float *features,
                          const uint32_t *feat_offsets, uint32_t n_chunks, const uint32_t *query_start_offsets,
                          const smhip_carried_anchor *carried, const uint32_t *carried_offsets,
                          smhip_chunk_result *results);
/* The chaining stage alone (spatial_index.cc:411-577: per-bucket std::sort, chaining DP, traceback, primary chains,
 * MAPQ) on anchors the caller hands in, sorted or not: chunk c's anchors = anchors[anchor_offsets[c] ..
 * anchor_offsets[c + 1]), each with its bucket (ref << 1 | strand).  Equivalent to GenerateChains called with an
 * empty query and these anchors as the previous chains'.  Chain anchors: smhip_map_download afterwards. */
int smhip_chain_anchors(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, const smhip_carried_anchor *anchors,
                        const uint32_t *anchor_offsets, uint32_t n_chunks, smhip_chunk_result *results);
/* ---- resident reads: SignalBatch::AddSignal's conversion on the device (signal_batch.cc:139-151;
 * the same loop for BLOW5 records at :195-207) ----
 * The host keeps a read as its raw ADC counts + (digitisation, range, offset); a read crosses PCIe
 * once, as int16.  The device converts pa = ((float)raw + offset) * (range / digitisation), drops
 * samples outside (30, 200) pA and compacts the survivors in order into the read's SLOT, from which
 * the batches then cut their chunks (the reference cuts its 4000-sample chunks from the FILTERED
 * array, sigmap.cc:637-652).  The context holds n_slots slots of keep_samples floats each (e.g.
 * max_num_chunks * 4000); the caller decides which read lives in which slot and when a slot is reused. */
int smhip_reads_configure(smhip_ctx *ctx, uint32_t n_slots, uint32_t keep_samples);
/* Read r = raw[raw_offsets[r] .. raw_offsets[r+1]) goes to slot slots[r] (slots == NULL: slot r).
 * pa_len[r] (optional) = filtered length of the WHOLE read -- what Signal::signal_values.size() is in
 * the reference (PAF column 2, and num_chunks = length / 4000, sigmap.cc:643) -- even when only the
 * first keep_samples of them are kept.  The raw samples are copied by DMA on a separate stream, overlapped
 * with the kernels of the previous batch, when `raw` is pinned memory (smhip_host_alloc).  With
 * pa_len == NULL the call only enqueues work: `raw` must then stay untouched until the next
 * smhip_map_sync (or any later call that returns results). */
int smhip_reads_upload_raw(smhip_ctx *ctx, const int16_t *raw, const uint64_t *raw_offsets,
                           const float *digitisation, const float *range, const float *offset,
                           const uint32_t *slots, uint32_t n_reads, uint32_t *pa_len);
/* PIECES of reads -- e.g. the one chunk of each read a streaming round maps (sigmap.cc:647-652) -- that live in PINNED
 * host memory (smhip_host_alloc), pulled over PCIe by a kernel of the context's stream: one launch for any number of
 * scattered pieces, no staging copy on the host, and only the samples that are mapped ever cross the link.  Piece r =
 * pieces[r][0 .. piece_len[r]) is converted and filtered like a read of smhip_reads_upload_raw and REPLACES the content
 * of slot slots[r] (slots == NULL: slot r).  expect_len (optional): the filtered length of every piece as the caller
 * counted it (SignalBatch's load phase: pa = ((float)raw + offset) * (range / digitisation), 30 < pa < 200, the same
 * f32 operations); a piece of which the device keeps another number of samples makes the next call that collects
 * results fail with SMHIP_ECAP -- a chunk cut by the host's count would not be the reference's chunk -- and the error
 * is STICKY: every later smhip_map_sync / smhip_map_download* / smhip_batch_select of the context fails the same way
 * until smhip_reads_upload_pieces is called again or the slots are refilled by smhip_reads_upload_raw (nothing computed
 * from the wrong chunks is ever handed out; a batch of smhip_batch_upload, which brings its own samples, is unaffected).
 * Every piece is range-checked on the host against the blocks of smhip_host_alloc (memory pinned by someone else:
 * one driver lookup per piece); a piece outside pinned memory is SMHIP_EINVAL, never an address given to the device.
 * Asynchronous: the arrays passed are free on return, the pieces must stay untouched until the batch that maps them
 * is collected. */
int smhip_reads_upload_pieces(smhip_ctx *ctx, const int16_t *const *pieces, const uint32_t *piece_len,
                              const float *digitisation, const float *range, const float *offset,
                              const uint32_t *slots, const uint32_t *expect_len, uint32_t n_pieces);
/* The next batch: chunk c = filtered samples [chunk_index[c] * chunk_size, + chunk_size) of slot
 * slots[c], clipped to the slot's length (sigmap.cc:647-652).  Other arguments as smhip_batch_upload.
 * Does not wait for the device: the caller's arrays go through pinned staging owned by the context and are free on
 * return.  The intended order is select -> run -> collect -> next select; a caller that selects again BEFORE the
 * previous batch was collected is still safe -- the call then waits (an event behind the copies that read the staging)
 * until those copies have finished before it overwrites or regrows the staging.  After a piece-length mismatch
 * (smhip_reads_upload_pieces, expect_len) it fails with SMHIP_ECAP until the pieces have been uploaded again. */
int smhip_batch_select(smhip_ctx *ctx, const uint32_t *slots, const uint32_t *chunk_index, uint32_t chunk_size,
                       uint32_t n_chunks, const uint32_t *query_start_offsets,
                       const smhip_carried_anchor *carried, const uint32_t *carried_offsets);
/* The filtered pA samples a slot holds (for parity tests against SignalBatch::AddSignal): up to cap
 * floats; n_kept = samples held (<= keep_samples), n_total = filtered length of the whole read. */
int smhip_reads_download_pa(smhip_ctx *ctx, uint32_t slot, float *pa, uint32_t cap, uint32_t *n_kept,
                            uint32_t *n_total);
/* Optional set-up step: allocate now the scratch a batch of n_chunks chunks with about
 * queries_per_chunk seeds each will need (it otherwise grows on first use; device allocations beyond a
 * few GB take 40 ms per GB on this platform -- tools/micro/malloc_cost.hip -- which a caller that times
 * its mapping phase wants outside that span, like the index load). */
int smhip_batch_reserve(smhip_ctx *ctx, uint32_t n_chunks, uint32_t queries_per_chunk);
/* The same with the parameters the caller will map with (NULL: the defaults): the fast hit slots per query are
 * resolved for THEIR search radius and hit cap -- from the same census of the index smhip_map_run resolves them
 * from, so the first batch finds its buffers at their final sizes and the census is taken once. */
int smhip_batch_reserve_for(smhip_ctx *ctx, const smhip_params *params, uint32_t n_chunks, uint32_t queries_per_chunk);
/* Pinned (page-locked) host memory for the raw reads: DMA at link rate, truly asynchronous copies. */
int smhip_host_alloc(void **out, size_t bytes);
void smhip_host_free(void *p);

/* Launch events -> seeds -> radius search -> sort -> chain on the context stream.  Two totals size the
 * later stages' buffers and only the device knows them: the number of queries (after event detection)
 * and the number of anchors (after the search).  The first batch of a context (and any batch while the
 * environment sets SMHIP_SYNC_RUN=1) reads them back: the call then waits twice and returns when the
 * search has finished, sort + chain still running.  Later batches SPECULATE: buffers are sized from the
 * per-chunk totals of the largest batch so far (+ 25 % + slack), a guard kernel on the device empties a
 * batch that outgrows them and raises a flag, and the call returns with every kernel enqueued and
 * nothing waited for.  smhip_map_sync / the download calls wait for the stream, find the flag and repeat
 * such a batch on the reading route (together with a smhip_refine_run enqueued behind it), starting from the
 * feature signals of the first pass: the records never depend on the route, and the read slots named by
 * smhip_batch_select may be refilled (smhip_reads_upload_raw, which queues its conversion behind the batch's
 * kernels) as soon as smhip_map_run has returned. */
int smhip_map_run(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs);
int smhip_map_sync(smhip_ctx *ctx);
/* results: n_chunks records. chain_anchors (nullable): anchors of ALL primary chains,
 * chain-major, end->start order, chunk c at [chain_anchor_offsets[c], ..[c+1]). */
int smhip_map_download(smhip_ctx *ctx, smhip_chunk_result *results,
                       smhip_carried_anchor *chain_anchors, uint64_t chain_anchor_cap,
                       uint64_t *chain_anchor_offsets /* n_chunks+1 */);
/* The chain anchors of SOME chunks of the batch -- the reads a streaming round carries into its next chunk
 * (spatial_index.cc:303-322), which the caller knows once it has applied the stop rules to the result records:
 * chain_anchor_offsets[i] .. [i + 1] = the anchors of all primary chains of chunk chunks[i], as smhip_map_download
 * hands them out for every chunk.  SMHIP_ECAP when chain_anchor_cap is too small (offsets are valid then). */
int smhip_map_download_anchors_of(smhip_ctx *ctx, const uint32_t *chunks, uint32_t n_listed, smhip_carried_anchor *chain_anchors,
                                  uint64_t chain_anchor_cap, uint64_t *chain_anchor_offsets);
/* EVERY primary chain of every chunk (the std::vector<SignalAnchorChain> GenerateChains hands back,
 * spatial_index.cc:552-576; the result record only holds the first SMHIP_TOPK), in rank order: chunk c's at
 * chains[chain_offsets[c] .. chain_offsets[c + 1]); their anchors are smhip_map_download's chain_anchors, chain after
 * chain (n_anchors each).  chains may be NULL (offsets only); SMHIP_ECAP when chain_cap is too small (offsets valid). */
int smhip_map_download_chains(smhip_ctx *ctx, smhip_chain *chains, uint64_t chain_cap, uint64_t *chain_offsets /* n_chunks+1 */);
/* upload + run + sync + download */
int smhip_map_chunks(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs,
                     const float *samples_pa, const uint32_t *sample_offsets, uint32_t n_chunks,
                     const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                     const uint32_t *carried_offsets, smhip_chunk_result *results);
/* Mean device time (ms) of the named kernel over the launches since the last
 * smhip_timing_reset(), measured with HIP events on the context stream.
 * names: "convert", "events", "seed_order" (the queries' visiting order), "seed_lookup" (the tree-walk
 * launch alone), "sort", "chain", "sdtw". */
/* ---- banded sDTW refinement of the best chain ----
 * The reference computes the z-scored expected signal of every contig strand at map time
 * (sigmap.cc:566-591) and aligns against it with Sigmap::sDTW (sigmap.cc:1259-1291) in its
 * DTWAlign/CWTAlign drivers.  Here the same recurrence, restricted to a band around the chain's
 * diagonal, refines chain 0 of every chunk of the last smhip_map_run; PAF columns 1-12 are not
 * altered, the distance is reported as an extra tag (dt:f:).
 * signals: concatenated float arrays in the order contig0 +, contig0 -, contig1 +, ...;
 * offsets[2*n_refs + 1] in elements. */
int smhip_reference_upload(smhip_ctx *ctx, const float *signals, const uint64_t *offsets, uint32_t n_refs);
/* Borrow the reference signal `owner` uploaded (another context of the SAME GPU, which must outlive this
 * one's use of it): the contexts of a GPU share one copy, like the index (smhip_index_export / _attach). */
int smhip_reference_share(smhip_ctx *ctx, smhip_ctx *owner);
/* Free and total device memory of the context's GPU (hipMemGetInfo): callers size their batches from it. */
int smhip_mem_info(smhip_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* For chunk c with >= 1 chain: target = strand signal[t0, t1) with t0 = max(0, start - pad),
 * t1 = min(len, end + dim + pad); query = the chunk's feature signal (F values); row t (1-based,
 * absolute index ta = t0 + t - 1) may use the 1-based columns [centre - band/2, centre + band/2]
 * clipped to [1, F], centre = 1 + qa + floor((ta - start) * (qb - qa) / max(1, end - start)), where
 * (start, qa) / (end, qb) are target/query positions of chain 0's first/last anchor (query
 * positions made chunk-local by subtracting query_start_offset, clamped at 0).
 * dist[c] = FLT_MAX and end_pos[c] = -1 when the chunk has no chain or the window exceeds 8192
 * rows; otherwise end_pos = absolute 1-based target row of the best end (t0 + t). */
int smhip_refine_run(smhip_ctx *ctx, uint32_t band, uint32_t pad);
int smhip_refine_download(smhip_ctx *ctx, float *dist, int64_t *end_pos);

/* The same refinement with fast_dtw.cc::DTW()'s recurrence and its path (smhip_dtw_path below, D5):
 * window and band exactly as smhip_refine_run.  Download: dist / end_pos as above but end_pos and
 * start_pos are 0-based absolute target indices of the path's last / first cell (-1: no alignment);
 * path_codes[path_off[c] .. + path_len[c]) = the path's codes from its first to its last cell
 * (smhost_print_alignment gives the reference's per-base CIGAR).  path_codes may be NULL (lengths
 * only); path_cap >= path_off[n_chunks] otherwise (SMHIP_ECAP tells that it was too small; the
 * offsets are valid then). */
int smhip_refine_path_run(smhip_ctx *ctx, uint32_t band, uint32_t pad);
int smhip_refine_path_download(smhip_ctx *ctx, float *dist, int64_t *end_pos, int64_t *start_pos,
                               uint32_t *path_len, uint64_t *path_off, uint8_t *path_codes, uint64_t path_cap);

/* Diagnostic counters of the last smhip_map_run: [0] stack-overflow flag, [1] queries that needed
 * the overflow pass, [2] overflow pool slots used, [3] pool-exhausted flag; of the context: [4] batches
 * that were repeated after a speculative run outgrew its buffers, [5] speculative runs so far, [6] how the last
 * smhip_map_run searched: 0 in the reference's traversal order (binary records), 1 the reference's own node set
 * (reference_walk), 2 order-free over quad records (nodes_visited then counts 128-byte quad records). */
int smhip_debug_counters(smhip_ctx *ctx, uint32_t out[16]);
/* The anchor sort of the last smhip_map_run: parts sorted by one wave with 32-bit keys [0], by one wave with 64-bit keys
 * [1], by a workgroup [2]; chunks that left the whole-line partition route for the exact one because a part outgrew
 * its slice [3] (sort.hip). */
int smhip_debug_sort_stats(smhip_ctx *ctx, uint32_t out[4]);
/* Device-side sweeps of the two division-free forms of this build against the expressions they replace, over every
 * float whose bit pattern lies in [lo_bits, hi_bits): the chain classifier's start-score factor
 * (float)(1 - 0.2 * (double)dist / (double)radius) (spatial_index.cc:437-438; chain.hip StartCoef) and the
 * t-statistics' x / (float)w (event.h:95-108; events.hip DivW, w = a window length).  *mismatches = values whose
 * results differ in any bit (tests/test_gpu_parity.py sweeps all non-negative distances below 8 radii and every
 * float of either sign; tools/micro/ holds the same sweeps for the host). */
int smhip_debug_start_coef_sweep(smhip_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, float radius, uint64_t *mismatches);
int smhip_debug_div_window_sweep(smhip_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint32_t w, uint64_t *mismatches);
int smhip_timing_enable(smhip_ctx *ctx, int on);
int smhip_timing_reset(smhip_ctx *ctx);
int smhip_timing_get(smhip_ctx *ctx, const char *kernel, double *mean_ms, uint64_t *launches);

/* ---- per-stage entry points (parity tests address each kernel in isolation) ---- */

/* The reference's CWT feature path for a batch of chunks: Sigmap::GenerateFeatureSignalUsingCWT,
 * sigmap.cc:1085-1098 = MAD normalisation (:1100-1129) -> one-scale DoG CWT (GenerateCWTSignal
 * :1157-1174 -> cwt.cc) -> z-score (:1131-1155) -> GeneratePeaks (:1176-1196, hysteresis SD/4).
 * scale0 is the CWT scale (the reference uses 8 for reads, sigmap.cc:947); 4 <= scale0 <= 64.
 * feat / positions: n_chunks x feat_stride (peaks beyond feat_stride are counted, not stored);
 * cwt_out (optional): n_chunks x 4096 floats, the CWT signal of every chunk.
 * Dormant in the reference's CLI (CWTAlign is disabled, sigmap.cc:1499-1504): offered as a stage,
 * not wired into smhip_map_run.  The CWT agrees with the reference's f32-FFT evaluation to that
 * FFT's rounding (about 2e-7 of the signal's maximum), not bit for bit. */
int smhip_cwt_features(smhip_ctx *ctx, const float *samples_pa, const uint32_t *sample_offsets, uint32_t n_chunks,
                       float scale0, uint32_t feat_stride, float *feat, uint32_t *positions, uint32_t *n_features,
                       float *cwt_out);

/* Sigmap::GenerateEvents, sigmap.cc:1048-1083 (DetectEvents event.h:226 + z-score
 * sigmap.cc:1131 + compression).  feat/stdv: n_chunks x feat_stride floats. */
int smhip_events(smhip_ctx *ctx, const smhip_params *p, const float *samples_pa,
                 const uint32_t *sample_offsets, uint32_t n_chunks, uint32_t feat_stride,
                 float *feat, float *stdv, uint32_t *n_features, uint32_t *n_events,
                 uint32_t *status);
/* The intermediates of the last smhip_events (or smhip_map_run) launch, for comparison with the
 * reference's DetectEvents stage by stage (event.h:58-68 prefix sums, :70-115 t-statistics of both
 * windows, :117-182 peaks, :184-224 events).  Every output pointer is optional.  n_peaks: per chunk;
 * peaks / ev_start / ev_length / ev_mean / ev_stdv: n_chunks x feat_stride (Event{start, length,
 * mean, stdv}, event.h:16-21; a chunk with < 2 peaks has no events); ps / pss / t1 / t2: n_chunks x
 * row_stride floats, entries [0, n] of each row valid (row_stride >= 4097). */
int smhip_events_raw(smhip_ctx *ctx, uint32_t n_chunks, uint32_t feat_stride, uint32_t row_stride,
                     uint32_t *n_peaks, uint32_t *peaks, uint32_t *ev_start, uint32_t *ev_length,
                     float *ev_mean, float *ev_stdv, float *ps, float *pss, float *t1, float *t2);
/* nanoflann radiusSearch, nanoflann.hpp:1278-1301, for n_queries windows of `dim` floats.
 * Hits of query i in traversal order at point_idx/dist[i*cap ..], counts[i] = total hits
 * (may exceed cap).  point_idx = original point-cloud index (the pair.first nanoflann
 * returns).  With p->search_mode == SMHIP_SEARCH_SET the mapping path's order-free walk over quad records
 * runs instead: the same set of hits per query, in that walk's order (smhip_debug_counters [6] tells which
 * walk ran: an index without inner nodes has no quad records and is walked in the reference's order). */
int smhip_seed_lookup(smhip_ctx *ctx, const smhip_params *p, const float *queries,
                      uint32_t n_queries, uint32_t cap, uint32_t *counts, uint64_t *point_idx,
                      float *dist, uint64_t *nodes_visited, uint64_t *points_tested);
/* Banded subsequence DTW refinement (recurrence of Sigmap::sDTW, sigmap.cc:1259-1291,
 * restricted to per-row column windows).  Problem i: target[t_off[i]..t_off[i+1]),
 * query[q_off[i]..q_off[i+1]); row r of problem i may use query columns
 * [lo[r], hi[r]] (1-based, inclusive; rows indexed globally like target). */
int smhip_sdtw_banded(smhip_ctx *ctx, const float *target, const uint64_t *t_off,
                      const float *query, const uint64_t *q_off, const uint32_t *lo,
                      const uint32_t *hi, uint32_t n_problems, float *dist, int64_t *end);

/* fast_dtw.cc::DTW() (fast_dtw.cc:74-140) over per-row column windows, with its path: the same
 * problem layout as smhip_sdtw_banded, but the recurrence has four candidates
 *   { D[t-1][q-1] + c, D[t][q-1] + c, D[t][q-1] + 2, D[t-1][q] + 2 },  c = |target - query|
 * (the "skip" moves cost a flat 2, fast_dtw.cc:76-77,113), the first minimum's index 0..3 is the
 * cell's path code (0 one-to-one, 1 one base to several signals, 2 skip one signal, 3 skip one base),
 * and a row can end the alignment only if its window reaches the last query column.  dist[i] /
 * end[i]: distance and 0-based target coordinate of the end row (fast_dtw.cc:138; FLT_MAX / -1 when
 * no row ends with a finite value).  path_codes[path_off[i] .. + path_len[i]): the codes along
 * generate_path's path (fast_dtw.cc:51-72) from its first cell (query column 0, target coordinate
 * start_target[i]) to the end cell; capacity path_off[i+1] - path_off[i] >= T_i + Q_i.
 * smhost_print_alignment (sigmap_host.h) turns the codes into the reference's per-base CIGAR. */
int smhip_dtw_path(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                   const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, uint32_t n_problems,
                   float *dist, int64_t *end, uint8_t *path_codes, const uint64_t *path_off,
                   uint32_t *path_len, int64_t *start_target);

#ifdef __cplusplus
}
#endif
#endif /* SIGMAP_HIP_H_ */
