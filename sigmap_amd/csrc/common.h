// common.h -- shared declarations of the gfx950 device side (libsigmap_hip.so).
// Compiled with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the FP pin of
// SURVEY.md section 0.4 applies to device code as well: no FMA contraction).
#ifndef SIGMAP_AMD_CSRC_COMMON_H_
#define SIGMAP_AMD_CSRC_COMMON_H_

#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/sigmap_hip.h"

namespace smhip {

constexpr int kMaxDim = 6;           // device kernels are specialised for the index dimension 6
constexpr int kMaxSamples = 4096;    // samples per chunk held in LDS (reference chunk = 4000)
constexpr int kFeatStride = 2736;    // >= 2/3 * kMaxSamples + 2: upper bound on events per chunk
constexpr int kStackDepth = 64;      // pending "other child" entries per tree walk
constexpr int kOvfLevels = 16;       // overflow segments per query (sizes fast_cap << level)
constexpr int kChainCap = 128;       // candidate chains per chunk that ChainFinalize ranks in LDS (more: global path)

struct Hit {  // 16 B: one radius-search result
  uint64_t position;  // Point.position of the matched reference window
  float dist;         // squared L2
  uint32_t slot;      // index in tree (leaf-major) order
};

struct RawReadMeta {  // one read of smhip_reads_upload_raw (32 B)
  uint64_t raw_off;   // first ADC sample in the staged raw buffer
  uint32_t n_raw;     // ADC samples
  uint32_t slot;      // resident slot the filtered pA samples go to
  float digitisation, range, offset;  // Signal{digitisation, range, offset}, signal_batch.h:14-25
  uint32_t pad;
};

struct PieceSrc {  // one piece of smhip_reads_upload_pieces (16 B)
  uint64_t src;  // address of its first ADC sample in pinned host memory
  uint32_t dst;  // first 16-byte unit of the piece in the staging buffer
  uint32_t n;    // samples
};

// Per-query hit slots (fast_cap hits of 16 B per query).  The whole array is 20-30 GB for a batch of
// 16 384 chunks, and a device allocation beyond 4 GB costs tens of ms per GB on this platform
// (tools/micro/malloc_chunks.hip: 8 x 3.9 GB in 1.4 ms, one 27 GB block in 3 s), so it is cut into
// segments of at most 2 GiB: queries [k << qshift, (k + 1) << qshift) live in segment k.
struct HitSlots {
  Hit *const *seg;   // device array of segment base pointers
  uint32_t qshift;   // queries per segment = 1 << qshift
  uint32_t fast_cap;
  __device__ __forceinline__ Hit *At(uint32_t q, uint32_t ord) const {
    return seg[q >> qshift] + ((size_t)(q & ((1u << qshift) - 1u)) * fast_cap + ord);
  }
};

// Resident read slots (keep floats each), cut into segments of at most 2 GiB like HitSlots.
struct SlotView {
  float *const *seg;  // device array of segment base pointers
  uint32_t sshift;    // slots per segment = 1 << sshift
  uint32_t keep;
  __device__ __forceinline__ float *At(uint32_t slot) const {
    return seg[slot >> sshift] + (size_t)(slot & ((1u << sshift) - 1u)) * keep;
  }
};

struct ChainTmp {
  float score;
  uint32_t ref, start, end, n_anchors, direction;
  uint32_t end_anchor;  // chunk-local index of the chain's last (highest target) anchor
  uint32_t list_off;    // chunk-local offset of the chain's anchor index list (end -> start)
};

#ifdef __HIPCC__
// Inclusive scans over the 64 lanes of a wave with DPP: shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then lane
// 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15, row mask 0xA) and lane 31 into rows 2 and 3 (row_bcast:31, row
// mask 0xC).  A lane without a source keeps the identity (`old` operand, bound_ctrl off).
__device__ __forceinline__ float WaveScanMax(float v) {
#define SM_STEP(CTRL, ROWS) \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-FLT_MAX), __float_as_int(v), CTRL, ROWS, 0xF, false)))
  SM_STEP(0x111, 0xF); SM_STEP(0x112, 0xF); SM_STEP(0x114, 0xF); SM_STEP(0x118, 0xF); SM_STEP(0x142, 0xA); SM_STEP(0x143, 0xC);
#undef SM_STEP
  return v;
}
__device__ __forceinline__ int WaveScanAdd(int v) {
#define SM_STEP(CTRL, ROWS) v += __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xF, false)
  SM_STEP(0x111, 0xF); SM_STEP(0x112, 0xF); SM_STEP(0x114, 0xF); SM_STEP(0x118, 0xF); SM_STEP(0x142, 0xA); SM_STEP(0x143, 0xC);
#undef SM_STEP
  return v;
}
#endif

#ifdef __HIPCC__
// Overflow segment of hit number fast_cap + k of a query (segments of fast_cap, 2 fast_cap, 4 fast_cap, ... hits:
// seed_lookup.hip AllocOverflowSegment): floor(log2(k / fast_cap + 1)).  fast_cap is wave-uniform and, whenever the
// library chose it, a power of two: a shift then -- the general 32-bit division is ~20 vector instructions, in the
// walker's store path and the partition's hit loads of a large reference, where most hits live in the pool.
__device__ __forceinline__ uint32_t OvfLevel(uint32_t k, uint32_t fast_cap) {
  const uint32_t ratio = (fast_cap & (fast_cap - 1u)) == 0u ? k >> (uint32_t)__builtin_ctz(fast_cap | 0x80000000u) : k / fast_cap;
  return 31u - (uint32_t)__clz((int)(ratio + 1u));
}
#endif

// Device view of the index (all pointers device memory; layouts: seed_lookup.hip header).
constexpr uint32_t kLeafTag = 0x80000000u;  // a node ref with this bit names a leaf (ordinal in the low 27 bits)

struct IndexView {
  const uint4 *inner;         // 4 x uint4 per INNER node: {divlow, divhigh, ref1 | feat << 28, ref2}, then the
                              // children's tight boxes, 24 x u16 (layout: seed_lookup.hip header)
  const uint32_t *leaf_tab;   // per leaf: (first slot of its point block / 16) | n << 27
  const float *leaf_pts;      // per-leaf 128-byte-aligned point blocks
  const uint32_t *leaf_vind;  // [n_slots / 2] Hit.slot -> original point index
  const float *bbox;          // [24] low0, high0, ... of nanoflann's root box; then base[6], step[6] of the
                              // grid the children's boxes are quantised on
  uint64_t n_inner, n_leaves, m_size;
  uint32_t root_ref;          // 0 (inner record 0), or kLeafTag for a tree that is a single leaf
  const uint4 *quads;         // 8 x uint4 per quad record (seed_lookup.hip: BuildQuads), null: none (no inner node,
  uint64_t n_quads;           // or an index too large for them); the root's quad is number 0
};

inline bool TraceAlloc() {
  // (development aid, stderr only; like every SMHIP_* switch it needs SMHIP_DEV=1 beside it.  Magic static: thread-safe)
  static const bool on = getenv("SMHIP_DEV") != nullptr && atoi(getenv("SMHIP_DEV")) == 1 && getenv("SMHIP_TRACE_ALLOC") != nullptr;
  return on;
}

#include "knobs.h"  // DevKnobs: the development knobs of a context (a header of its own: adding a knob leaves this one, and
                    // with it the source hash the counter profiles are paired by, unchanged)

// -DSMHIP_PHASE_PROF (tools/phase_prof.sh, never in the shipped library): thread 0 of every workgroup adds
// the clock ticks between the marks of an instrumented kernel to its translation unit's g_phase_prof[32]
// (read by smhip_debug_phase_prof / smhip_debug_chain_prof).
#ifdef SMHIP_PHASE_PROF
#define PHASE_DECL()                    \
  unsigned long long prof_acc[16] = {}; \
  unsigned long long prof_t = wall_clock64()
#define PHASE_INIT() prof_t = wall_clock64()
#define PHASE_MARK(k)                                 \
  do {                                                \
    const unsigned long long prof_n = wall_clock64(); \
    prof_acc[k] += prof_n - prof_t;                   \
    prof_t = prof_n;                                  \
  } while (0)
#define PHASE_FLUSH()                                                    \
  do {                                                                   \
    if (threadIdx.x == 0)                                                \
      for (int pk = 0; pk < 16; ++pk)                                    \
        if (prof_acc[pk]) atomicAdd(&g_phase_prof[pk], prof_acc[pk]);    \
  } while (0)
#else
#define PHASE_DECL() \
  do {               \
  } while (0)
#define PHASE_INIT() \
  do {               \
  } while (0)
#define PHASE_MARK(k) \
  do {                \
  } while (0)
#define PHASE_FLUSH() \
  do {                \
  } while (0)
#endif

// Grow-only device buffer.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t Reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + std::min<size_t>(bytes / 8, (size_t)256 << 20) + 256;  // (head room, bounded: buffers reach 40 GB)
    const bool trace = TraceAlloc();
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, want);
    if (trace)  // SMHIP_TRACE_ALLOC=1: device allocations beyond a few GB take tens of ms per GB here
      fprintf(stderr, "[smhip] hipMalloc %.3f GB: %.1f ms\n", (double)want / (1ull << 30),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (e == hipSuccess) cap = want;
    return e;
  }
  void Free() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

struct KernelTimer {
  std::string name;
  double total_ms = 0;
  uint64_t launches = 0;
};

}  // namespace smhip

struct smhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  // index
  bool have_index = false, index_owned = false;
  int dim = 0;
  smhip::IndexView index{};
  smhip::DevBuf ix_nodes, ix_tab, ix_pts, ix_vind, ix_bbox, ix_quads, scan_tmp;
  smhip::DevBuf ix_slot_leaf;     // block of eight point slots -> leaf ordinal (seed_lookup.hip OverCapSelectKernel), derived
  bool slot_leaf_valid = false;   // lazily from leaf_tab by the context itself, whichever way the index arrived
  int last_walk_mode = 0;         // search mode of the last smhip_map_run: 0 exact order, 1 census, 2 order-free (quad records)
  uint64_t leaf_pts_bytes = 0, leaf_vind_bytes = 0;
  smhip::DevBuf max_tpos;       // [0] largest target position of the resident index, [1] ... and of the batch's carried anchors
  bool max_tpos_valid = false;  // [0] computed for the index that is resident now
  // batch inputs
  uint32_t n_chunks = 0;
  uint64_t n_samples = 0, n_carried = 0;
  bool batch_ready = false, run_done = false;
  // chunk c of the batch = chunk_src[chunk_begin[c] .. + chunk_len[c]): chunk_src is `samples` (pA uploaded
  // with the batch, smhip_batch_upload), or null with chunk_begin = the chunk's device address / 4 (resident
  // reads in their slot segments, smhip_reads_upload_raw + smhip_batch_select)
  smhip::DevBuf samples, chunk_begin, chunk_len, qstart, carried, carried_off;
  const float *chunk_src = nullptr;
  bool batch_is_features = false;  // the batch holds feature signals (smhip_features_upload), not samples
  // resident reads (SignalBatch::AddSignal on the device): raw ADC staging, per-read records, segment
  // counts / bases, the slots' filtered pA samples (n_slots x reads_keep floats) and their lengths
  smhip::DevBuf reads_raw, reads_meta, reads_segstart, reads_seg, slot_len, reads_len, sel_slots, sel_chunk, pa_tab;
  std::vector<smhip::DevBuf> pa_segs;  // the slots' segments (SlotView); pa_tab = their pointers on the device
  std::vector<float *> h_pa_tab;
  uint32_t pa_sshift = 0;
  hipStream_t aux_stream = nullptr;    // independent kernels of a step beside `stream` (chain.hip: the wave DP), fork / join
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // by these events
  hipStream_t copy_stream = nullptr;   // H2D of raw reads, overlapped with the kernels on `stream`
  hipEvent_t copy_done = nullptr, convert_done = nullptr;
  uint32_t n_slots = 0, reads_keep = 0;
  void *h_meta = nullptr;              // pinned host staging of the per-read records + segment starts
  size_t h_meta_cap = 0;
  // smhip_reads_upload_pieces: its own pinned staging (records, segment starts, piece sources, expected lengths), the
  // event behind the copies that read it, the device copies of the last two, and the mismatch flag (pinned, written
  // by the device)
  void *h_pieces = nullptr;
  size_t h_pieces_cap = 0;
  hipEvent_t pieces_staged = nullptr;
  bool pieces_pending = false;
  smhip::DevBuf pieces_src, pieces_expect;
  uint32_t *d_carried_off = nullptr;  // the carried offsets of the batch: behind the query start offsets in `qstart`'s block
  uint32_t *h_piece_flag = nullptr;
  // carried anchors of smhip_batch_select staged in pinned memory (the call then returns without waiting)
  void *h_carried = nullptr;
  size_t h_carried_cap = 0;
  std::vector<uint64_t> h_chunk_begin; // scratch of smhip_batch_upload
  std::vector<uint32_t> h_chunk_len;
  // per-chunk intermediates.  ev_rows: where the last LaunchEvents put its scratch rows inside
  // ev_scratch (smhip_events_raw reads them back for the per-stage tests)
  struct EvRows {
    const float *ps = nullptr, *pss = nullptr, *t1 = nullptr, *t2 = nullptr, *ev_mean = nullptr, *ev_stdv = nullptr;
    const uint32_t *n_peaks = nullptr;
    const uint16_t *peaks = nullptr;
    uint32_t n_chunks = 0, feat_stride = 0, row_stride = 0;
  } ev_rows;
  smhip::DevBuf ev_scratch, feat, stdv, results, nq, q_off, chunk_anchor_cnt, anchor_off, n_groups;
  // per-query
  std::vector<smhip::DevBuf> region_segs;  // the hit slots' segments (HitSlots) ...
  std::vector<void *> h_region_tab;        // ... their pointers as last uploaded to region_tab
  smhip::DevBuf region_tab;
  smhip::DevBuf q_count, q_ovf_off, ovf_list, ovf_pool, counters, qrec, walk_stack, q_keys, q_order, q_hist, q_nodes, q_points;
  // per-anchor
  smhip::DevBuf a_tpos, a_qpos, a_dist, a_grp, keys, s_tpos, s_qpos, s_dist, s_grp, score, pred, used,
      group_start, chain_tmp, chain_cnt, out_anchors, out_anchor_cnt, dl_off, dl_compact, sort_tmp;
  const void *dl_chain_ranked = nullptr;     // the last batch's ranked candidate chains (chain.hip), primaries first ...
  const uint32_t *dl_chain_off = nullptr;    // ... chunk c's at 3 * dl_chain_off[c] (smhip_map_download_chains)
  uint64_t cap_queries = 0, cap_anchors = 0;
  uint64_t pool_hint = 0;  // overflow-pool size a previous batch needed
  bool walk_exact_only = false;  // batches with many queries beyond the 5000-hit cap: no set walk (seed_lookup.hip)
  uint64_t seen_queries = 0, seen_over_cap = 0;  // ... counted over the set-walk batches of this index
  // Census of the resident index (seed_lookup.hip IndexCensus): hit counts of a sample of its own points used as
  // queries at the default radius, sorted; taken on the first smhip_map_run after the index arrived (upload, attach
  // or broadcast alike).  Decides the automatic search mode and the automatic number of fast hit slots.
  std::vector<uint32_t> census_hits;
  bool census_valid = false;
  float census_radius = 0;
  smhip::DevKnobs knobs;
  // speculative batches (api.hip RunBatchSpeculative): queries / non-carried anchors per chunk of the largest
  // batch so far, the pinned record the stream reports a batch's totals and flags in, what a repeat needs
  uint64_t hint_q_per_chunk = 0, hint_a_per_chunk = 0;
  // the largest query / anchor capacities the grow-only buffers have been reserved for (res_fast: the fast hit slots per
  // query the query-sized ones were laid out with): a later, smaller batch may use all of it -- a batch of the junk
  // chunks of a streaming run has three times the queries per chunk its predecessors had
  uint64_t res_nq = 0, res_na = 0;
  uint32_t res_fast = 0;
  bool hint_valid = false, spec_pending = false;
  void *h_run_info = nullptr;
  void *h_stage = nullptr;   // pinned staging of small transfers to / from pageable caller memory (api.hip StageReserve)
  size_t h_stage_cap = 0;
  // recorded behind the last host-to-device copy that READS h_stage / h_carried: the host waits for it before it writes
  // either again or frees one of them (api.hip StageWait) -- a caller that selects its next batch without having
  // collected the previous one gets a wait, not a torn copy
  hipEvent_t stage_read = nullptr;
  bool stage_read_pending = false;
  // smhip_reads_upload_pieces found a piece whose filtered length the host and the device do not agree on: every call
  // that collects results fails until the next smhip_reads_upload_pieces / smhip_batch_select / smhip_batch_upload
  bool pieces_bad = false;
  smhip_params spec_params{};
  smhip_params last_params{};  // the parameters of the last smhip_map_run with every automatic choice resolved
  uint32_t spec_n_refs = 0, spec_reruns = 0, spec_runs = 0;
  uint32_t refine_band = 0, refine_pad = 0;
  // sDTW refinement
  smhip::DevBuf ref_sig, ref_off, refine_dist, refine_end;
  const float *d_ref_sig = nullptr;     // the reference signal in use: this context's (ref_sig) or one borrowed
  const uint64_t *d_ref_off = nullptr;  // from another context of the GPU (smhip_reference_share)
  // DTW-with-path refinement (smhip_refine_path_run): per-chunk / per-row window arrays and results
  smhip::DevBuf rp_rows, rp_cap, rp_toff, rp_poff, rp_tgt, rp_qoff, rp_qlen, rp_lo, rp_hi, rp_width, rp_celloff, rp_codes,
      rp_dist, rp_end, rp_path, rp_plen, rp_start;
  uint64_t rp_path_total = 0;
  bool refine_path_done = false;
  uint32_t ref_n = 0;
  bool refine_done = false;
  // host staging
  std::vector<uint64_t> h_anchor_off;
  // timing
  bool timing = false;
  std::vector<smhip::KernelTimer> timers;
  struct Pending { int timer; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace smhip {
// Per-stage device timers (smhip_timing_get names); a ScopedTimer brackets what is enqueued on the
// context stream during its lifetime with two HIP events.
constexpr const char *kTimerNames[] = {"events", "seed_order", "seed_lookup", "sort", "chain", "sdtw", "convert"};
enum { T_EVENTS = 0, T_ORDER, T_SEED, T_SORT, T_CHAIN, T_SDTW, T_CONVERT, T_COUNT };
struct ScopedTimer {
  smhip_ctx *ctx;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(smhip_ctx *c, int i) : ctx(c), id(i) {
    if (!ctx->timing) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!ctx->event_pool.empty()) {
        e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, ctx->stream);
  }
  ~ScopedTimer() {
    if (!ctx->timing) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({id, a, b});
  }
};

// seed_lookup.hip: fills ctx->census_hits (see smhip_ctx) with the hit counts of `samples` points of the index
hipError_t IndexCensus(smhip_ctx *ctx, const smhip_params &p, uint32_t samples);
// Per-device kernel attributes (dynamic LDS opt-ins): called by smhip_init for the context's device.
hipError_t InitSortAttributes();
hipError_t InitSdtwAttributes();
// Launchers implemented in the .hip files.  All enqueue on ctx->stream and return the
// launch status.
hipError_t LaunchLeafGather(smhip_ctx *ctx, int dim, const float *values, const uint64_t *positions,
                            const uint32_t *vind, const uint32_t *leaf_left, const uint32_t *leaf_tab,
                            uint64_t n_leaves, float *leaf_pts, uint32_t *slot_vind, float *leaf_box);
hipError_t LaunchInnerBoxes(smhip_ctx *ctx, const void *inner, const uint32_t *level_nodes, uint32_t count,
                            const float *leaf_box, float *inner_box);
// grid = {base[6], step[6]} from the root's tight box; then the children's boxes of every inner record
hipError_t LaunchPackBoxes(smhip_ctx *ctx, void *inner, uint64_t n_inner, uint32_t root_ref, const float *leaf_box,
                           const float *inner_box, float *grid, uint32_t *error_flag);
hipError_t LaunchEvents(smhip_ctx *ctx, const smhip_params &p, const float *samples, const uint64_t *cbegin,
                        const uint32_t *clen, uint32_t n_chunks, uint32_t feat_stride, float *feat, float *stdv,
                        smhip_chunk_result *results, uint32_t *nq);
hipError_t LaunchFeaturesIn(smhip_ctx *ctx, const smhip_params &p, const float *src, const uint64_t *cbegin,
                            const uint32_t *clen, uint32_t n_chunks, uint32_t feat_stride, float *feat,
                            smhip_chunk_result *results, uint32_t *nq);
hipError_t LaunchRestoreFront(smhip_ctx *ctx, uint32_t n_chunks, smhip_chunk_result *results, uint32_t *nq);
// event spans (start, length) of the last LaunchEvents' raw events, event.h:200-224
hipError_t LaunchEventSpans(smhip_ctx *ctx, const uint32_t *clen, uint32_t *ev_start, uint32_t *ev_length);
// signal.hip: raw ADC -> filtered pA of n_reads reads (n_segs segments in all) into their slots
uint32_t RawSegments(uint32_t n_raw);
hipError_t LaunchRawConvert(smhip_ctx *ctx, const int16_t *raw, const RawReadMeta *meta, const uint32_t *seg_start,
                            uint32_t n_reads, uint32_t n_segs, uint32_t *seg_count, uint32_t *seg_base, SlotView pa,
                            uint32_t *slot_len, uint32_t *read_len);
hipError_t LaunchGatherPieces(smhip_ctx *ctx, const PieceSrc *src, uint32_t n, void *staged);
hipError_t LaunchCheckPieceLen(smhip_ctx *ctx, const uint32_t *read_len, const uint32_t *expect, uint32_t n, uint32_t *host_flag);
hipError_t LaunchSelectChunks(smhip_ctx *ctx, const uint32_t *slots, const uint32_t *chunk_index, uint32_t n_chunks,
                              uint32_t chunk_size, SlotView pa, const uint32_t *slot_len, uint64_t *cbegin,
                              uint32_t *clen);
hipError_t LaunchScanU32(smhip_ctx *ctx, const uint32_t *in, uint32_t n, uint32_t *out_excl /*n+1*/);
hipError_t BuildQuads(smhip_ctx *ctx);  // seed_lookup.hip: ctx->index.quads from ctx->index.inner
hipError_t LaunchQueryExpand(smhip_ctx *ctx, const smhip_params &p, const float *feat, uint32_t feat_stride,
                             const uint32_t *q_off, uint32_t n_chunks, void *qrec);
hipError_t LaunchSeedLookup(smhip_ctx *ctx, const smhip_params &p, const void *qrec, const uint32_t *q_off,
                            uint32_t n_chunks, uint64_t n_queries_max, HitSlots region, uint32_t *q_count,
                            uint32_t *q_nodes, uint32_t *q_points, uint32_t *counters, Hit *pool,
                            uint32_t *ovf_tab, uint64_t pool_cap);
// explicit query list variant used by smhip_seed_lookup (queries given as n x 6 floats)
hipError_t LaunchSeedLookupRaw(smhip_ctx *ctx, const smhip_params &p, const float *queries, uint32_t n,
                               uint32_t cap, HitSlots region, uint32_t *q_count, uint32_t *q_nodes,
                               uint32_t *q_points, uint32_t *counters, void *qrec);
hipError_t LaunchAnchorCount(smhip_ctx *ctx, const smhip_params &p, const uint32_t *q_off,
                             const uint32_t *q_count, const uint32_t *q_nodes, const uint32_t *q_points,
                             const uint32_t *carried_off, uint32_t n_chunks, uint32_t *chunk_cnt,
                             smhip_chunk_result *results);
hipError_t LaunchScanU32ToU64(smhip_ctx *ctx, const uint32_t *in, uint32_t n, uint64_t *out_excl);
struct SortArgs {
  const uint32_t *q_off, *q_count, *ovf_tab, *qstart, *carried_off;
  const smhip_carried_anchor *carried;
  HitSlots region;
  const Hit *ovf_pool;
  const uint64_t *anchor_off;
  uint32_t fast_cap, nnp, step, n_refs;
  const uint32_t *span_bound;  // device: largest target position of the index and of this batch's carried anchors
  uint32_t *a_tpos, *a_qpos, *a_grp;
  float *a_dist;
  uint64_t *keys;
  uint32_t *s_tpos, *s_qpos, *s_grp;
  float *s_dist;
  uint32_t *group_start, *n_groups;
  smhip_chunk_result *results;
};
hipError_t LaunchSort(smhip_ctx *ctx, const SortArgs &a, uint32_t n_chunks, uint64_t n_anchors_total);
uint64_t SortRecordRoom(uint64_t n_anchors_total, uint32_t n_chunks);  // 16-byte records of SortArgs::keys
hipError_t LaunchIndexMaxTarget(smhip_ctx *ctx, uint32_t *out);
hipError_t LaunchCarriedMaxTarget(smhip_ctx *ctx, const smhip_carried_anchor *carried, uint64_t n_carried,
                                  const uint32_t *index_max, uint32_t *out);
struct ChainArgs {
  const uint64_t *anchor_off;
  const uint32_t *s_tpos, *s_qpos, *s_grp;
  const float *s_dist;
  const uint32_t *group_start, *n_groups;
  float *score;
  uint32_t *pred;
  uint8_t *used;
  ChainTmp *chain_tmp;           // candidate chains, chunk c at 3 * chain_off[c] (3 per bucket); set by LaunchChain
  ChainTmp *chain_sorted;        // same layout: ranked copy for chunks with > kChainCap candidates
  const uint32_t *chain_off;     // buckets before chunk c
  uint64_t *scratch;  // the sort's key buffer (2 keys per anchor), dead by now
  uint32_t *pending;  // per chunk: anchors that need the sequential DP, in index order (dead a_tpos)
  uint32_t *lists;    // per chunk: anchor index lists of the traced chains (dead a_qpos)
  uint64_t *item_lo, *item_hi;  // descriptors of the DP's items (dead a_dist / a_grp): chunk c's at [anchor_off[c] / 2, ..)
  smhip_chunk_result *results;
  smhip_carried_anchor *out_anchors;
  uint32_t *out_anchor_cnt;
};
hipError_t LaunchChain(smhip_ctx *ctx, const smhip_params &p, const ChainArgs &a, uint32_t n_chunks,
                       uint64_t n_anchors_total, uint32_t n_refs);
hipError_t LaunchChainCount(smhip_ctx *ctx, const smhip_chunk_result *results, uint32_t n_chunks, uint32_t *cnt);
hipError_t LaunchChainGather(smhip_ctx *ctx, const smhip_chunk_result *results, uint32_t n_chunks, const uint64_t *out_off,
                             smhip_chain *out);
hipError_t LaunchCwtFeatures(smhip_ctx *ctx, const float *samples, const uint64_t *cbegin, const uint32_t *clen,
                             uint32_t n_chunks,
                             const double *h, uint32_t L, float *cwt_rows, uint32_t feat_stride, float *feat,
                             uint32_t *pos, uint32_t *n_feat);
hipError_t LaunchSdtwChains(smhip_ctx *ctx, const smhip_chunk_result *results, const float *feat,
                            uint32_t feat_stride, const uint32_t *qstart, uint32_t n_chunks, const float *ref_sig,
                            const uint64_t *ref_off, uint32_t n_refs, uint32_t band, uint32_t pad, float *dist,
                            int64_t *end_pos);
hipError_t LaunchDtwPath(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                         const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, const uint64_t *cell_off,
                         uint8_t *codes, uint32_t n, float *dist, int64_t *end, uint8_t *path, const uint64_t *path_off,
                         uint32_t *path_len, int64_t *start_t, uint32_t max_t);
hipError_t LaunchChainWindowRows(smhip_ctx *ctx, const smhip_chunk_result *results, const uint32_t *qstart,
                                 uint32_t n_chunks, const uint64_t *ref_off, uint32_t n_refs, uint32_t pad, uint32_t *rows,
                                 uint32_t *path_cap);
hipError_t LaunchChainWindowFill(smhip_ctx *ctx, const smhip_chunk_result *results, const uint32_t *qstart,
                                 uint32_t feat_stride, uint32_t n_chunks, const uint64_t *ref_off, uint32_t n_refs,
                                 uint32_t band, uint32_t pad, const uint64_t *t_off, uint64_t *tgt_off, uint64_t *q_off,
                                 uint32_t *q_len, uint32_t *lo, uint32_t *hi, uint32_t *width);
hipError_t LaunchDtwPathChains(smhip_ctx *ctx, const float *ref_sig, const uint64_t *t_off, const uint64_t *tgt_off,
                               const float *feat, const uint64_t *q_off, const uint32_t *q_len, const uint32_t *lo,
                               const uint32_t *hi, const uint64_t *cell_off, uint8_t *codes, uint32_t n, float *dist,
                               int64_t *end, uint8_t *path, const uint64_t *path_off, uint32_t *path_len,
                               int64_t *start_t);
hipError_t LaunchSdtw(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                      const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, uint32_t n,
                      float *dist, int64_t *end, uint32_t max_t);
}  // namespace smhip

#endif
