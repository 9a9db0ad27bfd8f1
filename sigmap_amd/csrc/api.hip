// api.hip -- the C ABI of include/sigmap_hip.h: context, index residency, batch staging,
// kernel sequencing, timing.  Host code only (compiled by hipcc for the runtime API).
#include <string.h>

#include <algorithm>

#include <stdlib.h>

#include <math.h>

#include <dlfcn.h>

#include <map>
#include <mutex>

#include <chrono>

#include "common.h"

using namespace smhip;

namespace {
// name -> field of the development knobs (smhip_init reads SMHIP_<NAME>, smhip_dev_knob takes the lower-case name)
struct KnobName {
  const char *name;
  int DevKnobs::*field;
};
const KnobName kKnobNames[] = {{"lane_blocks_per_cu", &DevKnobs::lane_blocks_per_cu}, {"xcd_ranges", &DevKnobs::xcd_ranges},
                               {"scan_threshold", &DevKnobs::scan_threshold}, {"classes", &DevKnobs::classes},
                               {"bucket_bits", &DevKnobs::bucket_bits}, {"handout", &DevKnobs::handout},
                               {"sdtw_ring", &DevKnobs::sdtw_ring}, {"sync_run", &DevKnobs::sync_run},
                               {"spec_tight", &DevKnobs::spec_tight}, {"float_gap", &DevKnobs::float_gap},
                               {"sort_lines", &DevKnobs::sort_lines}, {"second_search", &DevKnobs::second_search},
                               {"classify_dense", &DevKnobs::classify_dense}, {"classify_resolve", &DevKnobs::classify_resolve},
                               {"walk_share", &DevKnobs::walk_share}, {"walk_grab", &DevKnobs::walk_grab},
                               {"walk_grab_upto", &DevKnobs::walk_grab_upto}, {"set_groups_per_cu", &DevKnobs::set_groups_per_cu}};
// the environment's knobs, honoured only behind SMHIP_DEV=1 (read once per context, in smhip_init)
DevKnobs ReadDevKnobs() {
  DevKnobs k;
  const char *dev = getenv("SMHIP_DEV");
  if (!dev || atoi(dev) != 1) return k;
  k.enabled = true;
  for (const KnobName &kn : kKnobNames) {
    std::string env = "SMHIP_";
    for (const char *c = kn.name; *c; ++c) env.push_back((char)toupper(*c));
    if (const char *e = getenv(env.c_str())) k.*(kn.field) = atoi(e);
  }
  if (const char *e = getenv("SMHIP_POOL_MIN")) k.pool_min = strtoll(e, nullptr, 10);
  return k;
}
}  // namespace

namespace {
int FinishRun(smhip_ctx *ctx);  // (defined with smhip_map_run: settles a speculative batch)
int ResolveParams(smhip_ctx *ctx, smhip_params *rp, uint32_t n_chunks);
int ReserveAnchorBuffers(smhip_ctx *ctx, uint64_t na_cap, uint32_t R);
int Fail(smhip_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->last_error = msg;
  return code;
}
// What a context has learnt from the batches of the index it held (search mode, capacities of the speculative route,
// overflow pool, the index census) does not carry over to another index.
void ForgetIndexHistory(smhip_ctx *ctx) {
  ctx->walk_exact_only = false;
  ctx->seen_queries = ctx->seen_over_cap = 0;
  ctx->hint_q_per_chunk = ctx->hint_a_per_chunk = 0;
  ctx->hint_valid = false;
  ctx->pool_hint = 0;
  ctx->census_valid = false;
  ctx->census_hits.clear();
  ctx->slot_leaf_valid = false;
}
int HipFail(smhip_ctx *ctx, hipError_t e, const char *what) {
  return Fail(ctx, e == hipErrorOutOfMemory ? SMHIP_ENOMEM : SMHIP_EHIP,
              std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(expr)                                   \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return HipFail(ctx, _e, #expr); \
  } while (0)

// Pinned host staging of the context (grow-only): small transfers to and from caller-owned pageable memory go
// through it as ONE DMA each -- a hipMemcpyAsync on pageable memory is cut into many staged pieces, ~19 us of
// blit kernel apiece on the stream (a 1.8 MB result download: a dozen of them).
// The host is about to write h_stage / h_carried (or to free one): the copies that read them have finished.
hipError_t StageWait(smhip_ctx *ctx) {
  if (!ctx->stage_read_pending) return hipSuccess;
  ctx->stage_read_pending = false;
  return hipEventSynchronize(ctx->stage_read);
}
hipError_t StageMarkRead(smhip_ctx *ctx) {
  hipError_t e = hipEventRecord(ctx->stage_read, ctx->stream);
  if (e == hipSuccess) ctx->stage_read_pending = true;
  return e;
}

hipError_t StageReserve(smhip_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->h_stage_cap) return hipSuccess;
  hipError_t w = StageWait(ctx);  // (nothing may still be reading the block that is freed)
  if (w != hipSuccess) return w;
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  ctx->h_stage = nullptr;
  ctx->h_stage_cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault);
  if (e == hipSuccess) ctx->h_stage_cap = want;
  return e;
}

void ResolveTimers(smhip_ctx *ctx) {
  for (auto &p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->timers[p.timer].total_ms += ms;
      ctx->timers[p.timer].launches += 1;
    }
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
  }
  ctx->pending.clear();
}

__global__ void HitsToArraysKernel(const Hit *__restrict__ region, const uint32_t *__restrict__ counts,
                                   uint32_t n, uint32_t cap, const uint32_t *__restrict__ leaf_vind,
                                   uint64_t *__restrict__ idx, float *__restrict__ dist) {
  const uint64_t total = (uint64_t)n * cap;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t q = (uint32_t)(i / cap), k = (uint32_t)(i % cap);
    if (k < counts[q]) {
      const Hit h = region[i];
      idx[i] = leaf_vind[h.slot];
      dist[i] = h.dist;
    }
  }
}

__global__ void CompactAnchorsKernel(const smhip_carried_anchor *__restrict__ src,
                                     const uint64_t *__restrict__ anchor_off,
                                     const uint64_t *__restrict__ out_off, smhip_carried_anchor *__restrict__ dst) {
  const uint32_t c = blockIdx.x;
  const uint64_t b = anchor_off[c], o = out_off[c];
  const uint32_t n = (uint32_t)(out_off[c + 1] - o);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[o + i] = src[b + i];
}

// the same for a LIST of chunks (smhip_map_download_anchors_of): counts of the listed chunks, then their anchors
__global__ void ListedAnchorCountKernel(const uint32_t *__restrict__ list, uint32_t n, const uint32_t *__restrict__ cnt,
                                        uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cnt[list[i]];
}
__global__ void CompactListedAnchorsKernel(const smhip_carried_anchor *__restrict__ src, const uint64_t *__restrict__ anchor_off,
                                           const uint32_t *__restrict__ list, const uint64_t *__restrict__ out_off,
                                           smhip_carried_anchor *__restrict__ dst) {
  const uint32_t i = blockIdx.x;
  const uint64_t b = anchor_off[list[i]], o = out_off[i];
  const uint32_t n = (uint32_t)(out_off[i + 1] - o);
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) dst[o + k] = src[b + k];
}

__global__ void OrStatusKernel(smhip_chunk_result *results, uint32_t n, const uint32_t *counters) {
  uint32_t bits = 0;
  if (counters[0]) bits |= SMHIP_ST_STACK_FULL;
  if (counters[3]) bits |= SMHIP_ST_HIT_POOL_FULL;
  if (!bits) return;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    results[i].status |= bits;
}

// query start offsets + carried anchors of a batch (shared by smhip_batch_upload / smhip_batch_select)
int UploadBatchSideInputs(smhip_ctx *ctx, uint32_t n_chunks, const uint32_t *query_start_offsets,
                          const smhip_carried_anchor *carried, const uint32_t *carried_offsets) {
  const uint64_t n_carried = carried_offsets && n_chunks ? carried_offsets[n_chunks] : 0;
  // [query start offsets (n)][carried offsets (n + 1)]: one device block, filled by ONE copy (absent arrays: zeros)
  HIPCHK(ctx->qstart.Reserve((2 * (uint64_t)n_chunks + 2) * sizeof(uint32_t)));
  ctx->d_carried_off = ctx->qstart.as<uint32_t>() + n_chunks;
  HIPCHK(ctx->carried.Reserve(std::max<uint64_t>(n_carried, 1) * sizeof(smhip_carried_anchor)));
  // small arrays through the pinned staging (words [2 n, 4 n + 2) of it: smhip_batch_select keeps its own two
  // arrays in front); the anchors themselves -- large, and only present in later chunk rounds -- directly
  HIPCHK(StageReserve(ctx, (4 * (size_t)n_chunks + 8) * sizeof(uint32_t)));
  uint32_t *st = static_cast<uint32_t *>(ctx->h_stage) + 2 * (size_t)n_chunks;
  HIPCHK(StageWait(ctx));  // (a no-op behind smhip_batch_select, which waited before it wrote the words in front)
  if (query_start_offsets) memcpy(st, query_start_offsets, (size_t)n_chunks * sizeof(uint32_t));
  else memset(st, 0, (size_t)n_chunks * sizeof(uint32_t));
  if (carried) memcpy(st + n_chunks, carried_offsets, ((size_t)n_chunks + 1) * sizeof(uint32_t));
  else memset(st + n_chunks, 0, ((size_t)n_chunks + 1) * sizeof(uint32_t));
  HIPCHK(hipMemcpyAsync(ctx->qstart.p, st, (2 * (uint64_t)n_chunks + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  if (carried) {
    // the anchors through pinned memory of the context: one DMA, and the caller's array is free on return without
    // anybody waiting for the stream.  (The staging is reused by the next batch's call: by then this batch has been
    // collected -- every route to its results synchronises the stream.)
    const size_t bytes = (size_t)n_carried * sizeof(smhip_carried_anchor);
    if (bytes > ctx->h_carried_cap) {  // (nothing reads the old block any more: StageWait above)
      if (ctx->h_carried) (void)hipHostFree(ctx->h_carried);
      ctx->h_carried = nullptr;
      ctx->h_carried_cap = 0;
      const size_t want = bytes + bytes / 2 + 4096;
      HIPCHK(hipHostMalloc(&ctx->h_carried, want, hipHostMallocDefault));
      ctx->h_carried_cap = want;
    }
    if (bytes) {
      memcpy(ctx->h_carried, carried, bytes);
      HIPCHK(hipMemcpyAsync(ctx->carried.p, ctx->h_carried, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  HIPCHK(StageMarkRead(ctx));
  return SMHIP_OK;
}

// The hit slots of nq queries (fast_cap each) as segments of <= 2 GiB; the pointer table goes to the device
// when a segment was (re)allocated.
int ReserveRegion(smhip_ctx *ctx, uint64_t nq, uint32_t fast_cap, HitSlots *out) {
  uint32_t qshift = 0;
  while (qshift < 31 && ((uint64_t)2 << qshift) * fast_cap * sizeof(Hit) <= (1ull << 31)) ++qshift;
  const uint64_t per_seg = 1ull << qshift;
  const size_t n_seg = (size_t)std::max<uint64_t>((nq + per_seg - 1) / per_seg, 1);
  if (ctx->region_segs.size() < n_seg) ctx->region_segs.resize(n_seg);
  std::vector<void *> tab(ctx->region_segs.size(), nullptr);
  for (size_t k = 0; k < n_seg; ++k) {
    const uint64_t q_here = std::min<uint64_t>(per_seg, nq - std::min<uint64_t>(nq, k * per_seg));
    HIPCHK(ctx->region_segs[k].Reserve(std::max<uint64_t>(q_here, 1) * fast_cap * sizeof(Hit)));
  }
  for (size_t k = 0; k < ctx->region_segs.size(); ++k) tab[k] = ctx->region_segs[k].p;
  if (tab != ctx->h_region_tab) {
    HIPCHK(ctx->region_tab.Reserve(tab.size() * sizeof(void *)));
    ctx->h_region_tab = tab;
    HIPCHK(hipMemcpyAsync(ctx->region_tab.p, ctx->h_region_tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice,
                          ctx->stream));
  }
  *out = HitSlots{ctx->region_tab.as<Hit *>(), qshift, fast_cap};
  return SMHIP_OK;
}
}  // namespace

extern "C" {

void smhip_default_params(smhip_params *p) {
  memset(p, 0, sizeof *p);
  p->window_length1 = 3;  // event.h:31-37
  p->window_length2 = 6;
  p->threshold1 = 4.30265f;
  p->threshold2 = 2.57058f;
  p->peak_height = 1.0f;
  p->compress_delta = 0.1;   // sigmap.cc:1075
  p->min_num_events = 50;    // sigmap.cc:660
  p->search_radius = 0.08f;  // sigmap.cc:1380
  p->step_size = 2;          // sigmap.cc:1384
  p->max_gap_length = 2000;  // spatial_index.cc:284-291
  p->max_target_gap_length = 5000;
  p->chaining_band_length = 5000;
  p->max_num_skips = 25;
  p->min_num_anchors = 2;
  p->num_best_chains = 3;
  p->num_nearest_points = 5000;
  p->min_chaining_score = 10;
  p->hits_per_query_fast = 0;  // automatic: from the census of the resident index (512 .. 2048)
  p->walk_blocks_per_cu = 0;
  p->search_mode = SMHIP_SEARCH_AUTO;
  p->set_walk_max_over_cap = 0.0005f;
}

int smhip_init(int device_ordinal, smhip_ctx **out) {
  if (!out) return SMHIP_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SMHIP_EHIP;
  if (device_ordinal < 0 || device_ordinal >= n) return SMHIP_EINVAL;
  if (hipSetDevice(device_ordinal) != hipSuccess) return SMHIP_EHIP;
  // dynamic-LDS opt-ins are per device: set them for this context's device (idempotent)
  if (InitSortAttributes() != hipSuccess || InitSdtwAttributes() != hipSuccess) return SMHIP_EHIP;
  smhip_ctx *ctx = new smhip_ctx;
  ctx->device = device_ordinal;
  ctx->knobs = ReadDevKnobs();
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return SMHIP_EHIP;
  }
  if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->copy_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->convert_done, hipEventDisableTiming) != hipSuccess) {
    smhip_destroy(ctx);
    return SMHIP_EHIP;
  }
  if (hipHostMalloc(&ctx->h_run_info, 256, hipHostMallocDefault) != hipSuccess) {  // RunInfo of the speculative batches
    ctx->h_run_info = nullptr;
    smhip_destroy(ctx);
    return SMHIP_EHIP;
  }
  {  // smhip_reads_upload_pieces: the event behind its staging copies, the mismatch word the device writes
    void *flag = nullptr;
    if (hipEventCreateWithFlags(&ctx->pieces_staged, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->stage_read, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(&flag, 64, hipHostMallocDefault) != hipSuccess) {
      smhip_destroy(ctx);
      return SMHIP_EHIP;
    }
    ctx->h_piece_flag = static_cast<uint32_t *>(flag);
    *ctx->h_piece_flag = 0;
  }
  ctx->timers.resize(T_COUNT);
  for (int i = 0; i < T_COUNT; ++i) ctx->timers[i].name = kTimerNames[i];
  *out = ctx;
  return SMHIP_OK;
}

void smhip_destroy(smhip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->aux_stream) {
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux_stream);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->copy_done) (void)hipEventDestroy(ctx->copy_done);
  if (ctx->convert_done) (void)hipEventDestroy(ctx->convert_done);
  if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
  if (ctx->h_pieces) (void)hipHostFree(ctx->h_pieces);
  if (ctx->h_piece_flag) (void)hipHostFree(ctx->h_piece_flag);
  if (ctx->h_carried) (void)hipHostFree(ctx->h_carried);
  if (ctx->pieces_staged) (void)hipEventDestroy(ctx->pieces_staged);
  if (ctx->stage_read) (void)hipEventDestroy(ctx->stage_read);
  ctx->pieces_src.Free();
  ctx->pieces_expect.Free();
  if (ctx->h_run_info) (void)hipHostFree(ctx->h_run_info);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (DevBuf &b : ctx->pa_segs) b.Free();
  DevBuf *bufs[] = {&ctx->reads_raw, &ctx->reads_meta, &ctx->reads_segstart, &ctx->reads_seg, &ctx->pa_tab,
                    &ctx->slot_len, &ctx->reads_len, &ctx->sel_slots, &ctx->sel_chunk, &ctx->chunk_begin, &ctx->chunk_len,
                    &ctx->dl_off, &ctx->dl_compact, &ctx->sort_tmp, &ctx->rp_rows, &ctx->rp_cap, &ctx->rp_toff, &ctx->rp_poff, &ctx->rp_tgt,
                    &ctx->rp_qoff, &ctx->rp_qlen, &ctx->rp_lo, &ctx->rp_hi, &ctx->rp_width, &ctx->rp_celloff, &ctx->rp_codes,
                    &ctx->rp_dist, &ctx->rp_end, &ctx->rp_path, &ctx->rp_plen, &ctx->rp_start,
                    &ctx->scan_tmp, &ctx->ix_slot_leaf, &ctx->ix_quads, &ctx->ix_nodes, &ctx->ix_tab, &ctx->ix_pts, &ctx->qrec, &ctx->walk_stack, &ctx->ev_scratch, &ctx->ref_sig, &ctx->ref_off, &ctx->refine_dist, &ctx->refine_end, &ctx->q_keys, &ctx->q_order, &ctx->q_hist, &ctx->q_nodes, &ctx->q_points, &ctx->ix_vind, &ctx->ix_bbox, &ctx->samples,
                    &ctx->qstart, &ctx->carried, &ctx->carried_off, &ctx->feat, &ctx->stdv,
                    &ctx->results, &ctx->nq, &ctx->q_off, &ctx->chunk_anchor_cnt, &ctx->anchor_off,
                    &ctx->n_groups, &ctx->q_count, &ctx->q_ovf_off, &ctx->region_tab, &ctx->ovf_list,
                    &ctx->ovf_pool, &ctx->counters, &ctx->max_tpos, &ctx->a_tpos, &ctx->a_qpos, &ctx->a_dist, &ctx->a_grp,
                    &ctx->keys, &ctx->s_tpos, &ctx->s_qpos, &ctx->s_dist, &ctx->s_grp, &ctx->score, &ctx->pred,
                    &ctx->used, &ctx->group_start, &ctx->chain_tmp, &ctx->chain_cnt, &ctx->out_anchors,
                    &ctx->out_anchor_cnt};
  for (DevBuf *b : bufs) b->Free();
  for (DevBuf &b : ctx->region_segs) b.Free();
  for (auto &p : ctx->pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *smhip_last_error(smhip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }
void *smhip_stream(smhip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int smhip_index_upload(smhip_ctx *ctx, int dim, const float *values, const uint64_t *positions,
                       uint64_t n_points, const uint32_t *vind, uint64_t m_size, const smhip_node *nodes,
                       uint64_t n_nodes, const float *root_bbox) {
  if (!ctx || !values || !positions || !vind || !nodes || !root_bbox) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (dim != kMaxDim) return Fail(ctx, SMHIP_EINVAL, "device kernels are built for index dimension 6");
  if (n_points < (uint64_t)dim || m_size != n_points - dim + 1 || n_nodes == 0 || n_nodes > 0xFFFFFFFFull)
    return Fail(ctx, SMHIP_EINVAL, "inconsistent index sizes");
  HIPCHK(hipSetDevice(ctx->device));
  // Walker structures (layouts: seed_lookup.hip header) from the pre-order node array: one
  // 64-byte record per INNER node, one table word and one point block of roundup(n, 8) 32-byte
  // slots (pairs of 128-byte lines: the table word has 27 bits for the block number) per leaf.  The boxes are computed on the device: leaves by the
  // gather, inner nodes level by level from the deepest one up, then quantised into the parents.
  std::vector<uint32_t> ref_of(n_nodes), depth(n_nodes, 0);
  uint64_t n_inner = 0, n_leaves = 0, n_slots = 0, leaf_points = 0;
  uint32_t max_depth = 0;
  // Record numbers: leaves by ordinal in tree order, inner nodes in pre-order among the inner nodes (a
  // node's first child is the next record, i.e. in the same 128-byte L2 line half of the time; numbering
  // siblings as adjacent, line-aligned pairs instead was measured and lost: DESIGN.md section 10).
  for (uint64_t i = 0; i < n_nodes; ++i) {
    if (nodes[i].feat >= 0) ref_of[i] = (uint32_t)n_inner++;
    else ref_of[i] = kLeafTag | (uint32_t)n_leaves++;
  }
  if (n_inner >= (1ull << 27) || n_leaves >= (1ull << 27))
    return Fail(ctx, SMHIP_EINVAL, "index too large for 27-bit node references");
  std::vector<uint32_t> h_inner(std::max<uint64_t>(n_inner, 1) * 16, 0), leaf_left(n_leaves), leaf_tab(n_leaves);
  for (uint64_t i = 0; i < n_nodes; ++i) {
    const smhip_node &nd = nodes[i];
    if (nd.feat >= 0) {
      if (nd.feat >= dim || i + 1 >= n_nodes || nd.child2 >= n_nodes || nd.child2 <= i + 1)
        return Fail(ctx, SMHIP_EINVAL, "malformed inner node (children must follow in pre-order)");
      uint32_t *o = &h_inner[(size_t)ref_of[i] * 16];
      o[0] = nd.a;
      o[1] = nd.b;
      o[2] = ref_of[i + 1] | ((uint32_t)nd.feat << 28);
      o[3] = ref_of[nd.child2];
      depth[i + 1] = depth[nd.child2] = depth[i] + 1;
      max_depth = std::max(max_depth, depth[i] + 1);
    } else {
      if (nd.b < nd.a || nd.b > m_size) return Fail(ctx, SMHIP_EINVAL, "leaf range outside the point list");
      // the scan phase tests one point per lane, 21 lanes per leaf (the reference builds with
      // max_leaf = 20, sigmap.cc:1426)
      if (nd.b - nd.a > 21) return Fail(ctx, SMHIP_EINVAL, "leaf with more than 21 points");
      if (n_slots >= (1ull << 30)) return Fail(ctx, SMHIP_EINVAL, "index too large for 27-bit leaf block numbers");
      const uint32_t n = nd.b - nd.a, l = ref_of[i] & ~kLeafTag;
      leaf_left[l] = nd.a;
      leaf_tab[l] = (uint32_t)(n_slots >> 3) | (n << 27);  // (32-byte slots, four to a line; blocks of two lines)
      n_slots += (n + 7u) & ~7u;
      leaf_points += n;
    }
  }
  if (leaf_points != m_size) return Fail(ctx, SMHIP_EINVAL, "leaves do not cover the point list exactly");
  // inner nodes grouped by depth (counting sort)
  std::vector<uint64_t> level_start(max_depth + 2, 0);
  for (uint64_t i = 0; i < n_nodes; ++i)
    if (nodes[i].feat >= 0) ++level_start[depth[i] + 1];
  for (uint32_t d = 0; d <= max_depth; ++d) level_start[d + 1] += level_start[d];
  std::vector<uint32_t> level_nodes(std::max<uint64_t>(n_inner, 1));
  {
    std::vector<uint64_t> cur(level_start.begin(), level_start.end() - 1);
    for (uint64_t i = 0; i < n_nodes; ++i)
      if (nodes[i].feat >= 0) level_nodes[cur[depth[i]]++] = ref_of[i];
  }
  const uint32_t root_ref = ref_of[0];
  ctx->leaf_pts_bytes = n_slots * 32 + 128;
  HIPCHK(ctx->ix_nodes.Reserve(std::max<uint64_t>(n_inner, 1) * 64));
  HIPCHK(ctx->ix_tab.Reserve(n_leaves * sizeof(uint32_t)));
  HIPCHK(ctx->ix_pts.Reserve(ctx->leaf_pts_bytes));
  HIPCHK(hipMemsetAsync(ctx->ix_pts.p, 0, ctx->leaf_pts_bytes, ctx->stream));
  ctx->leaf_vind_bytes = (n_slots + 32) * sizeof(uint32_t);
  HIPCHK(ctx->ix_vind.Reserve(ctx->leaf_vind_bytes));
  HIPCHK(hipMemsetAsync(ctx->ix_vind.p, 0, ctx->leaf_vind_bytes, ctx->stream));
  HIPCHK(ctx->ix_bbox.Reserve(4 * dim * sizeof(float)));
  // staging copies of the original-order point cloud and the exact boxes, freed after the build
  DevBuf d_values, d_positions, d_levels, d_vind, d_left, d_leaf_box, d_inner_box, d_flag;
  HIPCHK(d_values.Reserve(n_points * sizeof(float)));
  hipError_t e = d_positions.Reserve(n_points * sizeof(uint64_t));
  if (e == hipSuccess) e = d_levels.Reserve(level_nodes.size() * sizeof(uint32_t));
  if (e == hipSuccess) e = d_vind.Reserve(m_size * sizeof(uint32_t));
  if (e == hipSuccess) e = d_left.Reserve(n_leaves * sizeof(uint32_t));
  if (e == hipSuccess) e = d_leaf_box.Reserve(n_leaves * 12 * sizeof(float));
  if (e == hipSuccess) e = d_inner_box.Reserve(std::max<uint64_t>(n_inner, 1) * 12 * sizeof(float));
  if (e == hipSuccess) e = d_flag.Reserve(sizeof(uint32_t));
  auto cleanup = [&]() {
    for (DevBuf *b : {&d_values, &d_positions, &d_levels, &d_vind, &d_left, &d_leaf_box, &d_inner_box, &d_flag})
      b->Free();
  };
  if (e != hipSuccess) {
    cleanup();
    return HipFail(ctx, e, "staging allocation");
  }
#define HIPCHK2(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      cleanup();                                       \
      return HipFail(ctx, _e, #expr);                  \
    }                                                  \
  } while (0)
  HIPCHK2(hipMemcpyAsync(d_values.p, values, n_points * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(d_positions.p, positions, n_points * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(d_vind.p, vind, m_size * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(ctx->ix_nodes.p, h_inner.data(), h_inner.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(ctx->ix_tab.p, leaf_tab.data(), n_leaves * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(d_left.p, leaf_left.data(), n_leaves * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(d_levels.p, level_nodes.data(), level_nodes.size() * sizeof(uint32_t),
                         hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemcpyAsync(ctx->ix_bbox.p, root_bbox, 2 * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK2(hipMemsetAsync(d_flag.p, 0, sizeof(uint32_t), ctx->stream));
  HIPCHK2(LaunchLeafGather(ctx, dim, d_values.as<float>(), d_positions.as<uint64_t>(), d_vind.as<uint32_t>(),
                           d_left.as<uint32_t>(), ctx->ix_tab.as<uint32_t>(), n_leaves, ctx->ix_pts.as<float>(),
                           ctx->ix_vind.as<uint32_t>(), d_leaf_box.as<float>()));
  for (uint32_t d = max_depth + 1; d-- > 0;)
    HIPCHK2(LaunchInnerBoxes(ctx, ctx->ix_nodes.p, d_levels.as<uint32_t>() + level_start[d],
                             (uint32_t)(level_start[d + 1] - level_start[d]), d_leaf_box.as<float>(),
                             d_inner_box.as<float>()));
  HIPCHK2(LaunchPackBoxes(ctx, ctx->ix_nodes.p, n_inner, root_ref, d_leaf_box.as<float>(), d_inner_box.as<float>(),
                          ctx->ix_bbox.as<float>() + 2 * dim, d_flag.as<uint32_t>()));
  uint32_t pack_error = 0;
  HIPCHK2(hipMemcpyAsync(&pack_error, d_flag.p, sizeof pack_error, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK2(hipStreamSynchronize(ctx->stream));
#undef HIPCHK2
  cleanup();
  if (pack_error) return Fail(ctx, SMHIP_EINVAL, "point coordinates cannot be boxed on the 16-bit grid (non-finite values?)");
  ctx->dim = dim;
  ctx->index = IndexView{ctx->ix_nodes.as<uint4>(), ctx->ix_tab.as<uint32_t>(), ctx->ix_pts.as<float>(),
                         ctx->ix_vind.as<uint32_t>(), ctx->ix_bbox.as<float>(), n_inner, n_leaves, m_size, root_ref};
  ctx->have_index = true;
  ctx->max_tpos_valid = false;
  ctx->index_owned = true;
  ForgetIndexHistory(ctx);
  {
    const hipError_t qe = BuildQuads(ctx);  // the set walk's node array (seed_lookup.hip)
    if (qe != hipSuccess) {
      ctx->have_index = false;
      return HipFail(ctx, qe, "quad records");
    }
  }
  return SMHIP_OK;
}

int smhip_index_export(smhip_ctx *ctx, smhip_index_dev *out) {
  if (!ctx || !out) return SMHIP_EINVAL;
  if (!ctx->have_index) return Fail(ctx, SMHIP_ESTATE, "no index resident");
  memset(out, 0, sizeof *out);
  out->dim = ctx->dim;
  out->root_ref = ctx->index.root_ref;
  out->n_nodes = ctx->index.n_inner;
  out->n_leaves = ctx->index.n_leaves;
  out->m_size = ctx->index.m_size;
  out->nodes = (void *)ctx->index.inner;
  out->nodes_bytes = std::max<uint64_t>(ctx->index.n_inner, 1) * 64;
  out->leaf_tab = (void *)ctx->index.leaf_tab;
  out->leaf_tab_bytes = ctx->index.n_leaves * sizeof(uint32_t);
  out->leaf_pts = (void *)ctx->index.leaf_pts;
  out->leaf_pts_bytes = ctx->leaf_pts_bytes;
  out->leaf_vind = (void *)ctx->index.leaf_vind;
  out->leaf_vind_bytes = ctx->leaf_vind_bytes;
  out->bbox = (void *)ctx->index.bbox;
  out->bbox_bytes = 4 * ctx->dim * sizeof(float);
  out->quads = (void *)ctx->index.quads;
  out->quads_bytes = ctx->index.n_quads * 128;
  out->n_quads = ctx->index.n_quads;
  return SMHIP_OK;
}

int smhip_index_export_copy(smhip_ctx *ctx, int array_id, void *dst_device, uint64_t dst_bytes) {
  if (!ctx || !dst_device) return Fail(ctx, SMHIP_EINVAL, "null argument");
  smhip_index_dev d;
  int rc = smhip_index_export(ctx, &d);
  if (rc) return rc;
  const void *src[6] = {d.nodes, d.leaf_tab, d.leaf_pts, d.leaf_vind, d.bbox, d.quads};
  const uint64_t bytes[6] = {d.nodes_bytes, d.leaf_tab_bytes, d.leaf_pts_bytes, d.leaf_vind_bytes, d.bbox_bytes, d.quads_bytes};
  if (array_id < 0 || array_id > 5) return Fail(ctx, SMHIP_EINVAL, "array_id must be 0..5");
  if (dst_bytes < bytes[array_id]) return Fail(ctx, SMHIP_EINVAL, "destination smaller than the array");
  if (bytes[array_id] == 0) return SMHIP_OK;  // (an index without quad records)
  HIPCHK(hipSetDevice(ctx->device));
  // (hipMemcpyDefault: the destination may also be host memory -- the gloo test transport of bench.py)
  HIPCHK(hipMemcpyAsync(dst_device, src[array_id], bytes[array_id], hipMemcpyDefault, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SMHIP_OK;
}

// RCCL is loaded on first use: single-GPU runs never touch it.  (librccl.so.1; a process that has
// torch loaded gets torch's copy, the same way it gets torch's HIP runtime.)
}  // extern "C"
namespace {
struct Rccl {
  typedef void *comm_t;
  int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};
// `p` now holds a broadcast copy of `root`'s walker arrays in its own ix_* buffers
void AdoptReplica(smhip_ctx *p, const smhip_ctx *root) {
  p->dim = root->dim;
  p->index = IndexView{p->ix_nodes.as<uint4>(), p->ix_tab.as<uint32_t>(), p->ix_pts.as<float>(),
                       p->ix_vind.as<uint32_t>(), p->ix_bbox.as<float>(), root->index.n_inner, root->index.n_leaves,
                       root->index.m_size, root->index.root_ref,
                       root->index.n_quads ? p->ix_quads.as<uint4>() : nullptr, root->index.n_quads};
  p->leaf_pts_bytes = root->leaf_pts_bytes;
  p->leaf_vind_bytes = root->leaf_vind_bytes;
  p->have_index = true;
  p->max_tpos_valid = false;
  p->index_owned = true;
  ForgetIndexHistory(p);
}
const Rccl &LoadRccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      r.why = std::string("cannot load librccl: ") + dlerror();
      return;
    }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(h, "ncclBroadcast"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.Broadcast && r.GroupStart && r.GroupEnd && r.GetErrorString;
    if (!r.ok) r.why = "librccl lacks an expected symbol";
  });
  return r;
}
}  // namespace
extern "C" {

int smhip_index_bcast(smhip_ctx *const *ctxs, int n_gpus, int root) {
  if (!ctxs || n_gpus < 1 || root < 0 || root >= n_gpus) return SMHIP_EINVAL;
  for (int i = 0; i < n_gpus; ++i)
    if (!ctxs[i]) return SMHIP_EINVAL;
  smhip_ctx *ctx = ctxs[root];  // errors are reported on the root context
  if (!ctx->have_index) return Fail(ctx, SMHIP_ESTATE, "smhip_index_bcast: the root context has no index");
  for (int i = 0; i < n_gpus; ++i)
    for (int j = i + 1; j < n_gpus; ++j)
      if (ctxs[i]->device == ctxs[j]->device)
        return Fail(ctx, SMHIP_EINVAL, "smhip_index_bcast needs one context per GPU (contexts of one GPU share the index "
                                       "with smhip_index_export / smhip_index_attach)");
  if (n_gpus == 1) return SMHIP_OK;
  const Rccl &nc = LoadRccl();
  if (!nc.ok) return Fail(ctx, SMHIP_EHIP, nc.why);
  smhip_index_dev d;
  int rc = smhip_index_export(ctx, &d);
  if (rc) return rc;
  // the call walks over the peers' devices: whatever the exit, the caller's current device is what it was
  struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  } guard;
  const uint64_t bytes[6] = {d.nodes_bytes, d.leaf_tab_bytes, d.leaf_pts_bytes, d.leaf_vind_bytes, d.bbox_bytes, d.quads_bytes};
  void *root_ptr[6] = {d.nodes, d.leaf_tab, d.leaf_pts, d.leaf_vind, d.bbox, d.quads};
  // peers: their own buffers for the six walker arrays
  for (int i = 0; i < n_gpus; ++i) {
    if (i == root) continue;
    smhip_ctx *p = ctxs[i];
    if (hipSetDevice(p->device) != hipSuccess) return Fail(ctx, SMHIP_EHIP, "hipSetDevice(peer)");
    DevBuf *b[6] = {&p->ix_nodes, &p->ix_tab, &p->ix_pts, &p->ix_vind, &p->ix_bbox, &p->ix_quads};
    for (int a = 0; a < 6; ++a) {
      hipError_t e = b[a]->Reserve(bytes[a]);
      if (e != hipSuccess) return HipFail(ctx, e, "peer index allocation");
    }
  }
  std::vector<Rccl::comm_t> comms((size_t)n_gpus, nullptr);
  std::vector<int> devs((size_t)n_gpus);
  for (int i = 0; i < n_gpus; ++i) devs[i] = ctxs[i]->device;
  int nr = nc.CommInitAll(comms.data(), n_gpus, devs.data());
  if (nr != 0) {
    for (auto c : comms)
      if (c) (void)nc.CommDestroy(c);
    return Fail(ctx, SMHIP_EHIP, std::string("ncclCommInitAll: ") + nc.GetErrorString(nr));
  }
  // ONE broadcast per array, all ranks of this process inside one group call.  Every rank names ITS OWN
  // buffer as both send and receive buffer (in place): the root's is the source, a peer's send buffer is ignored by the
  // collective's semantics -- but RCCL compares the two pointers (the in-place test) and registers what it is given, so
  // a peer never hands it a pointer into another GPU's memory
  for (int a = 0; a < 6 && nr == 0; ++a) {
    if (bytes[a] == 0) continue;  // (an index without quad records)
    nr = nc.GroupStart();
    for (int i = 0; i < n_gpus && nr == 0; ++i) {
      smhip_ctx *p = ctxs[i];
      DevBuf *b[6] = {&p->ix_nodes, &p->ix_tab, &p->ix_pts, &p->ix_vind, &p->ix_bbox, &p->ix_quads};
      void *own = i == root ? root_ptr[a] : b[a]->p;
      if (hipSetDevice(p->device) != hipSuccess) nr = 1;
      if (nr == 0) nr = nc.Broadcast(own, own, bytes[a], /*ncclUint8*/ 1, root, comms[i], p->stream);
    }
    const int ge = nc.GroupEnd();
    if (nr == 0) nr = ge;
  }
  hipError_t he = hipSuccess;
  for (int i = 0; i < n_gpus; ++i) {
    if (hipSetDevice(ctxs[i]->device) != hipSuccess) he = hipErrorInvalidDevice;
    const hipError_t e = hipStreamSynchronize(ctxs[i]->stream);
    if (he == hipSuccess) he = e;
  }
  for (auto c : comms)
    if (c) (void)nc.CommDestroy(c);
  if (nr != 0) return Fail(ctx, SMHIP_EHIP, std::string("RCCL broadcast: ") + nc.GetErrorString(nr));
  if (he != hipSuccess) return HipFail(ctx, he, "index broadcast");
  for (int i = 0; i < n_gpus; ++i)
    if (i != root) AdoptReplica(ctxs[i], ctx);
  return SMHIP_OK;
}

int smhip_index_bcast_loopback(smhip_ctx *src, smhip_ctx *dst) {
  if (!src || !dst || src == dst) return SMHIP_EINVAL;
  smhip_ctx *ctx = src;  // errors are reported on the source context
  if (!src->have_index) return Fail(ctx, SMHIP_ESTATE, "smhip_index_bcast_loopback: the source context has no index");
  if (src->device != dst->device)
    return Fail(ctx, SMHIP_EINVAL, "smhip_index_bcast_loopback: both contexts must be on one GPU (peers: smhip_index_bcast)");
  const Rccl &nc = LoadRccl();
  if (!nc.ok) return Fail(ctx, SMHIP_EHIP, nc.why);
  smhip_index_dev d;
  int rc = smhip_index_export(src, &d);
  if (rc) return rc;
  HIPCHK(hipSetDevice(src->device));
  const uint64_t bytes[6] = {d.nodes_bytes, d.leaf_tab_bytes, d.leaf_pts_bytes, d.leaf_vind_bytes, d.bbox_bytes, d.quads_bytes};
  void *from[6] = {d.nodes, d.leaf_tab, d.leaf_pts, d.leaf_vind, d.bbox, d.quads};
  DevBuf *to[6] = {&dst->ix_nodes, &dst->ix_tab, &dst->ix_pts, &dst->ix_vind, &dst->ix_bbox, &dst->ix_quads};
  for (int a = 0; a < 6; ++a) HIPCHK(to[a]->Reserve(bytes[a]));
  Rccl::comm_t comm = nullptr;
  const int dev = src->device;
  int nr = nc.CommInitAll(&comm, 1, &dev);
  if (nr != 0) return Fail(ctx, SMHIP_EHIP, std::string("ncclCommInitAll(1): ") + nc.GetErrorString(nr));
  // the same call sequence as smhip_index_bcast: one grouped ncclBroadcast per array; with one rank and
  // sendbuff != recvbuff RCCL copies the root's buffer into the receive buffer
  for (int a = 0; a < 6 && nr == 0; ++a) {
    if (bytes[a] == 0) continue;
    nr = nc.GroupStart();
    if (nr == 0) nr = nc.Broadcast(from[a], to[a]->p, bytes[a], /*ncclUint8*/ 1, 0, comm, src->stream);
    const int ge = nc.GroupEnd();
    if (nr == 0) nr = ge;
  }
  const hipError_t he = hipStreamSynchronize(src->stream);
  (void)nc.CommDestroy(comm);
  if (nr != 0) return Fail(ctx, SMHIP_EHIP, std::string("RCCL broadcast: ") + nc.GetErrorString(nr));
  if (he != hipSuccess) return HipFail(ctx, he, "index broadcast (loopback)");
  AdoptReplica(dst, src);
  return SMHIP_OK;
}

int smhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int smhip_index_attach(smhip_ctx *ctx, const smhip_index_dev *in) {
  if (!ctx || !in) return SMHIP_EINVAL;
  if (in->dim != kMaxDim) return Fail(ctx, SMHIP_EINVAL, "device kernels are built for index dimension 6");
  if (!in->nodes || !in->leaf_tab || !in->leaf_pts || !in->leaf_vind || !in->bbox || in->n_leaves == 0)
    return Fail(ctx, SMHIP_EINVAL, "null index array");
  if (in->nodes_bytes < in->n_nodes * 64 || in->leaf_tab_bytes < in->n_leaves * 4 ||
      in->leaf_pts_bytes < in->m_size * 32 || in->leaf_vind_bytes < in->m_size * 4 || in->bbox_bytes < 96)
    return Fail(ctx, SMHIP_EINVAL, "index array smaller than its declared element count");
  if (in->n_nodes + 1 != in->n_leaves || (in->n_nodes == 0) != (in->root_ref == kLeafTag) ||
      (in->n_nodes != 0 && in->root_ref != 0))
    return Fail(ctx, SMHIP_EINVAL, "inconsistent node counts / root reference");
  if ((in->quads == nullptr) != (in->n_quads == 0) || in->quads_bytes < in->n_quads * 128 || in->n_quads > in->n_nodes)
    return Fail(ctx, SMHIP_EINVAL, "inconsistent quad records");
  ctx->dim = in->dim;
  ctx->index = IndexView{(const uint4 *)in->nodes, (const uint32_t *)in->leaf_tab, (const float *)in->leaf_pts,
                         (const uint32_t *)in->leaf_vind, (const float *)in->bbox, in->n_nodes, in->n_leaves,
                         in->m_size, in->root_ref, (const uint4 *)in->quads, in->n_quads};
  ctx->leaf_pts_bytes = in->leaf_pts_bytes;
  ctx->leaf_vind_bytes = in->leaf_vind_bytes;
  ctx->have_index = true;
  ctx->max_tpos_valid = false;
  ctx->index_owned = false;
  ForgetIndexHistory(ctx);
  return SMHIP_OK;
}

}  // extern "C"
namespace {
// smhip_batch_upload (chunks of pA samples) and smhip_features_upload (chunks of features) share everything but the
// per-chunk size limit and what EnqueueFront does with the chunks
int UploadChunks(smhip_ctx *ctx, bool features, const float *samples_pa, const uint32_t *sample_offsets, uint32_t n_chunks,
                 const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                 const uint32_t *carried_offsets);
}  // namespace
extern "C" {
int smhip_batch_upload(smhip_ctx *ctx, const float *samples_pa, const uint32_t *sample_offsets, uint32_t n_chunks,
                       const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                       const uint32_t *carried_offsets) {
  return UploadChunks(ctx, false, samples_pa, sample_offsets, n_chunks, query_start_offsets, carried, carried_offsets);
}

int smhip_features_upload(smhip_ctx *ctx, const float *features, const uint32_t *feat_offsets, uint32_t n_chunks,
                          const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                          const uint32_t *carried_offsets) {
  return UploadChunks(ctx, true, features, feat_offsets, n_chunks, query_start_offsets, carried, carried_offsets);
}

int smhip_generate_chains(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, const float *features,
                          const uint32_t *feat_offsets, uint32_t n_chunks, const uint32_t *query_start_offsets,
                          const smhip_carried_anchor *carried, const uint32_t *carried_offsets,
                          smhip_chunk_result *results) {
  int rc = smhip_features_upload(ctx, features, feat_offsets, n_chunks, query_start_offsets, carried, carried_offsets);
  if (rc) return rc;
  rc = smhip_map_run(ctx, p, n_refs);
  if (rc) return rc;
  rc = smhip_map_sync(ctx);
  if (rc) return rc;
  return smhip_map_download(ctx, results, nullptr, 0, nullptr);
}

int smhip_chain_anchors(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, const smhip_carried_anchor *anchors,
                        const uint32_t *anchor_offsets, uint32_t n_chunks, smhip_chunk_result *results) {
  if (!ctx || !anchors || !anchor_offsets) return Fail(ctx, SMHIP_EINVAL, "null argument");
  // chunks without a feature: their only anchors are the ones handed in, which take the carried anchors' route
  // into the sort (spatial_index.cc:303-322 puts previous chains' anchors into the buckets, :411-417 sorts them)
  const std::vector<uint32_t> no_features((size_t)n_chunks + 1, 0u);
  const float none = 0.0f;
  return smhip_generate_chains(ctx, p, n_refs, &none, no_features.data(), n_chunks, nullptr, anchors, anchor_offsets, results);
}
}  // extern "C"
namespace {
int UploadChunks(smhip_ctx *ctx, bool features, const float *samples_pa, const uint32_t *sample_offsets, uint32_t n_chunks,
                 const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                 const uint32_t *carried_offsets) {
  if (!ctx || !samples_pa || !sample_offsets) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if ((carried == nullptr) != (carried_offsets == nullptr))
    return Fail(ctx, SMHIP_EINVAL, "carried and carried_offsets must be given together");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->spec_pending) {  // a batch nobody collected: settle it before its inputs are replaced
    int rc0 = FinishRun(ctx);
    if (rc0 && !ctx->pieces_bad) return rc0;
  }
  ctx->pieces_bad = false;  // (this batch brings its own samples: a sticky piece-length error concerns the slots only)
  for (uint32_t c = 0; c < n_chunks; ++c) {
    if (sample_offsets[c + 1] < sample_offsets[c]) return Fail(ctx, SMHIP_EINVAL, "sample offsets must be non-decreasing");
    if (!features && sample_offsets[c + 1] - sample_offsets[c] > (uint32_t)kMaxSamples)
      return Fail(ctx, SMHIP_EINVAL, "a chunk exceeds 4096 samples (reference chunk size is 4000, sigmap.cc:639)");
    if (features && sample_offsets[c + 1] - sample_offsets[c] > (uint32_t)kFeatStride)
      return Fail(ctx, SMHIP_EINVAL, "a chunk exceeds 2736 features (the events of a 4096-sample chunk)");
  }
  const uint64_t n_samples = n_chunks ? sample_offsets[n_chunks] : 0;
  const uint64_t n_carried = carried_offsets && n_chunks ? carried_offsets[n_chunks] : 0;
  HIPCHK(ctx->samples.Reserve(std::max<uint64_t>(n_samples, 1) * sizeof(float)));
  HIPCHK(hipMemcpyAsync(ctx->samples.p, samples_pa, n_samples * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  ctx->h_chunk_begin.resize((size_t)n_chunks + 1);
  ctx->h_chunk_len.resize((size_t)n_chunks + 1);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    ctx->h_chunk_begin[c] = sample_offsets[c];
    ctx->h_chunk_len[c] = sample_offsets[c + 1] - sample_offsets[c];
  }
  HIPCHK(ctx->chunk_begin.Reserve(((uint64_t)n_chunks + 1) * sizeof(uint64_t)));
  HIPCHK(ctx->chunk_len.Reserve(((uint64_t)n_chunks + 1) * sizeof(uint32_t)));
  HIPCHK(hipMemcpyAsync(ctx->chunk_begin.p, ctx->h_chunk_begin.data(), (uint64_t)n_chunks * sizeof(uint64_t),
                        hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->chunk_len.p, ctx->h_chunk_len.data(), (uint64_t)n_chunks * sizeof(uint32_t),
                        hipMemcpyHostToDevice, ctx->stream));
  ctx->chunk_src = ctx->samples.as<float>();
  int rc = UploadBatchSideInputs(ctx, n_chunks, query_start_offsets, carried, carried_offsets);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));  // host buffers are caller-owned: safe to reuse on return
  ctx->n_chunks = n_chunks;
  ctx->n_samples = n_samples;
  ctx->n_carried = n_carried;
  ctx->batch_is_features = features;
  ctx->batch_ready = true;
  ctx->run_done = false;
  ctx->refine_done = false;
  ctx->refine_path_done = false;
  return SMHIP_OK;
}
// The blocks smhip_host_alloc handed out (start -> bytes): smhip_reads_upload_pieces checks EVERY piece against them on
// the host -- the device dereferences the pieces' addresses, and one it cannot reach is a GPU fault that kills the
// process, not an error code.  A lookup is a map search; a driver call per piece would cost more than the upload.
std::mutex g_pinned_mu;
std::map<uintptr_t, size_t> g_pinned;

// [addr, addr + bytes) inside one block of the registry?  `hint` caches the block of the previous piece (the pieces of
// a round mostly lie in a few arena blocks).
bool InPinnedRegistry(uintptr_t addr, size_t bytes, std::pair<uintptr_t, size_t> *hint) {
  if (addr >= hint->first && addr + bytes <= hint->first + hint->second) return true;
  auto it = g_pinned.upper_bound(addr);
  if (it == g_pinned.begin()) return false;
  --it;
  if (addr + bytes > it->first + it->second) return false;
  *hint = *it;
  return true;
}
}  // namespace
extern "C" {

int smhip_host_alloc(void **out, size_t bytes) {
  if (!out) return SMHIP_EINVAL;
  *out = nullptr;
  void *p = nullptr;
  const size_t n = std::max<size_t>(bytes, 1);
  if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return SMHIP_ENOMEM;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    g_pinned[(uintptr_t)p] = n;
  }
  *out = p;
  return SMHIP_OK;
}

void smhip_host_free(void *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    g_pinned.erase((uintptr_t)p);
  }
  (void)hipHostFree(p);
}

int smhip_reads_configure(smhip_ctx *ctx, uint32_t n_slots, uint32_t keep_samples) {
  if (!ctx) return SMHIP_EINVAL;
  if (n_slots == 0 || keep_samples == 0) return Fail(ctx, SMHIP_EINVAL, "n_slots and keep_samples must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // the slots in segments of <= 2 GiB (device allocations beyond 4 GB are slow: HitSlots, common.h)
  uint32_t sshift = 0;
  while (sshift < 31 && ((uint64_t)2 << sshift) * keep_samples * sizeof(float) <= (1ull << 31)) ++sshift;
  const uint64_t per_seg = 1ull << sshift;
  const size_t n_seg = (size_t)((n_slots + per_seg - 1) / per_seg);
  if (ctx->pa_segs.size() < n_seg) ctx->pa_segs.resize(n_seg);
  for (size_t k = 0; k < n_seg; ++k) {
    const uint64_t here = std::min<uint64_t>(per_seg, n_slots - k * per_seg);
    HIPCHK(ctx->pa_segs[k].Reserve(here * keep_samples * sizeof(float)));
  }
  ctx->h_pa_tab.assign(ctx->pa_segs.size(), nullptr);
  for (size_t k = 0; k < ctx->pa_segs.size(); ++k) ctx->h_pa_tab[k] = ctx->pa_segs[k].as<float>();
  HIPCHK(ctx->pa_tab.Reserve(ctx->h_pa_tab.size() * sizeof(float *)));
  HIPCHK(hipMemcpyAsync(ctx->pa_tab.p, ctx->h_pa_tab.data(), ctx->h_pa_tab.size() * sizeof(float *), hipMemcpyHostToDevice,
                        ctx->stream));
  ctx->pa_sshift = sshift;
  HIPCHK(ctx->slot_len.Reserve((uint64_t)n_slots * sizeof(uint32_t)));
  HIPCHK(hipMemsetAsync(ctx->slot_len.p, 0, (uint64_t)n_slots * sizeof(uint32_t), ctx->stream));
  ctx->n_slots = n_slots;
  ctx->reads_keep = keep_samples;
  return SMHIP_OK;
}

int smhip_reads_upload_raw(smhip_ctx *ctx, const int16_t *raw, const uint64_t *raw_offsets, const float *digitisation,
                           const float *range, const float *offset, const uint32_t *slots, uint32_t n_reads,
                           uint32_t *pa_len) {
  if (!ctx || !raw || !raw_offsets || !digitisation || !range || !offset) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (ctx->n_slots == 0) return Fail(ctx, SMHIP_ESTATE, "smhip_reads_upload_raw before smhip_reads_configure");
  if (n_reads == 0) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  ctx->pieces_bad = false;  // (slots are refilled: a sticky piece-length error of smhip_reads_upload_pieces ends here)
  // per-read records + segment starts, staged in pinned host memory owned by the context
  const size_t meta_bytes = (size_t)n_reads * sizeof(RawReadMeta), seg_bytes = ((size_t)n_reads + 1) * sizeof(uint32_t);
  HIPCHK(hipStreamSynchronize(ctx->copy_stream));  // the previous upload's staging is free again
  if (meta_bytes + seg_bytes > ctx->h_meta_cap) {
    if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
    ctx->h_meta = nullptr;
    ctx->h_meta_cap = 0;
    const size_t want = (meta_bytes + seg_bytes) * 2;
    HIPCHK(hipHostMalloc(&ctx->h_meta, want, hipHostMallocDefault));
    ctx->h_meta_cap = want;
  }
  RawReadMeta *meta = reinterpret_cast<RawReadMeta *>(ctx->h_meta);
  uint32_t *seg_start = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ctx->h_meta) + meta_bytes);
  const uint64_t base = raw_offsets[0];
  uint64_t n_segs = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    if (raw_offsets[r + 1] < raw_offsets[r]) return Fail(ctx, SMHIP_EINVAL, "raw offsets must be non-decreasing");
    const uint64_t n = raw_offsets[r + 1] - raw_offsets[r];
    if (n > 0xFFFFFFFFull) return Fail(ctx, SMHIP_EINVAL, "a read exceeds 2^32 samples");
    const uint32_t slot = slots ? slots[r] : r;
    if (slot >= ctx->n_slots) return Fail(ctx, SMHIP_EINVAL, "slot outside the configured range");
    if (!(digitisation[r] != 0.0f)) return Fail(ctx, SMHIP_EINVAL, "digitisation must be non-zero");
    meta[r] = RawReadMeta{raw_offsets[r] - base, (uint32_t)n, slot, digitisation[r], range[r], offset[r], 0u};
    seg_start[r] = (uint32_t)n_segs;
    n_segs += RawSegments((uint32_t)n);
    if (n_segs > 0x7FFFFFFFull) return Fail(ctx, SMHIP_EINVAL, "too many samples in one upload");
  }
  seg_start[n_reads] = (uint32_t)n_segs;
  const uint64_t n_raw = raw_offsets[n_reads] - base;
  HIPCHK(ctx->reads_raw.Reserve(std::max<uint64_t>(n_raw, 1) * sizeof(int16_t)));
  HIPCHK(ctx->reads_meta.Reserve(meta_bytes));
  HIPCHK(ctx->reads_segstart.Reserve(seg_bytes));
  HIPCHK(ctx->reads_seg.Reserve(n_segs * 2 * sizeof(uint32_t)));
  HIPCHK(ctx->reads_len.Reserve((uint64_t)n_reads * sizeof(uint32_t)));
  // H2D on the copy stream (overlaps whatever the compute stream is running: with pinned host memory,
  // smhip_host_alloc, the copy is a DMA at link rate), after the previous conversion has consumed
  // the staging buffer; the conversion on the compute stream waits for the copy.
  HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->convert_done, 0));
  HIPCHK(hipMemcpyAsync(ctx->reads_raw.p, raw + base, n_raw * sizeof(int16_t), hipMemcpyHostToDevice, ctx->copy_stream));
  HIPCHK(hipMemcpyAsync(ctx->reads_meta.p, meta, meta_bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  HIPCHK(hipMemcpyAsync(ctx->reads_segstart.p, seg_start, seg_bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  HIPCHK(hipEventRecord(ctx->copy_done, ctx->copy_stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->copy_done, 0));
  {
    ScopedTimer t(ctx, T_CONVERT);
    HIPCHK(LaunchRawConvert(ctx, ctx->reads_raw.as<int16_t>(), ctx->reads_meta.as<RawReadMeta>(),
                            ctx->reads_segstart.as<uint32_t>(), n_reads, (uint32_t)n_segs, ctx->reads_seg.as<uint32_t>(),
                            ctx->reads_seg.as<uint32_t>() + n_segs,
                            SlotView{ctx->pa_tab.as<float *>(), ctx->pa_sshift, ctx->reads_keep},
                            ctx->slot_len.as<uint32_t>(), ctx->reads_len.as<uint32_t>()));
  }
  HIPCHK(hipEventRecord(ctx->convert_done, ctx->stream));
  if (pa_len) {
    HIPCHK(hipMemcpyAsync(pa_len, ctx->reads_len.p, (uint64_t)n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return SMHIP_OK;
}

int smhip_reads_upload_pieces(smhip_ctx *ctx, const int16_t *const *pieces, const uint32_t *piece_len,
                              const float *digitisation, const float *range, const float *offset, const uint32_t *slots,
                              const uint32_t *expect_len, uint32_t n_pieces) {
  if (!ctx || !pieces || !piece_len || !digitisation || !range || !offset) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (ctx->n_slots == 0) return Fail(ctx, SMHIP_ESTATE, "smhip_reads_upload_pieces before smhip_reads_configure");
  if (n_pieces == 0) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  ctx->pieces_bad = false;  // new pieces: the slots' contents are replaced
  if (ctx->pieces_pending) {  // the previous call's staging has been read
    HIPCHK(hipEventSynchronize(ctx->pieces_staged));
    ctx->pieces_pending = false;
  }
  const size_t n = n_pieces;
  const size_t meta_bytes = n * sizeof(RawReadMeta), seg_bytes = (n + 1) * sizeof(uint32_t), src_bytes = n * sizeof(PieceSrc),
               exp_bytes = expect_len ? n * sizeof(uint32_t) : 0;
  const size_t stage_bytes = meta_bytes + src_bytes + seg_bytes + exp_bytes + 64;
  if (stage_bytes > ctx->h_pieces_cap) {
    if (ctx->h_pieces) (void)hipHostFree(ctx->h_pieces);
    ctx->h_pieces = nullptr;
    ctx->h_pieces_cap = 0;
    HIPCHK(hipHostMalloc(&ctx->h_pieces, stage_bytes * 2, hipHostMallocDefault));
    ctx->h_pieces_cap = stage_bytes * 2;
  }
  char *hp = static_cast<char *>(ctx->h_pieces);
  RawReadMeta *meta = reinterpret_cast<RawReadMeta *>(hp);
  PieceSrc *src = reinterpret_cast<PieceSrc *>(hp + meta_bytes);
  uint32_t *seg_start = reinterpret_cast<uint32_t *>(hp + meta_bytes + src_bytes);
  uint32_t *expect = reinterpret_cast<uint32_t *>(hp + meta_bytes + src_bytes + seg_bytes);
  uint64_t n_segs = 0, units = 0;
  for (uint32_t r = 0; r < n_pieces; ++r) {
    const uint64_t addr = (uint64_t)(uintptr_t)pieces[r];
    if ((addr & 1ull) || (!pieces[r] && piece_len[r])) return Fail(ctx, SMHIP_EINVAL, "a piece is not an int16 array");
    const uint32_t slot = slots ? slots[r] : r;
    if (slot >= ctx->n_slots) return Fail(ctx, SMHIP_EINVAL, "slot outside the configured range");
    if (!(digitisation[r] != 0.0f)) return Fail(ctx, SMHIP_EINVAL, "digitisation must be non-zero");
    const uint32_t phase = (uint32_t)(addr & 15ull) >> 1;  // the piece keeps its place inside a 16-byte unit
    if (units > 0xFFFFFFF0ull) return Fail(ctx, SMHIP_EINVAL, "too many samples in one upload");
    src[r] = PieceSrc{addr, (uint32_t)units, piece_len[r]};
    meta[r] = RawReadMeta{units * 8 + phase, piece_len[r], slot, digitisation[r], range[r], offset[r], 0u};
    units += ((uint64_t)phase + piece_len[r] + 7) / 8;
    seg_start[r] = (uint32_t)n_segs;
    n_segs += RawSegments(piece_len[r]);
    if (n_segs > 0x7FFFFFFFull) return Fail(ctx, SMHIP_EINVAL, "too many samples in one upload");
  }
  seg_start[n_pieces] = (uint32_t)n_segs;
  if (expect_len) memcpy(expect, expect_len, exp_bytes);
  // The device dereferences the pieces' addresses: memory it cannot reach would be a GPU fault, not an error code.
  // EVERY piece is checked: against the blocks of smhip_host_alloc (a map search), and a piece that lies in none of them
  // -- pinned by somebody else -- with a driver lookup of its first and last sample.
  {
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    std::pair<uintptr_t, size_t> hint{0, 0};
    for (uint32_t r = 0; r < n_pieces; ++r) {
      if (piece_len[r] == 0) continue;
      const uintptr_t a = (uintptr_t)pieces[r];
      const size_t bytes = (size_t)piece_len[r] * sizeof(int16_t);
      if (InPinnedRegistry(a, bytes, &hint)) continue;
      for (const int16_t *probe : {pieces[r], pieces[r] + (piece_len[r] - 1)}) {
        void *dev_view = nullptr;
        if (hipHostGetDevicePointer(&dev_view, const_cast<int16_t *>(probe), 0) != hipSuccess || dev_view == nullptr) {
          (void)hipGetLastError();
          return Fail(ctx, SMHIP_EINVAL, "smhip_reads_upload_pieces: every piece must lie in pinned host memory (smhip_host_alloc)");
        }
      }
    }
  }
  // (the staged samples are also what smhip_reads_upload_raw copies into: its DMA waits for convert_done, recorded below)
  HIPCHK(ctx->reads_raw.Reserve(std::max<uint64_t>(units, 1) * 16));
  // the four descriptor arrays lie in the staging one behind the other and travel as ONE copy into one device block
  // with the same layout (a small copy is a blit kernel of its own, ~5 us in a round's dependent chain; a streaming
  // round of a hundred reads is made of such)
  const size_t block_bytes = meta_bytes + src_bytes + seg_bytes + exp_bytes;
  HIPCHK(ctx->pieces_src.Reserve(block_bytes));
  HIPCHK(ctx->reads_seg.Reserve(n_segs * 2 * sizeof(uint32_t)));
  HIPCHK(ctx->reads_len.Reserve(n * sizeof(uint32_t)));
  char *dblock = static_cast<char *>(ctx->pieces_src.p);
  const RawReadMeta *d_meta = reinterpret_cast<const RawReadMeta *>(dblock);
  const PieceSrc *d_src = reinterpret_cast<const PieceSrc *>(dblock + meta_bytes);
  const uint32_t *d_segstart = reinterpret_cast<const uint32_t *>(dblock + meta_bytes + src_bytes);
  const uint32_t *d_expect = reinterpret_cast<const uint32_t *>(dblock + meta_bytes + src_bytes + seg_bytes);
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->convert_done, 0));  // (ordered anyway on this stream; cheap)
  HIPCHK(hipMemcpyAsync(dblock, hp, block_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->pieces_staged, ctx->stream));
  ctx->pieces_pending = true;
  {
    ScopedTimer t(ctx, T_CONVERT);
    HIPCHK(LaunchGatherPieces(ctx, d_src, n_pieces, ctx->reads_raw.p));
    HIPCHK(LaunchRawConvert(ctx, ctx->reads_raw.as<int16_t>(), d_meta,
                            d_segstart, n_pieces, (uint32_t)n_segs, ctx->reads_seg.as<uint32_t>(),
                            ctx->reads_seg.as<uint32_t>() + n_segs,
                            SlotView{ctx->pa_tab.as<float *>(), ctx->pa_sshift, ctx->reads_keep},
                            ctx->slot_len.as<uint32_t>(), ctx->reads_len.as<uint32_t>()));
    if (expect_len)
      HIPCHK(LaunchCheckPieceLen(ctx, ctx->reads_len.as<uint32_t>(), d_expect, n_pieces, ctx->h_piece_flag));
  }
  HIPCHK(hipEventRecord(ctx->convert_done, ctx->stream));
  return SMHIP_OK;
}

int smhip_batch_select(smhip_ctx *ctx, const uint32_t *slots, const uint32_t *chunk_index, uint32_t chunk_size,
                       uint32_t n_chunks, const uint32_t *query_start_offsets, const smhip_carried_anchor *carried,
                       const uint32_t *carried_offsets) {
  if (!ctx || (n_chunks && (!slots || !chunk_index))) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (ctx->n_slots == 0) return Fail(ctx, SMHIP_ESTATE, "smhip_batch_select before smhip_reads_configure");
  if ((carried == nullptr) != (carried_offsets == nullptr))
    return Fail(ctx, SMHIP_EINVAL, "carried and carried_offsets must be given together");
  if (chunk_size == 0 || chunk_size > (uint32_t)kMaxSamples)
    return Fail(ctx, SMHIP_EINVAL, "chunk_size must be 1..4096 (reference chunk size is 4000, sigmap.cc:639)");
  if (ctx->pieces_bad)
    return Fail(ctx, SMHIP_ECAP, "smhip_batch_select: the slots hold pieces whose filtered lengths the host and the device do not "
                                 "agree on (smhip_reads_upload_pieces, expect_len): upload the pieces again");
  for (uint32_t c = 0; c < n_chunks; ++c) {
    if (slots[c] >= ctx->n_slots) return Fail(ctx, SMHIP_EINVAL, "slot outside the configured range");
    if (((uint64_t)chunk_index[c] + 1) * chunk_size > ctx->reads_keep)
      return Fail(ctx, SMHIP_EINVAL, "chunk beyond the samples a slot keeps (keep_samples of smhip_reads_configure)");
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->spec_pending) {  // a batch nobody collected: settle it before its inputs are replaced
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  const uint64_t nb = std::max<uint64_t>(n_chunks, 1);
  HIPCHK(ctx->sel_slots.Reserve(2 * nb * sizeof(uint32_t)));  // [slots][chunk index]: one copy
  HIPCHK(ctx->chunk_begin.Reserve((nb + 1) * sizeof(uint64_t)));
  HIPCHK(ctx->chunk_len.Reserve((nb + 1) * sizeof(uint32_t)));
  // (through the pinned staging: [slots][chunk_index], and in UploadBatchSideInputs [qstart][carried_off] behind them)
  HIPCHK(StageReserve(ctx, (4 * (size_t)n_chunks + 8) * sizeof(uint32_t)));
  HIPCHK(StageWait(ctx));  // the previous batch's copies out of the staging have finished (they have, once it was collected)
  {
    uint32_t *st = static_cast<uint32_t *>(ctx->h_stage);
    memcpy(st, slots, (size_t)n_chunks * sizeof(uint32_t));
    memcpy(st + n_chunks, chunk_index, (size_t)n_chunks * sizeof(uint32_t));
    if (n_chunks)
      HIPCHK(hipMemcpyAsync(ctx->sel_slots.p, st, 2 * (uint64_t)n_chunks * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(LaunchSelectChunks(ctx, ctx->sel_slots.as<uint32_t>(), ctx->sel_slots.as<uint32_t>() + n_chunks, n_chunks, chunk_size,
                            SlotView{ctx->pa_tab.as<float *>(), ctx->pa_sshift, ctx->reads_keep},
                            ctx->slot_len.as<uint32_t>(), ctx->chunk_begin.as<uint64_t>(), ctx->chunk_len.as<uint32_t>()));
  ctx->chunk_src = nullptr;  // chunk_begin holds addresses (in floats)
  int rc = UploadBatchSideInputs(ctx, n_chunks, query_start_offsets, carried, carried_offsets);
  if (rc) return rc;
  // (no wait: every caller-owned array went through pinned staging of the context, so it is free on return; the
  // staging itself is next written by the following batch's calls, after this batch has been collected)
  ctx->batch_is_features = false;
  ctx->n_chunks = n_chunks;
  ctx->n_samples = (uint64_t)n_chunks * chunk_size;
  ctx->n_carried = carried_offsets && n_chunks ? carried_offsets[n_chunks] : 0;
  ctx->batch_ready = true;
  ctx->run_done = false;
  ctx->refine_done = false;
  ctx->refine_path_done = false;
  return SMHIP_OK;
}

int smhip_batch_reserve(smhip_ctx *ctx, uint32_t n_chunks, uint32_t queries_per_chunk) {
  return smhip_batch_reserve_for(ctx, nullptr, n_chunks, queries_per_chunk);
}

int smhip_batch_reserve_for(smhip_ctx *ctx, const smhip_params *params, uint32_t n_chunks, uint32_t queries_per_chunk) {
  if (!ctx) return SMHIP_EINVAL;
  if (n_chunks == 0 || queries_per_chunk == 0) return Fail(ctx, SMHIP_EINVAL, "counts must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  smhip_params p;
  if (params) p = *params;
  else smhip_default_params(&p);
  if (ctx->have_index) {  // (the automatic number of fast slots comes from the census of the index)
    int rc = ResolveParams(ctx, &p, n_chunks);
    if (rc) return rc;
  } else {
    p.hits_per_query_fast = 512;
  }
  const uint64_t R = n_chunks, nq = R * queries_per_chunk;
  // the buffers smhip_map_run sizes from the chunk and query counts (the anchor-sized ones depend on
  // the index and stay lazy); the per-query hit slots are by far the largest
  HIPCHK(ctx->feat.Reserve(R * kFeatStride * sizeof(float)));
  HIPCHK(ctx->results.Reserve(R * sizeof(smhip_chunk_result)));
  HIPCHK(ctx->q_count.Reserve(nq * sizeof(uint32_t)));
  HIPCHK(ctx->q_nodes.Reserve(nq * sizeof(uint32_t)));
  HIPCHK(ctx->q_points.Reserve(nq * sizeof(uint32_t)));
  HIPCHK(ctx->ovf_list.Reserve(nq * sizeof(uint32_t)));
  HIPCHK(ctx->q_ovf_off.Reserve(nq * kOvfLevels * sizeof(uint32_t)));
  {
    HitSlots hs;
    int rc = ReserveRegion(ctx, nq, p.hits_per_query_fast, &hs);
    if (rc) return rc;
  }
  HIPCHK(ctx->qrec.Reserve(nq * 32));
  HIPCHK(ctx->ovf_pool.Reserve((nq * 64 + (1u << 22)) * sizeof(Hit)));
  if (ctx->res_fast != p.hits_per_query_fast) ctx->res_nq = 0;
  ctx->res_fast = p.hits_per_query_fast;
  ctx->res_nq = std::max(ctx->res_nq, nq);
  // ... and the anchor-sized ones at the index's own estimate: the census' mean of min(hits, num_nearest_points) per
  // query, a quarter on top (what a speculative batch would ask for) -- within a sixth of the device; a batch that
  // needs more grows them as before
  if (ctx->have_index && !ctx->census_hits.empty()) {
    double sum = 0;
    for (uint32_t h : ctx->census_hits) sum += std::min<uint32_t>(h, (uint32_t)p.num_nearest_points);
    const uint64_t na = (uint64_t)((double)nq * (sum / (double)ctx->census_hits.size()) * 1.25) + (1u << 20);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && na * 110 < total_b / 6 && na * 110 * 2 < free_b) {
      int rc = ReserveAnchorBuffers(ctx, na, n_chunks);
      if (rc) return rc;
    }
  }
  return SMHIP_OK;
}

int smhip_reads_download_pa(smhip_ctx *ctx, uint32_t slot, float *pa, uint32_t cap, uint32_t *n_kept, uint32_t *n_total) {
  if (!ctx || !pa) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (slot >= ctx->n_slots) return Fail(ctx, SMHIP_EINVAL, "slot outside the configured range");
  HIPCHK(hipSetDevice(ctx->device));
  uint32_t total = 0;
  HIPCHK(hipMemcpyAsync(&total, ctx->slot_len.as<uint32_t>() + slot, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint32_t kept = std::min(total, ctx->reads_keep), n = std::min(kept, cap);
  const float *src = ctx->h_pa_tab[slot >> ctx->pa_sshift] +
                     (size_t)(slot & ((1u << ctx->pa_sshift) - 1u)) * ctx->reads_keep;
  HIPCHK(hipMemcpyAsync(pa, src, (size_t)n * sizeof(float),
                        hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (n_kept) *n_kept = kept;
  if (n_total) *n_total = total;
  return SMHIP_OK;
}

}  // extern "C"
namespace {
// ---- the stages of one batch, as enqueued on the context stream ----

// A step's kernels are sized by two totals only the device knows: the number of queries (after event
// detection) and the number of anchors (after the search).  smhip_map_run either reads them back
// (two stream synchronisations; the first batches of a context) or SPECULATES: the buffers get the
// capacities recent batches suggest, a guard kernel compares the real total with the capacity where
// it becomes known and -- beyond it -- empties the batch for the kernels downstream (which then touch
// nothing) and raises a flag; the whole chain is enqueued without the host ever waiting.  The flag,
// the totals and the pool state travel to pinned host memory behind the last kernel; whoever next
// needs the results (smhip_map_sync, the downloads) finds an overflow there and repeats the batch the
// synchronous way.  Results never depend on the route.
// pinned host mirror of the context's 40 counter words, copied behind the last kernel of a speculative batch:
// [0..25] the search's counters ([3] pool exhausted), [32] / [33] the guards' flags, [34] queries and
// [36..37] anchors as counted on the device
struct RunInfo {
  uint32_t w[40];
  uint64_t anchors() const { return (uint64_t)w[36] | ((uint64_t)w[37] << 32); }
};
static_assert(sizeof(RunInfo) <= 256, "smhip_init allocates 256 bytes for it");

template <typename OffT>
__global__ void GuardTotalKernel(uint32_t *__restrict__ counts, OffT *__restrict__ offsets, uint32_t n, OffT cap,
                                 uint32_t *__restrict__ flag, OffT *__restrict__ total_out) {
  const OffT total = offsets[n];
  if (threadIdx.x == 0) *total_out = total;  // the real total, also when the batch is emptied below
  if (total <= cap) return;
  __syncthreads();  // (one workgroup: every thread has read the total before anyone clears it)
  for (uint32_t i = threadIdx.x; i <= n; i += blockDim.x) {
    if (i < n) counts[i] = 0;
    offsets[i] = 0;
  }
  if (threadIdx.x == 0) *flag = 1u;
}

int EnqueueFront(smhip_ctx *ctx, const smhip_params *p, uint32_t R, bool reuse_features = false) {
  HIPCHK(ctx->feat.Reserve((uint64_t)R * kFeatStride * sizeof(float)));
  HIPCHK(ctx->results.Reserve((uint64_t)R * sizeof(smhip_chunk_result)));
  HIPCHK(ctx->nq.Reserve((uint64_t)R * sizeof(uint32_t)));
  HIPCHK(ctx->q_off.Reserve(((uint64_t)R + 1) * sizeof(uint32_t)));
  HIPCHK(ctx->chunk_anchor_cnt.Reserve((uint64_t)R * sizeof(uint32_t)));
  HIPCHK(ctx->anchor_off.Reserve(((uint64_t)R + 1) * sizeof(uint64_t)));
  HIPCHK(ctx->n_groups.Reserve((uint64_t)R * sizeof(uint32_t)));
  HIPCHK(ctx->out_anchor_cnt.Reserve((uint64_t)R * sizeof(uint32_t)));
  // counters: [0..31] the search's (zeroed before every walker launch); [32] / [33] guard flags of the
  // speculative route, [34] queries, [36..37] anchors as counted on the device
  HIPCHK(ctx->counters.Reserve(40 * sizeof(uint32_t)));
  HIPCHK(hipMemsetAsync(ctx->counters.as<uint32_t>() + 32, 0, 8 * sizeof(uint32_t), ctx->stream));
  if (reuse_features) {  // second pass over a batch: from the features of the first one (events.hip RestoreFrontKernel)
    HIPCHK(LaunchRestoreFront(ctx, R, ctx->results.as<smhip_chunk_result>(), ctx->nq.as<uint32_t>()));
  } else if (ctx->batch_is_features) {  // smhip_features_upload: the chunks ARE feature signals
    HIPCHK(LaunchFeaturesIn(ctx, *p, ctx->chunk_src, ctx->chunk_begin.as<uint64_t>(), ctx->chunk_len.as<uint32_t>(), R,
                            kFeatStride, ctx->feat.as<float>(), ctx->results.as<smhip_chunk_result>(), ctx->nq.as<uint32_t>()));
  } else {
    ScopedTimer t(ctx, T_EVENTS);
    HIPCHK(LaunchEvents(ctx, *p, ctx->chunk_src, ctx->chunk_begin.as<uint64_t>(), ctx->chunk_len.as<uint32_t>(), R, kFeatStride,
                        ctx->feat.as<float>(), nullptr, ctx->results.as<smhip_chunk_result>(), ctx->nq.as<uint32_t>()));
  }
  HIPCHK(LaunchScanU32(ctx, ctx->nq.as<uint32_t>(), R, ctx->q_off.as<uint32_t>()));
  return SMHIP_OK;
}

// per-query buffers for up to nq_cap queries, the query records, the search, the chunks' anchor counts
int EnqueueSearch(smhip_ctx *ctx, const smhip_params *p, uint32_t R, uint64_t nq_cap, uint64_t pool_cap, HitSlots *hit_slots) {
  smhip_chunk_result *results = ctx->results.as<smhip_chunk_result>();
  HIPCHK(ctx->q_count.Reserve(nq_cap * sizeof(uint32_t)));
  HIPCHK(ctx->q_nodes.Reserve(nq_cap * sizeof(uint32_t)));
  HIPCHK(ctx->q_points.Reserve(nq_cap * sizeof(uint32_t)));
  HIPCHK(ctx->ovf_list.Reserve(nq_cap * sizeof(uint32_t)));
  {
    int rc = ReserveRegion(ctx, nq_cap, p->hits_per_query_fast, hit_slots);
    if (rc) return rc;
  }
  HIPCHK(ctx->qrec.Reserve(nq_cap * 32));
  HIPCHK(ctx->q_ovf_off.Reserve(nq_cap * kOvfLevels * sizeof(uint32_t)));
  HIPCHK(ctx->ovf_pool.Reserve(pool_cap * sizeof(Hit)));
  if (ctx->res_fast != p->hits_per_query_fast) ctx->res_nq = 0;
  ctx->res_fast = p->hits_per_query_fast;
  ctx->res_nq = std::max(ctx->res_nq, nq_cap);
  HIPCHK(LaunchQueryExpand(ctx, *p, ctx->feat.as<float>(), kFeatStride, ctx->q_off.as<uint32_t>(), R, ctx->qrec.p));
  HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 32 * sizeof(uint32_t), ctx->stream));
  // (timers inside: "seed_order" = the visiting-order kernels, "seed_lookup" = the walker launch alone)
  HIPCHK(LaunchSeedLookup(ctx, *p, ctx->qrec.p, ctx->q_off.as<uint32_t>(), R, nq_cap, *hit_slots,
                          ctx->q_count.as<uint32_t>(), ctx->q_nodes.as<uint32_t>(), ctx->q_points.as<uint32_t>(),
                          ctx->counters.as<uint32_t>(), ctx->ovf_pool.as<Hit>(), ctx->q_ovf_off.as<uint32_t>(), pool_cap));
  HIPCHK(LaunchAnchorCount(ctx, *p, ctx->q_off.as<uint32_t>(), ctx->q_count.as<uint32_t>(), ctx->q_nodes.as<uint32_t>(),
                           ctx->q_points.as<uint32_t>(), ctx->d_carried_off, R,
                           ctx->chunk_anchor_cnt.as<uint32_t>(), results));
  HIPCHK(LaunchScanU32ToU64(ctx, ctx->chunk_anchor_cnt.as<uint32_t>(), R, ctx->anchor_off.as<uint64_t>()));
  return SMHIP_OK;
}

// the buffers EnqueueBack sizes from the anchor count
int ReserveAnchorBuffers(smhip_ctx *ctx, uint64_t na_cap, uint32_t R) {
  const uint64_t na = std::max<uint64_t>(na_cap, 1);
  DevBuf *b4[] = {&ctx->a_tpos, &ctx->a_qpos, &ctx->a_dist, &ctx->a_grp, &ctx->s_tpos, &ctx->s_qpos,
                  &ctx->s_dist, &ctx->s_grp, &ctx->score, &ctx->pred, &ctx->group_start};
  for (DevBuf *b : b4) HIPCHK(b->Reserve(na * 4));
  HIPCHK(ctx->keys.Reserve(std::max<uint64_t>(SortRecordRoom(na, R), na) * 2 * sizeof(uint64_t)));  // the sort's records (5/3 per anchor)
  HIPCHK(ctx->used.Reserve(na));
  HIPCHK(ctx->out_anchors.Reserve(na * sizeof(smhip_carried_anchor)));
  ctx->res_na = std::max(ctx->res_na, na_cap);
  return SMHIP_OK;
}

// per-anchor buffers for up to na_cap anchors, sort, chaining
int EnqueueBack(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, uint32_t R, uint64_t na_cap, const HitSlots &hit_slots) {
  smhip_chunk_result *results = ctx->results.as<smhip_chunk_result>();
  {
    int rc = ReserveAnchorBuffers(ctx, na_cap, R);
    if (rc) return rc;
  }
  SortArgs sa;
  sa.q_off = ctx->q_off.as<uint32_t>();
  sa.q_count = ctx->q_count.as<uint32_t>();
  sa.ovf_tab = ctx->q_ovf_off.as<uint32_t>();
  sa.qstart = ctx->qstart.as<uint32_t>();
  sa.carried_off = ctx->d_carried_off;
  sa.carried = ctx->carried.as<smhip_carried_anchor>();
  sa.region = hit_slots;
  sa.ovf_pool = ctx->ovf_pool.as<Hit>();
  sa.anchor_off = ctx->anchor_off.as<uint64_t>();
  sa.fast_cap = p->hits_per_query_fast;
  sa.nnp = (uint32_t)p->num_nearest_points;
  sa.step = (uint32_t)p->step_size;
  sa.n_refs = n_refs;
  HIPCHK(ctx->max_tpos.Reserve(2 * sizeof(uint32_t)));
  if (!ctx->max_tpos_valid) {  // once per index
    HIPCHK(LaunchIndexMaxTarget(ctx, ctx->max_tpos.as<uint32_t>()));
    ctx->max_tpos_valid = true;
  }
  HIPCHK(LaunchCarriedMaxTarget(ctx, sa.carried, ctx->n_carried, ctx->max_tpos.as<uint32_t>(),
                                ctx->max_tpos.as<uint32_t>() + 1));
  sa.span_bound = ctx->max_tpos.as<uint32_t>() + 1;
  sa.a_tpos = ctx->a_tpos.as<uint32_t>();
  sa.a_qpos = ctx->a_qpos.as<uint32_t>();
  sa.a_grp = ctx->a_grp.as<uint32_t>();
  sa.a_dist = ctx->a_dist.as<float>();
  sa.keys = ctx->keys.as<uint64_t>();
  sa.s_tpos = ctx->s_tpos.as<uint32_t>();
  sa.s_qpos = ctx->s_qpos.as<uint32_t>();
  sa.s_grp = ctx->s_grp.as<uint32_t>();
  sa.s_dist = ctx->s_dist.as<float>();
  sa.group_start = ctx->group_start.as<uint32_t>();
  sa.n_groups = ctx->n_groups.as<uint32_t>();
  sa.results = results;
  {
    ScopedTimer t(ctx, T_SORT);
    HIPCHK(LaunchSort(ctx, sa, R, na_cap));
  }
  ChainArgs ca;
  ca.anchor_off = sa.anchor_off;
  ca.s_tpos = sa.s_tpos;
  ca.s_qpos = sa.s_qpos;
  ca.s_grp = sa.s_grp;
  ca.s_dist = sa.s_dist;
  ca.group_start = sa.group_start;
  ca.n_groups = sa.n_groups;
  ca.score = ctx->score.as<float>();
  ca.pred = ctx->pred.as<uint32_t>();
  ca.used = ctx->used.as<uint8_t>();
  ca.chain_tmp = ca.chain_sorted = nullptr;  // sized and set by LaunchChain
  ca.chain_off = nullptr;
  ca.scratch = ctx->keys.as<uint64_t>();
  ca.pending = ctx->a_tpos.as<uint32_t>();  // the unsorted SoA is dead after the sort
  ca.lists = ctx->a_qpos.as<uint32_t>();
  ca.item_lo = ctx->a_dist.as<uint64_t>();
  ca.item_hi = ctx->a_grp.as<uint64_t>();
  ca.results = results;
  ca.out_anchors = ctx->out_anchors.as<smhip_carried_anchor>();
  ca.out_anchor_cnt = ctx->out_anchor_cnt.as<uint32_t>();
  {
    ScopedTimer t(ctx, T_CHAIN);
    HIPCHK(LaunchChain(ctx, *p, ca, R, na_cap, n_refs));
  }
  hipLaunchKernelGGL(OrStatusKernel, dim3(64), dim3(256), 0, ctx->stream, results, R, ctx->counters.as<uint32_t>());
  HIPCHK(hipGetLastError());
  return SMHIP_OK;
}

uint64_t FirstPoolCap(smhip_ctx *ctx, uint64_t nq_cap) {
  // Overflow pool (hits beyond a query's fast slots).  The walker keeps adding the sizes of the
  // segments it wants to the pool cursor even when the pool is exhausted, so a launch that ran out
  // of room reports the exact demand: the pool is then grown to it and the search repeated (once;
  // the size is remembered for the following batches).
  // (knob pool_min: makes the first size too small for the pool-growth test)
  const uint64_t pool_first = ctx->knobs.pool_min >= 0 ? (uint64_t)ctx->knobs.pool_min : nq_cap * 64 + (1u << 22);
  return std::min<uint64_t>(std::max<uint64_t>(pool_first, ctx->pool_hint), 0xFFFF0000ull);
}

// capacities of a speculative batch of R chunks: 25 % above the largest per-chunk totals seen, plus slack
uint64_t SpecCapQueries(const smhip_ctx *ctx, uint32_t R) {
  return std::min<uint64_t>((uint64_t)R * ctx->hint_q_per_chunk * 5 / 4 + 32768, 0xFFFFFFF0ull);
}
uint64_t SpecCapAnchors(const smhip_ctx *ctx, uint32_t R) {
  return (uint64_t)R * ctx->hint_a_per_chunk * 5 / 4 + ctx->n_carried + (1u << 20);
}

// Can the per-query / per-anchor buffers grow to these capacities?  (Their headroom above the exact totals is a
// convenience of the speculative route; a batch sized for the whole device -- the 250 Mb shard: 2.3 G anchors --
// must not fail for it.)  Bytes per query: hit slots, overflow table, records, keys, counters; per anchor: the
// unsorted and sorted columns, score, predecessor, group starts, sort records, flags, chain anchors.
bool CapacitiesAffordable(smhip_ctx *ctx, const smhip_params *p, uint64_t nq_cap, uint64_t na_cap) {
  const uint64_t per_query = (uint64_t)p->hits_per_query_fast * sizeof(Hit) + kOvfLevels * 4 + 32 * 2 + 4 * 6 + 64 * sizeof(Hit);
  const uint64_t per_anchor = 11 * 4 + 2 * sizeof(uint64_t) * 5 / 3 + 6 + 1 + sizeof(smhip_carried_anchor);
  uint64_t have_q = ctx->q_count.cap / 4, have_a = ctx->score.cap / 4;  // what these buffers hold already
  const uint64_t grow = (nq_cap > have_q ? (nq_cap - have_q) * per_query : 0) + (na_cap > have_a ? (na_cap - have_a) * per_anchor : 0);
  if (grow == 0) return true;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  // Head room only for contexts that stay small -- an eighth of the device with it -- and never more than a third
  // of what is free: the other contexts of the GPU have the same appetite, and a batch sized for the device
  // (100 Mb reference x 3 contexts: 250 GB) must still fit at its exact size.
  const uint64_t with_room = nq_cap * per_query + na_cap * per_anchor;
  return with_room < total_b / 8 && 3 * grow + (1ull << 30) < free_b;
}

// counters[7] of a set-walk batch = queries found beyond the hit cap (searched a second time).  Their share over the
// batches of this index so far -- at least 20 000 queries (some 90 chunks), so that one such query in a small tail
// batch decides nothing -- beyond set_walk_max_over_cap: every query in the reference's order from the next batch on.
void LearnOverCap(smhip_ctx *ctx, const smhip_params *p, uint64_t NQ, uint32_t over_cap) {
  if (ctx->last_walk_mode != 2) return;
  if ((uint32_t)p->num_nearest_points <= 8192u && !ctx->knobs.second_search) return;  // (ranked, not searched twice)
  ctx->seen_queries += NQ;
  ctx->seen_over_cap += over_cap;
  if (ctx->seen_queries >= 20000 && (double)ctx->seen_over_cap > (double)p->set_walk_max_over_cap * (double)ctx->seen_queries)
    ctx->walk_exact_only = true;
}

void LearnTotals(smhip_ctx *ctx, uint32_t R, uint64_t NQ, uint64_t NA) {
  const uint64_t non_carried = NA > ctx->n_carried ? NA - ctx->n_carried : 0;
  ctx->hint_q_per_chunk = std::max<uint64_t>(ctx->hint_q_per_chunk, (NQ + R - 1) / R);
  ctx->hint_a_per_chunk = std::max<uint64_t>(ctx->hint_a_per_chunk, (non_carried + R - 1) / R);
  ctx->hint_valid = true;
}

// the synchronous route: both totals are read back and the buffers sized exactly
int RunBatchSync(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, bool reuse_features = false) {
  const uint32_t R = ctx->n_chunks;
  const bool trace = TraceAlloc();  // SMHIP_TRACE_ALLOC=1: where the host waits on this route, on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&t0]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  int rc = EnqueueFront(ctx, p, R, reuse_features);
  if (rc) return rc;
  if (trace) fprintf(stderr, "[smhip] sync route, %u chunks: front enqueued %.2f ms\n", R, since());
  uint32_t NQ = 0;
  HIPCHK(hipMemcpyAsync(&NQ, ctx->q_off.as<uint32_t>() + R, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (trace) fprintf(stderr, "[smhip] sync route: front done %.2f ms (%u queries)\n", since(), NQ);
  // (buffers are reserved at the capacity a speculative batch of this size would ask for next, so that the
  // switch to that route does not reallocate -- a hipFree synchronises the whole device)
  ctx->hint_q_per_chunk = std::max<uint64_t>(ctx->hint_q_per_chunk, ((uint64_t)NQ + R - 1) / R);
  uint64_t nq_cap = std::max<uint64_t>(NQ, 1);
  if (CapacitiesAffordable(ctx, p, SpecCapQueries(ctx, R), 0)) nq_cap = std::max(nq_cap, SpecCapQueries(ctx, R));
  uint64_t pool_cap = FirstPoolCap(ctx, nq_cap);
  uint64_t NA = 0;
  HitSlots hit_slots;
  for (int attempt = 0;; ++attempt) {
    rc = EnqueueSearch(ctx, p, R, nq_cap, pool_cap, &hit_slots);
    if (rc) return rc;
    uint32_t hc[26] = {0};
    HIPCHK(hipMemcpyAsync(&NA, ctx->anchor_off.as<uint64_t>() + R, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(hc, ctx->counters.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
    if (trace) fprintf(stderr, "[smhip] sync route: search enqueued %.2f ms\n", since());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (trace) fprintf(stderr, "[smhip] sync route: search done %.2f ms\n", since());
    LearnOverCap(ctx, p, NQ, hc[7]);
    if (!hc[3] || attempt > 0) break;  // hc[3]: pool exhausted (kept as SMHIP_ST_HIT_POOL_FULL if it persists)
    const uint64_t demand = (((uint64_t)hc[25] << 32) | hc[24]) + (1u << 20);
    if (demand <= pool_cap || demand > 0xFFFF0000ull) break;  // level limit / beyond 32-bit offsets: report the flag
    pool_cap = demand;
    ctx->pool_hint = pool_cap;
  }
  LearnTotals(ctx, R, NQ, NA);
  uint64_t na_cap = NA;
  if (CapacitiesAffordable(ctx, p, 0, SpecCapAnchors(ctx, R))) na_cap = std::max(na_cap, SpecCapAnchors(ctx, R));
  rc = EnqueueBack(ctx, p, n_refs, R, na_cap, hit_slots);
  if (trace) fprintf(stderr, "[smhip] sync route: back enqueued %.2f ms (%llu anchors)\n", since(), (unsigned long long)NA);
  return rc;
}

// the speculative route: nothing is read back; capacities from the totals of earlier batches
int RunBatchSpeculative(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs) {
  const uint32_t R = ctx->n_chunks;
  uint64_t nq_cap = SpecCapQueries(ctx, R), na_cap = SpecCapAnchors(ctx, R);
  // ... or whatever the buffers already hold room for: a small batch after large ones (the tail of a streaming run, whose
  // junk chunks have several times the queries per chunk of the chunks before them) then never outgrows a capacity
  if (ctx->res_fast == p->hits_per_query_fast) nq_cap = std::max(nq_cap, std::min<uint64_t>(ctx->res_nq, 0xFFFFFFF0ull));
  na_cap = std::max(na_cap, ctx->res_na);
  // (knob spec_tight: 1 halves the query capacity, 2 the anchor capacity, so that the guards fire and the batch is
  // repeated on the synchronous route)
  if (ctx->knobs.spec_tight == 1) nq_cap = std::max<uint64_t>((uint64_t)R * ctx->hint_q_per_chunk / 2, 1);
  if (ctx->knobs.spec_tight == 2) na_cap = std::max<uint64_t>((uint64_t)R * ctx->hint_a_per_chunk / 2, 1);
  if (!ctx->h_run_info) HIPCHK(hipHostMalloc(&ctx->h_run_info, 256, hipHostMallocDefault));  // (smhip_init allocates it)
  int rc = EnqueueFront(ctx, p, R);
  if (rc) return rc;
  uint32_t *flags = ctx->counters.as<uint32_t>() + 32;  // [0] guard_q, [1] guard_a, [2] queries, [4..5] anchors
  // beyond the capacity: no queries for anybody downstream (the chunks then hold their carried anchors only)
  hipLaunchKernelGGL(GuardTotalKernel<uint32_t>, dim3(1), dim3(1024), 0, ctx->stream, ctx->nq.as<uint32_t>(),
                     ctx->q_off.as<uint32_t>(), R, (uint32_t)nq_cap, flags + 0, flags + 2);
  HIPCHK(hipGetLastError());
  HitSlots hit_slots;
  rc = EnqueueSearch(ctx, p, R, nq_cap, FirstPoolCap(ctx, nq_cap), &hit_slots);
  if (rc) return rc;
  hipLaunchKernelGGL(GuardTotalKernel<uint64_t>, dim3(1), dim3(1024), 0, ctx->stream, ctx->chunk_anchor_cnt.as<uint32_t>(),
                     ctx->anchor_off.as<uint64_t>(), R, na_cap, flags + 1, reinterpret_cast<uint64_t *>(flags + 4));
  HIPCHK(hipGetLastError());
  rc = EnqueueBack(ctx, p, n_refs, R, na_cap, hit_slots);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ctx->h_run_info, ctx->counters.p, sizeof(RunInfo), hipMemcpyDeviceToHost, ctx->stream));
  ctx->spec_pending = true;
  ++ctx->spec_runs;
  ctx->spec_params = *p;
  ctx->spec_n_refs = n_refs;
  return SMHIP_OK;
}

// Called by everything that hands results to the host: waits for the stream and, when a speculative batch
// outgrew a capacity (or the hit pool), maps the batch again the synchronous way -- and repeats the sDTW
// refinement that was enqueued behind it.
// the sticky piece-length error (FinishRun below): what every call that hands out results of the batch answers first
#define PIECES_BAD_MSG                                                                                                   \
  "smhip_reads_upload_pieces: the device's (30, 200) pA filter kept another number of samples of a piece than expect_len " \
  "says -- the chunk would not be the reference's chunk (upload the pieces and select the batch again)"
int FinishRun(smhip_ctx *ctx) {
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->h_piece_flag && *ctx->h_piece_flag) {
    // Sticky for the batch: its chunks are not the reference's chunks, so nothing computed from them may be handed out --
    // not by this call and not by a later one of a caller that ignored the error.  (A speculative batch's overflow record
    // is dropped with it: the batch has to be rebuilt from pieces anyway.)
    *ctx->h_piece_flag = 0;
    ctx->pieces_bad = true;
    ctx->spec_pending = false;
    ctx->run_done = false;
    ctx->batch_ready = false;
    ctx->refine_done = false;
    ctx->refine_path_done = false;
  }
  if (ctx->pieces_bad) return Fail(ctx, SMHIP_ECAP, PIECES_BAD_MSG);
  if (!ctx->spec_pending) return SMHIP_OK;
  ctx->spec_pending = false;
  const RunInfo *info = static_cast<const RunInfo *>(ctx->h_run_info);
  const bool overflow = info->w[32] || info->w[33] || info->w[3];
  LearnOverCap(ctx, &ctx->spec_params, info->w[34], info->w[7]);
  if (!overflow) {
    LearnTotals(ctx, std::max(ctx->n_chunks, 1u), info->w[34], info->anchors());
    return SMHIP_OK;
  }
  ++ctx->spec_reruns;
  // (from the features of the first pass: the read slots / sample buffer may hold other reads by now)
  int rc = RunBatchSync(ctx, &ctx->spec_params, ctx->spec_n_refs, /*reuse_features=*/true);
  if (rc) return rc;
  if (ctx->refine_done) rc = smhip_refine_run(ctx, ctx->refine_band, ctx->refine_pad);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SMHIP_OK;
}

// Share of the index census' sample with more than `cap` hits.
double CensusShareAbove(const smhip_ctx *ctx, uint32_t cap) {
  const auto &h = ctx->census_hits;
  if (h.empty()) return 0.0;
  return (double)(h.end() - std::upper_bound(h.begin(), h.end(), cap)) / (double)h.size();
}

// What the caller left automatic, decided from a census of the resident index (IndexCensus: the hit counts of 4096
// of its own points used as queries -- reads are noisy copies of such windows).
//  * hits_per_query_fast == 0: the power of two that holds three quarters of the sample's queries entirely, within
//    [512, 2048] and a sixth of the device (hits beyond the fast slots go to the shared overflow pool: correct at
//    any size, slower);
//  * search_mode AUTO: the set walk whenever the index has quad records.  Its queries beyond num_nearest_points hits
//    are settled by ranking the hits it found (seed_lookup.hip OverCapSelectKernel), which costs little however
//    many there are.  Only where that is not available (num_nearest_points > 8192, or the second_search knob) such
//    queries are searched a second time in the reference's order, one lane each: then an index whose census puts
//    more than set_walk_max_over_cap of the sample beyond the cap, or a context that has met as many in its
//    batches (LearnOverCap), searches everything in the reference's order at once.
int ResolveParams(smhip_ctx *ctx, smhip_params *rp, uint32_t n_chunks) {
  const bool want_fast = rp->hits_per_query_fast == 0;
  const bool ranked = (uint32_t)rp->num_nearest_points <= 8192u && !ctx->knobs.second_search;  // (kSelMaxKeep)
  const bool want_mode = rp->search_mode == SMHIP_SEARCH_AUTO && ctx->index.quads && !ranked && !ctx->walk_exact_only &&
                         !rp->reference_walk;
  if ((want_fast || want_mode) && (!ctx->census_valid || ctx->census_radius != rp->search_radius)) {
    HIPCHK(IndexCensus(ctx, *rp, 4096));
    ctx->census_valid = true;
    ctx->census_radius = rp->search_radius;
  }
  if (want_fast) {
    const auto &h = ctx->census_hits;
    const uint32_t q75 = h.empty() ? 0u : h[(h.size() * 3) / 4];
    uint32_t fast = 512;
    while (fast < 2048 && fast < q75) fast <<= 1;
    // ... within a sixth of the device for the slots of this batch's queries (about 288 per chunk until batches
    // have been seen)
    size_t free_b = 0, total_b = 0;
    if (fast > 512 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const uint64_t nq_est = (uint64_t)std::max(n_chunks, 1u) * (ctx->hint_q_per_chunk ? ctx->hint_q_per_chunk * 5 / 4 : 288);
      while (fast > 512 && nq_est * fast * sizeof(Hit) > total_b / 6) fast >>= 1;
    }
    rp->hits_per_query_fast = fast;
  }
  if (rp->search_mode == SMHIP_SEARCH_AUTO) {
    const bool exact = !ctx->index.quads || (!ranked && ctx->walk_exact_only) ||
                       (want_mode && CensusShareAbove(ctx, (uint32_t)rp->num_nearest_points) > (double)rp->set_walk_max_over_cap);
    rp->search_mode = exact ? SMHIP_SEARCH_EXACT : SMHIP_SEARCH_SET;
  }
  return SMHIP_OK;
}
}  // namespace
extern "C" {

int smhip_dev_knob(smhip_ctx *ctx, const char *name, long long value) {
  if (!ctx || !name) return SMHIP_EINVAL;
  if (!ctx->knobs.enabled)
    return Fail(ctx, SMHIP_ESTATE, "smhip_dev_knob: development knobs need SMHIP_DEV=1 in the environment of smhip_init");
  if (!strcmp(name, "pool_min")) {
    ctx->knobs.pool_min = value;
    return SMHIP_OK;
  }
  for (const KnobName &kn : kKnobNames)
    if (!strcmp(name, kn.name)) {
      ctx->knobs.*(kn.field) = (int)value;
      return SMHIP_OK;
    }
  return Fail(ctx, SMHIP_EINVAL, std::string("smhip_dev_knob: unknown knob ") + name);
}

int smhip_resolved_params(smhip_ctx *ctx, smhip_params *out) {
  if (!ctx || !out) return SMHIP_EINVAL;
  if (!ctx->run_done) return Fail(ctx, SMHIP_ESTATE, "smhip_resolved_params before smhip_map_run");
  *out = ctx->last_params;
  return SMHIP_OK;
}

int smhip_index_census(smhip_ctx *ctx, const smhip_params *p, uint32_t *hits, uint32_t cap, uint32_t *n_out) {
  if (!ctx || !p || !n_out) return SMHIP_EINVAL;
  if (!ctx->have_index) return Fail(ctx, SMHIP_ESTATE, "smhip_index_census before an index is resident");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->spec_pending) {
    int rc = FinishRun(ctx);
    if (rc) return rc;
  }
  if (!ctx->census_valid || ctx->census_radius != p->search_radius) {
    HIPCHK(IndexCensus(ctx, *p, 4096));
    ctx->census_valid = true;
    ctx->census_radius = p->search_radius;
  }
  *n_out = (uint32_t)ctx->census_hits.size();
  if (hits) memcpy(hits, ctx->census_hits.data(), std::min<size_t>(cap, ctx->census_hits.size()) * sizeof(uint32_t));
  return SMHIP_OK;
}

int smhip_map_run(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs) {
  if (!ctx || !p) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (!ctx->have_index) return Fail(ctx, SMHIP_ESTATE, "smhip_map_run before an index is resident");
  if (!ctx->batch_ready) return Fail(ctx, SMHIP_ESTATE, "smhip_map_run before smhip_batch_upload");
  if (n_refs == 0 || p->step_size < 1 || p->num_nearest_points < 1 || p->num_best_chains > 3)
    return Fail(ctx, SMHIP_EINVAL, "bad parameter");
  if (p->hits_per_query_fast > (1u << 20)) return Fail(ctx, SMHIP_EINVAL, "hits_per_query_fast must be 0 (automatic) .. 2^20");
  if (p->walk_blocks_per_cu > 8) return Fail(ctx, SMHIP_EINVAL, "walk_blocks_per_cu must be 0..8");
  if (p->search_mode > SMHIP_SEARCH_SET) return Fail(ctx, SMHIP_EINVAL, "search_mode must be SMHIP_SEARCH_AUTO / _EXACT / _SET");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->spec_pending) {  // the previous batch was never collected: settle it first
    int rc = FinishRun(ctx);
    if (rc) return rc;
  }
  const uint32_t R = ctx->n_chunks;
  ctx->run_done = false;
  if (R == 0) {
    ctx->run_done = true;
    return SMHIP_OK;
  }
  // The index census (first batch after the index arrived) resolves what the caller left automatic.
  smhip_params rp = *p;
  {
    const int rc0 = ResolveParams(ctx, &rp, R);
    if (rc0) return rc0;
  }
  p = &rp;
  ctx->last_params = rp;
  // (knob sync_run: always the synchronous route -- an A/B switch; results are the same)
  const bool speculate = ctx->hint_valid && !ctx->knobs.sync_run && ctx->knobs.pool_min < 0 &&
                         CapacitiesAffordable(ctx, p, SpecCapQueries(ctx, R), SpecCapAnchors(ctx, R));
  const int rc = speculate ? RunBatchSpeculative(ctx, p, n_refs) : RunBatchSync(ctx, p, n_refs);
  if (rc) return rc;
  ctx->run_done = true;
  return SMHIP_OK;
}

int smhip_map_sync(smhip_ctx *ctx) {
  if (!ctx) return SMHIP_EINVAL;
  HIPCHK(hipSetDevice(ctx->device));
  int rc = FinishRun(ctx);
  if (rc) return rc;
  ResolveTimers(ctx);
  return SMHIP_OK;
}

int smhip_map_download(smhip_ctx *ctx, smhip_chunk_result *results, smhip_carried_anchor *chain_anchors,
                       uint64_t chain_anchor_cap, uint64_t *chain_anchor_offsets) {
  if (!ctx || !results) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (ctx->pieces_bad) return Fail(ctx, SMHIP_ECAP, PIECES_BAD_MSG);
  if (!ctx->run_done) return Fail(ctx, SMHIP_ESTATE, "smhip_map_download before smhip_map_run");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t R = ctx->n_chunks;
  if (R == 0) {
    if (chain_anchor_offsets) chain_anchor_offsets[0] = 0;
    return SMHIP_OK;
  }
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  HIPCHK(StageReserve(ctx, (size_t)R * sizeof(smhip_chunk_result)));
  HIPCHK(hipMemcpyAsync(ctx->h_stage, ctx->results.p, (uint64_t)R * sizeof(smhip_chunk_result), hipMemcpyDeviceToHost,
                        ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(results, ctx->h_stage, (size_t)R * sizeof(smhip_chunk_result));
  ResolveTimers(ctx);
  if (chain_anchors && chain_anchor_offsets) {
    // (context-owned scratch: a hipFree here would synchronise the whole device, i.e. stall the other
    // contexts' streams)
    DevBuf &out_off = ctx->dl_off, &compact = ctx->dl_compact;
    HIPCHK(out_off.Reserve(((uint64_t)R + 1) * sizeof(uint64_t)));
    HIPCHK(LaunchScanU32ToU64(ctx, ctx->out_anchor_cnt.as<uint32_t>(), R, out_off.as<uint64_t>()));
    HIPCHK(hipMemcpyAsync(chain_anchor_offsets, out_off.p, ((uint64_t)R + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t total = chain_anchor_offsets[R];
    if (total > chain_anchor_cap) return Fail(ctx, SMHIP_ECAP, "chain_anchor_cap too small");
    if (total > 0) {
      HIPCHK(compact.Reserve(total * sizeof(smhip_carried_anchor)));
      hipLaunchKernelGGL(CompactAnchorsKernel, dim3(R), dim3(64), 0, ctx->stream,
                         ctx->out_anchors.as<smhip_carried_anchor>(), ctx->anchor_off.as<uint64_t>(),
                         out_off.as<uint64_t>(), compact.as<smhip_carried_anchor>());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(chain_anchors, compact.p, total * sizeof(smhip_carried_anchor), hipMemcpyDeviceToHost,
                            ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
  }
  return SMHIP_OK;
}

int smhip_map_download_anchors_of(smhip_ctx *ctx, const uint32_t *chunks, uint32_t n_listed, smhip_carried_anchor *chain_anchors,
                                  uint64_t chain_anchor_cap, uint64_t *chain_anchor_offsets) {
  if (!ctx || !chain_anchor_offsets || (n_listed && (!chunks || !chain_anchors))) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (ctx->pieces_bad) return Fail(ctx, SMHIP_ECAP, PIECES_BAD_MSG);
  if (!ctx->run_done) return Fail(ctx, SMHIP_ESTATE, "smhip_map_download_anchors_of before smhip_map_run");
  HIPCHK(hipSetDevice(ctx->device));
  chain_anchor_offsets[0] = 0;
  if (n_listed == 0 || ctx->n_chunks == 0) return SMHIP_OK;
  for (uint32_t i = 0; i < n_listed; ++i)
    if (chunks[i] >= ctx->n_chunks) return Fail(ctx, SMHIP_EINVAL, "listed chunk outside the batch");
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  const uint64_t n = n_listed;
  DevBuf &out_off = ctx->dl_off, &compact = ctx->dl_compact;
  // [offsets: n + 1 x u64][list: n x u32][counts: n x u32]
  HIPCHK(out_off.Reserve((n + 1) * sizeof(uint64_t) + 2 * n * sizeof(uint32_t)));
  uint32_t *d_list = reinterpret_cast<uint32_t *>(out_off.as<uint64_t>() + n + 1), *d_cnt = d_list + n;
  HIPCHK(StageReserve(ctx, (n + 1) * sizeof(uint64_t)));
  memcpy(ctx->h_stage, chunks, n * sizeof(uint32_t));
  HIPCHK(hipMemcpyAsync(d_list, ctx->h_stage, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(ListedAnchorCountKernel, dim3((n_listed + 255) / 256), dim3(256), 0, ctx->stream, d_list, n_listed,
                     ctx->out_anchor_cnt.as<uint32_t>(), d_cnt);
  HIPCHK(hipGetLastError());
  HIPCHK(LaunchScanU32ToU64(ctx, d_cnt, n_listed, out_off.as<uint64_t>()));
  HIPCHK(hipStreamSynchronize(ctx->stream));  // (the list has left the staging)
  HIPCHK(hipMemcpyAsync(ctx->h_stage, out_off.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(chain_anchor_offsets, ctx->h_stage, (n + 1) * sizeof(uint64_t));
  const uint64_t total = chain_anchor_offsets[n];
  if (total > chain_anchor_cap) return Fail(ctx, SMHIP_ECAP, "chain_anchor_cap too small");
  if (total == 0) return SMHIP_OK;
  HIPCHK(compact.Reserve(total * sizeof(smhip_carried_anchor)));
  hipLaunchKernelGGL(CompactListedAnchorsKernel, dim3(n_listed), dim3(64), 0, ctx->stream,
                     ctx->out_anchors.as<smhip_carried_anchor>(), ctx->anchor_off.as<uint64_t>(), d_list, out_off.as<uint64_t>(),
                     compact.as<smhip_carried_anchor>());
  HIPCHK(hipGetLastError());
  // (through the pinned staging: one DMA instead of the pieces a copy into pageable memory is cut into)
  HIPCHK(StageReserve(ctx, total * sizeof(smhip_carried_anchor)));
  HIPCHK(hipMemcpyAsync(ctx->h_stage, compact.p, total * sizeof(smhip_carried_anchor), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(chain_anchors, ctx->h_stage, total * sizeof(smhip_carried_anchor));
  return SMHIP_OK;
}

int smhip_map_download_chains(smhip_ctx *ctx, smhip_chain *chains, uint64_t chain_cap, uint64_t *chain_offsets) {
  if (!ctx || !chain_offsets) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (ctx->pieces_bad) return Fail(ctx, SMHIP_ECAP, PIECES_BAD_MSG);
  if (!ctx->run_done) return Fail(ctx, SMHIP_ESTATE, "smhip_map_download_chains before smhip_map_run");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t R = ctx->n_chunks;
  chain_offsets[0] = 0;
  if (R == 0) return SMHIP_OK;
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  DevBuf &out_off = ctx->dl_off, &compact = ctx->dl_compact;
  HIPCHK(out_off.Reserve(((uint64_t)R + 1) * sizeof(uint64_t) + (uint64_t)R * sizeof(uint32_t)));
  uint32_t *cnt = reinterpret_cast<uint32_t *>(out_off.as<uint64_t>() + R + 1);
  HIPCHK(LaunchChainCount(ctx, ctx->results.as<smhip_chunk_result>(), R, cnt));
  HIPCHK(LaunchScanU32ToU64(ctx, cnt, R, out_off.as<uint64_t>()));
  HIPCHK(hipMemcpyAsync(chain_offsets, out_off.p, ((uint64_t)R + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t total = chain_offsets[R];
  if (!chains) return SMHIP_OK;  // (sizes only)
  if (total > chain_cap) return Fail(ctx, SMHIP_ECAP, "chain_cap too small (chain_offsets[n_chunks] records are needed)");
  if (total == 0) return SMHIP_OK;
  if (!ctx->dl_chain_ranked) return Fail(ctx, SMHIP_ESTATE, "no chains resident");
  HIPCHK(compact.Reserve(total * sizeof(smhip_chain)));
  HIPCHK(LaunchChainGather(ctx, ctx->results.as<smhip_chunk_result>(), R, out_off.as<uint64_t>(), compact.as<smhip_chain>()));
  HIPCHK(hipMemcpyAsync(chains, compact.p, total * sizeof(smhip_chain), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SMHIP_OK;
}

int smhip_map_chunks(smhip_ctx *ctx, const smhip_params *p, uint32_t n_refs, const float *samples_pa,
                     const uint32_t *sample_offsets, uint32_t n_chunks, const uint32_t *query_start_offsets,
                     const smhip_carried_anchor *carried, const uint32_t *carried_offsets,
                     smhip_chunk_result *results) {
  int rc = smhip_batch_upload(ctx, samples_pa, sample_offsets, n_chunks, query_start_offsets, carried, carried_offsets);
  if (rc) return rc;
  rc = smhip_map_run(ctx, p, n_refs);
  if (rc) return rc;
  rc = smhip_map_sync(ctx);
  if (rc) return rc;
  return smhip_map_download(ctx, results, nullptr, 0, nullptr);
}

int smhip_reference_upload(smhip_ctx *ctx, const float *signals, const uint64_t *offsets, uint32_t n_refs) {
  if (!ctx || !signals || !offsets || n_refs == 0) return Fail(ctx, SMHIP_EINVAL, "null argument");
  for (uint32_t i = 0; i < 2 * n_refs; ++i)
    if (offsets[i + 1] < offsets[i]) return Fail(ctx, SMHIP_EINVAL, "offsets must be non-decreasing");
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t total = offsets[2 * n_refs];
  HIPCHK(ctx->ref_sig.Reserve(std::max<uint64_t>(total, 1) * sizeof(float)));
  HIPCHK(ctx->ref_off.Reserve((2ull * n_refs + 1) * sizeof(uint64_t)));
  HIPCHK(hipMemcpyAsync(ctx->ref_sig.p, signals, total * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->ref_off.p, offsets, (2ull * n_refs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                        ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->ref_n = n_refs;
  ctx->d_ref_sig = ctx->ref_sig.as<float>();
  ctx->d_ref_off = ctx->ref_off.as<uint64_t>();
  return SMHIP_OK;
}

int smhip_reference_share(smhip_ctx *ctx, smhip_ctx *owner) {
  if (!ctx || !owner || ctx == owner) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (owner->ref_n == 0 || owner->d_ref_sig != owner->ref_sig.as<float>())
    return Fail(ctx, SMHIP_ESTATE, "smhip_reference_share: the owner holds no reference signal of its own");
  if (owner->device != ctx->device) return Fail(ctx, SMHIP_EINVAL, "smhip_reference_share: contexts of one GPU only");
  ctx->ref_n = owner->ref_n;
  ctx->d_ref_sig = owner->d_ref_sig;
  ctx->d_ref_off = owner->d_ref_off;
  return SMHIP_OK;
}

int smhip_mem_info(smhip_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (!ctx) return SMHIP_EINVAL;
  HIPCHK(hipSetDevice(ctx->device));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return SMHIP_OK;
}

int smhip_refine_run(smhip_ctx *ctx, uint32_t band, uint32_t pad) {
  if (!ctx) return SMHIP_EINVAL;
  if (ctx->pieces_bad) return Fail(ctx, SMHIP_ECAP, PIECES_BAD_MSG);
  if (!ctx->run_done) return Fail(ctx, SMHIP_ESTATE, "smhip_refine_run before smhip_map_run");
  if (ctx->ref_n == 0) return Fail(ctx, SMHIP_ESTATE, "smhip_refine_run before smhip_reference_upload");
  if (band == 0) return Fail(ctx, SMHIP_EINVAL, "band must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t R = ctx->n_chunks;
  ctx->refine_done = false;
  ctx->refine_band = band;  // (FinishRun repeats the refinement behind a batch it had to map again)
  ctx->refine_pad = pad;
  if (R == 0) {
    ctx->refine_done = true;
    return SMHIP_OK;
  }
  HIPCHK(ctx->refine_dist.Reserve((uint64_t)R * sizeof(float)));
  HIPCHK(ctx->refine_end.Reserve((uint64_t)R * sizeof(int64_t)));
  {
    ScopedTimer t(ctx, T_SDTW);
    HIPCHK(LaunchSdtwChains(ctx, ctx->results.as<smhip_chunk_result>(), ctx->feat.as<float>(), kFeatStride,
                            ctx->qstart.as<uint32_t>(), R, ctx->d_ref_sig, ctx->d_ref_off,
                            ctx->ref_n, band, pad, ctx->refine_dist.as<float>(), ctx->refine_end.as<int64_t>()));
  }
  ctx->refine_done = true;
  return SMHIP_OK;
}

int smhip_refine_download(smhip_ctx *ctx, float *dist, int64_t *end_pos) {
  if (!ctx || !dist || !end_pos) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (!ctx->refine_done) return Fail(ctx, SMHIP_ESTATE, "smhip_refine_download before smhip_refine_run");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t R = ctx->n_chunks;
  if (R == 0) return SMHIP_OK;
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  HIPCHK(StageReserve(ctx, (size_t)R * (sizeof(float) + sizeof(int64_t))));
  int64_t *st_end = static_cast<int64_t *>(ctx->h_stage);
  float *st_dist = reinterpret_cast<float *>(st_end + R);
  HIPCHK(hipMemcpyAsync(st_dist, ctx->refine_dist.p, (uint64_t)R * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(st_end, ctx->refine_end.p, (uint64_t)R * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(dist, st_dist, (size_t)R * sizeof(float));
  memcpy(end_pos, st_end, (size_t)R * sizeof(int64_t));
  ResolveTimers(ctx);
  return SMHIP_OK;
}

int smhip_refine_path_run(smhip_ctx *ctx, uint32_t band, uint32_t pad) {
  if (!ctx) return SMHIP_EINVAL;
  if (ctx->pieces_bad) return Fail(ctx, SMHIP_ECAP, PIECES_BAD_MSG);
  if (!ctx->run_done) return Fail(ctx, SMHIP_ESTATE, "smhip_refine_path_run before smhip_map_run");
  if (ctx->ref_n == 0) return Fail(ctx, SMHIP_ESTATE, "smhip_refine_path_run before smhip_reference_upload");
  if (band == 0) return Fail(ctx, SMHIP_EINVAL, "band must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t R = ctx->n_chunks;
  ctx->refine_path_done = false;
  ctx->rp_path_total = 0;
  if (R == 0) {
    ctx->refine_path_done = true;
    return SMHIP_OK;
  }
  {  // (this call sizes its scratch from the batch's results: they must be final)
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  const smhip_chunk_result *results = ctx->results.as<smhip_chunk_result>();
  const uint64_t r1 = (uint64_t)R + 1;
  HIPCHK(ctx->rp_rows.Reserve(r1 * 4));
  HIPCHK(ctx->rp_cap.Reserve(r1 * 4));
  HIPCHK(ctx->rp_toff.Reserve(r1 * 8));
  HIPCHK(ctx->rp_poff.Reserve(r1 * 8));
  HIPCHK(ctx->rp_tgt.Reserve(r1 * 8));
  HIPCHK(ctx->rp_qoff.Reserve(r1 * 8));
  HIPCHK(ctx->rp_qlen.Reserve(r1 * 4));
  HIPCHK(ctx->rp_dist.Reserve(r1 * 4));
  HIPCHK(ctx->rp_end.Reserve(r1 * 8));
  HIPCHK(ctx->rp_plen.Reserve(r1 * 4));
  HIPCHK(ctx->rp_start.Reserve(r1 * 8));
  ScopedTimer t(ctx, T_SDTW);
  // rows and path capacity (T + Q) of every chunk's window, their prefix sums, the totals
  HIPCHK(LaunchChainWindowRows(ctx, results, ctx->qstart.as<uint32_t>(), R, ctx->d_ref_off, ctx->ref_n, pad,
                               ctx->rp_rows.as<uint32_t>(), ctx->rp_cap.as<uint32_t>()));
  HIPCHK(LaunchScanU32ToU64(ctx, ctx->rp_rows.as<uint32_t>(), R, ctx->rp_toff.as<uint64_t>()));
  HIPCHK(LaunchScanU32ToU64(ctx, ctx->rp_cap.as<uint32_t>(), R, ctx->rp_poff.as<uint64_t>()));
  uint64_t totals[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(&totals[0], ctx->rp_toff.as<uint64_t>() + R, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&totals[1], ctx->rp_poff.as<uint64_t>() + R, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t n_rows = totals[0], n_path = totals[1];
  if (n_rows > 0x7FFFFFFFull) return Fail(ctx, SMHIP_ECAP, "too many window rows in one batch");
  const uint64_t nr = std::max<uint64_t>(n_rows, 1);
  HIPCHK(ctx->rp_lo.Reserve(nr * 4));
  HIPCHK(ctx->rp_hi.Reserve(nr * 4));
  HIPCHK(ctx->rp_width.Reserve((nr + 1) * 4));
  HIPCHK(ctx->rp_celloff.Reserve((nr + 1) * 8));
  HIPCHK(ctx->rp_path.Reserve(std::max<uint64_t>(n_path, 1)));
  HIPCHK(LaunchChainWindowFill(ctx, results, ctx->qstart.as<uint32_t>(), kFeatStride, R, ctx->d_ref_off,
                               ctx->ref_n, band, pad, ctx->rp_toff.as<uint64_t>(), ctx->rp_tgt.as<uint64_t>(),
                               ctx->rp_qoff.as<uint64_t>(), ctx->rp_qlen.as<uint32_t>(), ctx->rp_lo.as<uint32_t>(),
                               ctx->rp_hi.as<uint32_t>(), ctx->rp_width.as<uint32_t>()));
  // cell offsets in 64 bits: up to 8192 rows of min(band + 1, features) cells per chunk pass 2^32 cells
  // in one batch with long carried chains or a wide band
  uint64_t n_cells = 0;
  if (n_rows > 0) {
    HIPCHK(LaunchScanU32ToU64(ctx, ctx->rp_width.as<uint32_t>(), (uint32_t)n_rows, ctx->rp_celloff.as<uint64_t>()));
    HIPCHK(hipMemcpyAsync(&n_cells, ctx->rp_celloff.as<uint64_t>() + n_rows, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n_cells > (1ull << 36))
      return Fail(ctx, SMHIP_ECAP, "more than 64 Gi window cells (one path-code byte each) in one batch: map fewer chunks "
                                   "per batch or narrow the band");
  }
  HIPCHK(ctx->rp_codes.Reserve(std::max<uint64_t>(n_cells, 1)));
  HIPCHK(LaunchDtwPathChains(ctx, ctx->d_ref_sig, ctx->rp_toff.as<uint64_t>(), ctx->rp_tgt.as<uint64_t>(),
                             ctx->feat.as<float>(), ctx->rp_qoff.as<uint64_t>(), ctx->rp_qlen.as<uint32_t>(),
                             ctx->rp_lo.as<uint32_t>(), ctx->rp_hi.as<uint32_t>(), ctx->rp_celloff.as<uint64_t>(),
                             ctx->rp_codes.as<uint8_t>(), R, ctx->rp_dist.as<float>(), ctx->rp_end.as<int64_t>(),
                             ctx->rp_path.as<uint8_t>(), ctx->rp_poff.as<uint64_t>(), ctx->rp_plen.as<uint32_t>(),
                             ctx->rp_start.as<int64_t>()));
  ctx->rp_path_total = n_path;
  ctx->refine_path_done = true;
  return SMHIP_OK;
}

int smhip_refine_path_download(smhip_ctx *ctx, float *dist, int64_t *end_pos, int64_t *start_pos, uint32_t *path_len,
                               uint64_t *path_off, uint8_t *path_codes, uint64_t path_cap) {
  if (!ctx || !dist || !end_pos || !start_pos || !path_len || !path_off) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (!ctx->refine_path_done) return Fail(ctx, SMHIP_ESTATE, "smhip_refine_path_download before smhip_refine_path_run");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t R = ctx->n_chunks;
  path_off[0] = 0;
  if (R == 0) return SMHIP_OK;
  // (everything but the codes is delivered also when path_cap is too small: the caller sizes its retry
  // from path_off[n_chunks], as the header says)
  const bool codes_fit = !path_codes || path_cap >= ctx->rp_path_total;
  std::vector<uint64_t> tgt((size_t)R), toff((size_t)R + 1);
  HIPCHK(hipMemcpyAsync(dist, ctx->rp_dist.p, (uint64_t)R * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(end_pos, ctx->rp_end.p, (uint64_t)R * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(start_pos, ctx->rp_start.p, (uint64_t)R * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(path_len, ctx->rp_plen.p, (uint64_t)R * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(path_off, ctx->rp_poff.p, ((uint64_t)R + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (path_codes && codes_fit && ctx->rp_path_total)
    HIPCHK(hipMemcpyAsync(path_codes, ctx->rp_path.p, ctx->rp_path_total, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ResolveTimers(ctx);
  // window-relative rows -> absolute positions on the strand's signal, like smhip_refine_download:
  // row r of chunk c is target index t0 + r, t0 = max(0, start - pad) = (window offset) - (strand offset)
  std::vector<uint64_t> ref_off((size_t)2 * ctx->ref_n + 1);
  HIPCHK(hipMemcpyAsync(ref_off.data(), ctx->d_ref_off, ref_off.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(tgt.data(), ctx->rp_tgt.p, (uint64_t)R * 8, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<smhip_chunk_result> res((size_t)R);
  HIPCHK(hipMemcpyAsync(res.data(), ctx->results.p, (uint64_t)R * sizeof(smhip_chunk_result), hipMemcpyDeviceToHost,
                        ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint32_t c = 0; c < R; ++c) {
    if (end_pos[c] < 0) continue;
    const smhip_chain &ch = res[c].chains[0];
    const uint64_t so = ref_off[2 * ch.ref + (ch.direction == 1 ? 0 : 1)];
    const int64_t t0 = (int64_t)(tgt[c] - so);
    end_pos[c] += t0;
    if (start_pos[c] >= 0) start_pos[c] += t0;
    if (path_codes && codes_fit) std::reverse(path_codes + path_off[c], path_codes + path_off[c] + path_len[c]);
  }
  if (!codes_fit) return Fail(ctx, SMHIP_ECAP, "path_cap too small (path_off[n_chunks] codes are needed)");
  return SMHIP_OK;
}

int smhip_debug_counters(smhip_ctx *ctx, uint32_t out[16]) {
  if (!ctx || !out) return SMHIP_EINVAL;
  memset(out, 0, 16 * sizeof(uint32_t));
  if (!ctx->counters.p) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  HIPCHK(hipMemcpyAsync(out, ctx->counters.p, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  out[4] = ctx->spec_reruns;
  out[5] = ctx->spec_runs;
  out[6] = (uint32_t)ctx->last_walk_mode;
  return SMHIP_OK;
}

int smhip_debug_sort_stats(smhip_ctx *ctx, uint32_t out[4]) {
  if (!ctx || !out) return SMHIP_EINVAL;
  memset(out, 0, 4 * sizeof(uint32_t));
  if (!ctx->sort_tmp.p || !ctx->run_done) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  HIPCHK(hipMemcpyAsync(out, ctx->sort_tmp.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));  // (LaunchSort: the counters lead the scratch)
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SMHIP_OK;
}

int smhip_timing_enable(smhip_ctx *ctx, int on) {
  if (!ctx) return SMHIP_EINVAL;
  ctx->timing = on != 0;
  return SMHIP_OK;
}

int smhip_timing_reset(smhip_ctx *ctx) {
  if (!ctx) return SMHIP_EINVAL;
  for (auto &t : ctx->timers) {
    t.total_ms = 0;
    t.launches = 0;
  }
  return SMHIP_OK;
}

int smhip_timing_get(smhip_ctx *ctx, const char *kernel, double *mean_ms, uint64_t *launches) {
  if (!ctx || !kernel || !mean_ms || !launches) return SMHIP_EINVAL;
  for (auto &t : ctx->timers) {
    if (t.name == kernel) {
      *launches = t.launches;
      *mean_ms = t.launches ? t.total_ms / (double)t.launches : 0.0;
      return SMHIP_OK;
    }
  }
  return Fail(ctx, SMHIP_EINVAL, "unknown kernel name");
}

int smhip_events(smhip_ctx *ctx, const smhip_params *p, const float *samples_pa, const uint32_t *sample_offsets,
                 uint32_t n_chunks, uint32_t feat_stride, float *feat, float *stdv, uint32_t *n_features,
                 uint32_t *n_events, uint32_t *status) {
  if (!ctx || !p || !feat || !n_features) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (feat_stride < (uint32_t)kFeatStride) return Fail(ctx, SMHIP_EINVAL, "feat_stride must be >= 2736");
  int rc = smhip_batch_upload(ctx, samples_pa, sample_offsets, n_chunks, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (n_chunks == 0) return SMHIP_OK;
  const uint64_t nf = (uint64_t)n_chunks * feat_stride;
  HIPCHK(ctx->feat.Reserve(nf * sizeof(float)));
  HIPCHK(ctx->stdv.Reserve(nf * sizeof(float)));
  HIPCHK(ctx->results.Reserve((uint64_t)n_chunks * sizeof(smhip_chunk_result)));
  HIPCHK(ctx->nq.Reserve((uint64_t)n_chunks * sizeof(uint32_t)));
  {
    ScopedTimer t(ctx, T_EVENTS);
    HIPCHK(LaunchEvents(ctx, *p, ctx->chunk_src, ctx->chunk_begin.as<uint64_t>(), ctx->chunk_len.as<uint32_t>(), n_chunks, feat_stride,
                        ctx->feat.as<float>(), ctx->stdv.as<float>(), ctx->results.as<smhip_chunk_result>(),
                        ctx->nq.as<uint32_t>()));
  }
  std::vector<smhip_chunk_result> res(n_chunks);
  HIPCHK(hipMemcpyAsync(feat, ctx->feat.p, nf * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (stdv) HIPCHK(hipMemcpyAsync(stdv, ctx->stdv.p, nf * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(res.data(), ctx->results.p, (uint64_t)n_chunks * sizeof(smhip_chunk_result),
                        hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ResolveTimers(ctx);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    n_features[c] = res[c].n_features;
    if (n_events) n_events[c] = res[c].n_events;
    if (status) status[c] = res[c].status;
  }
  ctx->run_done = false;
  return SMHIP_OK;
}

int smhip_events_raw(smhip_ctx *ctx, uint32_t n_chunks, uint32_t feat_stride, uint32_t row_stride, uint32_t *n_peaks,
                     uint32_t *peaks, uint32_t *ev_start, uint32_t *ev_length, float *ev_mean, float *ev_stdv,
                     float *ps, float *pss, float *t1, float *t2) {
  if (!ctx) return SMHIP_EINVAL;
  const auto &r = ctx->ev_rows;
  if (r.n_chunks == 0 || !ctx->batch_ready) return Fail(ctx, SMHIP_ESTATE, "smhip_events_raw before smhip_events / smhip_map_run");
  if (n_chunks != r.n_chunks || feat_stride != r.feat_stride)
    return Fail(ctx, SMHIP_EINVAL, "n_chunks / feat_stride differ from the last event-detection launch");
  if ((ps || pss || t1 || t2) && row_stride < (uint32_t)kMaxSamples + 1)
    return Fail(ctx, SMHIP_EINVAL, "row_stride must be >= 4097");
  HIPCHK(hipSetDevice(ctx->device));
  {
    int rc0 = FinishRun(ctx);
    if (rc0) return rc0;
  }
  const uint64_t evn = (uint64_t)n_chunks * feat_stride;
  DevBuf d_start, d_len;
  std::vector<uint16_t> h_peaks;
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t x) {
    if (e == hipSuccess) e = x;
  };
  if (ev_start || ev_length) {
    step(d_start.Reserve(evn * 4));
    step(d_len.Reserve(evn * 4));
    if (e == hipSuccess) step(hipMemsetAsync(d_start.p, 0, evn * 4, ctx->stream));
    if (e == hipSuccess) step(hipMemsetAsync(d_len.p, 0, evn * 4, ctx->stream));
    if (e == hipSuccess) step(LaunchEventSpans(ctx, ctx->chunk_len.as<uint32_t>(), d_start.as<uint32_t>(), d_len.as<uint32_t>()));
    if (e == hipSuccess && ev_start) step(hipMemcpyAsync(ev_start, d_start.p, evn * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (e == hipSuccess && ev_length) step(hipMemcpyAsync(ev_length, d_len.p, evn * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (e == hipSuccess && n_peaks) step(hipMemcpyAsync(n_peaks, r.n_peaks, (uint64_t)n_chunks * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess && ev_mean) step(hipMemcpyAsync(ev_mean, r.ev_mean, evn * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess && ev_stdv) step(hipMemcpyAsync(ev_stdv, r.ev_stdv, evn * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess && peaks) {
    h_peaks.resize((size_t)n_chunks * r.row_stride);
    step(hipMemcpyAsync(h_peaks.data(), r.peaks, h_peaks.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
  }
  const float *rows[4] = {r.ps, r.pss, r.t1, r.t2};
  float *outs[4] = {ps, pss, t1, t2};
  for (int k = 0; k < 4; ++k)
    if (e == hipSuccess && outs[k])
      step(hipMemcpy2DAsync(outs[k], (size_t)row_stride * 4, rows[k], (size_t)r.row_stride * 4,
                            (size_t)(kMaxSamples + 1) * 4, n_chunks, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipStreamSynchronize(ctx->stream));
  d_start.Free();
  d_len.Free();
  if (e != hipSuccess) return HipFail(ctx, e, "smhip_events_raw");
  if (peaks)  // rows of u16 sample indices -> n_chunks x feat_stride u32 (a chunk has at most feat_stride peaks)
    for (uint32_t c = 0; c < n_chunks; ++c)
      for (uint32_t k = 0; k < feat_stride; ++k)
        peaks[(size_t)c * feat_stride + k] = k < r.row_stride ? h_peaks[(size_t)c * r.row_stride + k] : 0;
  return SMHIP_OK;
}

int smhip_cwt_features(smhip_ctx *ctx, const float *samples_pa, const uint32_t *sample_offsets, uint32_t n_chunks,
                       float scale0, uint32_t feat_stride, float *feat, uint32_t *positions, uint32_t *n_features,
                       float *cwt_out) {
  if (!ctx || !feat || !positions || !n_features) return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (!(scale0 >= 4.0f && scale0 <= 64.0f)) return Fail(ctx, SMHIP_EINVAL, "scale0 must be in [4, 64]");
  if (feat_stride == 0) return Fail(ctx, SMHIP_EINVAL, "feat_stride must be positive");
  int rc = smhip_batch_upload(ctx, samples_pa, sample_offsets, n_chunks, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (n_chunks == 0) return SMHIP_OK;
  // impulse response of the reference's spectral filter (cwt.cc:1595-1630, 1762-1841) for the
  // padded length every chunk of up to 4096 samples gets (cwt.cc:1698-1700: 8192), in double
  const size_t npad = 8192;
  const double pi = 4.0 * atan(1.0), s = (double)scale0;
  const double norm = sqrt(2.0 * pi * s) * sqrt(1.0 / (0.75 * sqrt(pi)));  // Gamma(2.5) = 3 sqrt(pi) / 4
  const uint32_t L = (uint32_t)(9.0 * s) + 4;
  std::vector<double> D(npad / 2 + 1), h(L + 1);
  for (size_t k = 0; k <= npad / 2; ++k) {
    const double sk = s * ((double)k * 2.0 * pi / (double)npad);
    D[k] = norm * sk * sk * exp(-0.5 * sk * sk);
  }
  for (size_t j = 0; j <= L; ++j) {
    double acc = D[0] + ((npad / 2) * j % 2 ? -D[npad / 2] : D[npad / 2]);
    for (size_t k = 1; k < npad / 2; ++k) acc += 2.0 * D[k] * cos(2.0 * pi * (double)((k * j) % npad) / (double)npad);
    h[j] = acc / (double)npad;
  }
  DevBuf d_h, d_rows, d_feat, d_pos, d_n;
  auto free_all = [&]() {
    for (DevBuf *b : {&d_h, &d_rows, &d_feat, &d_pos, &d_n}) b->Free();
  };
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  const uint64_t nf = (uint64_t)n_chunks * feat_stride;
  step(d_h.Reserve((L + 1) * sizeof(double)));
  step(d_rows.Reserve((uint64_t)n_chunks * 4096 * sizeof(float)));
  step(d_feat.Reserve(nf * sizeof(float)));
  step(d_pos.Reserve(nf * sizeof(uint32_t)));
  step(d_n.Reserve((uint64_t)n_chunks * sizeof(uint32_t)));
  if (e == hipSuccess) step(hipMemcpyAsync(d_h.p, h.data(), (L + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemsetAsync(d_rows.p, 0, (uint64_t)n_chunks * 4096 * sizeof(float), ctx->stream));
  if (e == hipSuccess)
    step(LaunchCwtFeatures(ctx, ctx->chunk_src, ctx->chunk_begin.as<uint64_t>(), ctx->chunk_len.as<uint32_t>(), n_chunks, d_h.as<double>(), L,
                           d_rows.as<float>(), feat_stride, d_feat.as<float>(), d_pos.as<uint32_t>(),
                           d_n.as<uint32_t>()));
  if (e == hipSuccess) step(hipMemcpyAsync(feat, d_feat.p, nf * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(positions, d_pos.p, nf * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess)
    step(hipMemcpyAsync(n_features, d_n.p, (uint64_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess && cwt_out)
    step(hipMemcpyAsync(cwt_out, d_rows.p, (uint64_t)n_chunks * 4096 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipStreamSynchronize(ctx->stream));
  free_all();
  if (e != hipSuccess) return HipFail(ctx, e, "smhip_cwt_features");
  ctx->run_done = false;
  return SMHIP_OK;
}

int smhip_seed_lookup(smhip_ctx *ctx, const smhip_params *p, const float *queries, uint32_t n_queries, uint32_t cap,
                      uint32_t *counts, uint64_t *point_idx, float *dist, uint64_t *nodes_visited,
                      uint64_t *points_tested) {
  if (!ctx || !p || !queries || !counts || !point_idx || !dist || cap == 0)
    return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (!ctx->have_index) return Fail(ctx, SMHIP_ESTATE, "smhip_seed_lookup before an index is resident");
  if (n_queries == 0) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  if (p->walk_blocks_per_cu > 8) return Fail(ctx, SMHIP_EINVAL, "walk_blocks_per_cu must be 0..8");
  if (p->search_mode > SMHIP_SEARCH_SET) return Fail(ctx, SMHIP_EINVAL, "search_mode must be SMHIP_SEARCH_AUTO / _EXACT / _SET");
  DevBuf d_q, d_region, d_cnt, d_nodes, d_pts, d_idx, d_dist, d_qrec, d_tab;
  const uint64_t total = (uint64_t)n_queries * cap;
  auto free_all = [&]() {
    for (DevBuf *b : {&d_q, &d_region, &d_cnt, &d_nodes, &d_pts, &d_idx, &d_dist, &d_qrec, &d_tab}) b->Free();
  };
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  step(d_q.Reserve((uint64_t)n_queries * 6 * sizeof(float)));
  step(d_region.Reserve(total * sizeof(Hit)));
  step(d_tab.Reserve(sizeof(void *)));
  void *h_tab[1] = {d_region.p};  // one segment holds every query of this test entry point (qshift 31)
  if (e == hipSuccess) step(hipMemcpyAsync(d_tab.p, h_tab, sizeof h_tab, hipMemcpyHostToDevice, ctx->stream));
  step(d_cnt.Reserve((uint64_t)n_queries * 4));
  step(d_nodes.Reserve((uint64_t)n_queries * 4));
  step(d_pts.Reserve((uint64_t)n_queries * 4));
  step(d_idx.Reserve(total * 8));
  step(d_dist.Reserve(total * 4));
  step(d_qrec.Reserve((uint64_t)n_queries * 32));
  step(ctx->counters.Reserve(32 * sizeof(uint32_t)));
  if (e == hipSuccess) step(hipMemcpyAsync(d_q.p, queries, (uint64_t)n_queries * 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemsetAsync(ctx->counters.p, 0, 32 * sizeof(uint32_t), ctx->stream));
  if (e == hipSuccess) {
    step(LaunchSeedLookupRaw(ctx, *p, d_q.as<float>(), n_queries, cap, HitSlots{d_tab.as<Hit *>(), 31u, cap}, d_cnt.as<uint32_t>(),
                             d_nodes.as<uint32_t>(), d_pts.as<uint32_t>(), ctx->counters.as<uint32_t>(), d_qrec.p));
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(HitsToArraysKernel, dim3(1024), dim3(256), 0, ctx->stream, d_region.as<Hit>(),
                       d_cnt.as<uint32_t>(), n_queries, cap, ctx->index.leaf_vind, d_idx.as<uint64_t>(),
                       d_dist.as<float>());
    step(hipGetLastError());
  }
  std::vector<uint32_t> hn(n_queries), hp(n_queries);
  uint32_t hc[16] = {0};
  if (e == hipSuccess) step(hipMemcpyAsync(counts, d_cnt.p, (uint64_t)n_queries * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(point_idx, d_idx.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(dist, d_dist.p, total * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(hn.data(), d_nodes.p, (uint64_t)n_queries * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(hp.data(), d_pts.p, (uint64_t)n_queries * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(hc, ctx->counters.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipStreamSynchronize(ctx->stream));
  free_all();
  if (e != hipSuccess) return HipFail(ctx, e, "smhip_seed_lookup");
  ResolveTimers(ctx);
  for (uint32_t i = 0; i < n_queries; ++i) {
    if (nodes_visited) nodes_visited[i] = hn[i];
    if (points_tested) points_tested[i] = hp[i];
  }
  if (hc[0]) return Fail(ctx, SMHIP_ECAP, "tree walk stack overflow (tree deeper than 64 pending visits)");
  return SMHIP_OK;
}

int smhip_sdtw_banded(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query,
                      const uint64_t *q_off, const uint32_t *lo, const uint32_t *hi, uint32_t n_problems, float *dist,
                      int64_t *end) {
  if (!ctx || !target || !t_off || !query || !q_off || !lo || !hi || !dist || !end)
    return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (n_problems == 0) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t nt = t_off[n_problems], nqv = q_off[n_problems];
  uint32_t max_t = 0;
  for (uint32_t i = 0; i < n_problems; ++i) {
    if (t_off[i + 1] < t_off[i] || q_off[i + 1] < q_off[i]) return Fail(ctx, SMHIP_EINVAL, "offsets must be non-decreasing");
    max_t = std::max<uint32_t>(max_t, (uint32_t)std::min<uint64_t>(t_off[i + 1] - t_off[i], 0xFFFFFFFFull));
  }
  DevBuf d_t, d_q, d_to, d_qo, d_lo, d_hi, d_d, d_e;
  auto free_all = [&]() {
    for (DevBuf *b : {&d_t, &d_q, &d_to, &d_qo, &d_lo, &d_hi, &d_d, &d_e}) b->Free();
  };
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  step(d_t.Reserve(std::max<uint64_t>(nt, 1) * 4));
  step(d_q.Reserve(std::max<uint64_t>(nqv, 1) * 4));
  step(d_to.Reserve(((uint64_t)n_problems + 1) * 8));
  step(d_qo.Reserve(((uint64_t)n_problems + 1) * 8));
  step(d_lo.Reserve(std::max<uint64_t>(nt, 1) * 4));
  step(d_hi.Reserve(std::max<uint64_t>(nt, 1) * 4));
  step(d_d.Reserve((uint64_t)n_problems * 4));
  step(d_e.Reserve((uint64_t)n_problems * 8));
  if (e == hipSuccess) step(hipMemcpyAsync(d_t.p, target, nt * 4, hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(d_q.p, query, nqv * 4, hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(d_to.p, t_off, ((uint64_t)n_problems + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(d_qo.p, q_off, ((uint64_t)n_problems + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(d_lo.p, lo, nt * 4, hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(d_hi.p, hi, nt * 4, hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) {
    ScopedTimer t(ctx, T_SDTW);
    step(LaunchSdtw(ctx, d_t.as<float>(), d_to.as<uint64_t>(), d_q.as<float>(), d_qo.as<uint64_t>(),
                    d_lo.as<uint32_t>(), d_hi.as<uint32_t>(), n_problems, d_d.as<float>(), d_e.as<int64_t>(), max_t));
  }
  if (e == hipSuccess) step(hipMemcpyAsync(dist, d_d.p, (uint64_t)n_problems * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipMemcpyAsync(end, d_e.p, (uint64_t)n_problems * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (e == hipSuccess) step(hipStreamSynchronize(ctx->stream));
  free_all();
  if (e != hipSuccess) return HipFail(ctx, e, "smhip_sdtw_banded");
  ResolveTimers(ctx);
  return SMHIP_OK;
}

int smhip_dtw_path(smhip_ctx *ctx, const float *target, const uint64_t *t_off, const float *query, const uint64_t *q_off,
                   const uint32_t *lo, const uint32_t *hi, uint32_t n_problems, float *dist, int64_t *end,
                   uint8_t *path_codes, const uint64_t *path_off, uint32_t *path_len, int64_t *start_target) {
  if (!ctx || !target || !t_off || !query || !q_off || !lo || !hi || !dist || !end || !path_codes || !path_off ||
      !path_len || !start_target)
    return Fail(ctx, SMHIP_EINVAL, "null argument");
  if (n_problems == 0) return SMHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t nt = t_off[n_problems], nqv = q_off[n_problems], np_total = path_off[n_problems];
  uint32_t max_t = 0;
  std::vector<uint64_t> cell_off(std::max<uint64_t>(nt, 1) + 1);
  uint64_t cells = 0;
  for (uint32_t i = 0; i < n_problems; ++i) {
    if (t_off[i + 1] < t_off[i] || q_off[i + 1] < q_off[i] || path_off[i + 1] < path_off[i])
      return Fail(ctx, SMHIP_EINVAL, "offsets must be non-decreasing");
    const uint64_t T = t_off[i + 1] - t_off[i], Q = q_off[i + 1] - q_off[i];
    if (path_off[i + 1] - path_off[i] < T + Q) return Fail(ctx, SMHIP_EINVAL, "path capacity must be >= T + Q per problem");
    max_t = std::max<uint32_t>(max_t, (uint32_t)std::min<uint64_t>(T, 0xFFFFFFFFull));
    for (uint64_t r = t_off[i]; r < t_off[i + 1]; ++r) {
      const uint64_t a = std::max<uint32_t>(lo[r], 1u), b = std::min<uint64_t>(hi[r], Q);
      cell_off[r] = cells;
      if (b >= a) cells += b - a + 1;
    }
  }
  DevBuf d_t, d_q, d_to, d_qo, d_lo, d_hi, d_co, d_codes, d_d, d_e, d_path, d_po, d_pl, d_st;
  auto free_all = [&]() {
    for (DevBuf *b : {&d_t, &d_q, &d_to, &d_qo, &d_lo, &d_hi, &d_co, &d_codes, &d_d, &d_e, &d_path, &d_po, &d_pl, &d_st}) b->Free();
  };
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  const uint64_t n1 = (uint64_t)n_problems + 1;
  step(d_t.Reserve(std::max<uint64_t>(nt, 1) * 4));
  step(d_q.Reserve(std::max<uint64_t>(nqv, 1) * 4));
  step(d_to.Reserve(n1 * 8));
  step(d_qo.Reserve(n1 * 8));
  step(d_lo.Reserve(std::max<uint64_t>(nt, 1) * 4));
  step(d_hi.Reserve(std::max<uint64_t>(nt, 1) * 4));
  step(d_co.Reserve(std::max<uint64_t>(nt, 1) * 8));
  step(d_codes.Reserve(std::max<uint64_t>(cells, 1)));
  step(d_d.Reserve((uint64_t)n_problems * 4));
  step(d_e.Reserve((uint64_t)n_problems * 8));
  step(d_path.Reserve(std::max<uint64_t>(np_total, 1)));
  step(d_po.Reserve(n1 * 8));
  step(d_pl.Reserve((uint64_t)n_problems * 4));
  step(d_st.Reserve((uint64_t)n_problems * 8));
  auto h2d = [&](DevBuf &b, const void *src, uint64_t bytes) {
    if (e == hipSuccess && bytes) step(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  };
  h2d(d_t, target, nt * 4);
  h2d(d_q, query, nqv * 4);
  h2d(d_to, t_off, n1 * 8);
  h2d(d_qo, q_off, n1 * 8);
  h2d(d_lo, lo, nt * 4);
  h2d(d_hi, hi, nt * 4);
  h2d(d_co, cell_off.data(), nt * 8);
  h2d(d_po, path_off, n1 * 8);
  if (e == hipSuccess) {
    ScopedTimer t(ctx, T_SDTW);
    step(LaunchDtwPath(ctx, d_t.as<float>(), d_to.as<uint64_t>(), d_q.as<float>(), d_qo.as<uint64_t>(), d_lo.as<uint32_t>(),
                       d_hi.as<uint32_t>(), d_co.as<uint64_t>(), d_codes.as<uint8_t>(), n_problems, d_d.as<float>(),
                       d_e.as<int64_t>(), d_path.as<uint8_t>(), d_po.as<uint64_t>(), d_pl.as<uint32_t>(),
                       d_st.as<int64_t>(), max_t));
  }
  auto d2h = [&](void *dst, DevBuf &b, uint64_t bytes) {
    if (e == hipSuccess && bytes) step(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  };
  d2h(dist, d_d, (uint64_t)n_problems * 4);
  d2h(end, d_e, (uint64_t)n_problems * 8);
  d2h(path_codes, d_path, np_total);
  d2h(path_len, d_pl, (uint64_t)n_problems * 4);
  d2h(start_target, d_st, (uint64_t)n_problems * 8);
  if (e == hipSuccess) step(hipStreamSynchronize(ctx->stream));
  free_all();
  if (e != hipSuccess) return HipFail(ctx, e, "smhip_dtw_path");
  ResolveTimers(ctx);
  for (uint32_t i = 0; i < n_problems; ++i)  // the device walked end -> start: std::reverse (fast_dtw.cc:71)
    std::reverse(path_codes + path_off[i], path_codes + path_off[i] + path_len[i]);
  return SMHIP_OK;
}
}  // extern "C"
