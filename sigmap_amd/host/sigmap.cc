// sigmap.cc -- the `sigmap` command line (same flags and defaults as the reference driver,
// sigmap.cc:1329-1526) over the MI355X hot path.
//
//   sigmap -i -r ref.fa -p model.txt -o PREFIX [-d 6] [-l 20]
//   sigmap -m -r ref.fa -p model.txt -x PREFIX -s READS_DIR -o out.paf [-t N] [development flags]
//
// -i is Sigmap::ConstructIndex (sigmap.cc:999-1046).  -m is Sigmap::StreamingMap
// (sigmap.cc:549-880) with the per-read taskloop turned into chunk rounds: round c sends chunk c
// of every still-active read to the GPU in one batch (include/sigmap_hip.h), then the unchanged
// early-stop / output rules run on the host (mapping_logic.cc).  Reads come from FAST5 files (single-
// and multi-read, through the HDF5 C API, fast5_reader.cc), BLOW5 files (blow5_reader.cc: the binary
// SLOW5 layout parsed directly, slow5lib is not vendored) or this build's .sgr container (signal_batch.h).
#include <omp.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sigmap_hip.h"
#include "mapping_logic.h"
#include "output_tools.h"
#include "pore_model.h"
#include "sequence_batch.h"
#include "sigmap.h"
#include "signal_batch.h"
#include "spatial_index.h"
#include "utils.h"

namespace sigmap {
void BuildIndex(SequenceBatch &ref, const PoreModel &pore_model, SpatialIndex &index);  // host_abi.cc

void Sigmap::ConstructIndex() {
  PoreModel pore_model;
  pore_model.Load(pore_model_file_path_);
  SequenceBatch reference;
  reference.InitializeLoading(reference_file_path_);
  reference.LoadAllSequences();
  SpatialIndex index(dimension_, max_leaf_, omp_get_max_threads(), output_file_path_);
  BuildIndex(reference, pore_model, index);
  index.Save();
}

// Pinned arena of the raw reads (SignalBatch::KeepRaw): a run of reads is one DMA.
void *Sigmap::PinnedAlloc(size_t bytes) {
  void *p = nullptr;
  return smhip_host_alloc(&p, bytes) == SMHIP_OK ? p : nullptr;
}
void Sigmap::PinnedFree(void *p) { smhip_host_free(p); }

// share_device (development option --gpus-share-device): the N "GPUs" are all GPU `device` and borrow
// one resident index, so that the read sharding, the per-GPU lanes and the merge of `--gpus N` can be
// exercised on a one-GPU machine (RCCL itself refuses two ranks on one device).
void Sigmap::StreamingMap(int device, int num_gpus, size_t max_batch_chunks, int sdtw_band, int num_streams,
                  bool sdtw_cigar, bool share_device, bool warm_up) {
  num_gpus = std::max(1, num_gpus);
  // (num_streams < 0: up to that many contexts per GPU, as many as the reads need -- the command line's default)
  const bool auto_lanes = num_streams < 0;
  num_streams = std::max(1, auto_lanes ? -num_streams : num_streams);
  if (device < 0 || device + (share_device ? 1 : num_gpus) > smhip_device_count())
    ExitWithMessage("No usable MI355X GPU for --device " + std::to_string(device) + " --gpus " +
                    std::to_string(num_gpus) + " (" + std::to_string(smhip_device_count()) +
                    " visible): this build has no CPU mapping path.");
  double t_phase = GetRealTime();
  auto phase = [&t_phase](const char *what) {  // set-up timeline on stderr, like the reference's "Loaded ... in"
    const double now = GetRealTime();
    std::cerr << what << " in " << now - t_phase << "s.\n";
    t_phase = now;
  };
  const uint32_t chunk_size = 4000;
  const uint32_t max_chunks = (uint32_t)std::max(0, params_.max_num_chunks);
  SignalBatch reads;
  // ADC counts stay 2 bytes per sample in pinned memory; the GPU pulls, converts and filters the ONE chunk of every
  // read a round maps (smhip_reads_upload_pieces).  What the reference's load phase computes per read -- the filtered
  // length (PAF column 2, num_chunks) -- and where the chunks of the filtered array begin in the raw one are counted
  // while loading (SignalBatch::AddSignal's arithmetic), so only samples that are mapped ever cross PCIe.
  reads.KeepRaw(&Sigmap::PinnedAlloc, &Sigmap::PinnedFree, (size_t)256 << 20, chunk_size, max_chunks);
  reads.InitializeLoading(signal_directory_);
  const size_t num_reads = reads.LoadAllReadSignals();
  phase("Loaded read signals (first device use included)");
  SequenceBatch reference;  // names and lengths for PAF; the pore model is only needed by -i
  reference.InitializeLoading(reference_file_path_);
  const uint32_t num_refs = reference.LoadAllSequences();
  std::vector<uint32_t> ref_lengths(num_refs);
  for (uint32_t i = 0; i < num_refs; ++i) ref_lengths[i] = reference.GetSequenceLengthAt(i);
  SpatialIndex index(1000, std::vector<int>(1000, 5000), reference_index_file_path_);
  index.Load();
  phase("Loaded reference and index");

  // ctxs[g][k]: context k (one HIP stream each) of GPU device + g.  The first context of GPU 0
  // uploads the index; the first contexts of the other GPUs receive it by one RCCL broadcast per
  // array (smhip_index_bcast); the further contexts of a GPU borrow its arrays.
  // Reads are independent (one taskloop over all reads, sigmap.cc:618-666): every GPU maps a
  // contiguous block of the reads (read i -> GPU i * G / n), every context of a GPU takes batches
  // from that block, driven by its own host thread.
  std::vector<std::vector<smhip_ctx *>> ctxs((size_t)num_gpus, std::vector<smhip_ctx *>((size_t)num_streams, nullptr));
  for (int g = 0; g < num_gpus; ++g)
    for (auto &c : ctxs[g])
      if (smhip_init(share_device ? device : device + g, &c) != SMHIP_OK)
        ExitWithMessage("smhip_init failed for GPU " + std::to_string(device + g) +
                        ": this build has no CPU mapping path.");
  if (smhip_index_upload(ctxs[0][0], index.dimension(), index.values().data(), index.positions().data(),
                         index.values().size(), index.tree().vind.data(), index.tree().m_size,
                         index.tree().nodes.data(), index.tree().nodes.size(), index.tree().bbox.data()) != SMHIP_OK)
    ExitWithMessage(std::string("Index upload failed: ") + smhip_last_error(ctxs[0][0]));
  if (num_gpus > 1 && share_device) {
    smhip_index_dev shared;
    if (smhip_index_export(ctxs[0][0], &shared) != SMHIP_OK)
      ExitWithMessage(std::string("Index export failed: ") + smhip_last_error(ctxs[0][0]));
    for (int g = 1; g < num_gpus; ++g)
      if (smhip_index_attach(ctxs[g][0], &shared) != SMHIP_OK)
        ExitWithMessage(std::string("Index attach failed: ") + smhip_last_error(ctxs[g][0]));
  } else if (num_gpus > 1) {
    const double t0 = GetRealTime();
    std::vector<smhip_ctx *> firsts;
    for (int g = 0; g < num_gpus; ++g) firsts.push_back(ctxs[g][0]);
    if (smhip_index_bcast(firsts.data(), num_gpus, 0) != SMHIP_OK)
      ExitWithMessage(std::string("Index broadcast failed: ") + smhip_last_error(ctxs[0][0]));
    std::cerr << "Broadcast index to " << num_gpus << " GPUs in " << GetRealTime() - t0 << "s.\n";
  }
  for (int g = 0; g < num_gpus; ++g) {
    smhip_index_dev shared;
    if (smhip_index_export(ctxs[g][0], &shared) != SMHIP_OK)
      ExitWithMessage(std::string("Index export failed: ") + smhip_last_error(ctxs[g][0]));
    for (size_t k = 1; k < ctxs[g].size(); ++k)
      if (smhip_index_attach(ctxs[g][k], &shared) != SMHIP_OK)
        ExitWithMessage(std::string("Index attach failed: ") + smhip_last_error(ctxs[g][k]));
  }
  phase("Index resident on the device(s)");
  if (sdtw_band > 0) {  // z-scored expected signal of every contig strand (sigmap.cc:566-591)
    PoreModel pore_model;
    pore_model.Load(pore_model_file_path_);
    SignalBatch ref_signals;
    for (uint32_t i = 0; i < num_refs; ++i) reference.PrepareNegativeSequenceAt(i);
    ref_signals.ConvertSequencesToSignals(reference, pore_model, num_refs);
    std::vector<float> sig;
    std::vector<uint64_t> off(1, 0);
    for (uint32_t i = 0; i < num_refs; ++i) {
      const Signal &s = ref_signals.GetSignalAt(i);
      ZscoreNormalize(s.signal_values.data(), s.signal_values.size(), sig);
      off.push_back(sig.size());
      ZscoreNormalize(s.negative_signal_values.data(), s.negative_signal_values.size(), sig);
      off.push_back(sig.size());
    }
    // one copy per GPU: its first context uploads, the others borrow (like the index)
    for (int g = 0; g < num_gpus; ++g) {
      smhip_ctx *owner = share_device ? ctxs[0][0] : ctxs[g][0];
      for (smhip_ctx *c : ctxs[g]) {
        const int rc = c == owner ? smhip_reference_upload(c, sig.data(), off.data(), num_refs)
                                  : smhip_reference_share(c, owner);
        if (rc != SMHIP_OK) ExitWithMessage(std::string("Reference signal upload failed: ") + smhip_last_error(c));
      }
    }
  }
  smhip_params dp;
  smhip_default_params(&dp);
  dp.search_radius = params_.search_radius;
  dp.step_size = params_.read_seeding_step_size;

  struct ReadState {
    uint32_t chunk_index = 0;  // next chunk to process
    uint32_t num_events = 0;
    smhip_chunk_result last{};       // chains after the last GenerateChains call (zero = none yet)
    float dt = 0;                    // sDTW distance of those chains' best (when --sdtw-band)
    bool has_dt = false;
    std::string cigar;               // per-base CIGAR of DTW()'s banded path (when --sdtw-cigar)
    double start_time = 0;
  };
  std::vector<ReadState> st(num_reads);
  std::vector<uint32_t> num_chunks(num_reads, 0);
  for (size_t r = 0; r < num_reads; ++r) {
    memset(&st[r].last, 0, sizeof(smhip_chunk_result));
    num_chunks[r] = (uint32_t)(reads.GetSignalLengthAt(r) / chunk_size);  // sigmap.cc:643
  }
  // every context gets an equal share of the batch budget, so that `max_batch_chunks` bounds the
  // chunks in flight on a GPU as it does with one context; a slot holds the one chunk of a read that a round maps
  const uint32_t keep = chunk_size;
  size_t lane_budget = std::max<size_t>(1, max_batch_chunks / (size_t)num_streams);
  lane_budget = std::max<size_t>(1, std::min<size_t>(lane_budget, ((size_t)8 << 30) / ((size_t)keep * sizeof(float))));
  {
    // ... and within the memory the GPU has left: per chunk in flight a context needs its read slot, ~288 queries x
    // (8 KB of hit slots + ~1.4 KB of per-query records and overflow pool) and ~60 bytes per anchor (estimated
    // from the index size); half of the free memory is left to the anchor-sized buffers of large references and
    // to growth.  (A default budget that an MI355X takes in its stride fails on a smaller device otherwise.)
    uint64_t free_b = 0, total_b = 0;
    if (smhip_mem_info(ctxs[0][0], &free_b, &total_b) == SMHIP_OK && free_b > 0) {
      // anchors per query = mean over the census of the index (its own points as queries) of min(hits, cap);
      // fast hit slots as the device will choose them from the same census (sigmap_hip.h)
      std::vector<uint32_t> census(4096);
      uint32_t n_census = 0;
      double hits_per_query = 1.7e-6 * (double)index.tree().m_size;  // (fallback: uniform random genome, 41 at 12 Mb, 850 at 250 Mb)
      uint32_t fast_slots = 512;
      if (smhip_index_census(ctxs[0][0], &dp, census.data(), (uint32_t)census.size(), &n_census) == SMHIP_OK && n_census > 0) {
        double sum = 0;
        for (uint32_t i = 0; i < n_census; ++i) sum += std::min<uint32_t>(census[i], (uint32_t)dp.num_nearest_points);
        hits_per_query = sum / n_census;
        const uint32_t q75 = census[((size_t)n_census * 3) / 4];
        while (fast_slots < 2048 && fast_slots < q75) fast_slots <<= 1;
      }
      const size_t contexts_here = (size_t)num_streams * (share_device ? (size_t)num_gpus : 1u);
      // (development hook: SIGMAP_DEV_FREE_MB pretends the device has that much left -- the shrink path on a small index)
      if (const char *fake = getenv("SIGMAP_DEV_FREE_MB")) free_b = std::min<uint64_t>(free_b, (uint64_t)atoll(fake) << 20);
      const size_t fit = ChunkBudgetForMemory(free_b, keep, fast_slots, hits_per_query, contexts_here);
      if (fit < lane_budget) {
        std::cerr << "Chunks in flight per context limited to " << std::max<size_t>(fit, 1) << " of " << lane_budget
                  << " by the device's free memory (" << (free_b >> 20) << " MiB).\n";
        lane_budget = std::max<size_t>(fit, 1);
      }
    }
  }
  // ... and no more than a context's share of its GPU's block of reads: reads are admitted in input order by whichever
  // context has free slots, and a context with room for the whole block would take all of it and leave the GPU's other
  // contexts -- whose kernels fill the gaps between its chunk rounds -- without work
  // Contexts in use per GPU: all of them when the GPU's reads outnumber one context's slots -- the other context's
  // kernels then fill the gaps between a context's chunk rounds, and its uploads overlap them --, ONE when they fit a
  // single context (default only; an explicit --streams is taken at its word): the rounds of multi-chunk reads are a
  // chain whose length the longest read sets, two half-sized batches cost more than one whole (a batch of 5 000
  // chunks is 20 % dearer per chunk than one of 10 000), and two chains side by side lengthen each other.  Measured,
  // 10 000 reads of 8 / 30 chunks: 35.8 / 73.7 ms on one context, 39.7 - 45.8 / 94.1 on two; 50 000 reads of 8
  // (more than a context's slots either way): 121.3 / 119.7 ms.
  size_t lanes_used = (size_t)num_streams;
  {
    const size_t block_reads = (num_reads + (size_t)num_gpus - 1) / (size_t)num_gpus;
    if (auto_lanes) lanes_used = ContextsForReads(block_reads, lane_budget, (size_t)num_streams);
    lane_budget = std::max<size_t>(1, std::min(lane_budget, (block_reads + lanes_used - 1) / lanes_used));
  }
  // (the other contexts stay idle: created and holding the index, never given read slots or work)
  // read slots of every context: set-up like the index, before the mapping span
  for (auto &gpu : ctxs)
    for (smhip_ctx *c : std::vector<smhip_ctx *>(gpu.begin(), gpu.begin() + (long)lanes_used))
      if (smhip_reads_configure(c, (uint32_t)lane_budget, keep) != SMHIP_OK ||
          smhip_batch_reserve_for(c, &dp, (uint32_t)lane_budget, 288) != SMHIP_OK)  // ~225 seeds per 4000-sample chunk
        ExitWithMessage(std::string("Read slots / batch buffers for ") + std::to_string(lane_budget) +
                        " chunks in flight per context (--batch-chunks / --streams): " + smhip_last_error(c));
  phase("Reserved read slots and batch buffers");
  const double t_warm = GetRealTime();
  if (warm_up) {
    // Warm-up, part of the set-up like the index load: every context maps one round of first chunks -- as many as it
    // will have in flight, taken from its GPU's first reads (again and again when those are fewer) -- and throws the
    // results away.  What a context's FIRST batch otherwise pays inside the mapping span, measured at 40 ms of a 60 ms
    // job of 10 000 reads (profiles/EXPERIMENTS.md, round 5): the scratch memory the runtime attaches to a hardware
    // queue at the first kernel that needs it, buffers that grow (a hipFree waits for the whole device), the route with
    // two host read-backs that sizes the later, speculative batches.  Results never depend on it.
    // The reference's own span (sigmap.cc:617, 869 "Finished mapping in") starts right after loading and has no such
    // step: the seconds spent here are printed next to the span and in STREAM_STATS (warmup_s), and --no-warmup maps
    // without it -- the span then holds the first-batch costs, the comparison the reference's figure is.
    std::vector<std::thread> warm;
    std::vector<std::string> failed(ctxs.size() * (size_t)num_streams);
    for (int g = 0; g < num_gpus; ++g)
      for (size_t k = 0; k < lanes_used; ++k)
        warm.emplace_back([&, g, k] {
          smhip_ctx *ctx = ctxs[g][k];
          size_t lo = 0, hi = 0;
          ShardRange(num_reads, g, num_gpus, &lo, &hi);
          std::vector<const int16_t *> pieces;
          std::vector<uint32_t> plen, expect, slots, zeros;
          std::vector<float> dig, rng, ofs;
          for (size_t r = lo; r < hi && pieces.size() < lane_budget; ++r) {
            const Signal &sig = reads.GetSignalAt(r);
            if (sig.chunk_bounds.size() < 2) continue;  // (shorter than a chunk)
            pieces.push_back(sig.raw);
            plen.push_back(sig.chunk_bounds[1]);
            dig.push_back(sig.digitisation); rng.push_back(sig.range); ofs.push_back(sig.offset);
          }
          if (pieces.empty() || max_chunks == 0) return;
          const size_t have = pieces.size();
          for (size_t i = have; i < lane_budget; ++i) {
            pieces.push_back(pieces[i % have]); plen.push_back(plen[i % have]);
            dig.push_back(dig[i % have]); rng.push_back(rng[i % have]); ofs.push_back(ofs[i % have]);
          }
          const uint32_t n = (uint32_t)pieces.size();
          expect.assign(n, chunk_size); zeros.assign(n, 0u); slots.resize(n);
          for (uint32_t i = 0; i < n; ++i) slots[i] = i;
          std::vector<smhip_chunk_result> res(n);
          int rc = smhip_reads_upload_pieces(ctx, pieces.data(), plen.data(), dig.data(), rng.data(), ofs.data(), nullptr,
                                             expect.data(), n);
          if (rc == SMHIP_OK) rc = smhip_batch_select(ctx, slots.data(), zeros.data(), chunk_size, n, zeros.data(), nullptr, nullptr);
          if (rc == SMHIP_OK) rc = smhip_map_run(ctx, &dp, num_refs);
          if (rc == SMHIP_OK && sdtw_band > 0) rc = smhip_refine_run(ctx, (uint32_t)sdtw_band, 32);
          if (rc == SMHIP_OK && sdtw_band > 0 && sdtw_cigar) rc = smhip_refine_path_run(ctx, (uint32_t)sdtw_band, 32);
          if (rc == SMHIP_OK) rc = smhip_map_sync(ctx);
          if (rc == SMHIP_OK) rc = smhip_map_download(ctx, res.data(), nullptr, 0, nullptr);
          if (rc == SMHIP_OK) {  // ... and the anchors of every chunk, as a later round may ask for them
            std::vector<uint64_t> off(n + 1);
            std::vector<smhip_carried_anchor> anch(1);
            rc = smhip_map_download_anchors_of(ctx, slots.data(), n, anch.data(), 0, off.data());
            if (rc == SMHIP_ECAP) {
              anch.resize(off[n]);
              rc = smhip_map_download_anchors_of(ctx, slots.data(), n, anch.data(), anch.size(), off.data());
            }
          }
          if (rc != SMHIP_OK) failed[(size_t)g * (size_t)num_streams + k] = smhip_last_error(ctx);
        });
    for (auto &w : warm) w.join();
    for (const auto &f : failed)
      if (!f.empty()) ExitWithMessage("GPU warm-up batch failed: " + f);
  }
  // A lane's large host buffers -- result records, the anchors downloaded after a round, the carry-over of the next --
  // sized and TOUCHED here: a vector that grows inside the span pays for its fresh pages there (measured: 0.5 ms in the
  // first round's anchor download, 0.6 ms where the carry-over first reached 150 000 anchors).
  struct LaneScratch {
    std::vector<smhip_carried_anchor> carried, next_carried, out_anchors;
    std::vector<smhip_chunk_result> results;
  };
  std::vector<LaneScratch> lane_scratch((size_t)num_gpus * (size_t)num_streams);
  for (int g = 0; g < num_gpus; ++g)
    for (size_t k = 0; k < lanes_used; ++k) {
      LaneScratch &sc = lane_scratch[(size_t)g * (size_t)num_streams + k];
      const size_t anchors_room = std::min<size_t>((size_t)1 << 20, std::max<size_t>(lane_budget * 64, 4096));
      sc.out_anchors.resize((size_t)1 << 20);  // (the download's first capacity)
      sc.carried.resize(anchors_room); sc.carried.clear();
      sc.next_carried.resize(anchors_room); sc.next_carried.clear();
      sc.results.resize(lane_budget); sc.results.clear();
    }
  const double warm_span = warm_up ? GetRealTime() - t_warm : 0.0;
  phase(warm_up ? "Warmed up the device contexts (one discarded round of first chunks each)"
                : "Host buffers of the lanes touched (no warm-up round: --no-warmup)");
  const double t_map = GetRealTime();
  // GPU g maps reads [block[g], block[g + 1]) (contiguous blocks); reads are admitted
  // in order and stay with the context that admitted them (chain carry-over is per read)
  std::vector<size_t> block((size_t)num_gpus + 1), cursor((size_t)num_gpus);
  for (int g = 0; g <= num_gpus; ++g) ShardRange(num_reads, g, num_gpus, g < num_gpus ? &block[g] : nullptr, nullptr);
  block[num_gpus] = num_reads;
  for (int g = 0; g < num_gpus; ++g) cursor[g] = block[g];
  std::vector<std::mutex> cursor_mutex((size_t)num_gpus);
  // Where a lane's wall time goes, chunk round by chunk round (SIGMAP_STREAM_STATS=1: one `STREAM_STATS {json}` line on
  // stderr after the mapping span; tools/stream_bench.py).  "device" = from the first call that enqueues a round's work
  // to the return of smhip_map_sync -- the host thread only waits there; everything else is host time between rounds.
  struct LaneStats {
    double admit = 0, build = 0, device = 0, download = 0, anchors = 0, decide = 0;
    uint64_t rounds = 0, chunks = 0, carried_up = 0, anchors_down = 0, continued = 0, max_batch = 0, spec_runs = 0, spec_reruns = 0;
  };
  const bool want_stats = getenv("SIGMAP_STREAM_STATS") != nullptr;
  const bool want_rounds = want_stats && atoi(getenv("SIGMAP_STREAM_STATS")) >= 2;  // ... and one line per chunk round
  std::vector<std::string> round_lines;
  LaneStats total_stats;
  std::mutex stats_mutex;
  auto lane = [&](int g, smhip_ctx *ctx) {
    LaneStats ls, prev;
    double t_mark = GetRealTime();
    auto lap = [&](double &acc) {
      const double now = GetRealTime();
      acc += now - t_mark;
      t_mark = now;
    };
    std::vector<size_t> batch, in_flight;
    std::vector<uint32_t> slots, chunk_idx, qstart, carried_off, plen, expect, want, want_at;
    std::vector<const int16_t *> pieces;
    std::vector<float> dig, rng, ofs;
    // (the large ones come from the set-up: grown inside the span, their first pages cost a millisecond per round)
    LaneScratch &scratch = lane_scratch[(size_t)g * (size_t)num_streams +
                                        (size_t)(std::find(ctxs[g].begin(), ctxs[g].end(), ctx) - ctxs[g].begin())];
    std::vector<smhip_carried_anchor> &carried = scratch.carried, &out_anchors = scratch.out_anchors;
    std::vector<smhip_chunk_result> &results = scratch.results;
    std::vector<uint64_t> out_off;
    std::vector<float> dts, pdist;
    std::vector<int64_t> dte, pend, pstart;
    std::vector<uint32_t> plen_path;
    std::vector<uint64_t> poff;
    std::vector<uint8_t> pcodes;
    // the carry-over of the reads in flight (the anchors of their chains, spatial_index.cc:303-322), in in_flight order:
    // next_off[i] .. next_off[i + 1] of next_carried belongs to in_flight[i].  Built while a round's results are looked
    // at, handed to the next round as it is: an anchor is copied once on its way from one download to the next upload.
    std::vector<smhip_carried_anchor> &next_carried = scratch.next_carried;
    std::vector<uint32_t> next_off(1, 0u);
    auto finish = [&](size_t r) {
      ReadState &s = st[r];
      s.start_time = (GetRealTime() - s.start_time) * 1000;
    };
    while (true) {
      size_t a = 0, b = 0;
      {  // admit new reads up to the chunks this context may have in flight
        std::lock_guard<std::mutex> lock(cursor_mutex[g]);
        a = cursor[g];
        b = std::min(block[g + 1], a + (lane_budget - std::min(lane_budget, in_flight.size())));
        cursor[g] = b;
      }
      const double now = GetRealTime();
      for (size_t r = a; r < b; ++r) {
        st[r].start_time = now;
        if (num_chunks[r] == 0 || max_chunks == 0) {
          finish(r);
        } else {
          in_flight.push_back(r);
          next_off.push_back(next_off.back());  // (a new read carries nothing)
        }
      }
      lap(ls.admit);
      if (in_flight.empty()) {
        if (a == b) break;  // nothing in flight, nothing left to admit
        continue;
      }
      batch = in_flight;
      const size_t nb = batch.size();
      pieces.resize(nb); plen.resize(nb); expect.assign(nb, chunk_size); dig.resize(nb); rng.resize(nb); ofs.resize(nb);
      slots.resize(nb); chunk_idx.assign(nb, 0u); qstart.resize(nb);
      carried.swap(next_carried);
      carried_off.swap(next_off);  // (nb + 1 entries)
      next_carried.clear();
      next_off.assign(1, 0u);
      for (size_t bi = 0; bi < nb; ++bi) {
        const size_t r = batch[bi];
        const Signal &sig = reads.GetSignalAt(r);
        const uint32_t c = st[r].chunk_index;  // (< min(num_chunks, max_chunks): chunk_bounds has c + 2 entries)
        pieces[bi] = sig.raw + sig.chunk_bounds[c];
        plen[bi] = sig.chunk_bounds[c + 1] - sig.chunk_bounds[c];
        dig[bi] = sig.digitisation; rng[bi] = sig.range; ofs[bi] = sig.offset;
        slots[bi] = (uint32_t)bi;  // a slot lives for one round: chunk bi of the batch
        qstart[bi] = st[r].num_events;
      }
      results.resize(nb);
      out_off.resize(nb + 1);
      ls.rounds += 1;
      ls.chunks += nb;
      ls.carried_up += carried.size();
      ls.max_batch = std::max<uint64_t>(ls.max_batch, nb);
      lap(ls.build);
      // the round's chunks: pulled from the pinned arena by the device, converted, filtered (exactly chunk_size survivors
      // each, by the load phase's count), one slot per chunk
      int rc = smhip_reads_upload_pieces(ctx, pieces.data(), plen.data(), dig.data(), rng.data(), ofs.data(), nullptr,
                                         expect.data(), (uint32_t)nb);
      if (rc == SMHIP_OK)
        rc = smhip_batch_select(ctx, slots.data(), chunk_idx.data(), chunk_size, (uint32_t)nb, qstart.data(),
                                carried.empty() ? nullptr : carried.data(), carried.empty() ? nullptr : carried_off.data());
      if (rc == SMHIP_OK) rc = smhip_map_run(ctx, &dp, num_refs);
      if (rc == SMHIP_OK && sdtw_band > 0) rc = smhip_refine_run(ctx, (uint32_t)sdtw_band, 32);
      if (rc == SMHIP_OK && sdtw_band > 0 && sdtw_cigar) rc = smhip_refine_path_run(ctx, (uint32_t)sdtw_band, 32);
      if (rc == SMHIP_OK) rc = smhip_map_sync(ctx);
      lap(ls.device);
      dts.resize(batch.size());
      dte.resize(batch.size());
      if (rc == SMHIP_OK && sdtw_band > 0) rc = smhip_refine_download(ctx, dts.data(), dte.data());
      if (rc == SMHIP_OK && sdtw_band > 0 && sdtw_cigar) {
        pdist.resize(batch.size()); pend.resize(batch.size()); pstart.resize(batch.size());
        plen_path.resize(batch.size()); poff.resize(batch.size() + 1);
        rc = smhip_refine_path_download(ctx, pdist.data(), pend.data(), pstart.data(), plen_path.data(), poff.data(), nullptr, 0);
        if (rc == SMHIP_OK) {
          pcodes.resize(std::max<uint64_t>(poff[batch.size()], 1));
          rc = smhip_refine_path_download(ctx, pdist.data(), pend.data(), pstart.data(), plen_path.data(), poff.data(),
                                          pcodes.data(), pcodes.size());
        }
      }
      // the result records first; the anchors of the chains (the next chunk's carry-over,
      // spatial_index.cc:303-322) only when some read of the batch goes on to another chunk
      if (rc == SMHIP_OK) rc = smhip_map_download(ctx, results.data(), nullptr, 0, nullptr);
      if (rc != SMHIP_OK) ExitWithMessage(std::string("GPU mapping failed: ") + smhip_last_error(ctx));
      lap(ls.download);
      // the chains' anchors -- the next chunk's carry-over, spatial_index.cc:303-322 -- of the reads that go on to
      // another chunk with NEW chains (a chunk with <= 50 features keeps the chains it came with, sigmap.cc:660), and of
      // those only: two reads in three stop here
      want.clear();
      want_at.assign(batch.size(), 0xFFFFFFFFu);
      for (size_t bi = 0; bi < batch.size(); ++bi) {
        const size_t r = batch[bi];
        const smhip_chunk_result &res = results[bi];
        if (res.n_features > 50 && !ShouldStopMapping(res, params_) && st[r].chunk_index + 1 < num_chunks[r] &&
            st[r].chunk_index + 1 < max_chunks) {
          want_at[bi] = (uint32_t)want.size();
          want.push_back((uint32_t)bi);
        }
      }
      if (!want.empty()) {
        uint64_t cap = std::max<uint64_t>(out_anchors.size(), 1 << 20);
        out_off.resize(want.size() + 1);
        while (rc == SMHIP_OK) {
          out_anchors.resize(cap);
          rc = smhip_map_download_anchors_of(ctx, want.data(), (uint32_t)want.size(), out_anchors.data(), cap, out_off.data());
          if (rc != SMHIP_ECAP) break;
          cap = std::max<uint64_t>(cap * 2, out_off[want.size()]);
          rc = SMHIP_OK;
        }
        if (rc != SMHIP_OK) ExitWithMessage(std::string("GPU mapping failed: ") + smhip_last_error(ctx));
        ls.anchors_down += out_off[want.size()];
      }
      lap(ls.anchors);
      in_flight.clear();
      for (size_t bi = 0; bi < batch.size(); ++bi) {
        const size_t r = batch[bi];
        ReadState &s = st[r];
        const smhip_chunk_result &res = results[bi];
        // a device-side capacity was exceeded: the records of this chunk may differ from the
        // reference's, so nothing is written from them (the flags are not expected on any input)
        if (res.status & ~SMHIP_ST_FEW_PEAKS)
          ExitWithMessage(std::string("read ") + reads.GetSignalNameAt(r) + ": device capacity flag " +
                          std::to_string(res.status) + " (include/sigmap_hip.h, SMHIP_ST_*)");
        bool stop = false;
        if (res.n_features > 50) {  // sigmap.cc:660: GenerateChains only then; chains persist otherwise
          s.last = res;
          s.has_dt = sdtw_band > 0 && dte[bi] >= 0;
          s.dt = dts[bi];
          s.cigar.clear();
          if (sdtw_band > 0 && sdtw_cigar && pend[bi] >= 0) s.cigar = PrintAlignment(pcodes.data() + poff[bi], plen_path[bi]);
          s.num_events += res.n_features;
          stop = ShouldStopMapping(res, params_);
        }
        if (!stop) {
          ++s.chunk_index;
          if (s.chunk_index < num_chunks[r] && s.chunk_index < max_chunks) {
            in_flight.push_back(r);
            // its carry-over: the new chains' anchors, or -- a chunk with <= 50 features -- the ones it came with
            if (want_at[bi] != 0xFFFFFFFFu)
              next_carried.insert(next_carried.end(), out_anchors.begin() + out_off[want_at[bi]],
                                  out_anchors.begin() + out_off[want_at[bi] + 1]);
            else
              next_carried.insert(next_carried.end(), carried.begin() + carried_off[bi], carried.begin() + carried_off[bi + 1]);
            next_off.push_back((uint32_t)next_carried.size());
            continue;
          }
        }
        finish(r);
      }
      ls.continued += in_flight.size();
      lap(ls.decide);
      if (want_rounds) {
        char line[256];
        snprintf(line, sizeof line, "STREAM_ROUND ctx %p t %.3f ms: %zu chunks (%zu carried anchors), ms: admit %.3f build %.3f device %.3f "
                 "results %.3f anchors %.3f decide %.3f, %zu go on", (void *)ctx, (GetRealTime() - t_map) * 1e3, batch.size(), carried.size(),
                 (ls.admit - prev.admit) * 1e3, (ls.build - prev.build) * 1e3, (ls.device - prev.device) * 1e3,
                 (ls.download - prev.download) * 1e3, (ls.anchors - prev.anchors) * 1e3, (ls.decide - prev.decide) * 1e3, in_flight.size());
        std::lock_guard<std::mutex> lock(stats_mutex);
        round_lines.push_back(line);
        prev = ls;
      }
    }
    {
      uint32_t dc[16] = {0};  // [4] batches mapped a second time after a speculative run outgrew a buffer, [5] speculative runs
      if (smhip_debug_counters(ctx, dc) == SMHIP_OK) {
        ls.spec_reruns = dc[4];
        ls.spec_runs = dc[5];
      }
    }
    std::lock_guard<std::mutex> lock(stats_mutex);
    total_stats.spec_runs += ls.spec_runs; total_stats.spec_reruns += ls.spec_reruns;
    total_stats.admit += ls.admit; total_stats.build += ls.build; total_stats.device += ls.device;
    total_stats.download += ls.download; total_stats.anchors += ls.anchors; total_stats.decide += ls.decide;
    total_stats.rounds += ls.rounds; total_stats.chunks += ls.chunks; total_stats.carried_up += ls.carried_up;
    total_stats.anchors_down += ls.anchors_down; total_stats.continued += ls.continued;
    total_stats.max_batch = std::max(total_stats.max_batch, ls.max_batch);
  };
  {
    std::vector<std::thread> workers;
    for (int g = 0; g < num_gpus; ++g)
      for (size_t k = 0; k < lanes_used; ++k)
        if (g || k) workers.emplace_back(lane, g, ctxs[g][k]);
    lane(0, ctxs[0][0]);
    for (auto &w : workers) w.join();
  }
  const double map_span = GetRealTime() - t_map;
  std::cerr << "Finished mapping in " << map_span << ", # reads: " << num_reads << "\n";
  if (warm_up)
    std::cerr << "(set-up before that span: " << warm_span << " s of warm-up on the job's own first chunks, results discarded; "
              << "--no-warmup maps without it)\n";
  if (want_stats) {
    const LaneStats &t = total_stats;
    const double lanes = (double)num_gpus * (double)lanes_used;
    char buf[1024];
    snprintf(buf, sizeof buf,
             "STREAM_STATS {\"reads\": %zu, \"span_s\": %.6f, \"warmup_s\": %.6f, \"lanes\": %d, \"chunks_in_flight_per_lane\": %zu, \"rounds\": %llu, "
             "\"chunks\": %llu, \"max_batch\": %llu, \"carried_anchors_uploaded\": %llu, \"anchors_downloaded\": %llu, "
             "\"speculative_batches\": %llu, \"batches_mapped_twice\": %llu, "
             "\"lane_seconds\": {\"admit_upload\": %.6f, \"build_batch\": %.6f, \"device\": %.6f, \"download_results\": %.6f, "
             "\"download_anchors\": %.6f, \"decide\": %.6f}}",
             num_reads, map_span, warm_span, (int)lanes, lane_budget, (unsigned long long)t.rounds, (unsigned long long)t.chunks,
             (unsigned long long)t.max_batch, (unsigned long long)t.carried_up, (unsigned long long)t.anchors_down,
             (unsigned long long)t.spec_runs, (unsigned long long)t.spec_reruns, t.admit, t.build,
             t.device, t.download, t.anchors, t.decide);
    std::cerr << buf << "\n";
    for (const auto &l : round_lines) std::cerr << l << "\n";
  }

  // sigmap.cc:197-215: rows grouped by reference index, reads in input order inside a group
  // (whichever GPU and context mapped a read)
  std::vector<std::vector<std::string>> rows(std::max<uint32_t>(num_refs, 1));
  for (size_t r = 0; r < num_reads; ++r) {
    ReadState &s = st[r];
    uint32_t chunk_index = s.chunk_index;  // sigmap.cc:690-693
    if (chunk_index > 0 && (chunk_index == num_chunks[r] || chunk_index == max_chunks)) --chunk_index;
    uint32_t bucket = 0;
    const PAFMapping m = MakeMapping(s.last, (uint32_t)r, reads.GetSignalNameAt(r),
                                     (uint32_t)reads.GetSignalLengthAt(r), chunk_index, s.num_events, ref_lengths,
                                     params_, true, s.start_time, &bucket, s.has_dt ? &s.dt : nullptr,
                                     sdtw_cigar ? &s.cigar : nullptr);
    rows[bucket].push_back(FormatPAF(m, num_refs ? reference.GetSequenceNameAt(bucket) : "*",
                                     num_refs ? ref_lengths[bucket] : 0));
  }
  FILE *out = fopen(output_file_path_.c_str(), "w");
  if (!out) ExitWithMessage("Cannot open output file " + output_file_path_);
  for (auto &g : rows)
    for (auto &line : g) fputs(line.c_str(), out);
  fclose(out);
  for (int g = num_gpus; g-- > 0;)
    for (size_t k = ctxs[g].size(); k-- > 0;) smhip_destroy(ctxs[g][k]);  // borrowers first
}

// Development mode `--features events|cwt -s DIR -o OUT` (no index, no reference): the feature signal the
// mapping path (events: Sigmap::GenerateEvents, sigmap.cc:1048-1083) or the reference's dormant CWT path
// (cwt: Sigmap::GenerateFeatureSignalUsingCWT, sigmap.cc:1085-1098, scale0 = 8 as for reads, :947) derives from
// every 4000-sample chunk of every read, computed on the GPU, as text: `read<TAB>chunk<TAB>k<TAB>position<TAB>value`
// with %f values like the reference's disabled debugging dumps (EventsToText, sigmap.cc:1198-1235, which writes
// the z-scored event means of whole reads, one `%f` per line).  position = sample index of a CWT peak / valley
// inside its chunk; 0 for events (the compression drops the spans).
void Sigmap::DumpFeatures(const std::string &kind, int device, const std::string &signal_directory,
                         const std::string &output_file_path) {
  if (kind != "events" && kind != "cwt") ExitWithMessage("--features takes `events` or `cwt`");
  SignalBatch reads;
  reads.InitializeLoading(signal_directory);
  const size_t num_reads = reads.LoadAllReadSignals();  // host conversion: pA, (30, 200) filter
  smhip_ctx *ctx = nullptr;
  if (smhip_init(device, &ctx) != SMHIP_OK) ExitWithMessage("No usable GPU (smhip_init failed).");
  smhip_params dp;
  smhip_default_params(&dp);
  FILE *out = fopen(output_file_path.c_str(), "w");
  if (!out) ExitWithMessage("Cannot open output file " + output_file_path);
  const uint32_t chunk_size = 4000, stride = 2736, batch_chunks = 4096;
  std::vector<float> samples, feat((size_t)batch_chunks * stride), stdv;
  std::vector<uint32_t> offsets(1, 0), pos((size_t)batch_chunks * stride), nf(batch_chunks), owner, chunk_of;
  auto flush = [&]() {
    const uint32_t n = (uint32_t)owner.size();
    if (n == 0) return;
    int rc;
    if (kind == "cwt") {
      rc = smhip_cwt_features(ctx, samples.data(), offsets.data(), n, 8.0f, stride, feat.data(), pos.data(), nf.data(),
                              nullptr);
    } else {
      stdv.resize((size_t)n * stride);
      rc = smhip_events(ctx, &dp, samples.data(), offsets.data(), n, stride, feat.data(), stdv.data(), nf.data(), nullptr,
                        nullptr);
    }
    if (rc != SMHIP_OK) ExitWithMessage(std::string("GPU feature extraction failed: ") + smhip_last_error(ctx));
    for (uint32_t c = 0; c < n; ++c)
      for (uint32_t k = 0; k < std::min(nf[c], stride); ++k)
        fprintf(out, "%s\t%u\t%u\t%u\t%f\n", reads.GetSignalNameAt(owner[c]), chunk_of[c], k,
                kind == "cwt" ? pos[(size_t)c * stride + k] : 0u, feat[(size_t)c * stride + k]);
    samples.clear();
    offsets.assign(1, 0);
    owner.clear();
    chunk_of.clear();
  };
  for (size_t r = 0; r < num_reads; ++r) {
    const std::vector<float> &v = reads.GetSignalAt(r).signal_values;
    for (uint32_t c = 0; (size_t)(c + 1) * chunk_size <= v.size(); ++c) {  // whole chunks only (sigmap.cc:643)
      samples.insert(samples.end(), v.begin() + (size_t)c * chunk_size, v.begin() + (size_t)(c + 1) * chunk_size);
      offsets.push_back((uint32_t)samples.size());
      owner.push_back((uint32_t)r);
      chunk_of.push_back(c);
      if (owner.size() == batch_chunks) flush();
    }
  }
  flush();
  fclose(out);
  smhip_destroy(ctx);
  std::cerr << "Wrote the " << kind << " features of " << num_reads << " reads.\n";
}


namespace {
struct Args {
  std::map<std::string, std::string> kv;
  bool Has(const std::string &k) const { return kv.count(k) != 0; }
  std::string Str(const std::string &k) const { return kv.at(k); }
};
const char *kHelp =
    "Map ONT raw signal data\nUsage:\n  sigmap [OPTION...]\n\n Indexing options:\n"
    "  -i, --build-index    Build spatial index for reference\n  -d, --dimension INT  Dimension of spatial index [6]\n"
    "  -l, --max-leaf INT   Max leaf of spatial index [20]\n\n Mapping options:\n  -m, --map              Map signal data\n"
    "      --step-size INT    Seeding step size in reads [2]\n  -t, --num-threads INT  # threads for mapping [1] (host side)\n\n"
    " Input options:\n  -r, --ref FILE         Reference file\n  -p, --pore-model FILE  Pore model file\n"
    "  -x, --ref-index FILE   Reference index file\n  -s, --sig-dir DIR      Signal data directory (FAST5 / BLOW5 / .sgr files)\n\n"
    " Output options:\n  -o, --output arg  Output file\n\n Development options:\n"
    "      --search-radius FLT         Search radius for each seed [0.08]\n"
    "      --max-num-chunks INT        Max # chunks before stop trying to map a read [30]\n"
    "      --min-num-anchors INT       Min # anchors to stop mapping [10]\n"
    "      --min-num-anchors-output INT  Min # anchors to output mappings [10]\n"
    "      --stop-mapping FLOAT        best/second-best chaining score ratio to stop mapping [1.4]\n"
    "      --stop-mapping-output FLOAT best/second-best ratio to output mappings [1.2]\n"
    "      --stop-mapping-mean FLOAT   best/mean chaining score ratio to stop mapping [5]\n"
    "      --stop-mapping-mean-output FLOAT  best/mean ratio to output mappings [5]\n"
    "      --sdtw-band INT             banded sDTW refinement of the best chain, adds dt:f: [0 = off]\n"
    "      --sdtw-cigar                with --sdtw-band: also the banded DTW() path's per-base CIGAR, cg:Z:\n"
    "      --device INT                first GPU ordinal [0]\n"
    "      --gpus INT                  GPUs to map on: the index is broadcast over xGMI (RCCL), reads shard [1]\n"
    "      --gpus-share-device         (development) the --gpus are all GPU --device and share one index\n"
    "      --batch-chunks INT          chunks in flight per GPU [32768]\n"
    "      --streams INT               device contexts / HIP streams that share the index and the chunk budget\n"
    "                                  [up to 2: one when a GPU's reads fit one context's share of --batch-chunks]\n"
    "      --no-warmup                 no discarded round of first chunks before the mapping span (the span then holds the\n"
    "                                  contexts' first-batch costs, like the reference's own)\n"
    "      --features events|cwt       (development) with -s DIR -o FILE: dump every chunk's feature signal as text\n"
    "  -h, --help  Print help\n";
}  // namespace

void ParseArgsAndRun(int argc, char *argv[]) {
  static const std::map<std::string, std::string> shorts = {
      {"i", "build-index"}, {"d", "dimension"}, {"l", "max-leaf"}, {"m", "map"}, {"t", "num-threads"},
      {"r", "ref"}, {"p", "pore-model"}, {"x", "ref-index"}, {"s", "sig-dir"}, {"o", "output"}, {"h", "help"}};
  static const std::vector<std::string> flags = {"build-index", "map", "help", "sdtw-cigar", "gpus-share-device", "no-warmup"};
  // every option this program knows: anything else is an error, like the reference's cxxopts parser
  // (sigmap.cc:1378 `options.parse` throws on an unknown option)
  static const std::vector<std::string> valued = {
      "dimension", "max-leaf", "step-size", "num-threads", "ref", "pore-model", "ref-index", "sig-dir", "output",
      "search-radius", "max-num-chunks", "min-num-anchors", "min-num-anchors-output", "stop-mapping",
      "stop-mapping-output", "stop-mapping-mean", "stop-mapping-mean-output", "sdtw-band", "device", "gpus",
      "batch-chunks", "streams", "features"};
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    if (k.rfind("--", 0) == 0) k = k.substr(2);
    else if (k.rfind("-", 0) == 0 && shorts.count(k.substr(1))) k = shorts.at(k.substr(1));
    else ExitWithMessage("Unknown argument " + k);
    std::string v = "1";
    const size_t eq = k.find('=');
    if (eq != std::string::npos) {
      v = k.substr(eq + 1);
      k = k.substr(0, eq);
    }
    const bool is_flag = std::find(flags.begin(), flags.end(), k) != flags.end();
    if (!is_flag && std::find(valued.begin(), valued.end(), k) == valued.end())
      ExitWithMessage("Option '" + std::string(argv[i]) + "' does not exist");
    if (!is_flag && eq == std::string::npos) {
      if (i + 1 >= argc) ExitWithMessage("Option '" + k + "' is missing an argument");
      v = argv[++i];
    }
    a.kv[k] = v;
  }
  MappingParams p;
  if (a.Has("search-radius")) p.search_radius = std::stof(a.Str("search-radius"));
  if (a.Has("step-size")) p.read_seeding_step_size = std::stoi(a.Str("step-size"));
  if (a.Has("max-num-chunks")) p.max_num_chunks = std::stoi(a.Str("max-num-chunks"));
  if (a.Has("min-num-anchors")) p.stop_mapping_min_num_anchors = std::stoi(a.Str("min-num-anchors"));
  if (a.Has("min-num-anchors-output")) p.output_mapping_min_num_anchors = std::stoi(a.Str("min-num-anchors-output"));
  if (a.Has("stop-mapping")) p.stop_mapping_ratio = std::stof(a.Str("stop-mapping"));
  if (a.Has("stop-mapping-output")) p.output_mapping_ratio = std::stof(a.Str("stop-mapping-output"));
  if (a.Has("stop-mapping-mean")) p.stop_mapping_mean_ratio = std::stof(a.Str("stop-mapping-mean"));
  if (a.Has("stop-mapping-mean-output")) p.output_mapping_mean_ratio = std::stof(a.Str("stop-mapping-mean-output"));
  const int num_threads = a.Has("num-threads") ? std::stoi(a.Str("num-threads")) : 1;
  auto need = [&](const char *k, const char *msg) {
    if (!a.Has(k)) ExitWithMessage(msg);
    return a.Str(k);
  };
  if (a.Has("features")) {
    const std::string dir = need("sig-dir", "No signal data directory specified!");
    const std::string out = need("output", "No output file specified!");
    Sigmap::DumpFeatures(a.Str("features"), a.Has("device") ? std::stoi(a.Str("device")) : 0, dir, out);
  } else if (a.Has("build-index")) {
    const int dimension = a.Has("dimension") ? std::stoi(a.Str("dimension")) : 6;
    const int max_leaf = a.Has("max-leaf") ? std::stoi(a.Str("max-leaf")) : 20;
    std::cerr << "Dimension: " << dimension << ", max leaf: " << max_leaf << "\n";
    // The device kernels are specialised (include/sigmap_hip.h, smhip_index_upload): refuse here,
    // not after an hour of index construction, what `-m` could not load.
    if (dimension != 6)
      ExitWithMessage("This build maps with index dimension 6 only (-d " + std::to_string(dimension) +
                      " would build an index that -m cannot use).");
    if (max_leaf < 1 || max_leaf > 21)
      ExitWithMessage("This build maps with at most 21 points per leaf (-l " + std::to_string(max_leaf) +
                      " would build an index that -m cannot use).");
    const std::string ref = need("ref", "No reference file specified!");
    const std::string pm = need("pore-model", "No pore model file specified!");
    const std::string out = need("output", "No output file specified!");
    Sigmap(dimension, max_leaf, ref, pm, out).ConstructIndex();
  } else if (a.Has("map")) {
    std::cerr << "Number of threads: " << num_threads << "\n";
    const std::string ref = need("ref", "No reference file specified!");
    const std::string pm = need("pore-model", "No pore model file specified!");
    const std::string idx = need("ref-index", "No reference index file specified!");
    const std::string dir = need("sig-dir", "No signal data directory specified!");
    const std::string out = need("output", "No output file specified!");
    omp_set_num_threads(std::max(1, num_threads));
    const int device = a.Has("device") ? std::stoi(a.Str("device")) : 0;
    const size_t batch = a.Has("batch-chunks") ? (size_t)std::stoul(a.Str("batch-chunks")) : 32768;
    const int sdtw_band = a.Has("sdtw-band") ? std::stoi(a.Str("sdtw-band")) : 0;
    const int streams = a.Has("streams") ? std::stoi(a.Str("streams")) : -2;  // (default: up to two, as the reads need)
    const int gpus = a.Has("gpus") ? std::stoi(a.Str("gpus")) : 1;
    Sigmap(p, num_threads, ref, pm, dir, idx, out).StreamingMap(device, gpus, batch, sdtw_band, streams, a.Has("sdtw-cigar"),
                                                               a.Has("gpus-share-device"), !a.Has("no-warmup"));
  } else {
    std::cerr << kHelp;
  }
}
}  // namespace sigmap

int main(int argc, char *argv[]) {
  // The HIP runtime multiplexes a process's streams over 4 hardware queues unless told otherwise; the contexts of a
  // GPU (--streams, three streams each) then wait for each other's kernels.  Read when the runtime initialises, i.e.
  // at the first HIP call below; a value in the caller's environment wins.
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  sigmap::ParseArgsAndRun(argc, argv);
  return 0;
}
