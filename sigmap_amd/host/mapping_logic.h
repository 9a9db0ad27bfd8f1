// mapping_logic.h -- the per-read control logic of Sigmap::StreamingMap that stays on the
// host (sigmap.cc:631-866): early-stop test after each chunk, output test, coordinates,
// tag string.  Inputs are the device results of include/sigmap_hip.h.
#ifndef SIGMAP_AMD_MAPPINGLOGIC_H_
#define SIGMAP_AMD_MAPPINGLOGIC_H_
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sigmap_hip.h"
#include "output_tools.h"

namespace sigmap {
struct MappingParams {  // CLI defaults, sigmap.cc:1380-1419
  float search_radius = 0.08f;
  int read_seeding_step_size = 2;
  int max_num_chunks = 30;
  int stop_mapping_min_num_anchors = 10;
  int output_mapping_min_num_anchors = 10;
  float stop_mapping_ratio = 1.4f;
  float output_mapping_ratio = 1.2f;
  float stop_mapping_mean_ratio = 5;
  float output_mapping_mean_ratio = 5;
};

// Contiguous block of items for part `rank` of `world` (read i -> GPU i * world / n):
// [*lo, *hi).  Either pointer may be null.
void ShardRange(size_t n_items, int rank, int world, size_t *lo, size_t *hi);

// Chunks one device context may have in flight so that `contexts` of them fit the GPU's free memory: per chunk a context
// needs its slot of `keep_samples` floats, ~288 queries x (16 bytes x fast hit slots + ~1.4 KB of per-query records and
// overflow pool) and ~110 bytes per anchor (288 x hits_per_query, capped at the reference's 5000 per query: what the
// sort's and the chain stage's buffers are reserved at) + 120 KB of per-chunk rows, and 1.5 GB whatever its budget (the
// walker's stack room, the capacities no batch goes below); four tenths of the free memory are left to growth and to
// the other users of the device.  At least 1.
// (12 Mb reference, 41 hits per query, 512 slots: 4.2 MB per chunk; 250 Mb, 850 hits, 1024 slots: 32 MB per chunk.
// Round 5, eight "GPUs" folded onto one device at 250 Mb -- sixteen contexts: the earlier form, 60 bytes per anchor and
// no fixed part, let each have 321 chunks and the warm-up ran out of memory.)
size_t ChunkBudgetForMemory(uint64_t free_bytes, uint32_t keep_samples, uint32_t fast_slots, double hits_per_query,
                            size_t contexts);

// Device contexts of a GPU that `sigmap -m` gives work when the number is its to choose (no --streams on the command
// line): one when the GPU's block of reads fits a single context's read slots (`lane_budget`: the rounds of multi-chunk
// reads are a dependent chain -- two half-sized chains side by side are slower than one), otherwise as
// many as the reads fill, up to `max_contexts`.  At least 1.
size_t ContextsForReads(size_t block_reads, size_t lane_budget, size_t max_contexts);

// Per-base CIGAR of a DTW path (print_alignment, fast_dtw.cc:160-209): codes 0 one-to-one, 1 one base
// to several signals, 2 skip one signal, 3 skip one base; "(2M)(1M3I)(1D)...".  The format, including
// that the run still open when the path ends is not written, is the reference's.
std::string PrintAlignment(const uint8_t *codes, size_t n);

// true = stop feeding chunks of this read (sigmap.cc:667-687).  Only meaningful when the
// chunk produced more than 50 features (the caller checks n_features like sigmap.cc:660).
bool ShouldStopMapping(const smhip_chunk_result &r, const MappingParams &p);

// Build the PAF record after the last processed chunk (sigmap.cc:690-865).
// chunk_index = index of the last chunk counted the way the reference counts it
// (sigmap.cc:690-693), num_events = features consumed over all chunks with > 50 features.
// include_time_tag: prepend "mt:f:<ms>" like the reference (wall clock; excluded from parity).
// sdtw_distance: when given, appended as the extra tag dt:f: (banded sDTW refinement of chain 0;
// not part of the reference's PAF).  cigar: when given, appended as cg:Z: (print_alignment's per-base
// CIGAR of the banded fast_dtw.cc::DTW() path over the same window).
PAFMapping MakeMapping(const smhip_chunk_result &r, uint32_t read_id, const std::string &read_name,
                       uint32_t read_length, uint32_t chunk_index, uint32_t num_events,
                       const std::vector<uint32_t> &ref_lengths, const MappingParams &p,
                       bool include_time_tag, double mapping_time_ms, uint32_t *out_ref_bucket,
                       const float *sdtw_distance = nullptr, const std::string *cigar = nullptr);
}  // namespace sigmap
#endif
