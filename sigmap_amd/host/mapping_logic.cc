#include "mapping_logic.h"

namespace sigmap {
size_t ChunkBudgetForMemory(uint64_t free_bytes, uint32_t keep_samples, uint32_t fast_slots, double hits_per_query,
                            size_t contexts) {
  const double per_chunk = (double)keep_samples * sizeof(float) + 288.0 * (16.0 * fast_slots + 1400.0) +
                           288.0 * std::min(std::max(hits_per_query, 0.0), 5000.0) * 110.0 + 120000.0;
  const double fixed = 1.5e9;  // per context, whatever its budget: the walker's 1 GB of stack room, the capacities a batch never goes below
  const double share = (double)free_bytes * 0.6 / (double)std::max<size_t>(contexts, 1) - fixed;
  const double fit = share / per_chunk;
  return fit < 1.0 ? 1 : (size_t)fit;
}

size_t ContextsForReads(size_t block_reads, size_t lane_budget, size_t max_contexts) {
  const size_t budget = std::max<size_t>(lane_budget, 1);
  const size_t need = std::max<size_t>(1, (block_reads + budget - 1) / budget);
  return std::max<size_t>(1, std::min(need, std::max<size_t>(max_contexts, 1)));
}

void ShardRange(size_t n_items, int rank, int world, size_t *lo, size_t *hi) {
  if (world < 1) world = 1;
  // item i belongs to part floor(i * world / n): part r starts at ceil(r * n / world)
  auto start = [&](int r) { return r >= world ? n_items : (size_t)(((unsigned __int128)n_items * (unsigned)r + world - 1) / world); };
  if (lo) *lo = start(rank);
  if (hi) *hi = start(rank + 1);
}

std::string PrintAlignment(const uint8_t *codes, size_t n) {
  if (n == 0) return "";
  // One string per base.  `state` is the kind of run in progress: 1 = signals matched to the current
  // base (a code 0 cell opens it, code 1 cells extend it: both print as M), 2 = skipped signals (I),
  // 3 = a skipped base (D).
  static const char kLetter[] = {'?', 'M', 'I', 'D'};
  std::vector<std::string> bases;
  int state;
  size_t run = 1;
  if (codes[0] == 3) {
    bases.push_back("1D");
    state = 3;
  } else {
    state = codes[0] == 0 ? 1 : 2;
  }
  bases.emplace_back();
  for (size_t i = 1; i < n; ++i) {
    const int c = codes[i];
    if (c == 1 || c == 2) {  // still inside the current base
      if (c == state) {
        ++run;
      } else {
        bases.back() += std::to_string(run) + kLetter[state];
        run = 1;
        state = c;
      }
      continue;
    }
    // code 0 / 3: the current base is complete, the next one starts (matched / skipped)
    bases.back() += std::to_string(run) + kLetter[state];
    state = c == 0 ? 1 : 3;
    if (i != n - 1) {
      bases.emplace_back();
      run = 1;
    }
  }
  std::string cigar;
  for (const auto &base : bases) cigar += "(" + base + ")";
  return cigar;
}

bool ShouldStopMapping(const smhip_chunk_result &r, const MappingParams &p) {
  if (r.n_chains >= 2) {
    if (r.chains[0].score / r.chains[1].score >= p.stop_mapping_ratio) return true;
    float mean_chain_score = r.sum_score;
    mean_chain_score /= r.n_chains;
    if (r.chains[0].score >= p.stop_mapping_mean_ratio * mean_chain_score) return true;
  } else if (r.n_chains == 1 && r.chains[0].n_anchors >= (uint32_t)p.stop_mapping_min_num_anchors) {
    return true;
  }
  return false;
}

PAFMapping MakeMapping(const smhip_chunk_result &r, uint32_t read_id, const std::string &read_name,
                       uint32_t read_length, uint32_t chunk_index, uint32_t num_events,
                       const std::vector<uint32_t> &ref_lengths, const MappingParams &p,
                       bool include_time_tag, double mapping_time_ms, uint32_t *out_ref_bucket,
                       const float *sdtw_distance, const std::string *cigar) {
  const uint32_t bp_per_sec = 450, sample_rate = 4000, chunk_size = 4000;
  const float read_position_scale =
      ((float)(chunk_index + 1) * chunk_size / num_events) / ((float)sample_rate / bp_per_sec);
  float mean_chain_score = r.sum_score;  // 0 when there are no chains -> 0/0 = NaN below
  mean_chain_score /= r.n_chains;
  std::string tags;
  if (include_time_tag) tags.append("mt:f:" + std::to_string(mapping_time_ms) + "\t");
  tags.append("ci:i:" + std::to_string(chunk_index + 1));
  tags.append("\tsl:i:" + std::to_string(read_length));
  if (r.n_chains >= 1) {
    const smhip_chain &c0 = r.chains[0];
    float ad = r.c0_sum_dist, at = r.c0_sum_tgap, aq = r.c0_sum_qgap;
    ad /= c0.n_anchors;
    at /= c0.n_anchors;
    aq /= c0.n_anchors;
    tags.append("\tcm:i:" + std::to_string(c0.n_anchors));
    tags.append("\tnc:i:" + std::to_string(r.n_chains));
    tags.append("\ts1:f:" + std::to_string(c0.score));
    tags.append("\ts2:f:" + std::to_string(r.n_chains > 1 ? r.chains[1].score : 0.0f));
    tags.append("\tsm:f:" + std::to_string(mean_chain_score));
    tags.append("\tad:f:" + std::to_string(ad));
    tags.append("\tat:f:" + std::to_string(at));
    tags.append("\taq:f:" + std::to_string(aq));
    if (sdtw_distance) tags.append("\tdt:f:" + std::to_string(*sdtw_distance));
    if (cigar && !cigar->empty()) tags.append("\tcg:Z:" + *cigar);
  }
  const bool output =
      (r.n_chains >= 2 && (r.chains[0].score / r.chains[1].score >= p.output_mapping_ratio ||
                           r.chains[0].score >= p.output_mapping_mean_ratio * mean_chain_score)) ||
      (r.n_chains == 1 && r.chains[0].n_anchors >= (uint32_t)p.output_mapping_min_num_anchors);
  if (output) {
    const smhip_chain &c0 = r.chains[0];
    *out_ref_bucket = c0.ref;
    return PAFMapping{read_id,
                      read_name,
                      read_length,
                      (uint32_t)(read_position_scale * r.c0_last_qpos),
                      (uint32_t)(read_position_scale * r.c0_first_qpos),
                      c0.direction == 1 ? c0.start : (uint32_t)(ref_lengths[c0.ref] + 1 - c0.end),
                      (uint32_t)(c0.end - c0.start + 1),
                      (uint8_t)c0.mapq,
                      (uint8_t)(c0.direction == 1 ? 1 : 0),
                      1,
                      tags};
  }
  *out_ref_bucket = 0;
  return PAFMapping{read_id, read_name, read_length, 0, 0, 0, 0, 61, 0, 1, tags};
}
}  // namespace sigmap
