// host_abi.cc -- extern "C" surface of the host library (include/sigmap_host.h).
#include <omp.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/sigmap_host.h"
#include "mapping_logic.h"
#include "pore_model.h"
#include "sequence_batch.h"
#include "sigmap.h"
#include "signal_batch.h"
#include "spatial_index.h"

using namespace sigmap;

struct smhost_index {
  SpatialIndex *si;
};

namespace sigmap {
// The ConstructIndex pipeline shared by the CLI (-i) and the C ABI.
void BuildIndex(SequenceBatch &ref, const PoreModel &pore_model, SpatialIndex &index) {
  const uint32_t n = ref.GetNumSequences();
  for (uint32_t i = 0; i < n; ++i) ref.PrepareNegativeSequenceAt(i);
  std::vector<std::vector<bool>> pos_masked, neg_masked;
  GenerateMaskedPositions(index.dimension() + pore_model.GetKmerSize() - 1, 0.0002f, ref, pos_masked,
                          neg_masked);
  SignalBatch ref_signals;
  ref_signals.ConvertSequencesToSignals(ref, pore_model, n);
  std::vector<std::vector<float>> pos(n), neg(n);
#pragma omp parallel for schedule(dynamic)
  for (uint32_t i = 0; i < n; ++i) {
    const Signal &s = ref_signals.GetSignalAt(i);
    ZscoreNormalize(s.signal_values.data(), s.signal_values.size(), pos[i]);
    ZscoreNormalize(s.negative_signal_values.data(), s.negative_signal_values.size(), neg[i]);
  }
  index.Construct(n, pos_masked, neg_masked, pos, neg);
}
}  // namespace sigmap

extern "C" {

smhost_index *smhost_index_build(const float *level_means, int kmer_size, const char *const *seqs,
                                 const uint32_t *seq_lens, uint32_t n_seqs, int dim, int max_leaf,
                                 int num_threads) {
  if (num_threads < 1) num_threads = omp_get_max_threads();
  PoreModel pm;
  pm.SetTable(kmer_size, level_means);
  SequenceBatch sb;
  for (uint32_t i = 0; i < n_seqs; ++i)
    sb.AddSequence("ref" + std::to_string(i), std::string(seqs[i], seq_lens[i]));
  smhost_index *ix = new smhost_index;
  ix->si = new SpatialIndex(dim, max_leaf, num_threads, "");
  omp_set_num_threads(num_threads);
  BuildIndex(sb, pm, *ix->si);
  return ix;
}

smhost_index *smhost_index_from_points(const uint64_t *positions, const float *values,
                                       uint64_t n_points, int dim, int max_leaf, int num_threads) {
  if (num_threads < 1) num_threads = omp_get_max_threads();
  if (n_points < (uint64_t)dim) return nullptr;
  smhost_index *ix = new smhost_index;
  ix->si = new SpatialIndex(dim, max_leaf, num_threads, "");
  ix->si->SetPointCloud(std::vector<uint64_t>(positions, positions + n_points),
                        std::vector<float>(values, values + n_points));
  ix->si->BuildTree();
  return ix;
}

smhost_index *smhost_index_load(const char *prefix) {
  smhost_index *ix = new smhost_index;
  ix->si = new SpatialIndex(1000, std::vector<int>(), prefix);
  ix->si->Load();
  return ix;
}

int smhost_index_save(const smhost_index *ix, const char *prefix) {
  ix->si->set_index_file_path_prefix(prefix);
  ix->si->Save();
  return 0;
}

void smhost_index_free(smhost_index *ix) {
  if (!ix) return;
  delete ix->si;
  delete ix;
}

int smhost_index_get(const smhost_index *ix, smhost_index_view *out) {
  if (!ix || !out) return -1;
  const SpatialIndex &si = *ix->si;
  out->dim = si.dimension();
  out->max_leaf = si.max_leaf();
  out->depth = si.tree().depth;
  out->pad = 0;
  out->n_points = si.values().size();
  out->m_size = si.tree().m_size;
  out->n_nodes = si.tree().nodes.size();
  out->values = si.values().data();
  out->positions = si.positions().data();
  out->vind = si.tree().vind.data();
  out->nodes = si.tree().nodes.data();
  out->bbox = si.tree().bbox.data();
  return 0;
}

uint64_t smhost_level_means(const float *level_means, int kmer_size, const char *seq,
                            uint32_t seq_len, float *out) {
  PoreModel pm;
  pm.SetTable(kmer_size, level_means);
  const std::string s(seq, seq_len);  // guarantees the NUL the rolling hash reads
  const std::vector<float> v = pm.GetLevelMeansAt(s.c_str(), 0, seq_len);
  memcpy(out, v.data(), v.size() * sizeof(float));
  return v.size();
}

uint64_t smhost_reference_signal(const float *level_means, int kmer_size, const char *seq,
                                 uint32_t seq_len, int negative_strand, float *out) {
  PoreModel pm;
  pm.SetTable(kmer_size, level_means);
  SequenceBatch sb;
  sb.AddSequence("r", std::string(seq, seq_len));
  sb.PrepareNegativeSequenceAt(0);
  const char *s = negative_strand ? sb.GetNegativeSequenceAt(0).c_str() : sb.GetSequenceAt(0);
  const std::vector<float> lv = pm.GetLevelMeansAt(s, 0, seq_len);
  std::vector<float> z;
  ZscoreNormalize(lv.data(), lv.size(), z);
  memcpy(out, z.data(), z.size() * sizeof(float));
  return z.size();
}

struct smhost_signals {
  SignalBatch batch;
};

int smhost_have_fast5(void) { return SignalBatch::HaveFAST5() ? 1 : 0; }

smhost_signals *smhost_signals_load(const char *path) {
  if (!path) return nullptr;
  smhost_signals *s = new smhost_signals;
  s->batch.InitializeLoading(path);
  s->batch.LoadAllReadSignals();
  s->batch.FinalizeLoading();
  return s;
}

void smhost_signals_free(smhost_signals *s) { delete s; }

uint64_t smhost_signals_count(const smhost_signals *s) { return s ? s->batch.GetNumSignals() : 0; }

int smhost_signals_get(const smhost_signals *s, uint64_t i, const char **id, const float **pa, uint64_t *n,
                       float *digitisation, float *range, float *offset) {
  if (!s || i >= s->batch.GetNumSignals()) return -1;
  const Signal &g = s->batch.GetSignalAt(i);
  if (id) *id = g.id.c_str();
  if (pa) *pa = g.signal_values.data();
  if (n) *n = g.signal_values.size();
  if (digitisation) *digitisation = g.digitisation;
  if (range) *range = g.range;
  if (offset) *offset = g.offset;
  return 0;
}

int smhost_write_fast5(const char *path, int multi_read, uint32_t n_reads, const char *const *ids,
                       const int16_t *const *raws, const uint32_t *raw_lens, float digitisation,
                       float range, float offset) {
  if (!path || !ids || !raws || !raw_lens) return -1;
  std::vector<std::string> idv(n_reads);
  std::vector<std::vector<int16_t>> rv(n_reads);
  for (uint32_t r = 0; r < n_reads; ++r) {
    idv[r] = ids[r];
    rv[r].assign(raws[r], raws[r] + raw_lens[r]);
  }
  return SignalBatch::WriteFAST5(path, multi_read != 0, idv, digitisation, range, offset, rv) ? 0 : -1;
}

int smhost_write_blow5(const char *path, int record_press, int signal_press, uint32_t n_reads, const char *const *ids,
                       const int16_t *const *raws, const uint32_t *raw_lens, double digitisation, double range,
                       double offset) {
  if (!path || !ids || !raws || !raw_lens) return -1;
  std::vector<std::string> idv(n_reads);
  std::vector<std::vector<int16_t>> rv(n_reads);
  for (uint32_t r = 0; r < n_reads; ++r) {
    idv[r] = ids[r];
    rv[r].assign(raws[r], raws[r] + raw_lens[r]);
  }
  return SignalBatch::WriteBLOW5(path, record_press, signal_press, idv, digitisation, range, offset, rv) ? 0 : -1;
}

int smhost_have_zstd(void) { return SignalBatch::HaveZstd() ? 1 : 0; }

uint32_t smhost_raw_to_pa(const int16_t *raw, uint32_t n_raw, float digitisation, float range,
                          float offset, float *out) {
  SignalBatch b;
  b.AddSignal("", digitisation, range, offset, raw, n_raw);
  const std::vector<float> &v = b.GetSignalAt(0).signal_values;
  memcpy(out, v.data(), v.size() * sizeof(float));
  return (uint32_t)v.size();
}

uint32_t smhost_chunk_bounds(const int16_t *raw, uint32_t n_raw, float digitisation, float range, float offset,
                             uint32_t chunk_size, uint32_t max_chunks, uint32_t *bounds, uint32_t cap,
                             uint64_t *filtered_length) {
  SignalBatch b;
  b.KeepRaw([](size_t n) { return malloc(n); }, [](void *p) { free(p); }, (size_t)1 << 20, chunk_size, max_chunks);
  b.AddSignal("", digitisation, range, offset, raw, n_raw);
  const Signal &s = b.GetSignalAt(0);
  if (filtered_length) *filtered_length = s.filtered_length;
  const uint32_t n = (uint32_t)std::min<size_t>(s.chunk_bounds.size(), cap);
  if (bounds) memcpy(bounds, s.chunk_bounds.data(), (size_t)n * sizeof(uint32_t));
  return n;
}

void smhost_default_map_params(smhost_map_params *p) {
  const MappingParams d;
  p->search_radius = d.search_radius;
  p->step_size = d.read_seeding_step_size;
  p->max_num_chunks = d.max_num_chunks;
  p->stop_min_anchors = d.stop_mapping_min_num_anchors;
  p->output_min_anchors = d.output_mapping_min_num_anchors;
  p->stop_ratio = d.stop_mapping_ratio;
  p->output_ratio = d.output_mapping_ratio;
  p->stop_mean_ratio = d.stop_mapping_mean_ratio;
  p->output_mean_ratio = d.output_mapping_mean_ratio;
}

static MappingParams ToParams(const smhost_map_params *p) {
  MappingParams m;
  m.search_radius = p->search_radius;
  m.read_seeding_step_size = p->step_size;
  m.max_num_chunks = p->max_num_chunks;
  m.stop_mapping_min_num_anchors = p->stop_min_anchors;
  m.output_mapping_min_num_anchors = p->output_min_anchors;
  m.stop_mapping_ratio = p->stop_ratio;
  m.output_mapping_ratio = p->output_ratio;
  m.stop_mapping_mean_ratio = p->stop_mean_ratio;
  m.output_mapping_mean_ratio = p->output_mean_ratio;
  return m;
}

int64_t smhost_print_alignment(const uint8_t *codes, uint64_t n, char *out, uint64_t cap) {
  const std::string c = sigmap::PrintAlignment(codes, (size_t)n);
  if (out && cap) {
    const size_t k = std::min<size_t>(c.size(), (size_t)cap - 1);
    memcpy(out, c.data(), k);
    out[k] = 0;
  }
  return (int64_t)c.size();
}

void smhost_shard_range(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi) {
  size_t a = 0, b = 0;
  sigmap::ShardRange((size_t)n_items, rank, world, &a, &b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

uint64_t smhost_chunk_budget(uint64_t free_bytes, uint32_t keep_samples, uint32_t fast_slots, double hits_per_query,
                             uint32_t contexts) {
  return (uint64_t)sigmap::ChunkBudgetForMemory(free_bytes, keep_samples, fast_slots, hits_per_query, contexts);
}

uint64_t smhost_contexts_for_reads(uint64_t block_reads, uint64_t lane_budget, uint64_t max_contexts) {
  return (uint64_t)sigmap::ContextsForReads((size_t)block_reads, (size_t)lane_budget, (size_t)max_contexts);
}

int smhost_should_stop(const smhip_chunk_result *r, const smhost_map_params *p) {
  return ShouldStopMapping(*r, ToParams(p)) ? 1 : 0;
}

int smhost_format_paf(const smhip_chunk_result *r, const char *read_name, uint32_t read_length,
                      uint32_t chunk_index, uint32_t num_events, const char *const *ref_names,
                      const uint32_t *ref_lengths, uint32_t n_refs, const smhost_map_params *p,
                      char *buf, size_t cap, uint32_t *ref_bucket) {
  std::vector<uint32_t> lens(ref_lengths, ref_lengths + n_refs);
  uint32_t bucket = 0;
  const PAFMapping m = MakeMapping(*r, 0, read_name, read_length, chunk_index, num_events, lens,
                                   ToParams(p), false, 0.0, &bucket);
  const std::string line = FormatPAF(m, ref_names[bucket], ref_lengths[bucket]);
  if (ref_bucket) *ref_bucket = bucket;
  if (line.size() + 1 > cap) return -1;
  memcpy(buf, line.c_str(), line.size() + 1);
  return (int)line.size();
}

int smhost_format_paf_record(const char *read_name, uint32_t read_length, uint32_t read_start, uint32_t read_end,
                             uint32_t frag_start, uint32_t frag_length, uint32_t mapq, uint32_t direction,
                             const char *tags, const char *ref_name, uint32_t ref_length, char *buf, size_t cap) {
  if (!read_name || !tags || !ref_name || !buf) return -1;
  // (mapq and direction pass through the widths of the reference's record: 6 bits and 1 bit, output_tools.h:26)
  const PAFMapping m{0, read_name, read_length, read_start, read_end, frag_start, frag_length,
                     (uint8_t)(mapq & 63u), (uint8_t)(direction & 1u), 1, tags};
  const std::string line = FormatPAF(m, ref_name, ref_length);
  if (line.size() + 1 > cap) return -1;
  memcpy(buf, line.c_str(), line.size() + 1);
  return (int)line.size();
}

struct smhost_sequences {
  SequenceBatch batch;
};
smhost_sequences *smhost_sequences_load(const char *path) {
  if (!path) return nullptr;
  smhost_sequences *s = new smhost_sequences;
  s->batch.InitializeLoading(path);
  const uint32_t n = s->batch.LoadAllSequences();
  for (uint32_t i = 0; i < n; ++i) s->batch.PrepareNegativeSequenceAt(i);
  s->batch.FinalizeLoading();
  return s;
}
void smhost_sequences_free(smhost_sequences *s) { delete s; }
uint32_t smhost_sequences_count(const smhost_sequences *s) { return s ? s->batch.GetNumSequences() : 0; }
uint64_t smhost_sequences_num_bases(const smhost_sequences *s) { return s ? s->batch.GetNumBases() : 0; }
int smhost_sequences_get(const smhost_sequences *s, uint32_t i, const char **name, const char **bases,
                         uint32_t *length, const char **negative) {
  if (!s || i >= s->batch.GetNumSequences()) return -1;
  if (name) *name = s->batch.GetSequenceNameAt(i);
  if (bases) *bases = s->batch.GetSequenceAt(i);
  if (length) *length = s->batch.GetSequenceLengthAt(i);
  if (negative) *negative = s->batch.GetNegativeSequenceAt(i).c_str();
  return 0;
}

// ---- the per-chunk class methods (device_methods.cc) behind a C face, for the ctypes tests ----
int smhost_index_bind_device(smhost_index *ix, void *smhip_context) {
  if (!ix) return -1;
  ix->si->BindDevice(static_cast<smhip_ctx *>(smhip_context));
  return 0;
}

int64_t smhost_generate_chains(smhost_index *ix, const float *query, uint32_t n_query, uint32_t query_start_offset, int step,
                               float radius, uint32_t n_refs, const smhip_chain *prev_chains, uint32_t n_prev,
                               const smhip_carried_anchor *prev_anchors, smhip_chain *out_chains, uint32_t chain_cap,
                               smhip_carried_anchor *out_anchors, uint64_t anchor_cap, uint64_t *n_out_anchors) {
  if (!ix || (n_query && !query) || !out_chains || !out_anchors || !n_out_anchors) return -1;
  std::vector<SignalAnchorChain> chains;
  size_t at = 0;
  for (uint32_t i = 0; i < n_prev; ++i) {
    const smhip_chain &c = prev_chains[i];
    SignalAnchorChain ch{c.score, c.ref, c.start, c.end, c.n_anchors, (uint8_t)c.mapq, c.direction == 1 ? Positive : Negative, {}};
    for (uint32_t k = 0; k < c.n_anchors; ++k, ++at)
      ch.anchors.push_back(SignalAnchor{prev_anchors[at].target_position, prev_anchors[at].query_position, prev_anchors[at].distance});
    chains.push_back(std::move(ch));
  }
  const std::vector<float> q(query, query + n_query), stdvs;
  ix->si->GenerateChains(q, stdvs, query_start_offset, step, radius, n_refs, chains);
  uint64_t total = 0;
  for (const auto &c : chains) total += c.anchors.size();
  *n_out_anchors = total;
  if (chains.size() > chain_cap || total > anchor_cap) return -2;
  at = 0;
  for (size_t i = 0; i < chains.size(); ++i) {
    const SignalAnchorChain &c = chains[i];
    out_chains[i] = smhip_chain{c.score, c.reference_sequence_index, c.start_position, c.end_position, c.num_anchors, c.mapq,
                                c.direction == Positive ? 1u : 0u, 0u};
    for (const SignalAnchor &a : c.anchors)
      out_anchors[at++] = smhip_carried_anchor{(c.reference_sequence_index << 1) | (c.direction == Positive ? 0u : 1u),
                                               a.target_position, a.query_position, a.distance};
  }
  return (int64_t)chains.size();
}

int64_t smhost_generate_events(void *smhip_context, const float *signal_pa, uint64_t n, uint64_t start, uint64_t end, float *feat,
                               float *stdvs, uint32_t cap, uint32_t stdvs_already) {
  if (!signal_pa || !feat || !stdvs) return -1;
  Sigmap sm(MappingParams(), 1, "", "", "", "", "");
  if (smhip_context) sm.BindDevice(static_cast<smhip_ctx *>(smhip_context));
  Signal sig;
  sig.signal_values.assign(signal_pa, signal_pa + n);
  std::vector<float> f(7, 1.0f);                       // stale content: the method clears it
  std::vector<float> s(stdvs_already, -1.0f);          // ... and appends to this one
  sm.GenerateEvents(start, end, sig, f, s);
  if (f.size() > cap || s.size() > cap) return -2;
  memcpy(feat, f.data(), f.size() * sizeof(float));
  memcpy(stdvs, s.data(), s.size() * sizeof(float));
  return (int64_t)f.size();
}
}
