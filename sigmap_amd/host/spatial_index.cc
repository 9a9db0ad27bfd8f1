#include "spatial_index.h"

#include <omp.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include "sequence_batch.h"
#include "utils.h"

namespace sigmap {
void SpatialIndex::GeneratePointCloudOnOneDirection(Direction direction, uint32_t signal_index,
                                                    const std::vector<std::vector<bool>> &is_masked,
                                                    const float *signal_values, size_t signal_length,
                                                    int step_size) {
  if (signal_length < (uint32_t)dimension_) return;
  const uint64_t strand = direction == Positive ? 0 : 1;
  const std::vector<bool> &masked = is_masked[signal_index];
  for (size_t p = 0; p < signal_length - dimension_ + 1; p += step_size) {
    if (masked[p]) continue;
    if (p == 0 || values_.empty() || std::fabs(signal_values[p] - values_.back()) > 0.01) {
      positions_.push_back(((((uint64_t)signal_index) << 32 | (uint32_t)p) << 1) | strand);
      values_.push_back(signal_values[p]);
    }
  }
}

void SpatialIndex::BuildTree() {
  tree_.Build(values_.data(), values_.size(), dimension_, max_leaf_, num_threads_);
}

void SpatialIndex::SetPointCloud(std::vector<uint64_t> positions, std::vector<float> values) {
  positions_ = std::move(positions);
  values_ = std::move(values);
}

void SpatialIndex::Construct(size_t num_signals, const std::vector<std::vector<bool>> &positive_is_masked,
                             const std::vector<std::vector<bool>> &negative_is_masked,
                             const std::vector<std::vector<float>> &positive_signals,
                             const std::vector<std::vector<float>> &negative_signals) {
  const double t0 = GetRealTime();
  size_t total = 0;
  for (const auto &s : positive_signals) total += s.size();
  positions_.reserve(2 * total);
  values_.reserve(2 * total);
  for (size_t i = 0; i < num_signals; ++i)
    GeneratePointCloudOnOneDirection(Positive, (uint32_t)i, positive_is_masked,
                                     positive_signals[i].data(), positive_signals[i].size(), 1);
  for (size_t i = 0; i < num_signals; ++i)
    GeneratePointCloudOnOneDirection(Negative, (uint32_t)i, negative_is_masked,
                                     negative_signals[i].data(), negative_signals[i].size(), 1);
  std::cerr << "Collected " << values_.size() << " points.\n";
  BuildTree();
  std::cerr << "Built spatial index.\n";
  std::cerr << "Built index successfully in " << GetRealTime() - t0 << "s.\n";
}

void SpatialIndex::Save() {
  const double t0 = GetRealTime();
  FILE *f = fopen((index_file_path_prefix_ + ".pt").c_str(), "wb");
  if (!f) ExitWithMessage("Cannot write " + index_file_path_prefix_ + ".pt");
  const int32_t dim = dimension_, ml = max_leaf_;
  const uint64_t n = values_.size();
  fwrite(&dim, 4, 1, f);
  fwrite(&ml, 4, 1, f);
  fwrite(&n, 8, 1, f);
  struct Rec { uint64_t position; float value; uint32_t pad; };
  std::vector<Rec> buf(1 << 16);
  for (uint64_t i = 0; i < n; i += buf.size()) {
    const uint64_t c = std::min<uint64_t>(buf.size(), n - i);
    for (uint64_t k = 0; k < c; ++k) buf[k] = Rec{positions_[i + k], values_[i + k], 0};
    fwrite(buf.data(), sizeof(Rec), c, f);
  }
  fclose(f);
  if (!tree_.Save(index_file_path_prefix_ + ".si")) ExitWithMessage("Cannot write " + index_file_path_prefix_ + ".si");
  std::cerr << "Saved in " << GetRealTime() - t0 << "s.\n";
}

void SpatialIndex::Load() {
  const double t0 = GetRealTime();
  FILE *f = fopen((index_file_path_prefix_ + ".pt").c_str(), "rb");
  if (!f) ExitWithMessage("Cannot open " + index_file_path_prefix_ + ".pt");
  int32_t dim = 0, ml = 0;
  uint64_t n = 0;
  if (fread(&dim, 4, 1, f) != 1 || fread(&ml, 4, 1, f) != 1 || fread(&n, 8, 1, f) != 1)
    ExitWithMessage("Truncated point cloud file");
  dimension_ = dim;
  max_leaf_ = ml;
  positions_.resize(n);
  values_.resize(n);
  struct Rec { uint64_t position; float value; uint32_t pad; };
  std::vector<Rec> buf(1 << 16);
  for (uint64_t i = 0; i < n; i += buf.size()) {
    const uint64_t c = std::min<uint64_t>(buf.size(), n - i);
    if (fread(buf.data(), sizeof(Rec), c, f) != c) ExitWithMessage("Truncated point cloud file");
    for (uint64_t k = 0; k < c; ++k) {
      positions_[i + k] = buf[k].position;
      values_[i + k] = buf[k].value;
    }
  }
  fclose(f);
  std::cerr << "Load point cloud successfully! dim: " << dimension_ << ", max leaf: " << max_leaf_
            << ", point cloud size: " << n << "\n";
  if (!tree_.Load(index_file_path_prefix_ + ".si")) ExitWithMessage("Cannot load " + index_file_path_prefix_ + ".si");
  if (tree_.dim != dimension_ || tree_.m_size != n - dimension_ + 1)
    ExitWithMessage("Index files do not belong together (.pt vs .si)");
  std::cerr << "Loaded index successfully in " << GetRealTime() - t0 << "s.\n";
}

namespace {
// count table: dense array for k <= 14 (4^k u32 counters), hash map above that
struct KmerCounts {
  int k;
  std::vector<uint32_t> dense;
  std::unordered_map<uint64_t, uint64_t> sparse;
  explicit KmerCounts(int k_) : k(k_) {
    if (k <= 14) dense.assign((size_t)1 << (2 * k), 0);
  }
  void Add(uint64_t key) {
    if (!dense.empty()) ++dense[key]; else ++sparse[key];
  }
  uint64_t Get(uint64_t key) const {
    if (!dense.empty()) return dense[key];
    auto it = sparse.find(key);
    return it == sparse.end() ? 0 : it->second;
  }
};

template <typename F>
void ForEachCanonicalKmer(const char *seq, uint32_t len, int k, F &&f) {
  const uint64_t shift = 2 * (uint64_t)(k - 1);
  const uint64_t mask = (((uint64_t)1) << (2 * k)) - 1;
  uint64_t fw = 0, rv = 0;
  int unambiguous = 0;
  for (uint32_t p = 0; p < len; ++p) {
    const uint8_t b = CharToUint8(seq[p]);
    if (b < 4) {
      fw = ((fw << 2) | b) & mask;
      rv = (rv >> 2) | (((uint64_t)(3 ^ b)) << shift);
      if (++unambiguous >= k) f(p, true, fw < rv ? fw : rv);
    } else {
      unambiguous = 0;
      fw = rv = 0;
      f(p, false, 0);
    }
  }
}
}  // namespace

void GenerateMaskedPositions(int kmer_size, float frequency, const SequenceBatch &sb,
                             std::vector<std::vector<bool>> &positive_is_masked,
                             std::vector<std::vector<bool>> &negative_is_masked) {
  const double t0 = GetRealTime();
  KmerCounts counts(kmer_size);
  uint64_t num_kmers = 0;
  const uint32_t n = sb.GetNumSequences();
  for (uint32_t si = 0; si < n; ++si)
    ForEachCanonicalKmer(sb.GetSequenceAt(si), sb.GetSequenceLengthAt(si), kmer_size,
                         [&](uint32_t, bool valid, uint64_t key) {
                           if (valid) {
                             counts.Add(key);
                             ++num_kmers;
                           }
                         });
  std::cerr << "# " << kmer_size << "mers: " << num_kmers << "\n";
  uint64_t num_masked = 0;
  auto mask_strand = [&](const char *seq, uint32_t len, std::vector<bool> &masked) {
    masked.assign(len - kmer_size + 1, false);
    ForEachCanonicalKmer(seq, len, kmer_size, [&](uint32_t p, bool valid, uint64_t key) {
      if (valid) {
        const bool m = (counts.Get(key) / (float)num_kmers) > frequency;
        masked[p + 1 - kmer_size] = m;
        num_masked += m;
      } else if (p >= (uint32_t)kmer_size - 1) {
        masked[p + 1 - kmer_size] = true;
        ++num_masked;
      }
    });
  };
  positive_is_masked.resize(n);
  negative_is_masked.resize(n);
  for (uint32_t si = 0; si < n; ++si) {
    mask_strand(sb.GetSequenceAt(si), sb.GetSequenceLengthAt(si), positive_is_masked[si]);
    mask_strand(sb.GetNegativeSequenceAt(si).c_str(), sb.GetSequenceLengthAt(si), negative_is_masked[si]);
  }
  std::cerr << "Mask " << num_masked << " high frequency kmer in " << GetRealTime() - t0 << "s.\n";
}
}  // namespace sigmap
