#include "pore_model.h"

#include <fstream>
#include <sstream>

#include "utils.h"

namespace sigmap {
void PoreModel::Load(const std::string &path) {
  const double t0 = GetRealTime();
  std::ifstream in(path);
  if (!in) ExitWithMessage("Cannot open pore model file " + path);
  std::string line;
  int num_kmers = 0;
  while (getline(in, line)) {
    if (line.empty() || line[0] == '#' || line.compare(0, 4, "kmer") == 0) continue;
    std::stringstream ss(line);
    std::string kmer;
    float level_mean = 0;
    ss >> kmer >> level_mean;
    if (kmer_size_ == 0) {
      kmer_size_ = (int)kmer.length();
      level_means_.assign((size_t)1 << (2 * kmer_size_), 0.0f);
    }
    if ((int)kmer.length() != kmer_size_) ExitWithMessage("Inconsistent k-mer length in pore model");
    level_means_[GenerateSeedFromSequence(kmer.data(), kmer.length(), 0, kmer_size_)] = level_mean;
    ++num_kmers;
  }
  std::cerr << "Loaded " << num_kmers << " kmers in " << GetRealTime() - t0 << "s.\n";
}

void PoreModel::SetTable(int kmer_size, const float *level_means) {
  kmer_size_ = kmer_size;
  level_means_.assign(level_means, level_means + ((size_t)1 << (2 * kmer_size)));
}

std::vector<float> PoreModel::GetLevelMeansAt(const char *sequence, uint32_t start_position,
                                              uint32_t end_position) const {
  std::vector<float> out;
  const int32_t signal_length = (int32_t)(end_position - start_position - kmer_size_ + 1);
  if (signal_length <= 0) return out;
  out.reserve(signal_length);
  const uint32_t mask = ((uint32_t)1 << (2 * kmer_size_)) - 1;
  uint32_t h = (uint32_t)GenerateSeedFromSequence(sequence, (size_t)start_position + signal_length,
                                                  start_position, kmer_size_);
  out.push_back(level_means_[h]);
  for (uint32_t p = start_position + 1; p < end_position - kmer_size_ + 1; ++p) {
    const uint8_t b = CharToUint8(sequence[p + kmer_size_]);
    h = ((h << 2) | (b < 4 ? b : 0)) & mask;
    out.push_back(level_means_[h]);
  }
  return out;
}
}  // namespace sigmap
