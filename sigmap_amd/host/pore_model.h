// pore_model.h -- k-mer -> expected current table (API of the reference's PoreModel,
// pore_model.h:19-33: Load, GetLevelMeansAt, GetKmerSize).
#ifndef SIGMAP_AMD_POREMODEL_H_
#define SIGMAP_AMD_POREMODEL_H_
#include <cstdint>
#include <string>
#include <vector>

namespace sigmap {
class PoreModel {
 public:
  // Text table "kmer<ws>level_mean[<ws>...]"; lines starting with '#' or "kmer" are headers.
  void Load(const std::string &pore_model_file_path);
  void SetTable(int kmer_size, const float *level_means);  // 4^k entries
  // Expected current for sequence[start, end): one value per k-mer start.
  // Matches the reference's arithmetic including its rolling-hash indexing
  // (pore_model.cc:67-79): value 0 is k-mer [start, start+k); value p >= 1 shifts in
  // sequence[p + k] (so base start+k is never used and the last step reads the string's
  // terminating NUL, counted as A).  `sequence` must therefore be NUL-terminated.
  std::vector<float> GetLevelMeansAt(const char *sequence, uint32_t start_position,
                                     uint32_t end_position) const;
  int GetKmerSize() const { return kmer_size_; }
  const std::vector<float> &level_means() const { return level_means_; }

 protected:
  int kmer_size_ = 0;
  std::vector<float> level_means_;
};
}  // namespace sigmap
#endif
