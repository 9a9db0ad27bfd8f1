// sequence_batch.h -- reference genome container (API of the reference's SequenceBatch as
// used by the driver: sequence_batch.h:34-55,84-140: InitializeLoading, LoadAllSequences,
// PrepareNegativeSequenceAt, Get{Sequence,SequenceLength,SequenceName,NegativeSequence}At).
// Own FASTA/FASTQ reader on zlib (plain or gzip input); the reference's kseq.h is third party.
#ifndef SIGMAP_AMD_SEQUENCEBATCH_H_
#define SIGMAP_AMD_SEQUENCEBATCH_H_
#include <cstdint>
#include <string>
#include <vector>

namespace sigmap {
class SequenceBatch {
 public:
  void InitializeLoading(const std::string &sequence_file_path);
  uint32_t LoadAllSequences();  // skips empty records; returns number of sequences
  void FinalizeLoading() {}
  void AddSequence(const std::string &name, const std::string &bases);  // in-memory feed
  void PrepareNegativeSequenceAt(uint32_t i);  // reverse complement, N stays N
  uint32_t GetNumSequences() const { return (uint32_t)names_.size(); }
  uint64_t GetNumBases() const { return num_bases_; }
  const char *GetSequenceAt(uint32_t i) const { return seqs_[i].c_str(); }
  uint32_t GetSequenceLengthAt(uint32_t i) const { return (uint32_t)seqs_[i].size(); }
  const char *GetSequenceNameAt(uint32_t i) const { return names_[i].c_str(); }
  const std::string &GetNegativeSequenceAt(uint32_t i) const { return neg_seqs_[i]; }

 protected:
  std::string path_;
  uint64_t num_bases_ = 0;
  std::vector<std::string> names_, seqs_, neg_seqs_;
};
}  // namespace sigmap
#endif
