#include "sequence_batch.h"

#include <zlib.h>

#include "utils.h"

namespace sigmap {
void SequenceBatch::InitializeLoading(const std::string &path) { path_ = path; }

void SequenceBatch::AddSequence(const std::string &name, const std::string &bases) {
  names_.push_back(name);
  seqs_.push_back(bases);
  neg_seqs_.emplace_back();
  num_bases_ += bases.size();
}

uint32_t SequenceBatch::LoadAllSequences() {
  const double t0 = GetRealTime();
  gzFile f = gzopen(path_.c_str(), "r");
  if (f == NULL) ExitWithMessage("Cannot find sequence file!");
  std::vector<char> buf(1 << 20);
  std::string line, name, bases;
  bool in_record = false, in_quality = false;
  size_t qual_left = 0;
  auto flush = [&]() {
    if (in_record && !bases.empty()) AddSequence(name, bases);
    in_record = false;
    bases.clear();
  };
  auto handle_line = [&](const std::string &l) {
    if (in_quality) {  // FASTQ quality block: as many characters as bases
      qual_left = l.size() >= qual_left ? 0 : qual_left - l.size();
      if (qual_left == 0) {
        in_quality = false;
        flush();
      }
      return;
    }
    if (l.empty()) return;
    if (l[0] == '>' || (l[0] == '@' && !in_record)) {
      flush();
      size_t e = 1;
      while (e < l.size() && l[e] != ' ' && l[e] != '\t') ++e;
      name = l.substr(1, e - 1);
      in_record = true;
    } else if (l[0] == '+' && in_record) {
      in_quality = true;
      qual_left = bases.size();
      if (qual_left == 0) in_quality = false;
    } else if (in_record) {
      bases += l;
    }
  };
  int n;
  while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
    for (int i = 0; i < n; ++i) {
      const char c = buf[i];
      if (c == '\n') {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        handle_line(line);
        line.clear();
      } else {
        line.push_back(c);
      }
    }
  }
  if (!line.empty()) handle_line(line);
  flush();
  gzclose(f);
  std::cerr << "Number of sequences: " << names_.size() << ".\n";
  std::cerr << "Number of bases: " << num_bases_ << ".\n";
  std::cerr << "Loaded all sequences successfully in " << GetRealTime() - t0 << "s.\n";
  return (uint32_t)names_.size();
}

void SequenceBatch::PrepareNegativeSequenceAt(uint32_t i) {
  const std::string &s = seqs_[i];
  std::string &neg = neg_seqs_[i];
  neg.resize(s.size());
  for (size_t k = 0; k < s.size(); ++k) {
    const uint8_t b = CharToUint8(s[s.size() - 1 - k]);
    neg[k] = b < 4 ? Uint8ToChar(3 ^ b) : 'N';
  }
}
}  // namespace sigmap
