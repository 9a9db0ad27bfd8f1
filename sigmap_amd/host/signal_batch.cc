#include "signal_batch.h"

#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "utils.h"

namespace sigmap {
void SignalBatch::KeepRaw(ArenaAlloc alloc, ArenaFree free_fn, size_t block_bytes, uint32_t chunk_size, uint32_t max_chunks) {
  raw_mode_ = true;
  alloc_ = alloc;
  free_ = free_fn;
  block_bytes_ = block_bytes;
  chunk_size_ = chunk_size;
  max_chunks_ = max_chunks;
}

// Survivors of the (30, 200) pA filter among raw[0 .. n) -- signal_batch.cc:139-151's arithmetic: f32 add, f32 multiply,
// two comparisons -- and the raw positions at which the filtered array reaches chunk_size, 2 chunk_size, ... samples.
static size_t CountFiltered(const int16_t *raw, size_t n, float offset, float scale, uint32_t chunk_size, uint32_t max_chunks,
                            std::vector<uint32_t> *bounds) {
  size_t kept = 0, next = chunk_size;  // `next`: the filtered length at which the next chunk is complete
  bounds->assign(1, 0u);
  constexpr size_t kBlock = 512;
  for (size_t i0 = 0; i0 < n; i0 += kBlock) {
    const size_t m = std::min(kBlock, n - i0);
    uint32_t cnt = 0;
    for (size_t i = 0; i < m; ++i) {  // (vectorises: no dependence but the sum)
      const float pa = ((float)raw[i0 + i] + offset) * scale;
      cnt += (pa > 30 && pa < 200) ? 1u : 0u;
    }
    if (chunk_size && bounds->size() <= max_chunks && kept + cnt >= next) {  // a chunk ends inside this block
      size_t k = kept;
      for (size_t i = 0; i < m; ++i) {
        const float pa = ((float)raw[i0 + i] + offset) * scale;
        if (pa > 30 && pa < 200 && ++k == next) {
          bounds->push_back((uint32_t)(i0 + i + 1));
          next += chunk_size;
          if (bounds->size() > max_chunks) break;
        }
      }
    }
    kept += cnt;
  }
  return kept;
}

SignalBatch::~SignalBatch() {
  for (auto &b : blocks_)
    if (b.base && free_) free_(b.base);
}

void SignalBatch::AddSignal(const std::string &id, float digitisation, float range, float offset,
                            const int16_t *raw, size_t n_raw) {
  Signal s;
  s.id = id;
  s.digitisation = digitisation;
  s.range = range;
  s.offset = offset;
  if (raw_mode_) {  // keep the ADC counts; the device converts (smhip_reads_upload_raw)
    if (blocks_.empty() || blocks_.back().used + n_raw > blocks_.back().cap) {
      const size_t cap = std::max(block_bytes_ / sizeof(int16_t), n_raw);
      int16_t *base = static_cast<int16_t *>(alloc_(std::max<size_t>(cap, 1) * sizeof(int16_t)));
      if (!base) ExitWithMessage("Cannot allocate the raw-signal arena.");
      blocks_.push_back(Block{base, cap, 0});
    }
    Block &b = blocks_.back();
    memcpy(b.base + b.used, raw, n_raw * sizeof(int16_t));
    s.raw = b.base + b.used;
    s.n_raw = n_raw;
    s.raw_block = (uint32_t)(blocks_.size() - 1);
    b.used += n_raw;
    if (chunk_size_)
      s.filtered_length = CountFiltered(s.raw, n_raw, offset, range / digitisation, chunk_size_, max_chunks_, &s.chunk_bounds);
    signals_.push_back(std::move(s));
    return;
  }
  s.signal_values.reserve(n_raw);
  const float scale = range / digitisation;
  for (size_t i = 0; i < n_raw; ++i) {
    const float pa = ((float)raw[i] + offset) * scale;
    if (pa > 30 && pa < 200) s.signal_values.push_back(pa);
  }
  signals_.push_back(std::move(s));
}

void SignalBatch::AddSignalsFromSGR(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) ExitWithMessage("Cannot open signal file " + path);
  char magic[4];
  uint32_t n = 0;
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "SGR1", 4) != 0 || fread(&n, 4, 1, f) != 1)
    ExitWithMessage("Not an SGR1 signal file: " + path);
  std::vector<int16_t> raw;
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t id_len = 0, n_raw = 0;
    float dig = 0, range = 0, offset = 0;
    if (fread(&id_len, 4, 1, f) != 1) ExitWithMessage("Truncated signal file " + path);
    std::string id(id_len, '\0');
    if (fread(&id[0], 1, id_len, f) != id_len || fread(&dig, 4, 1, f) != 1 ||
        fread(&range, 4, 1, f) != 1 || fread(&offset, 4, 1, f) != 1 || fread(&n_raw, 4, 1, f) != 1)
      ExitWithMessage("Truncated signal file " + path);
    raw.resize(n_raw);
    if (fread(raw.data(), 2, n_raw, f) != n_raw) ExitWithMessage("Truncated signal file " + path);
    AddSignal(id, dig, range, offset, raw.data(), n_raw);
  }
  fclose(f);
}

bool SignalBatch::WriteSGR(const std::string &path, const std::vector<std::string> &ids,
                           float digitisation, float range, float offset,
                           const std::vector<std::vector<int16_t>> &raws) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const uint32_t n = (uint32_t)ids.size();
  fwrite("SGR1", 1, 4, f);
  fwrite(&n, 4, 1, f);
  for (uint32_t r = 0; r < n; ++r) {
    const uint32_t id_len = (uint32_t)ids[r].size(), n_raw = (uint32_t)raws[r].size();
    fwrite(&id_len, 4, 1, f);
    fwrite(ids[r].data(), 1, id_len, f);
    fwrite(&digitisation, 4, 1, f);
    fwrite(&range, 4, 1, f);
    fwrite(&offset, 4, 1, f);
    fwrite(&n_raw, 4, 1, f);
    fwrite(raws[r].data(), 2, n_raw, f);
  }
  return fclose(f) == 0;
}

// signal_batch.cc:24-67: every file under the directory (sub-directories included) whose path
// contains ".fast5" is read as FAST5, else one whose path contains ".blow5" as BLOW5 (:46-55);
// ".sgr" files are this build's own container.  Files are taken in sorted path order (the reference
// follows readdir order).
static void CollectSignalFiles(const std::string &dir_path, std::vector<std::string> *fast5,
                               std::vector<std::string> *blow5, std::vector<std::string> *sgr, int depth) {
  DIR *dir = opendir(dir_path.c_str());
  if (!dir) return;
  struct dirent *ent;
  std::vector<std::string> subdirs;
  while ((ent = readdir(dir)) != nullptr) {
    const std::string name = ent->d_name;
    if (name == "." || name == "..") continue;
    const std::string path = dir_path + "/" + name;
    DIR *sub = opendir(path.c_str());
    if (sub) {
      closedir(sub);
      subdirs.push_back(path);
    } else if (path.find(".fast5") != std::string::npos) {
      fast5->push_back(path);
    } else if (path.find(".blow5") != std::string::npos) {
      blow5->push_back(path);
    } else if (name.size() > 4 && name.compare(name.size() - 4, 4, ".sgr") == 0) {
      sgr->push_back(path);
    }
  }
  closedir(dir);
  if (depth < 8)
    for (const auto &p : subdirs) CollectSignalFiles(p, fast5, blow5, sgr, depth + 1);
}

size_t SignalBatch::LoadAllReadSignals() {
  const double t0 = GetRealTime();
  std::vector<std::string> fast5, blow5, sgr;
  DIR *dir = opendir(signal_directory_.c_str());
  if (dir) {
    closedir(dir);
    CollectSignalFiles(signal_directory_, &fast5, &blow5, &sgr, 0);
    std::sort(fast5.begin(), fast5.end());
    std::sort(blow5.begin(), blow5.end());
    std::sort(sgr.begin(), sgr.end());
  } else if (signal_directory_.find(".fast5") != std::string::npos) {
    fast5.push_back(signal_directory_);  // a single file
  } else if (signal_directory_.find(".blow5") != std::string::npos) {
    blow5.push_back(signal_directory_);
  } else {
    sgr.push_back(signal_directory_);
  }
  for (const auto &p : fast5) AddSignalsFromFAST5(p);  // signal_batch.cc:61-66: FAST5 files first, then BLOW5
  for (const auto &p : blow5) AddSignalsFromBLOW5(p);
  for (const auto &p : sgr) AddSignalsFromSGR(p);
  std::cerr << "Loaded " << signals_.size() << " reads in " << GetRealTime() - t0 << "s.\n";
  return signals_.size();
}

void SignalBatch::ConvertSequencesToSignals(const SequenceBatch &sequence_batch,
                                            const PoreModel &pore_model, size_t num_sequences) {
  const double t0 = GetRealTime();
  for (size_t i = 0; i < num_sequences; ++i) {
    const size_t len = sequence_batch.GetSequenceLengthAt(i);
    Signal s;
    s.id = sequence_batch.GetSequenceNameAt(i);
    s.signal_values = pore_model.GetLevelMeansAt(sequence_batch.GetSequenceAt(i), 0, (uint32_t)len);
    s.negative_signal_values =
        pore_model.GetLevelMeansAt(sequence_batch.GetNegativeSequenceAt(i).c_str(), 0, (uint32_t)len);
    signals_.push_back(std::move(s));
  }
  std::cerr << "Convert " << num_sequences << " sequences to signals in " << GetRealTime() - t0
            << "s.\n";
}

float ZscoreNormalize(const float *x, size_t n, std::vector<float> &out) {
  double mean = 0;
  for (size_t i = 0; i < n; ++i) mean += x[i];
  mean /= n;
  double SD = 0;
  for (size_t i = 0; i < n; ++i) SD += (x[i] - mean) * (x[i] - mean);
  SD /= (n - 1);
  SD = sqrt(SD);
  out.reserve(out.size() + n);
  for (size_t i = 0; i < n; ++i) out.push_back((float)((x[i] - mean) / SD));
  return (float)SD;
}
}  // namespace sigmap
