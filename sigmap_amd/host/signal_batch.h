// signal_batch.h -- raw-signal container (API of the reference's SignalBatch/Signal,
// signal_batch.h:14-57: InitializeLoading, LoadAllReadSignals, GetSignalAt,
// GetSignalNameAt, GetSignalLengthAt, FinalizeLoading, ConvertSequencesToSignals).
//
// On-disk read containers: FAST5 (single- and multi-read, fast5_reader.cc, when HDF5 is present
// at build time), BLOW5 (blow5_reader.cc: the binary SLOW5 layout parsed directly, slow5lib is not
// vendored) and this build's own flat format (".sgr", below).  The pA conversion and (30,200) pA
// filter that define what the hot path sees are implemented here.
//
//   .sgr : "SGR1" | u32 n_reads | n_reads x { u32 id_len | id bytes | f32 digitisation |
//          f32 range | f32 offset | u32 n_raw | i16 raw[n_raw] }       (little endian)
#ifndef SIGMAP_AMD_SIGNALBATCH_H_
#define SIGMAP_AMD_SIGNALBATCH_H_
#include <cstdint>
#include <string>
#include <vector>

#include "pore_model.h"
#include "sequence_batch.h"

namespace sigmap {
struct Signal {
  std::string id;
  float digitisation = 0, range = 0, offset = 0;
  std::vector<float> signal_values;           // pA, filtered
  std::vector<float> negative_signal_values;  // reference minus strand only
  // Raw mode (SignalBatch::KeepRaw): the read stays as its ADC counts in the batch's arena and the
  // conversion above runs on the GPU (smhip_reads_upload_raw); signal_values stays empty and the
  // filtered length is filled in by the mapper when the device has counted it.
  const int16_t *raw = nullptr;
  size_t n_raw = 0;
  uint32_t raw_block = 0;        // arena block holding the samples
  size_t filtered_length = 0;
  // ... and, when the batch was told the chunk size (KeepRaw), where the chunks the mapper cuts out of the FILTERED
  // array (sigmap.cc:637-652) begin in the RAW one: chunk c = the survivors of raw[chunk_bounds[c] .. chunk_bounds[c + 1]),
  // exactly chunk_size of them; min(filtered_length / chunk_size, max_chunks) + 1 entries.  Counted at load time with
  // AddSignal's own arithmetic, so the mapper can send the device one chunk's samples at a time.
  std::vector<uint32_t> chunk_bounds;
  size_t GetSignalLength() const { return raw ? filtered_length : signal_values.size(); }
};

class SignalBatch {
 public:
  void InitializeLoading(const std::string &signal_directory) { signal_directory_ = signal_directory; }
  void FinalizeLoading() {}
  size_t LoadAllReadSignals();  // every *.sgr under the directory (or the file itself)
  // pA = (raw + offset) * (range / digitisation); kept iff 30 < pA < 200
  void AddSignal(const std::string &id, float digitisation, float range, float offset,
                 const int16_t *raw, size_t n_raw);
  void AddSignalsFromSGR(const std::string &path);
  void AddSignalsFromFAST5(const std::string &path);  // signal_batch.cc:69-151
  void AddSignalsFromBLOW5(const std::string &path);  // signal_batch.cc:156-210 (blow5_reader.cc)
  static bool HaveFAST5();
  static bool HaveZstd();
  // record_press: 0 none, 1 zlib, 2 zstd; signal_press: 0 none, 1 svb-zd (tests / generator)
  static bool WriteBLOW5(const std::string &path, int record_press, int signal_press, const std::vector<std::string> &ids,
                         double digitisation, double range, double offset,
                         const std::vector<std::vector<int16_t>> &raws);
  static bool WriteFAST5(const std::string &path, bool multi_read, const std::vector<std::string> &ids,
                         float digitisation, float range, float offset,
                         const std::vector<std::vector<int16_t>> &raws);
  static bool WriteSGR(const std::string &path, const std::vector<std::string> &ids,
                       float digitisation, float range, float offset,
                       const std::vector<std::vector<int16_t>> &raws);
  void ConvertSequencesToSignals(const SequenceBatch &sequence_batch, const PoreModel &pore_model,
                                 size_t num_sequences);
  size_t GetNumSignals() const { return signals_.size(); }
  const Signal &GetSignalAt(size_t i) const { return signals_[i]; }
  const char *GetSignalNameAt(size_t i) const { return signals_[i].id.data(); }
  size_t GetSignalLengthAt(size_t i) const { return signals_[i].GetSignalLength(); }

  // Raw mode: AddSignal keeps the ADC counts (2 bytes per sample) instead of converting on the host.
  // The samples of consecutive reads are packed into large arena blocks obtained from `alloc` (the
  // CLI passes the device library's pinned allocator, so a run of reads is one DMA); a read never
  // straddles two blocks.  Call before loading.
  typedef void *(*ArenaAlloc)(size_t bytes);
  typedef void (*ArenaFree)(void *p);
  // chunk_size > 0: every read also gets its filtered length and chunk_bounds (above) at load time.
  void KeepRaw(ArenaAlloc alloc, ArenaFree free_fn, size_t block_bytes = (size_t)256 << 20, uint32_t chunk_size = 0,
               uint32_t max_chunks = 0);
  bool raw_mode() const { return raw_mode_; }
  void SetFilteredLengthAt(size_t i, size_t n) { signals_[i].filtered_length = n; }
  // arena block b: base pointer (int16) and samples used
  const int16_t *RawBlockBase(uint32_t b) const { return blocks_[b].base; }
  ~SignalBatch();

 protected:
  std::string signal_directory_;
  std::vector<Signal> signals_;
  struct Block {
    int16_t *base;
    size_t cap, used;
  };
  bool raw_mode_ = false;
  ArenaAlloc alloc_ = nullptr;
  ArenaFree free_ = nullptr;
  size_t block_bytes_ = 0;
  uint32_t chunk_size_ = 0, max_chunks_ = 0;
  std::vector<Block> blocks_;
};

// (x - mean) / SD with mean and SD = sqrt(sum((x-mean)^2)/(n-1)) accumulated in double in
// index order, narrowed to float (Sigmap::GenerateZscoreNormalizedSignal, sigmap.cc:1131-1155).
float ZscoreNormalize(const float *x, size_t n, std::vector<float> &out);
}  // namespace sigmap
#endif
