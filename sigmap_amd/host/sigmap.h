// sigmap.h -- the reference's driver class (sigmap.h:23-120 of the reference) over the MI355X hot path.
// ConstructIndex / StreamingMap / DumpFeatures are defined in sigmap.cc (the command line, which links the device
// library directly); GenerateEvents -- the per-chunk support function of the reference's class surface -- lives in
// the host library (device_methods.cc) and reaches the device through the bridge of device_api.h.
#ifndef SIGMAP_AMD_SIGMAP_H_
#define SIGMAP_AMD_SIGMAP_H_
#include <cstddef>
#include <string>
#include <vector>

#include "mapping_logic.h"
#include "signal_batch.h"

struct smhip_ctx;

namespace sigmap {
class Sigmap {
 public:
  // for index construction
  Sigmap(int dimension, int max_leaf, const std::string &reference_file_path,
         const std::string &pore_model_file_path, const std::string &output_file_path)
      : dimension_(dimension), max_leaf_(max_leaf), reference_file_path_(reference_file_path),
        pore_model_file_path_(pore_model_file_path), output_file_path_(output_file_path) {}
  // for mapping
  Sigmap(const MappingParams &p, int num_threads, const std::string &reference_file_path,
         const std::string &pore_model_file_path, const std::string &signal_directory,
         const std::string &reference_index_file_path, const std::string &output_file_path)
      : params_(p), num_threads_(num_threads), reference_file_path_(reference_file_path),
        pore_model_file_path_(pore_model_file_path), signal_directory_(signal_directory),
        reference_index_file_path_(reference_index_file_path), output_file_path_(output_file_path) {}
  ~Sigmap();

  void ConstructIndex();
  // Pinned arena of the raw reads (SignalBatch::KeepRaw): a run of reads is one DMA.
  static void *PinnedAlloc(size_t bytes);
  static void PinnedFree(void *p);
  void StreamingMap(int device, int num_gpus, size_t max_batch_chunks, int sdtw_band, int num_streams, bool sdtw_cigar,
                    bool share_device, bool warm_up = true);
  static void DumpFeatures(const std::string &kind, int device, const std::string &signal_directory,
                           const std::string &output_file_path);

  // Sigmap::GenerateEvents(start, end, signal, feature_signal, feature_signal_stdvs), sigmap.h:76 / sigmap.cc:1048-1083
  // of the reference: event detection, z-score and 0.1-compression of signal.signal_values[start, end) on the GPU
  // (smhip_events with one chunk; at most 4096 samples -- the reference's callers pass its 4000-sample chunks).
  // Like the reference it CLEARS feature_signal and APPENDS to feature_signal_stdvs (the caller clears those,
  // sigmap.cc:650-651).  The device context: BindDevice's, or one created on GPU 0 at the first call; without a
  // usable GPU the process ends with a message (utils.h:67 behaviour) -- there is no CPU implementation.
  void GenerateEvents(size_t start, size_t end, const Signal &signal, std::vector<float> &feature_signal,
                      std::vector<float> &feature_signal_stdvs);
  void BindDevice(smhip_ctx *ctx) { ctx_ = ctx; }

 protected:
  int dimension_ = 6, max_leaf_ = 20;
  MappingParams params_;
  int num_threads_ = 1;
  std::string reference_file_path_, pore_model_file_path_, signal_directory_, reference_index_file_path_,
      output_file_path_;
  smhip_ctx *ctx_ = nullptr;      // device context of GenerateEvents
  bool owns_ctx_ = false;
};
}  // namespace sigmap
#endif
