// spatial_index.h -- host side of the reference point-cloud index (API of the reference's
// SpatialIndex, spatial_index.h:47-117): construction, `.pt`/`.si` files, and the bridge
// to the device-resident copy.  The query side (GenerateChains, spatial_index.cc:276-577)
// runs on the GPU behind include/sigmap_hip.h -- batched in the command line (sigmap.cc), and as the
// reference's own per-query method below (device_methods.cc: a one-chunk forward); there is no CPU
// implementation of it in the product.
#ifndef SIGMAP_AMD_SPATIALINDEX_H_
#define SIGMAP_AMD_SPATIALINDEX_H_
#include <cstdint>
#include <string>
#include <vector>

#include "kdtree.h"

struct smhip_ctx;

namespace sigmap {
enum Direction { Negative = 0, Positive = 1 };

struct SignalAnchor {
  uint32_t target_position, query_position;
  float distance;
};

struct SignalAnchorChain {
  float score;
  uint32_t reference_sequence_index, start_position, end_position, num_anchors;
  uint8_t mapq;
  Direction direction;
  std::vector<SignalAnchor> anchors;  // chain end -> chain start
};

class SpatialIndex {
 public:
  // for read mapping (same signature as the reference's mapping constructor)
  SpatialIndex(int min_num_seeds_required_for_mapping, const std::vector<int> &max_seed_frequencies,
               const std::string &index_file_path_prefix)
      : index_file_path_prefix_(index_file_path_prefix) {
    (void)min_num_seeds_required_for_mapping;
    (void)max_seed_frequencies;
  }
  // for index construction
  SpatialIndex(int dimension, int max_leaf, int num_threads, const std::string &index_file_path_prefix)
      : dimension_(dimension), max_leaf_(max_leaf), num_threads_(num_threads),
        index_file_path_prefix_(index_file_path_prefix) {}

  // Appends the points of one strand of one reference signal: position p is kept iff it is
  // unmasked and (p == 0 || cloud empty || |v[p] - last kept value| > 0.01); key layout
  // ((ref << 32 | p) << 1) | strand, strand 0 = +.
  void GeneratePointCloudOnOneDirection(Direction direction, uint32_t signal_index,
                                        const std::vector<std::vector<bool>> &is_masked,
                                        const float *signal_values, size_t signal_length,
                                        int step_size);
  void Construct(size_t num_signals, const std::vector<std::vector<bool>> &positive_is_masked,
                 const std::vector<std::vector<bool>> &negative_is_masked,
                 const std::vector<std::vector<float>> &positive_signals,
                 const std::vector<std::vector<float>> &negative_signals);
  void Save();  // PREFIX.pt + PREFIX.si, reference formats
  void Load();
  ~SpatialIndex();

  // ---- query side, the reference's signature (spatial_index.h:87-92): seeds of query_signal -> radius search ->
  // anchors (+ the anchors of the chains handed in) -> sort -> chaining; `chains`: in = the previous chunk's chains,
  // out = the new primary chains with their anchors (end -> start) and chain 0's MAPQ.  Runs on the GPU
  // (smhip_generate_chains with one chunk).  BindDevice uploads this index to `ctx` (null: a context of its own on
  // GPU 0, which the first GenerateChains also does by itself).  query_signal_stdvs is unused, as in the reference.
  // Errors end the process with a message (utils.h:67 behaviour of the reference).
  void BindDevice(smhip_ctx *ctx);
  void GenerateChains(const std::vector<float> &query_signal, const std::vector<float> &query_signal_stdvs,
                      uint32_t query_start_offset, int query_point_cloud_step_size, float search_radius,
                      size_t num_target_signals, std::vector<SignalAnchorChain> &chains);

  void set_index_file_path_prefix(const std::string &p) { index_file_path_prefix_ = p; }
  int dimension() const { return dimension_; }
  int max_leaf() const { return max_leaf_; }
  const std::vector<float> &values() const { return values_; }
  const std::vector<uint64_t> &positions() const { return positions_; }
  const FlatKdTree &tree() const { return tree_; }
  // Set the point cloud directly (tests / tools).
  void SetPointCloud(std::vector<uint64_t> positions, std::vector<float> values);
  void BuildTree();

 protected:
  int dimension_ = 6;
  int max_leaf_ = 20;
  int num_threads_ = 1;
  std::string index_file_path_prefix_;
  std::vector<uint64_t> positions_;  // Point.position
  std::vector<float> values_;        // Point.value
  FlatKdTree tree_;
  smhip_ctx *ctx_ = nullptr;         // device context the query side runs on (BindDevice)
  bool owns_ctx_ = false;
};

// 11-mer style high-frequency mask (Sigmap::GenerateMaskedPositions, sigmap.cc:19-185):
// canonical k-mers with count/total > frequency are masked at their start position; the
// window ENDING at an ambiguous base is masked too.
void GenerateMaskedPositions(int kmer_size, float frequency, const class SequenceBatch &sequence_batch,
                             std::vector<std::vector<bool>> &positive_is_masked,
                             std::vector<std::vector<bool>> &negative_is_masked);
}  // namespace sigmap
#endif
