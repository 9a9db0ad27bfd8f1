// kdtree.h -- host-side k-d tree over the sliding-window point cloud.
//
// Produces exactly the tree nanoflann 1.3.2 builds for SigmapAdaptor<float> (reference:
// nanoflann.hpp:858-1004 divideTree/middleSplit_/planeSplit, :1196-1210 buildIndex,
// sigmap_adaptor.h:89-97), but (a) directly in the flattened pre-order 16-byte node layout
// the GPU walks (include/sigmap_hip.h smhip_node), (b) with subtrees built in parallel
// (OpenMP tasks; subtrees own disjoint ranges of the permutation array, so the result is
// identical to the sequential build), and (c) with 32-bit point indices.
// Reads and writes the reference's `.si` format (nanoflann.hpp:1025-1072).
#ifndef SIGMAP_AMD_KDTREE_H_
#define SIGMAP_AMD_KDTREE_H_
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sigmap_hip.h"

namespace sigmap {
struct FlatKdTree {
  int dim = 0;
  uint64_t leaf_max = 0;
  uint64_t m_size = 0;               // number of k-d points = n_values - dim + 1
  std::vector<float> bbox;           // low0, high0, low1, high1, ...
  std::vector<uint32_t> vind;        // tree order -> point index
  std::vector<smhip_node> nodes;     // pre-order; child1 = self + 1
  int depth = 0;

  // values: the 1-D point-cloud value array; k-d point i = values[i .. i+dim)
  void Build(const float *values, uint64_t n_values, int dim, int leaf_max, int num_threads);
  bool Save(const std::string &path) const;  // reference `.si`
  bool Load(const std::string &path);        // reference `.si`
  int ComputeDepth() const;
};
}  // namespace sigmap
#endif
