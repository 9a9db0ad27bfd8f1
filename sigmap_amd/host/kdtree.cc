// kdtree.cc -- parallel construction of the reference's k-d tree (divideTree / middleSplit_ /
// planeSplit), producing the SAME tree as nanoflann 1.3.2 does: node boxes, split values, the
// permutation `vind` and therefore the bytes of the .si file are the parity surface, and they depend
// on the exact order of the swaps of nanoflann's Hoare-style partition.  PlaneSplit and MiddleSplit
// below therefore follow nanoflann's planeSplit / middleSplit_ step for step
// (reference: src/nanoflann.hpp:858-1004).
//
// nanoflann is Copyright 2008-2009 Marius Muja, 2008-2009 David G. Lowe, 2011-2016 Jose Luis Blanco,
// distributed under the BSD license (the licence text is in the header of nanoflann.hpp); the two
// routines named above are restatements of that algorithm, everything else here (the task-parallel
// build, the flat pre-order node array) is this project's own.
#include "kdtree.h"

#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace sigmap {
namespace {
constexpr int kMaxDim = 16;
constexpr uint64_t kParallelCutoff = 1u << 16;  // subtrees smaller than this build serially

inline uint32_t F2U(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
inline float U2F(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct Builder {
  const float *values;
  int dim;
  uint64_t leaf_max;
  uint32_t *vind;

  inline float Pt(uint32_t idx, int d) const { return values[(uint64_t)idx + d]; }

  void MinMax(const uint32_t *ind, uint64_t count, int el, float &mn, float &mx) const {
    mn = mx = Pt(ind[0], el);
    for (uint64_t i = 1; i < count; ++i) {
      const float v = Pt(ind[i], el);
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
  }

  // Three-way partition around cutval: [< cutval | == cutval | > cutval]; lim1/lim2 are the
  // two boundaries.  Hoare-style sweeps with an unsigned right cursor that never passes 0.
  void PlaneSplit(uint32_t *ind, uint64_t count, int cutfeat, float cutval, uint64_t &lim1,
                  uint64_t &lim2) const {
    uint64_t left = 0, right = count - 1;
    for (;;) {
      while (left <= right && Pt(ind[left], cutfeat) < cutval) ++left;
      while (right && left <= right && Pt(ind[right], cutfeat) >= cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left;
      --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && Pt(ind[left], cutfeat) <= cutval) ++left;
      while (right && left <= right && Pt(ind[right], cutfeat) > cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left;
      --right;
    }
    lim2 = left;
  }

  // Widest-spread dimension among those within (1-1e-5) of the widest bbox span; cut at the
  // bbox midpoint clamped to the data; split index pulled toward count/2.
  void MiddleSplit(uint32_t *ind, uint64_t count, uint64_t &index, int &cutfeat, float &cutval,
                   const float *lo, const float *hi) const {
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int i = 1; i < dim; ++i) {
      const float span = hi[i] - lo[i];
      if (span > max_span) max_span = span;
    }
    float max_spread = -1;
    cutfeat = 0;
    float cmn = 0, cmx = 0;
    bool have_minmax = false;
    for (int i = 0; i < dim; ++i) {
      const float span = hi[i] - lo[i];
      if (span > (1 - EPS) * max_span) {
        float mn, mx;
        MinMax(ind, count, i, mn, mx);
        const float spread = mx - mn;
        if (spread > max_spread) {
          cutfeat = i;
          max_spread = spread;
          cmn = mn;
          cmx = mx;
          have_minmax = true;
        }
      }
    }
    if (!have_minmax) MinMax(ind, count, cutfeat, cmn, cmx);  // degenerate box: dimension 0
    const float split_val = (lo[cutfeat] + hi[cutfeat]) / 2;
    if (split_val < cmn)
      cutval = cmn;
    else if (split_val > cmx)
      cutval = cmx;
    else
      cutval = split_val;
    uint64_t lim1, lim2;
    PlaneSplit(ind, count, cutfeat, cutval, lim1, lim2);
    if (lim1 > count / 2)
      index = lim1;
    else if (lim2 < count / 2)
      index = lim2;
    else
      index = count / 2;
  }

  void LeafBox(uint64_t left, uint64_t right, float *lo, float *hi) const {
    for (int i = 0; i < dim; ++i) lo[i] = hi[i] = Pt(vind[left], i);
    for (uint64_t k = left + 1; k < right; ++k)
      for (int i = 0; i < dim; ++i) {
        const float v = Pt(vind[k], i);
        if (lo[i] > v) lo[i] = v;
        if (hi[i] < v) hi[i] = v;
      }
  }

  // Appends the subtree over vind[left,right) to `out` in pre-order.  child2 is stored
  // RELATIVE to the node's own slot while building (made absolute at the end), so subtree
  // vectors can be concatenated without patching.  lo/hi: bounding box, in-out.
  void BuildSerial(uint64_t left, uint64_t right, float *lo, float *hi,
                   std::vector<smhip_node> &out) const {
    const size_t self = out.size();
    out.push_back(smhip_node{});
    if ((right - left) <= leaf_max) {
      out[self] = smhip_node{-1, (uint32_t)left, (uint32_t)right, 0};
      LeafBox(left, right, lo, hi);
      return;
    }
    uint64_t idx;
    int cutfeat;
    float cutval;
    MiddleSplit(vind + left, right - left, idx, cutfeat, cutval, lo, hi);
    float llo[kMaxDim], lhi[kMaxDim], rlo[kMaxDim], rhi[kMaxDim];
    memcpy(llo, lo, sizeof(float) * dim);
    memcpy(lhi, hi, sizeof(float) * dim);
    memcpy(rlo, lo, sizeof(float) * dim);
    memcpy(rhi, hi, sizeof(float) * dim);
    lhi[cutfeat] = cutval;
    rlo[cutfeat] = cutval;
    BuildSerial(left, left + idx, llo, lhi, out);
    const uint32_t rel_child2 = (uint32_t)(out.size() - self);
    BuildSerial(left + idx, right, rlo, rhi, out);
    out[self] = smhip_node{cutfeat, F2U(lhi[cutfeat]), F2U(rlo[cutfeat]), rel_child2};
    for (int i = 0; i < dim; ++i) {
      lo[i] = std::min(llo[i], rlo[i]);
      hi[i] = std::max(lhi[i], rhi[i]);
    }
  }

  void BuildParallel(uint64_t left, uint64_t right, float *lo, float *hi,
                     std::vector<smhip_node> &out) const {
    if (right - left < kParallelCutoff) {
      BuildSerial(left, right, lo, hi, out);
      return;
    }
    uint64_t idx;
    int cutfeat;
    float cutval;
    MiddleSplit(vind + left, right - left, idx, cutfeat, cutval, lo, hi);
    float llo[kMaxDim], lhi[kMaxDim], rlo[kMaxDim], rhi[kMaxDim];
    memcpy(llo, lo, sizeof(float) * dim);
    memcpy(lhi, hi, sizeof(float) * dim);
    memcpy(rlo, lo, sizeof(float) * dim);
    memcpy(rhi, hi, sizeof(float) * dim);
    lhi[cutfeat] = cutval;
    rlo[cutfeat] = cutval;
    std::vector<smhip_node> lv, rv;
#pragma omp task shared(lv, llo, lhi) firstprivate(left, idx)
    BuildParallel(left, left + idx, llo, lhi, lv);
#pragma omp task shared(rv, rlo, rhi) firstprivate(left, right, idx)
    BuildParallel(left + idx, right, rlo, rhi, rv);
#pragma omp taskwait
    out.reserve(out.size() + 1 + lv.size() + rv.size());
    out.push_back(smhip_node{cutfeat, F2U(lhi[cutfeat]), F2U(rlo[cutfeat]), (uint32_t)(1 + lv.size())});
    out.insert(out.end(), lv.begin(), lv.end());
    out.insert(out.end(), rv.begin(), rv.end());
    for (int i = 0; i < dim; ++i) {
      lo[i] = std::min(llo[i], rlo[i]);
      hi[i] = std::max(lhi[i], rhi[i]);
    }
  }
};
}  // namespace

void FlatKdTree::Build(const float *values, uint64_t n_values, int dim_, int leaf_max_,
                       int num_threads) {
  if (dim_ < 1 || dim_ > kMaxDim) throw std::runtime_error("FlatKdTree: unsupported dimension");
  if (n_values < (uint64_t)dim_) throw std::runtime_error("FlatKdTree: fewer values than dimensions");
  dim = dim_;
  leaf_max = (uint64_t)leaf_max_;
  m_size = n_values - (uint64_t)dim + 1;
  if (m_size >= 0xFFFFFFFFull) throw std::runtime_error("FlatKdTree: more than 2^32-1 points");
  vind.resize(m_size);
  for (uint64_t i = 0; i < m_size; ++i) vind[i] = (uint32_t)i;
  std::vector<float> lo(dim), hi(dim);
  for (int d = 0; d < dim; ++d) lo[d] = hi[d] = values[d];
  // root bounding box over every window (parallel reduction: min/max are order-free)
#pragma omp parallel num_threads(num_threads)
  {
    float tlo[kMaxDim], thi[kMaxDim];
    for (int d = 0; d < dim; ++d) tlo[d] = thi[d] = values[d];
#pragma omp for schedule(static)
    for (uint64_t k = 1; k < m_size; ++k)
      for (int d = 0; d < dim; ++d) {
        const float v = values[k + d];
        if (v < tlo[d]) tlo[d] = v;
        if (v > thi[d]) thi[d] = v;
      }
#pragma omp critical
    for (int d = 0; d < dim; ++d) {
      if (tlo[d] < lo[d]) lo[d] = tlo[d];
      if (thi[d] > hi[d]) hi[d] = thi[d];
    }
  }
  nodes.clear();
  Builder b{values, dim, leaf_max, vind.data()};
#pragma omp parallel num_threads(num_threads)
#pragma omp single
  b.BuildParallel(0, m_size, lo.data(), hi.data(), nodes);
  for (size_t i = 0; i < nodes.size(); ++i)
    if (nodes[i].feat >= 0) nodes[i].child2 += (uint32_t)i;  // relative -> absolute
  bbox.resize(2 * dim);
  for (int d = 0; d < dim; ++d) {
    bbox[2 * d] = lo[d];
    bbox[2 * d + 1] = hi[d];
  }
  depth = ComputeDepth();
}

int FlatKdTree::ComputeDepth() const {
  // iterative pre-order walk: depth of node i = depth(parent) + 1
  if (nodes.empty()) return 0;
  std::vector<std::pair<uint32_t, int>> stack;  // (child2 index, depth) pending
  int best = 0, d = 1;
  size_t i = 0;
  while (true) {
    best = std::max(best, d);
    if (nodes[i].feat >= 0) {
      stack.emplace_back(nodes[i].child2, d + 1);
      ++i;
      ++d;
    } else {
      if (stack.empty()) break;
      i = stack.back().first;
      d = stack.back().second;
      stack.pop_back();
    }
  }
  return best;
}

namespace {
#pragma pack(push, 1)
struct DiskNode {  // 32-byte on-disk node of the `.si` stream
  union {
    struct { uint64_t left, right; } lr;
    struct { int32_t divfeat; float divlow, divhigh; uint32_t unused; } sub;
  } u;
  uint64_t child1, child2;  // stale heap addresses in reference files; 0 <=> NULL
};
#pragma pack(pop)
static_assert(sizeof(DiskNode) == 32, "disk node must be 32 bytes");
}  // namespace

bool FlatKdTree::Save(const std::string &path) const {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const uint64_t m = m_size, cnt = (uint64_t)dim, lm = leaf_max;
  const int32_t d32 = dim;
  fwrite(&m, 8, 1, f);
  fwrite(&d32, 4, 1, f);
  fwrite(&cnt, 8, 1, f);
  fwrite(bbox.data(), 4, 2 * (size_t)dim, f);
  fwrite(&lm, 8, 1, f);
  fwrite(&m, 8, 1, f);
  {
    std::vector<uint64_t> buf(1 << 16);
    for (uint64_t i = 0; i < m; i += buf.size()) {
      const uint64_t n = std::min<uint64_t>(buf.size(), m - i);
      for (uint64_t k = 0; k < n; ++k) buf[k] = vind[i + k];
      fwrite(buf.data(), 8, n, f);
    }
  }
  for (size_t i = 0; i < nodes.size(); ++i) {
    DiskNode d;
    memset(&d, 0, sizeof d);
    if (nodes[i].feat < 0) {
      d.u.lr.left = nodes[i].a;
      d.u.lr.right = nodes[i].b;
    } else {
      d.u.sub.divfeat = nodes[i].feat;
      d.u.sub.divlow = U2F(nodes[i].a);
      d.u.sub.divhigh = U2F(nodes[i].b);
      d.child1 = i + 1;            // any non-NULL value: loaders only test for NULL
      d.child2 = nodes[i].child2;
    }
    fwrite(&d, sizeof d, 1, f);
  }
  return fclose(f) == 0;
}

bool FlatKdTree::Load(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  uint64_t m = 0, cnt = 0, lm = 0, vn = 0;
  int32_t d32 = 0;
  bool ok = fread(&m, 8, 1, f) == 1 && fread(&d32, 4, 1, f) == 1 && fread(&cnt, 8, 1, f) == 1;
  if (ok && (cnt != (uint64_t)d32 || d32 < 1 || d32 > kMaxDim)) ok = false;
  if (ok) {
    bbox.resize(2 * cnt);
    ok = fread(bbox.data(), 4, 2 * cnt, f) == 2 * cnt;
  }
  ok = ok && fread(&lm, 8, 1, f) == 1 && fread(&vn, 8, 1, f) == 1 && vn == m && m < 0xFFFFFFFFull;
  if (ok) {
    vind.resize(vn);
    std::vector<uint64_t> buf(1 << 16);
    for (uint64_t i = 0; i < vn && ok; i += buf.size()) {
      const uint64_t n = std::min<uint64_t>(buf.size(), vn - i);
      ok = fread(buf.data(), 8, n, f) == n;
      for (uint64_t k = 0; k < n; ++k) vind[i + k] = (uint32_t)buf[k];
    }
  }
  if (!ok) {
    fclose(f);
    return false;
  }
  dim = d32;
  leaf_max = lm;
  m_size = m;
  // The stream is a pre-order dump, so child1 = self+1 and child2 = first slot after
  // child1's subtree: one pass with a stack of inner nodes still waiting for child2.
  nodes.clear();
  std::vector<uint32_t> waiting;
  DiskNode d;
  long pending = 1;  // subtrees still to read
  while (pending > 0 && fread(&d, sizeof d, 1, f) == 1) {
    const uint32_t self = (uint32_t)nodes.size();
    --pending;
    if (d.child1 == 0 && d.child2 == 0) {
      nodes.push_back(smhip_node{-1, (uint32_t)d.u.lr.left, (uint32_t)d.u.lr.right, 0});
      // a finished subtree: if it was some node's child1, the next slot is that node's child2
      while (!waiting.empty()) {
        const uint32_t w = waiting.back();
        if (nodes[w].child2 == 0) {
          nodes[w].child2 = (uint32_t)nodes.size();
          break;
        }
        waiting.pop_back();
      }
    } else {
      nodes.push_back(smhip_node{d.u.sub.divfeat, F2U(d.u.sub.divlow), F2U(d.u.sub.divhigh), 0});
      waiting.push_back(self);
      pending += 2;
    }
  }
  fclose(f);
  if (pending != 0) return false;
  depth = ComputeDepth();
  return true;
}
}  // namespace sigmap
