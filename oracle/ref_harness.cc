// ref_harness.cc -- thin extern "C" shim around the REAL reference code.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it calls is compiled from
// the reference sources where they lie (/root/reference/src, passed as -I by
// oracle/Makefile) -- nothing from the reference is copied into this repository.
// Output goes to oracle/_ref/libsigmap_ref.so (git-ignored, travels with gpurun).
//
// Only translation units that compile with the headers really present in this image are
// used: event.h, nanoflann.hpp, sigmap_adaptor.h, pore_model.{h,cc} (+ utils.h, cwt.h,
// and the genuine HDF5 headers under /opt/conda/include that utils.h includes).
// spatial_index.cc / sigmap.cc / fast_dtw.cc include <slow5/slow5.h>, which does not
// exist here, so they are NOT built (no stand-in headers are written).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cwt.h"
#include "event.h"
#include "nanoflann.hpp"
#include "pore_model.h"
#include "sigmap_adaptor.h"

using sigmap::Event;

extern "C" {

// event.h:226 DetectEvents with event_detection_defaults; dumps every intermediate.
// Arrays must hold n+1 entries.  Returns number of events (or -1 when the reference
// would run out of bounds: fewer than 2 peaks, event.h:213-223).
long ref_detect_events(const float *x, size_t n, float *ps, float *pss, float *t1, float *t2,
                       uint64_t *peaks, size_t *n_peaks, uint64_t *ev_start,
                       uint64_t *ev_length, float *ev_mean, float *ev_stdv) {
  std::vector<float> vps, vpss, vt1, vt2;
  std::vector<size_t> vpeaks;
  std::vector<Event> events;
  vps.reserve(n + 1);
  vpss.reserve(n + 1);
  sigmap::ComputePrefixSumAndPrefixSumSquares(x, n, vps, vpss);
  sigmap::ComputeTStat(vps.data(), vpss.data(), n, sigmap::event_detection_defaults.window_length1, vt1);
  sigmap::ComputeTStat(vps.data(), vpss.data(), n, sigmap::event_detection_defaults.window_length2, vt2);
  sigmap::Detector sd = {-1, FLT_MAX, vt1.data(), n, sigmap::event_detection_defaults.threshold1,
                         sigmap::event_detection_defaults.window_length1, 0, -1, FLT_MAX, false};
  sigmap::Detector ld = {-1, FLT_MAX, vt2.data(), n, sigmap::event_detection_defaults.threshold2,
                         sigmap::event_detection_defaults.window_length2, 0, -1, FLT_MAX, false};
  sigmap::GeneratePeaksUsingMultiWindows(&sd, &ld, sigmap::event_detection_defaults.peak_height, vpeaks);
  if (ps) memcpy(ps, vps.data(), vps.size() * 4);
  if (pss) memcpy(pss, vpss.data(), vpss.size() * 4);
  if (t1) memcpy(t1, vt1.data(), vt1.size() * 4);
  if (t2) memcpy(t2, vt2.data(), vt2.size() * 4);
  *n_peaks = vpeaks.size();
  for (size_t i = 0; i < vpeaks.size(); ++i) peaks[i] = vpeaks[i];
  if (vpeaks.size() < 2) return -1;
  sigmap::CreateEvents(vpeaks.data(), vpeaks.size(), vps.data(), vpss.data(), n, events);
  for (size_t i = 0; i < events.size(); ++i) {
    ev_start[i] = events[i].start;
    ev_length[i] = events[i].length;
    ev_mean[i] = events[i].mean;
    ev_stdv[i] = events[i].stdv;
  }
  return (long)events.size();
}

// The whole thing through the public entry point, for a second check.
long ref_detect_events_entry(const float *x, size_t n, float *ev_mean, float *ev_stdv) {
  std::vector<float> ps, pss, t1, t2;
  std::vector<size_t> peaks;
  std::vector<Event> events;
  // guard the reference's out-of-bounds case first
  {
    uint64_t *pk = new uint64_t[n + 1];
    uint64_t *es = new uint64_t[n + 1], *el = new uint64_t[n + 1];
    float *em = new float[n + 1], *sd = new float[n + 1];
    size_t np = 0;
    long r = ref_detect_events(x, n, nullptr, nullptr, nullptr, nullptr, pk, &np, es, el, em, sd);
    delete[] pk; delete[] es; delete[] el; delete[] em; delete[] sd;
    if (r < 0) return -1;
  }
  sigmap::DetectEvents(x, n, sigmap::event_detection_defaults, ps, pss, t1, t2, peaks, events);
  for (size_t i = 0; i < events.size(); ++i) {
    ev_mean[i] = events[i].mean;
    ev_stdv[i] = events[i].stdv;
  }
  return (long)events.size();
}

struct RefTree {
  std::vector<Point> cloud;
  SigmapAdaptor<float> *adaptor;
};
typedef SigmapAdaptor<float>::index_t index_t;

// spatial_index.cc:95-97: SigmapAdaptor + buildIndex
void *ref_kdtree_build(const float *values, const uint64_t *positions, size_t n, int dim,
                       int max_leaf) {
  RefTree *t = new RefTree;
  t->cloud.reserve(n);
  for (size_t i = 0; i < n; ++i) t->cloud.emplace_back(positions ? positions[i] : 0, values[i]);
  t->adaptor = new SigmapAdaptor<float>(dim, t->cloud, max_leaf);
  t->adaptor->index->buildIndex();
  return t;
}

// spatial_index.cc:124-128
int ref_kdtree_save(void *h, const char *path) {
  RefTree *t = (RefTree *)h;
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  t->adaptor->index->saveIndex(f);
  fclose(f);
  return 0;
}

// spatial_index.cc:150-155
void *ref_kdtree_load(const char *path, const float *values, const uint64_t *positions,
                      size_t n, int dim, int max_leaf) {
  RefTree *t = new RefTree;
  t->cloud.reserve(n);
  for (size_t i = 0; i < n; ++i) t->cloud.emplace_back(positions ? positions[i] : 0, values[i]);
  FILE *f = fopen(path, "rb");
  if (!f) { delete t; return nullptr; }
  t->adaptor = new SigmapAdaptor<float>(dim, t->cloud, max_leaf);
  t->adaptor->index->loadIndex(f);
  fclose(f);
  return t;
}

void ref_kdtree_free(void *h) {
  RefTree *t = (RefTree *)h;
  delete t->adaptor;
  delete t;
}

// spatial_index.cc:324,367-368: radiusSearch with params.sorted = false
size_t ref_radius_search(void *h, const float *q, float radius, uint64_t *idx, float *dist,
                         size_t cap) {
  RefTree *t = (RefTree *)h;
  nanoflann::SearchParams params;
  params.sorted = false;
  std::vector<std::pair<size_t, float> > res;
  size_t n = t->adaptor->index->radiusSearch(q, radius, res, params);
  for (size_t i = 0; i < n && i < cap; ++i) {
    idx[i] = res[i].first;
    dist[i] = res[i].second;
  }
  return n;
}

static void dump_rec(index_t::Node *node, std::vector<int32_t> &is_leaf,
                     std::vector<uint64_t> &a, std::vector<uint64_t> &b, std::vector<float> &lo,
                     std::vector<float> &hi) {
  bool leaf = node->child1 == NULL && node->child2 == NULL;
  is_leaf.push_back(leaf ? 1 : 0);
  if (leaf) {
    a.push_back(node->node_type.lr.left);
    b.push_back(node->node_type.lr.right);
    lo.push_back(0);
    hi.push_back(0);
  } else {
    a.push_back((uint64_t)node->node_type.sub.divfeat);
    b.push_back(0);
    lo.push_back(node->node_type.sub.divlow);
    hi.push_back(node->node_type.sub.divhigh);
    dump_rec(node->child1, is_leaf, a, b, lo, hi);
    dump_rec(node->child2, is_leaf, a, b, lo, hi);
  }
}

// Pre-order dump for semantic tree comparison.  Call with NULL arrays to get the count.
size_t ref_kdtree_dump(void *h, int32_t *is_leaf, uint64_t *a, uint64_t *b, float *lo,
                       float *hi, uint64_t *vind, float *bbox /*2*dim: low,high pairs*/) {
  RefTree *t = (RefTree *)h;
  index_t *ix = t->adaptor->index;
  std::vector<int32_t> vl;
  std::vector<uint64_t> va, vb;
  std::vector<float> vlo, vhi;
  dump_rec(ix->root_node, vl, va, vb, vlo, vhi);
  if (is_leaf) {
    memcpy(is_leaf, vl.data(), vl.size() * 4);
    memcpy(a, va.data(), va.size() * 8);
    memcpy(b, vb.data(), vb.size() * 8);
    memcpy(lo, vlo.data(), vlo.size() * 4);
    memcpy(hi, vhi.data(), vhi.size() * 4);
    for (size_t i = 0; i < ix->vind.size(); ++i) vind[i] = ix->vind[i];
    for (int d = 0; d < ix->dim; ++d) {
      bbox[2 * d] = ix->root_bbox[d].low;
      bbox[2 * d + 1] = ix->root_bbox[d].high;
    }
  }
  return vl.size();
}

// pore_model.cc:11-45 Load + :57-81 GetLevelMeansAt
long ref_level_means(const char *model_path, const char *seq, uint32_t start, uint32_t end,
                     float *out, int *kmer_size) {
  sigmap::PoreModel pm;
  pm.Load(model_path);
  *kmer_size = pm.GetKmerSize();
  std::vector<float> v = pm.GetLevelMeansAt(seq, start, end);
  memcpy(out, v.data(), v.size() * 4);
  return (long)v.size();
}

// sigmap.cc:1157-1174 GenerateCWTSignal's calls into cwt.cc (cwt_init :1633, setCWTScales :1711,
// cwt :1735, cwavelet :1762, DoG daughter :1595-1630): one-scale DoG (m = 2) CWT, real part.
void ref_cwt_dog(const float *x, size_t n, float scale0, float *out) {
  char wave[] = "dog";
  char type[] = "pow";
  sigmap::cwt_object wt = sigmap::cwt_init(wave, 2.0, (int)n, 1, 1);
  sigmap::setCWTScales(wt, scale0, 1, type, 2.0);
  sigmap::cwt(wt, x);
  for (size_t i = 0; i < n; ++i) out[i] = wt->output[i].re;
  sigmap::cwt_free(wt);
}
}
