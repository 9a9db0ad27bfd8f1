// ref_harness.cc -- thin extern "C" shim around the REAL reference code.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it calls is compiled from
// the reference sources where they lie (/root/reference/src, passed as -I by
// oracle/Makefile) -- nothing from the reference is copied into this repository.
// Output goes to oracle/_ref/libsigmap_ref.so (git-ignored, travels with gpurun).
//
// Only translation units that compile with the headers really present in this image are
// used: event.h, nanoflann.hpp, sigmap_adaptor.h, pore_model.{h,cc}, cwt.{h,cc},
// output_tools.h + sequence_batch.{h,cc} (+ kseq.h, utils.h, zlib, and the genuine HDF5
// headers under /opt/conda/include that utils.h includes).
// spatial_index.cc / sigmap.cc / signal_batch.cc / fast_dtw.cc include <slow5/slow5.h>
// (through signal_batch.h:6), which does not exist here, so they are NOT built (no stand-in
// headers are written).  output_tools.h only reaches sequence_batch.h (output_tools.h:12),
// which needs no slow5: the PAF text and the FASTA loader ARE built (round 4).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cwt.h"
#include "event.h"
#include "nanoflann.hpp"
#include "output_tools.h"
#include "pore_model.h"
#include "sequence_batch.h"
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

// ---- sequence_batch.{h,cc}: the reference's own FASTA / FASTQ loader (kseq over zlib) as the driver
// uses it, sigmap.cc:559-568: InitializeLoading, LoadAllSequences, PrepareNegativeSequenceAt per contig.
struct RefSeqs {
  sigmap::SequenceBatch batch;
  uint32_t n;
};
void *ref_fasta_load(const char *path) {
  RefSeqs *r = new RefSeqs;
  r->batch.InitializeLoading(path);
  r->n = r->batch.LoadAllSequences();
  for (uint32_t i = 0; i < r->n; ++i) r->batch.PrepareNegativeSequenceAt(i);
  r->batch.FinalizeLoading();
  return r;
}
uint32_t ref_fasta_count(void *h) { return ((RefSeqs *)h)->n; }
uint64_t ref_fasta_num_bases(void *h) { return ((RefSeqs *)h)->batch.GetNumBases(); }
// name / bases / reverse complement of contig i (pointers into the batch: valid until ref_fasta_free)
int ref_fasta_get(void *h, uint32_t i, const char **name, const char **seq, uint32_t *len, const char **neg) {
  RefSeqs *r = (RefSeqs *)h;
  if (i >= r->n) return -1;
  *name = r->batch.GetSequenceNameAt(i);
  *seq = r->batch.GetSequenceAt(i);
  *len = r->batch.GetSequenceLengthAt(i);
  *neg = r->batch.GetNegativeSequenceAt(i).c_str();
  return 0;
}
void ref_fasta_free(void *h) { delete (RefSeqs *)h; }

// ---- output_tools.h: one PAFMapping (the record sigmap.cc:745-768 / :862-865 emplace) written by the real
// PAFOutputTools<PAFMapping> into `tmp_path`, read back into out.  mapped != 0: AppendMapping
// (output_tools.h:336-354), else AppendUnmappedRead (:200-210) -- the choice OutputMappingsInVector makes
// (sigmap.cc:205-213) is the caller's.  mapq goes through the record's 6-bit field, direction through its 1-bit
// field (output_tools.h:26).  Returns the line's length, or -1.
static long paf_line(void *fasta, int mapped, uint32_t rid, uint32_t read_id, const char *read_name,
                     uint32_t read_length, uint32_t read_start, uint32_t read_end, uint32_t frag_start,
                     uint32_t frag_length, uint32_t mapq, uint32_t direction, const char *tags, const char *tmp_path,
                     char *out, size_t cap) {
  RefSeqs *r = (RefSeqs *)fasta;
  sigmap::PAFOutputTools<sigmap::PAFMapping> tools;
  sigmap::OutputTools<sigmap::PAFMapping> &base = tools;  // (AppendMapping is private in the specialisation, public here)
  base.InitializeMappingOutput(tmp_path);
  sigmap::PAFMapping m{read_id, std::string(read_name), read_length, read_start, read_end, frag_start, frag_length,
                       (uint8_t)mapq, (uint8_t)direction, (uint8_t)1, std::string(tags)};
  if (mapped) base.AppendMapping(rid, r->batch, m);
  else base.AppendUnmappedRead(rid, r->batch, m);
  base.FinalizeMappingOutput();
  FILE *f = fopen(tmp_path, "rb");
  if (!f) return -1;
  const size_t n = fread(out, 1, cap - 1, f);
  fclose(f);
  out[n] = 0;
  return (long)n;
}
long ref_paf_mapped(void *fasta, uint32_t rid, uint32_t read_id, const char *read_name, uint32_t read_length,
                    uint32_t read_start, uint32_t read_end, uint32_t frag_start, uint32_t frag_length, uint32_t mapq,
                    uint32_t direction, const char *tags, const char *tmp_path, char *out, size_t cap) {
  return paf_line(fasta, 1, rid, read_id, read_name, read_length, read_start, read_end, frag_start, frag_length, mapq,
                  direction, tags, tmp_path, out, cap);
}
long ref_paf_unmapped(void *fasta, uint32_t rid, uint32_t read_id, const char *read_name, uint32_t read_length,
                      uint32_t mapq, const char *tags, const char *tmp_path, char *out, size_t cap) {
  return paf_line(fasta, 0, rid, read_id, read_name, read_length, 0, 0, 0, 0, mapq, 0, tags, tmp_path, out, cap);
}
// The tag values are std::to_string(float / double / integer) in sigmap.cc:730-741: libstdc++'s own "%f"
// (nan / inf / -nan included), exposed so that the goldens' tag strings are composed by the same function.
int ref_to_string_float(float v, char *out, size_t cap) { return snprintf(out, cap, "%s", std::to_string(v).c_str()); }
int ref_to_string_double(double v, char *out, size_t cap) { return snprintf(out, cap, "%s", std::to_string(v).c_str()); }
}
