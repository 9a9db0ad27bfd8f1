// device_methods.cc -- the per-chunk methods of the reference's class surface as single-chunk
// forwards to the batched device entry points: SpatialIndex::GenerateChains (spatial_index.h:87-92 of the
// reference) -> smhip_generate_chains, Sigmap::GenerateEvents (sigmap.h:76) -> smhip_events.  A maintainer who keeps
// the reference's per-read loop binds these; the batched path of sigmap.cc does not go through them.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "device_api.h"
#include "sigmap.h"
#include "spatial_index.h"
#include "utils.h"

namespace sigmap {
namespace {
std::string OwnDirectory() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&OwnDirectory), &info) && info.dli_fname) {
    const std::string path = info.dli_fname;
    const size_t slash = path.rfind('/');
    if (slash != std::string::npos) return path.substr(0, slash + 1);
  }
  return "";
}
}  // namespace

const DeviceApi &Device() {
  static DeviceApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // a process that already holds the device library (the command line links it) gets that copy
    void *h = dlopen("libsigmap_hip.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen((OwnDirectory() + "libsigmap_hip.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libsigmap_hip.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) ExitWithMessage(std::string("Cannot load libsigmap_hip.so (this build has no CPU mapping path): ") + dlerror());
    auto need = [&](const char *name) {
      void *p = dlsym(h, name);
      if (!p) ExitWithMessage(std::string("libsigmap_hip.so lacks ") + name);
      return p;
    };
#define SM_BIND(field, name) api.field = reinterpret_cast<decltype(api.field)>(need(#name))
    SM_BIND(init, smhip_init);
    SM_BIND(destroy, smhip_destroy);
    SM_BIND(last_error, smhip_last_error);
    SM_BIND(default_params, smhip_default_params);
    SM_BIND(index_upload, smhip_index_upload);
    SM_BIND(generate_chains, smhip_generate_chains);
    SM_BIND(map_download, smhip_map_download);
    SM_BIND(map_download_chains, smhip_map_download_chains);
    SM_BIND(events, smhip_events);
#undef SM_BIND
  });
  return api;
}

smhip_ctx *NewDeviceContext(int device_ordinal) {
  smhip_ctx *ctx = nullptr;
  if (Device().init(device_ordinal, &ctx) != SMHIP_OK || !ctx)
    ExitWithMessage("No usable MI355X GPU (smhip_init failed): this build has no CPU mapping path.");
  return ctx;
}

// ------------------------------------------------------------------ SpatialIndex: the query side
SpatialIndex::~SpatialIndex() {
  if (owns_ctx_ && ctx_) Device().destroy(ctx_);
}

void SpatialIndex::BindDevice(smhip_ctx *ctx) {
  if (owns_ctx_ && ctx_) Device().destroy(ctx_);
  ctx_ = ctx;
  owns_ctx_ = false;
  if (!ctx_) {
    ctx_ = NewDeviceContext(0);
    owns_ctx_ = true;
  }
  const int rc = Device().index_upload(ctx_, dimension_, values_.data(), positions_.data(), values_.size(), tree_.vind.data(),
                                       tree_.m_size, tree_.nodes.data(), tree_.nodes.size(), tree_.bbox.data());
  if (rc != SMHIP_OK) ExitWithMessage(std::string("Index upload failed: ") + Device().last_error(ctx_));
}

void SpatialIndex::GenerateChains(const std::vector<float> &query_signal, const std::vector<float> &query_signal_stdvs,
                                  uint32_t query_start_offset, int query_point_cloud_step_size, float search_radius,
                                  size_t num_target_signals, std::vector<SignalAnchorChain> &chains) {
  (void)query_signal_stdvs;  // (unused by the reference as well, spatial_index.cc:276-577)
  if (!ctx_) BindDevice(nullptr);
  const DeviceApi &d = Device();
  smhip_params p;
  d.default_params(&p);
  p.step_size = query_point_cloud_step_size;
  p.search_radius = search_radius;
  // the chains handed in: their anchors go into the buckets first (spatial_index.cc:301-322)
  std::vector<smhip_carried_anchor> carried;
  for (const SignalAnchorChain &c : chains)
    for (const SignalAnchor &a : c.anchors)
      carried.push_back(smhip_carried_anchor{(c.reference_sequence_index << 1) | (c.direction == Positive ? 0u : 1u),
                                             a.target_position, a.query_position, a.distance});
  const uint32_t feat_off[2] = {0, (uint32_t)query_signal.size()}, car_off[2] = {0, (uint32_t)carried.size()};
  const float none = 0.0f;
  smhip_chunk_result res;
  int rc = d.generate_chains(ctx_, &p, (uint32_t)num_target_signals, query_signal.empty() ? &none : query_signal.data(), feat_off, 1,
                             &query_start_offset, carried.empty() ? nullptr : carried.data(), carried.empty() ? nullptr : car_off,
                             &res);
  if (rc != SMHIP_OK) ExitWithMessage(std::string("GenerateChains failed on the device: ") + d.last_error(ctx_));
  if (res.status & ~SMHIP_ST_FEW_PEAKS)
    ExitWithMessage("GenerateChains: a device capacity was exceeded (status " + std::to_string(res.status) + ")");
  chains.clear();
  if (res.n_chains == 0) return;
  std::vector<smhip_chain> recs(res.n_chains);
  uint64_t coff[2] = {0, 0}, aoff[2] = {0, 0};
  rc = d.map_download_chains(ctx_, recs.data(), recs.size(), coff);
  size_t total_anchors = 0;
  for (const smhip_chain &c : recs) total_anchors += c.n_anchors;
  std::vector<smhip_carried_anchor> anchors(std::max<size_t>(total_anchors, 1));
  if (rc == SMHIP_OK) rc = d.map_download(ctx_, &res, anchors.data(), anchors.size(), aoff);
  if (rc != SMHIP_OK || coff[1] != res.n_chains || aoff[1] != total_anchors)
    ExitWithMessage(std::string("GenerateChains: download failed: ") + d.last_error(ctx_));
  size_t at = 0;
  for (const smhip_chain &c : recs) {
    SignalAnchorChain out{c.score, c.ref, c.start, c.end, c.n_anchors, (uint8_t)c.mapq, c.direction == 1 ? Positive : Negative, {}};
    out.anchors.reserve(c.n_anchors);
    for (uint32_t k = 0; k < c.n_anchors; ++k, ++at)
      out.anchors.push_back(SignalAnchor{anchors[at].target_position, anchors[at].query_position, anchors[at].distance});
    chains.push_back(std::move(out));
  }
}

// ------------------------------------------------------------------ Sigmap::GenerateEvents
Sigmap::~Sigmap() {
  if (owns_ctx_ && ctx_) Device().destroy(ctx_);
}

void Sigmap::GenerateEvents(size_t start, size_t end, const Signal &signal, std::vector<float> &feature_signal,
                            std::vector<float> &feature_signal_stdvs) {
  if (!(start < end) || end > signal.GetSignalLength() || signal.signal_values.size() < end)
    ExitWithMessage("GenerateEvents: the range must lie inside the signal's pA samples");  // (the reference asserts, sigmap.cc:1051-1053)
  if (end - start > 4096) ExitWithMessage("GenerateEvents: at most 4096 samples per call (the reference's chunks hold 4000)");
  if (!ctx_) {
    ctx_ = NewDeviceContext(0);
    owns_ctx_ = true;
  }
  const DeviceApi &d = Device();
  smhip_params p;
  d.default_params(&p);
  const uint32_t stride = 2736, off[2] = {0, (uint32_t)(end - start)};
  std::vector<float> feat(stride), stdv(stride);
  uint32_t nf = 0, ne = 0, status = 0;
  const int rc = d.events(ctx_, &p, signal.signal_values.data() + start, off, 1, stride, feat.data(), stdv.data(), &nf, &ne, &status);
  if (rc != SMHIP_OK) ExitWithMessage(std::string("GenerateEvents failed on the device: ") + d.last_error(ctx_));
  feature_signal.assign(feat.begin(), feat.begin() + nf);
  feature_signal_stdvs.insert(feature_signal_stdvs.end(), stdv.begin(), stdv.begin() + nf);
}
}  // namespace sigmap
