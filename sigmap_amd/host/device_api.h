// device_api.h -- how the host library (libsigmap_host.so, plain g++) reaches the device library
// (libsigmap_hip.so): the entry points of include/sigmap_hip.h it needs, resolved on first use with dlopen from
// the directory this library was loaded from (a process that links libsigmap_hip.so, like the `sigmap` command,
// gets the copy it already has).  Without the library or without a GPU the process ends with a message, like
// the reference's ExitWithMessage (utils.h:67): the product has no CPU implementation of the hot path.
#ifndef SIGMAP_AMD_DEVICE_API_H_
#define SIGMAP_AMD_DEVICE_API_H_
#include "../../include/sigmap_hip.h"

namespace sigmap {
struct DeviceApi {
  decltype(&smhip_init) init;
  decltype(&smhip_destroy) destroy;
  decltype(&smhip_last_error) last_error;
  decltype(&smhip_default_params) default_params;
  decltype(&smhip_index_upload) index_upload;
  decltype(&smhip_generate_chains) generate_chains;
  decltype(&smhip_map_download) map_download;
  decltype(&smhip_map_download_chains) map_download_chains;
  decltype(&smhip_events) events;
};
const DeviceApi &Device();        // exits with a message when the device library cannot be loaded
smhip_ctx *NewDeviceContext(int device_ordinal);  // exits with a message when there is no usable GPU
}  // namespace sigmap
#endif
