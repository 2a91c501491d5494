// Backend factory of the product library: trackers run on the gfx950 device layer, and only on it.
#include "track/HipBackend.h"

using namespace feature_tracker;

FrontendBackend* ftk_make_backend(int device, const ft_limits& limits, void** device_ctx) {
    HipBackend* b = new HipBackend(device, limits);   // throws when no device context can be created
    if (device_ctx) *device_ctx = b->ctx();
    return b;
}

const char* ftk_backend_label() { return "hip-gfx950"; }
