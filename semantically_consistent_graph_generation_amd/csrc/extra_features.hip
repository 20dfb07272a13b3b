// placeholder, replaced below
#include "../../include/scgg.h"
#include "kernels.h"
extern "C" int scgg_adjacency_features(const int8_t*, const uint8_t*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                                       float*, float*, float*, float*, void*) {
  return scgg::fail("adjacency_features: not built yet");
}
