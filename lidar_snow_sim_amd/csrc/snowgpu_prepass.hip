// snowgpu_prepass.hip -- placeholder, replaced below by the real device prepass / wet-ground kernels.
#include <hip/hip_runtime.h>
#include "sg_prepass.h"
extern "C" int sg_prepass_run(SgPrepassScratch *, const void *, int, const int64_t *, int, int64_t, const double *, double, double *, int32_t *, void *) { return -1; }
extern "C" int sg_wet_run(SgPrepassScratch *, const void *, int, const int64_t *, int, int64_t, const double *, const SgWetParams *, double *, int32_t *, int64_t *, int32_t *, int32_t *, void *) { return -1; }
extern "C" void sg_prepass_release(SgPrepassScratch *) {}
