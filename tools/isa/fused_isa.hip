// probe TU: only the headline instantiation of the one-pass step, for fast ISA / register inspection
#include "wdf_clipper_fused.h"
namespace wdf {
template __global__ void clipper_fused_tp_kernel<false, true, true, false, v2f, 1>(
    const float*, const float*, const float*, float, int, int, const float*, float, int64_t, float*, const float*,
    float*, float*, float*, float*, TpStatus*, TpCtl*, float*, int, unsigned*, unsigned*, float, int64_t, int64_t, int64_t, int64_t, int,
    double*, FusedOut, int64_t, double*, int);
}
