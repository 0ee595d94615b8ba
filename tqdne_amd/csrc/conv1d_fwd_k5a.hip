// k = 5 forward launches without dropout and without the fused skip conv (prologues none | GN | GN + SiLU)
#define TQ_STAMP_OWNER
#include "conv1d_kernel.hpp"

namespace tq {
int conv_launch_fwd_k5(const ConvArgs& a, hipStream_t s) {
    const bool gn = a.flags & TQ_CONV_GN, silu = a.flags & TQ_CONV_SILU, drop = a.flags & TQ_CONV_DROPOUT;
    if ((!gn && silu) || (drop && !silu)) return TQ_ERR_ARG;  // supported prologues: none | GN | GN+SiLU | GN+SiLU+dropout
    if (a.sx0 || drop) return conv_launch_fwd_k5b(a, s);     // fused 1x1 skip conv / dropout prologue: conv1d_fwd_k5b.hip
    if (gn && silu) return dispatch_tile<5, 1, 0, 0, 2>(a, s);
    if (gn) return dispatch_tile<5, 1, 0, 0, 1>(a, s);
    return dispatch_tile<5, 1, 0, 0, 0>(a, s);
}
}  // namespace tq
