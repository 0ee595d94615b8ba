// k = 5 forward launches with the dropout prologue and / or the ResBlock's fused 1x1 skip conv (training forward, conv2)
#include "conv1d_kernel.hpp"

namespace tq {
int conv_launch_fwd_k5b(const ConvArgs& a, hipStream_t s) {
    const bool gn = a.flags & TQ_CONV_GN, silu = a.flags & TQ_CONV_SILU, drop = a.flags & TQ_CONV_DROPOUT;
    if (a.sx0) {  // fused 1x1 skip conv: built for the ResBlock's second conv (k = 5, GN + SiLU [+ dropout])
        if (gn && silu && drop) return dispatch_tile<5, 1, 0, 0, 3, true>(a, s);
        if (gn && silu) return dispatch_tile<5, 1, 0, 0, 2, true>(a, s);
        return TQ_ERR_SHAPE;
    }
    if (gn && silu && drop) return dispatch_tile<5, 1, 0, 0, 3>(a, s);
    return TQ_ERR_ARG;
}
}  // namespace tq
