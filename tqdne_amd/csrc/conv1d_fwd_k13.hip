// k = 1 and k = 3 forward launches (attention projections incl. the pre-split qkv epilogue, Encoder / Decoder convs)
#include "conv1d_kernel.hpp"

namespace tq {
int conv_launch_fwd_k1(const ConvArgs& a, hipStream_t s) { return dispatch_act<1>(a, s); }
int conv_launch_fwd_k3(const ConvArgs& a, hipStream_t s) { return dispatch_act<3>(a, s); }
int conv_launch_qkv(const ConvArgs& a, hipStream_t s) {  // qkv projection with the pre-split K / V epilogue: k = 1, plain or folded-GN prologue
    if (a.flags & TQ_CONV_GN) return dispatch_tile<1, 1, 0, 2, 1>(a, s);
    return dispatch_tile<1, 1, 0, 2, 0>(a, s);
}
}  // namespace tq
