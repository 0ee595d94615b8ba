// stride-2 (Downsample) and nearest-x2 (Upsample) convolutions: raw inputs, no prologue
#include "conv1d_kernel.hpp"

namespace tq {
int conv_launch_resample(const ConvArgs& a, int ktaps, int stride, hipStream_t s) {
    if (stride == 2) return dispatch_tile<3, 2, 0, 0, 0>(a, s);
    if (ktaps == 5) return dispatch_tile<5, 1, 1, 0, 0>(a, s);
    if (ktaps == 3) return dispatch_tile<3, 1, 1, 0, 0>(a, s);
    return TQ_ERR_SHAPE;
}
}  // namespace tq
