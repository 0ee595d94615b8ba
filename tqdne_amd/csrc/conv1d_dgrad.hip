// data gradients (EPI == 1) of the stride-1 convolutions
#include "conv1d_kernel.hpp"

namespace tq {
int conv_launch_dgrad(const ConvArgs& a, int ktaps, hipStream_t s) {
    switch (ktaps) {
        case 1: return dispatch_tile<1, 1, 0, 1, 0>(a, s);
        case 3: return dispatch_tile<3, 1, 0, 1, 0>(a, s);
        default: return dispatch_tile<5, 1, 0, 1, 0>(a, s);
    }
}
}  // namespace tq
