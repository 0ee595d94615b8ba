// Micro-benchmark: HBM write rate of the conv epilogue's store pattern (16 rows x 64 B per wave-instruction, the other 64-B
// half of each 128-B line written by a later instruction) against full-line patterns (8 rows x 128 B per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>

// y is (rows, C) fp32.  One wave owns 16-row x 32-channel blocks (like one (mi, ni) accumulator pair column).
template <int MODE>
__global__ __launch_bounds__(256) void k(float* y, int rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves_c = C / 32;                       // waves along channels
    const long gw = (long)blockIdx.x * 4 + wave;       // global wave id
    const long rblk = gw / nwaves_c;                   // 128-row block
    const int cw = (int)(gw % nwaves_c) * 32;
    if (rblk * 128 >= rows) return;
    const float4 v = make_float4(lane, 1.f, 2.f, 3.f);
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
        if (MODE == 0) {  // as the conv epilogue: lane&15 = row, lane>>4 = 4-channel group; two instructions per 16 rows
            const long r = rblk * 128 + tb * 16 + (lane & 15);
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
                *reinterpret_cast<float4*>(y + r * C + cw + cbk * 16 + 4 * (lane >> 4)) = v;
        } else {          // full lines: lane&7 = 4-channel group (8 x 16 B = 128 B), lane>>3 = row; two instructions per 16 rows
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long r = rblk * 128 + tb * 16 + h * 8 + (lane >> 3);
                *reinterpret_cast<float4*>(y + r * C + cw + 4 * (lane & 7)) = v;
            }
        }
    }
}

template <int MODE>
void run(int C) {
    const long rows = 64L * 4096 * 64 / C;  // 64 MiB of output whatever C
    float* y;
    hipMalloc(&y, rows * C * 4);
    const long nwaves = rows / 128 * (C / 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<(nwaves + 3) / 4, 256>>>(y, rows, C);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<MODE><<<(nwaves + 3) / 4, 256>>>(y, rows, C);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d C %3d: %.1f us per 64 MiB  = %.0f GB/s\n", MODE, C, ms / 20 * 1e3, rows * C * 4.0 / (ms / 20) * 1e-6);
    hipFree(y);
}

int main() {
    for (int C : {64, 128, 256}) { run<0>(C); run<1>(C); }
    return 0;
}
