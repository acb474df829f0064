// One instantiation of the producer / consumer dense 3x3 kernels per translation unit (dense_conv_ws_kernels.h; dense_conv_ws.hip
// holds the entry logic): dense_conv3x3_ws_kernel<4, 2>.
#include "dense_conv_ws_kernels.h"

void dc_ws_launch_32_42(dim3 grid, dim3 block, hipStream_t stream, const float* x, const uint16_t* split_weight, int B, int H, int W, int cin,
                         int cout, int tx, int ty, float* y, int ystride, int prow, int pcol, double* stats, const uint32_t* amax_x,
                         const uint32_t* amax_weight, DcBnBwd bn, const float* zero_page, DcSlices sl) {
    hipLaunchKernelGGL((dense_conv3x3_ws_kernel<4, 2>), grid, block, 0, stream, x, split_weight, B, H, W, cin, cout, tx, ty, y, ystride, prow, pcol, stats,
                       amax_x, amax_weight, bn, zero_page, sl);
}
