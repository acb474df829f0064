// Shared by dense_conv.hip (the lock-step forms of the dense 3x3 convolution, its weight gradient and the entry points) and
// dense_conv_ws.hip (the producer / consumer form of the two-plane forward / backward-data convolution).
#pragma once
#include "gga_common.h"
#include "conv_planes.h"

#define DC_TW 32                             // pixels of a tile row (the M of the 32x32x16 matrix instruction)
#define DC_HW (DC_TW + 2)                    // halo row
#define DC_CK 16                             // input channels per chunk
#define DC_WS_MAX_CIN 4096                    // input channels the producer / consumer form takes (its zero page: one float per channel)
#define DC_ROWB 48                           // bytes per LDS row (16 x 16-bit + pad: conflict-free ds_read_b128)

// Backward-data launches whose result is the gradient of a BatchNorm + ReLU output z = relu(bn(y)) take the reduce pass
// of that BatchNorm's backward into their epilogue: the tile is masked by the ReLU (recomputed from y, gamma, beta and
// the saved statistics exactly as the forward pass computed it: gga_bn_scale_shift) before it is stored, and the tile's
// per-channel sums of g and g * xhat go to `stats` in the layout of the forward statistics. y: the BatchNorm's input,
// channel block of this launch, pixel stride ystride floats; gamma / beta / mean / invstd: of that channel block.
struct DcBnBwd {
    const float* y;
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* invstd;
    int ystride;
};

// Several 128-channel slices of one convolution's output in one launch of the producer / consumer form (n <= 1: one output)
#define DC_MAX_SLICES 16
struct DcSlices {
    int n;
    const uint16_t* w[DC_MAX_SLICES];        // packed weight operand of the slice
    float* y[DC_MAX_SLICES];                 // first output column of the slice (pixel stride: the launch's)
    double* stats[DC_MAX_SLICES];            // the slice's per-tile BatchNorm sums, or null with the launch's `stats`
};

// rows of a tile (= of a row of `stats`) for a launch of this shape and arithmetic; H, W of the tile space
int dc_tile_rows(int B, int H, int W, int cout, int planes);
// workgroups of a producer / consumer launch over n_tiles tiles (all slices) = rows of each slice's `stats`
int64_t dc_ws_grid(int64_t n_tiles);
// whether dense_conv_ws.hip runs launches of this arithmetic (two fp16 planes, GGA_DC_WS != 0)
bool dc_ws_enabled(int planes);
// the producer / consumer form: same arguments as gga_dense_conv3x3_bn_bwd after its checks (H, W, prow, pcol of the tile space)
int dc_launch_ws(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout, float* y, int ystride, int prow,
                 int pcol, double* stats, const uint32_t* amax_x, const uint32_t* amax_weight, DcBnBwd bn, const DcSlices* slices,
                 hipStream_t stream);
