// a4 / a5: the 3x3 stride-1 convolutions of the BEV trunk and the head (mmdet3d/models/backbones/second.py:58-63,
// dense_heads/centerpoint_head.py:58-68), forward and backward-data on two fp16 operand planes - the producer / consumer form.
//
// Why a second form. The lock-step kernel of dense_conv.hip has every wave stage operands AND multiply: at each chunk head all
// waves wait for the halo loads, split them, write LDS and meet at a barrier while the matrix pipes idle, and every stage's
// weight copy sits between the MFMAs of the wave that issues it. Ablation builds of that kernel (round 4, profiles/r04_dense_conv_ablation.txt:
// staging compiled out, MFMAs and fragment reads kept) run 19-26 % faster than the kernel itself, and a barrier-per-stage
// LDS-fed MFMA stream of the same shape (tools_dev/micro/mfma_sustained.hip) holds 0.58-0.62 of the nominal 2.5 PFLOP/s on
// random data where the kernel reaches 0.35-0.37. Here the two jobs belong to different waves of one 512-thread workgroup per CU:
//   waves 0-3 (one per SIMD), consumers: MT image rows x NT 32-column tiles each (128 accumulator registers), two fragment
//     sets - a stage multiplies the set the previous stage read for it and reads the next stage's between its own MFMAs
//     (2 MFMAs : 1 ds_read_b128, pinned with sched_group_barrier); nothing else in their loop but the stage barrier;
//   waves 4-7, producers: the weight stage two stages ahead global -> registers -> LDS (three LDS buffers, two register
//     sets, requested three stages ahead), and the NEXT chunk's halo image - loaded a chunk ahead, split into the two fp16
//     planes and written into the other of two LDS halo images, a few pieces per stage, so that no stage carries a chunk head.
// A workgroup walks tiles blockIdx.x, + gridDim.x, ... as one uninterrupted stream of stages: the producers stage the next
// tile's first chunk during the current tile's last one, and only the consumers' epilogue interrupts the matrix pipes.
// LDS images, packed weight layout (dense_pack_weight_kernel / weight_bank.hip), arithmetic and epilogues (scale back, BatchNorm
// statistics, the BatchNorm-backward mask + sums of DcBnBwd) are those of dense_conv3x3_x9_kernel; results differ from it only
// in the order of the per-tile statistics' partial sums.
// What it bought (EXPERIMENTS.md 6d): 7-22 % stand-alone on dense random inputs, 3 % per launch inside the step - the matrix pipe went from
// 0.45-0.50 to 0.50-0.61 busy and the clock fell from 2.0-2.2 to 1.84-1.97 GHz with it: under this arithmetic the part is
// power-managed (the same stream on zero operands runs a third faster), so a better-fed pipe is paid back in frequency.
#include <stdlib.h>

#include "dense_conv.h"

#define GGA_MAX_DEVICES 64
#define DC_WS_DEFAULT_MFMA 32                 // consumer waves' matrix instruction unless GGA_DC_WS_MFMA says otherwise (16: 16x16x32)
__device__ __attribute__((aligned(16))) float dc_zero_page[DC_WS_MAX_CIN];      // what a halo piece outside the image is read from

// launchers of the four instantiations, one translation unit each (dense_conv_ws_i{32,16}_{42,24}.hip)
#define DC_WS_LAUNCH_ARGS dim3 grid, dim3 block, hipStream_t stream, const float* x, const uint16_t* split_weight, int B, int H, int W, int cin,    \
                          int cout, int tx, int ty, float* y, int ystride, int prow, int pcol, double* stats, const uint32_t* amax_x,             \
                          const uint32_t* amax_weight, DcBnBwd bn, const float* zero_page, DcSlices sl
void dc_ws_launch_32_42(DC_WS_LAUNCH_ARGS);
void dc_ws_launch_32_24(DC_WS_LAUNCH_ARGS);
void dc_ws_launch_16_42(DC_WS_LAUNCH_ARGS);
void dc_ws_launch_16_24(DC_WS_LAUNCH_ARGS);
#define DC_WS_PASS grid, block, stream, x, (const uint16_t*)split_weight, B, H, W, cin, cout, tx, ty, y, ystride, prow, pcol, stats, amax_x, amax_weight, bn, zero_page, sl

bool dc_ws_enabled(int planes) {         // read per call: a test compares the two forms within one process
    const char* e = getenv("GGA_DC_WS");
    return planes == 2 && !(e && atoi(e) == 0);
}

// workgroups of a launch = rows of its `stats` (per slice)
int64_t dc_ws_grid(int64_t n_tiles) {
    static const int max_grid = getenv("GGA_DC_WS_GRID") ? atoi(getenv("GGA_DC_WS_GRID")) : 256;
    return n_tiles < max_grid ? n_tiles : max_grid;
}

int dc_launch_ws(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout, float* y, int ystride, int prow,
                 int pcol, double* stats, const uint32_t* amax_x, const uint32_t* amax_weight, DcBnBwd bn, const DcSlices* slices,
                 hipStream_t stream) {
    const int trows = cout == 128 ? 8 : 16;
    const int tx = (W + DC_TW - 1) / DC_TW, ty = (H + trows - 1) / trows;
    DcSlices sl;
    sl.n = 0;
    if (slices) sl = *slices;
    const int64_t n_tiles = (int64_t)B * tx * ty * (sl.n > 1 ? sl.n : 1);
    GGA_REQUIRE(n_tiles < 2147483647ll, "gga_dense_conv3x3: too many tiles");
    // one workgroup per CU (512 threads at 256 registers); more tiles than CUs: persistent workgroups, tiles b, b + grid, ...
    static const float* zero_pages[GGA_MAX_DEVICES] = {};           // per device: the address of dc_zero_page (a lookup, not an allocation)
    int dev = 0;
    GGA_CHECK_HIP(hipGetDevice(&dev), "hipGetDevice");
    GGA_REQUIRE(dev >= 0 && dev < GGA_MAX_DEVICES, "gga_dense_conv3x3: device %d", dev);
    if (!zero_pages[dev]) {
        void* p = nullptr;
        GGA_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(dc_zero_page)), "hipGetSymbolAddress(dc_zero_page)");
        zero_pages[dev] = (const float*)p;
    }
    const float* zero_page = zero_pages[dev];
    const dim3 grid((unsigned)dc_ws_grid(n_tiles)), block(512);
    // matrix instruction of the consumer waves: 16x16x32 (needs whole quads of chunks: cin % 64 == 0) or 32x32x16; GGA_DC_WS_MFMA=32 / 16: A/B switch
    const char* mfma_env = getenv("GGA_DC_WS_MFMA");          // read per call: a test compares the forms within one process
    const int mfma = mfma_env ? atoi(mfma_env) : DC_WS_DEFAULT_MFMA;
    if (mfma == 16 && cin % 64 == 0) {
        if (cout == 128) dc_ws_launch_16_42(DC_WS_PASS);
        else dc_ws_launch_16_24(DC_WS_PASS);
        GGA_CHECK_LAUNCH("dense_conv3x3_ws16_kernel");
        return GGA_OK;
    }
    if (cout == 128) dc_ws_launch_32_42(DC_WS_PASS);
    else dc_ws_launch_32_24(DC_WS_PASS);
    GGA_CHECK_LAUNCH("dense_conv3x3_ws_kernel");
    return GGA_OK;
}
