// a5 (output convs of SeparateHead): 3x3 convolution, 64 input channels -> 1..4 output channels,
// stride 1, pad 1, + bias, on a channels-last input, forward and weight gradient (gfx950).
//
// Reference: the last layer of every head branch, mmdet3d/models/dense_heads/centerpoint_head.py:
// 70-79 (build_conv_layer(conv_cfg, head_conv, classes, kernel_size=final_kernel, padding=1,
// bias=True)) — 15 of them per step (reg 2, height 1, dim 3, rot 2, heatmap 1 channels x 3 tasks).
// With 1-3 output channels these are not GEMM-shaped: a matrix kernel spends its time on a
// 64-576-wide reduction for a 3-wide output (measured: MIOpen implicit-GEMM 0.34 ms fwd, 0.38 ms
// wgrad per conv on a 219 MB input = 0.6 TB/s). They are HBM/L2-bound streaming reductions:
//   fwd    one lane per output pixel: 9 x 16 float4 loads of its input rows (L1/L2 serve the 9x
//          neighbour reuse), weights broadcast from LDS, planar NCHW output written coalesced.
//   wgrad  one lane per input channel: the wave walks its pixels, 9 coalesced 256 B loads per
//          pixel, 27 accumulators per lane, one atomicAdd per weight per wave at the end.
// Backward-data (writes the 219 MB input gradient, already bandwidth-bound) stays with MIOpen.
#include "gga_common.h"

#define HC_CIN 64
#define HC_MAXCO 4

// Tile = HC_TR rows x HC_TW pixels of one image; its (HC_TR+2) x (HC_TW+2) x 64 input halo is staged
// in LDS with a pixel stride of 68 floats: 16 B aligned for b128 access, and (68 mod 64 = 4) makes
// the lane-per-pixel b128 reads of the forward bank-conflict free; the lane-per-channel reads of the
// weight gradient are contiguous.
#define HC_TR 4
#define HC_TW 32
#define HC_PS 68                      // LDS pixel stride in floats
#define HC_HR (HC_TR + 2)
#define HC_HW (HC_TW + 2)

struct HcTile { int b, y0, x0; };

__device__ __forceinline__ HcTile hc_tile(int64_t t, int tiles_x, int tiles_y) {
    HcTile r;
    const int per_img = tiles_x * tiles_y;
    r.b = (int)(t / per_img);
    const int rem = (int)(t - (int64_t)r.b * per_img);
    r.y0 = (rem / tiles_x) * HC_TR;
    r.x0 = (rem % tiles_x) * HC_TW;
    return r;
}

// cooperative, coalesced load of the halo tile (zero outside the image)
__device__ __forceinline__ void hc_load_tile(const float* __restrict__ x, HcTile t, int H, int W, float* __restrict__ lds) {
    for (int i = threadIdx.x; i < HC_HR * HC_HW * (HC_CIN / 4); i += 256) {
        const int q = i & 15, pix = i >> 4;
        const int hr = pix / HC_HW, hx = pix - hr * HC_HW;
        const int iy = t.y0 + hr - 1, ix = t.x0 + hx - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = *reinterpret_cast<const float4*>(x + (((int64_t)t.b * H + iy) * W + ix) * HC_CIN + q * 4);
        *reinterpret_cast<float4*>(lds + (int64_t)pix * HC_PS + q * 4) = v;
    }
}

// x: [B, H, W, 64] (channels-last memory of a [B,64,H,W] tensor); w: [cout][64][3][3]; y: [B, cout, H, W]
// thread = (tile pixel, half of the input channels); weights are broadcast b128 reads from LDS,
// the two halves are summed with one shuffle.
template <int COUT>
__global__ __launch_bounds__(256) void headconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int B, int H, int W,
                                                          int tiles_x, int tiles_y, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float lds[HC_HR * HC_HW * HC_PS];
    __shared__ __attribute__((aligned(16))) float wl[9 * COUT * HC_CIN];     // [off][co][ci]: b128 = 4 ci of one co
    const HcTile t = hc_tile(blockIdx.x, tiles_x, tiles_y);
    for (int i = threadIdx.x; i < 9 * COUT * HC_CIN; i += 256) {
        const int ci = i & 63, oc = i >> 6;
        const int off = oc / COUT, co = oc - off * COUT;
        wl[i] = w[((int64_t)co * HC_CIN + ci) * 9 + off];
    }
    hc_load_tile(x, t, H, W, lds);
    __syncthreads();
    const int half = threadIdx.x & 1, pid = threadIdx.x >> 1;      // pid 0..127 = tile pixel
    const int ty = pid / HC_TW, tx = pid - ty * HC_TW;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    // (kept rolled: a full unroll lets the scheduler hoist all 72*(1+COUT) b128 loads and spill)
#pragma unroll 1
    for (int off = 0; off < 9; ++off) {
            const int ky = off / 3, kx = off - ky * 3;
            const float* src = lds + ((ty + ky) * HC_HW + (tx + kx)) * HC_PS + half * 32;
            const float* wsrc = wl + off * COUT * HC_CIN + half * 32;
#pragma unroll 2
            for (int q = 0; q < 8; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(src + q * 4);
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float4 wv = *reinterpret_cast<const float4*>(wsrc + co * HC_CIN + q * 4);   // broadcast read
                    acc[co] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                }
            }
        }
    const int oy = t.y0 + ty, ox = t.x0 + tx;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float s = acc[co] + __shfl_xor(acc[co], 1, 64);
        if (half == 0 && oy < H && ox < W)
            y[(((int64_t)t.b * COUT + co) * H + oy) * W + ox] = s + (bias ? bias[co] : 0.0f);
    }
}

// dW[co][ci][off] = sum_p x[p+off][ci] * dy[co][p]; dbias[co] = sum_p dy[co][p].
// Persistent workgroups walk the tiles; wave = tile row, lane = input channel; per-block partial
// sums go to `partials` and a second kernel adds them in a fixed order (no atomics).
template <int COUT>
__global__ __launch_bounds__(256) void headconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int B, int H, int W, int tiles_x, int tiles_y,
                                                            int64_t n_tiles, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) float lds[HC_HR * HC_HW * HC_PS];
    __shared__ float gds[HC_TR * HC_TW * COUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[9][COUT];
#pragma unroll
    for (int o = 0; o < 9; ++o)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[o][co] = 0.0f;
    float bsum[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) bsum[co] = 0.0f;
    for (int64_t ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        const HcTile t = hc_tile(ti, tiles_x, tiles_y);
        __syncthreads();                               // previous tile fully consumed
        hc_load_tile(x, t, H, W, lds);
        for (int i = threadIdx.x; i < HC_TR * HC_TW * COUT; i += 256) {
            const int co = i / (HC_TR * HC_TW), pid = i - co * (HC_TR * HC_TW);
            const int oy = t.y0 + pid / HC_TW, ox = t.x0 + pid % HC_TW;
            gds[pid * COUT + co] = (oy < H && ox < W) ? dy[(((int64_t)t.b * COUT + co) * H + oy) * W + ox] : 0.0f;
        }
        __syncthreads();
        for (int tx = 0; tx < HC_TW; ++tx) {
            float g[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) { g[co] = gds[(wave * HC_TW + tx) * COUT + co]; bsum[co] += g[co]; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = lds[((wave + ky) * HC_HW + (tx + kx)) * HC_PS + lane];
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[ky * 3 + kx][co] += v * g[co];
                }
        }
    }
    // block reduce over the 4 waves through LDS, then one partial row per block:
    // layout [block][co][ci][off] (+ COUT bias sums at the end)
    __syncthreads();
    float* red = lds;                                   // 4 * 9 * COUT * 64 floats <= tile buffer
#pragma unroll
    for (int o = 0; o < 9; ++o)
#pragma unroll
        for (int co = 0; co < COUT; ++co) red[((wave * 9 + o) * COUT + co) * 64 + lane] = acc[o][co];
    if (lane == 0)
#pragma unroll
        for (int co = 0; co < COUT; ++co) gds[wave * COUT + co] = bsum[co];
    __syncthreads();
    float* out = partials + (int64_t)blockIdx.x * (COUT * HC_CIN * 9 + COUT);
    for (int i = threadIdx.x; i < 9 * COUT * 64; i += 256) {
        const int ci = i & 63, oc = i >> 6;              // oc = o * COUT + co
        const int o = oc / COUT, co = oc - o * COUT;
        const float s = (red[(0 * 9 * COUT + oc) * 64 + ci] + red[(1 * 9 * COUT + oc) * 64 + ci]) +
                        (red[(2 * 9 * COUT + oc) * 64 + ci] + red[(3 * 9 * COUT + oc) * 64 + ci]);
        out[((int64_t)co * HC_CIN + ci) * 9 + o] = s;
    }
    if (threadIdx.x < COUT)
        out[COUT * HC_CIN * 9 + threadIdx.x] = (gds[threadIdx.x] + gds[COUT + threadIdx.x]) +
                                               (gds[2 * COUT + threadIdx.x] + gds[3 * COUT + threadIdx.x]);
}

// one wavefront per output value: lanes stride over the block partials (fixed order)
__global__ __launch_bounds__(256) void headconv_wgrad_final_kernel(const float* __restrict__ partials, int nblocks,
                                                                  int n_w, int cout, float* __restrict__ dw,
                                                                  float* __restrict__ dbias) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_w + cout) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += (double)partials[(int64_t)b * (n_w + cout) + i];
    s = wave_sum(s);
    if (lane != 0) return;
    if (i < n_w) dw[i] = (float)s;
    else if (dbias) dbias[i - n_w] = (float)s;
}

#define HC_WGRAD_BLOCKS 512

static int headconv_check(const char* fn, int B, int H, int W, int cin, int cout) {
    GGA_REQUIRE(B >= 1 && H >= 1 && W >= 1, "%s: bad sizes", fn);
    GGA_REQUIRE(cin == HC_CIN && cout >= 1 && cout <= HC_MAXCO,
                "%s: specialised for %d input channels and 1..%d output channels (got %d -> %d)", fn, HC_CIN, HC_MAXCO,
                cin, cout);
    return GGA_OK;
}

extern "C" size_t gga_head_conv3x3_workspace_bytes(int cout) {
    return (size_t)HC_WGRAD_BLOCKS * ((size_t)cout * HC_CIN * 9 + cout) * sizeof(float);
}

extern "C" int gga_head_conv3x3_fwd(const float* x, const float* weight, const float* bias, int B, int H, int W, int cin,
                                    int cout, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = headconv_check("gga_head_conv3x3_fwd", B, H, W, cin, cout)) return rc;
    GGA_REQUIRE(x && weight && y, "gga_head_conv3x3_fwd: null pointer argument");
    const int tx = (W + HC_TW - 1) / HC_TW, ty = (H + HC_TR - 1) / HC_TR;
    const dim3 grid((unsigned)((int64_t)B * tx * ty)), block(256);
#define HC_F(CO) hipLaunchKernelGGL(headconv_fwd_kernel<CO>, grid, block, 0, stream, x, weight, bias, B, H, W, tx, ty, y)
    switch (cout) { case 1: HC_F(1); break; case 2: HC_F(2); break; case 3: HC_F(3); break; default: HC_F(4); }
#undef HC_F
    GGA_CHECK_LAUNCH("headconv_fwd_kernel");
    return GGA_OK;
}

extern "C" int gga_head_conv3x3_wgrad(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                                      float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = headconv_check("gga_head_conv3x3_wgrad", B, H, W, cin, cout)) return rc;
    GGA_REQUIRE(x && grad_y && grad_weight && workspace, "gga_head_conv3x3_wgrad: null pointer argument");
    if (workspace_bytes < gga_head_conv3x3_workspace_bytes(cout)) {
        gga_set_error("gga_head_conv3x3_wgrad: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    const int tx = (W + HC_TW - 1) / HC_TW, ty = (H + HC_TR - 1) / HC_TR;
    const int64_t n_tiles = (int64_t)B * tx * ty;
    const int nb = (int)(n_tiles < HC_WGRAD_BLOCKS ? n_tiles : HC_WGRAD_BLOCKS);
    float* partials = (float*)workspace;
#define HC_W(CO) hipLaunchKernelGGL(headconv_wgrad_kernel<CO>, dim3(nb), dim3(256), 0, stream, x, grad_y, B, H, W, tx, ty, n_tiles, partials)
    switch (cout) { case 1: HC_W(1); break; case 2: HC_W(2); break; case 3: HC_W(3); break; default: HC_W(4); }
#undef HC_W
    GGA_CHECK_LAUNCH("headconv_wgrad_kernel");
    const int n_w = cout * HC_CIN * 9;
    hipLaunchKernelGGL(headconv_wgrad_final_kernel, dim3((n_w + cout + 3) / 4), dim3(256), 0, stream, partials, nb, n_w,
                       cout, grad_weight, grad_bias);
    GGA_CHECK_LAUNCH("headconv_wgrad_final_kernel");
    return GGA_OK;
}
