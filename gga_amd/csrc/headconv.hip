// a5 (output convs of SeparateHead): 3x3 convolution, 64 input channels -> 1..4 output channels,
// stride 1, pad 1, + bias, on a channels-last input, forward and weight gradient (gfx950).
//
// Reference: the last layer of every head branch, mmdet3d/models/dense_heads/centerpoint_head.py:
// 70-79 (build_conv_layer(conv_cfg, head_conv, classes, kernel_size=final_kernel, padding=1,
// bias=True)) — 15 of them per step (reg 2, height 1, dim 3, rot 2, heatmap 1 channels x 3 tasks).
// With 1-3 output channels these are not GEMM-shaped as they stand: a matrix kernel spends its
// time on a 576-wide reduction for a 3-wide output (measured: MIOpen implicit-GEMM 0.34 ms fwd,
// 0.38 ms wgrad per conv on a 219 MB input = 0.6 TB/s; a lane-per-pixel VALU kernel over an LDS
// halo tile: 0.165 / 0.147 ms, bound by LDS reads). Swapping the roles makes them GEMMs with
// N = 9 taps x COUT <= 32 and the convolution a shifted sum of the result (see the kernels).
// Three generations live in this file: round 1 (one tile per workgroup, described next), round 2 (persistent, rows parked
// in LDS: headconv_fwd_kernel / headconv_wgrad_kernel - kept for maps beyond the 32-bit tile offsets of the newer ones and
// for the A/B, GGA_HEADCONV_PARKED) and round 3, the shipped ones (headconv_fwd16_kernel / headconv_wgrad16_kernel: operand
// straight from the load into v_mfma_f32_16x16x4_f32, every load a tile ahead with counted waits; 0.091 -> 0.056-0.063 ms
// forward, 0.100 -> 0.047-0.055 ms weight gradient at 16 x 248 x 216: EXPERIMENTS.md 6c).
// Backward-data is never materialised on the train path: the gradient w.r.t. the (never stored)
// normalised activation is recomputed from the 1-4 channel grad_y inside the BatchNorm backward of
// the branch (headtail_bwd_kernel below).
#include "gga_common.h"

#define HC_CIN 64
#define HC_MAXCO 4

// Forward on the matrix cores, with the roles swapped so that the tiny output width is not the
// GEMM's N: Z[p][n] = sum_ci x[p][ci] * w[co][ci][off] for n = off*COUT + co (9*COUT <= 32 columns,
// one 32-wide tile) is a [pixels x 64] x [64 x 32] product on v_mfma_f32_32x32x2_f32, and the
// convolution is the 9-tap shifted sum y[p][co] = bias + sum_off Z[p + off][off*COUT + co].
//   x: [B, H, W, 64] (channels-last memory of a [B,64,H,W] tensor); w: [cout][64][3][3]; y: [B, cout, H, W]
// A 512-thread workgroup owns 8 x 32 output pixels; its 10 x 34 halo pixels are dealt to the 8
// waves in groups of 32. Lane (r = lane%32, h = lane/32) loads channels 32h..32h+31 of pixel r of
// its group straight from global memory (128 contiguous bytes, no LDS staging of x) and keeps the
// weights of column r for the same channels in registers, so K is walked in the order
// (32h + s) on both operands. Z goes to LDS (pixel stride 33 floats), then one thread per
// (output pixel, channel) adds its nine taps. Each input pixel is read once per tile
// (halo 1.33x), 32 MFMAs per 32 pixels.
// (That was the round-1 forward kernel. Cycle stamps per workgroup at
// 16 x 248 x 216 had shown ~25 k of its ~60 k cycles waiting for the first pixel group's rows, ~20 k for the second group's -
// which only three of the eight waves have -, 2 k of matrix products per group and 1-8 k of epilogue.)
#define HM_TR 8
#define HM_TW 32
#define HM_HR (HM_TR + 2)
#define HM_HW (HM_TW + 2)
#define HM_NPIX (HM_HR * HM_HW)
#define HM_NGRP ((HM_NPIX + 31) / 32)
#define HM_ZS 33

// Round 2 (headconv_fwd_kernel): the kernel is PERSISTENT and keeps one pixel group's rows in flight per wave at all times. A workgroup owns
// 13 x 32 output pixels; its 15 x 34 = 510 halo pixels are exactly 16 groups of 32, two per wave. A wave walks the stream
// of its groups (tile after tile): park the rows that arrived (whole pixel rows, lane l = 16-byte piece l of four rows per
// load instruction, optional input affine + ReLU on the lane's four fixed channels) in its own LDS region, read them back
// pixel-per-lane (lane (r, h): channels 32h .. 32h+31 of pixel r; 272-byte rows are conflict-free both ways), REQUEST THE
// NEXT GROUP (possibly of the next tile), multiply, write the Z block. After a tile's second group: barrier, nine-tap sums,
// barrier. The region holds the parked rows first and the wave's two Z blocks after (8704 B), so a workgroup needs 70 KB
// and 119 registers: two per CU. The one-tile-per-workgroup form had every workgroup of a round load (25 k of its 60 k
// cycles, 12 us under a full memory pipe), then multiply, with the pipe idle meanwhile: 141-150 us per call with cold
// caches against 114-118 us now (104 -> 77 us with the input resident in the MALL); tools_dev/bench_headconv_cold.py.
#define HP_TR 13
#define HP_HR (HP_TR + 2)
#define HP_NPIX (HP_HR * HM_HW)                        // 510
#define HP_NGRP 16
#define HM_SROW 272                                    // bytes per parked pixel row (256 + 16)
#define HM_WREG (32 * HM_SROW)                         // per-wave LDS region: 8704 B >= two Z blocks of 32 x 33 floats

template <int COUT>
__global__ __launch_bounds__(512, 2) void headconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int B, int H, int W,
                                                          int tiles_x, int tiles_y, int n_tiles, int cout_total, int co_base,
                                                          const float* __restrict__ in_ss, int64_t xs, float* __restrict__ y) {
    typedef float acc16 __attribute__((ext_vector_type(16)));
    static_assert(2 * 32 * HM_ZS * 4 <= HM_WREG && (HP_NPIX + 31) / 32 == HP_NGRP, "tile geometry");
    __shared__ __attribute__((aligned(16))) unsigned char reg[8 * HM_WREG];
    const int per_img = tiles_x * tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    unsigned char* mine = reg + wave * HM_WREG;

    // B operand: column n = r -> (off, co); zero columns beyond 9*COUT. Built once per workgroup as wl[channel][32 columns]
    // in LDS and read per MFMA (lane (r, h) of k-step s: wl[32h + s][r], conflict-free): 32 registers less per lane than
    // holding the column, which is what lets two 512-thread workgroups share a CU
    __shared__ float wl[HC_CIN * 32];
    for (int i = threadIdx.x; i < HC_CIN * 32; i += 512) {
        const int c = i >> 5, n = i & 31;
        const bool used = n < 9 * COUT;
        const int off = used ? n / COUT : 0, co = used ? n - off * COUT : 0;
        wl[i] = used ? w[((int64_t)(co_base + co) * HC_CIN + c) * 9 + off] : 0.0f;
    }
    __syncthreads();
    const float* wcol = wl + (32 * h) * 32 + r;
    // the lane's piece of a pixel row: channels 4 * (lane % 16) .. +3, fixed for every load
    const int piece = lane & 15, prow_ = lane >> 4;                    // 4 pixel rows per load instruction
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_ss) {
        sc = *reinterpret_cast<const float4*>(in_ss + 4 * piece);
        sf = *reinterpret_cast<const float4*>(in_ss + HC_CIN + 4 * piece);
    }
    float4 ld[8];
#define HM_FETCH_(P) (*reinterpret_cast<const float4*>(P))
    // request the rows of group G of tile T (8 loads in flight per lane), zero outside the image / the halo
#define HM_LOAD(T, G) {                                                                                               \
        const int tb_ = (T) / per_img, trem_ = (T) - tb_ * per_img;                                                   \
        const int ty0_ = (trem_ / tiles_x) * HP_TR, tx0_ = (trem_ % tiles_x) * HM_TW;                                 \
        int pv_ = prow_;                                                                                              \
        asm volatile("" : "+v"(pv_));      /* keep the 16 row positions from being hoisted out of the tile loop (32 registers) */ \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
            const int hp = (G) * 32 + 4 * i + pv_;                                                                    \
            const int hr = hp / HM_HW, hx = hp - hr * HM_HW;                                                          \
            const int iy = ty0_ + hr - 1, ix = tx0_ + hx - 1;                                                         \
            const bool ok = hp < HP_NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;                 \
            float4 v = ok ? HM_FETCH_(x + (((int64_t)tb_ * H + iy) * W + ix) * xs + 4 * piece)                        \
                          : make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
            if (ok && in_ss) {                           /* input = relu(x * scale + shift); zero padding stays zero */ \
                v.x = fmaxf(fmaf(v.x, sc.x, sf.x), 0.0f); v.y = fmaxf(fmaf(v.y, sc.y, sf.y), 0.0f);                   \
                v.z = fmaxf(fmaf(v.z, sc.z, sf.z), 0.0f); v.w = fmaxf(fmaf(v.w, sc.w, sf.w), 0.0f);                   \
            }                                                                                                         \
            ld[i] = v; } }
    // park the group in the wave's region, read it back pixel-per-lane (a wave is in lockstep: no barrier, the LDS
    // counter orders the write before the read)
#define HM_PARK(XA) {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(mine + (4 * i + prow_) * HM_SROW + piece * 16) = ld[i]; \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) XA[q] = *reinterpret_cast<const float4*>(mine + r * HM_SROW + h * 128 + q * 16); }
#define HM_MMA(XA, ACC) {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) ACC[i] = 0.0f;                                                 \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                               \
            ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(XA[q].x, wcol[(4 * q + 0) * 32], ACC, 0, 0, 0);                \
            ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(XA[q].y, wcol[(4 * q + 1) * 32], ACC, 0, 0, 0);                \
            ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(XA[q].z, wcol[(4 * q + 2) * 32], ACC, 0, 0, 0);                \
            ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(XA[q].w, wcol[(4 * q + 3) * 32], ACC, 0, 0, 0);                \
        } }
    // D layout of 32x32x2: register v of lane l holds row (v/4)*8 + (l/32)*4 + v%4, column l%32
#define HM_ZOUT(U, ACC) { float* zg = reinterpret_cast<float*>(mine) + (U) * 32 * HM_ZS + r;                          \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) zg[((v >> 2) * 8 + h * 4 + (v & 3)) * HM_ZS] = ACC[v]; }
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    HM_LOAD(tile, wave)
    for (; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / per_img;
        const int rem = tile - b * per_img;
        const int y0 = (rem / tiles_x) * HP_TR, x0 = (rem % tiles_x) * HM_TW;
        float4 xa[8];
        acc16 acc, acc2;
        HM_PARK(xa)
        HM_LOAD(tile, wave + 8)                           // in flight while the first group multiplies
        HM_MMA(xa, acc)
        HM_PARK(xa)                                       // the region is free for the Z blocks from here on
        if (tile + (int)gridDim.x < n_tiles) { HM_LOAD(tile + (int)gridDim.x, wave) }     // across the barriers and the epilogue
        HM_ZOUT(0, acc)
        HM_MMA(xa, acc2)
        HM_ZOUT(1, acc2)
        __syncthreads();
        for (int i = threadIdx.x; i < HP_TR * HM_TW * COUT; i += 512) {
            const int co = i / (HP_TR * HM_TW), pid = i - co * (HP_TR * HM_TW);
            const int ty = pid / HM_TW, tx = pid - ty * HM_TW;
            const int oy = y0 + ty, ox = x0 + tx;
            if (oy >= H || ox >= W) continue;
            float s = bias ? bias[co_base + co] : 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int hp = (ty + ky) * HM_HW + tx + kx;           // halo pixel -> group g = hp / 32, kept by wave g % 8
                    const int g = hp >> 5;
                    s += reinterpret_cast<const float*>(reg + (g & 7) * HM_WREG)[((g >> 3) * 32 + (hp & 31)) * HM_ZS + (ky * 3 + kx) * COUT + co];
                }
            y[(((int64_t)b * cout_total + co_base + co) * H + oy) * W + ox] = s;
        }
        __syncthreads();                                  // the regions are parked into again
    }
#undef HM_LOAD
#undef HM_PARK
#undef HM_MMA
#undef HM_ZOUT
#undef HM_FETCH_
}

// Round 3: the forward kernel without the LDS parking step. Cycle ablations of the parked form at 16 x 248 x 216 out of a
// 960-channel tensor (tools_dev/exp_headconv.py: 91 us as shipped, 69 without the matrix products, 59 without the global
// reads, 31 with neither) had shown its three parts running one after another, not side by side: a wave waited for its
// group, parked it, multiplied, and only one group per wave was ever in flight. Here the operand comes straight from the
// load: v_mfma_f32_16x16x4_f32 wants A[i][k] from lane (i = lane % 16, k = lane / 16), so lane (i, hq) loads the float4
// at channels 16j + 4hq .. +3 of pixel i (16 pixel rows x 64 contiguous bytes per instruction, j = 0..3) and feeds its
// four floats to four k-steps whose B rows are the channels (16j + 4hq' + e), hq' = 0..3 - any channel order serves as
// long as both operands use it. No LDS round trip, no register copy, the address of every halo slot is a per-lane constant
// plus the tile's origin, and BOTH pixel groups of the next tile are requested a whole tile ahead (16 loads in flight per
// lane). Z is kept compactly (9 * COUT columns, odd stride) for the whole tile; a one-channel branch needs one 16-column
// product instead of two.
template <int COUT, bool AFF>
__global__ __launch_bounds__(512, 4) void headconv_fwd16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int H, int W,
                                                            int tiles_x, int tiles_y, int n_tiles, int cout_total, int co_base,
                                                            const float* __restrict__ in_ss, int64_t xs, float* __restrict__ y) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NT = COUT == 1 ? 1 : 2;               // 16-column tiles of Z
    constexpr int ZS = (9 * COUT) | 1;                  // floats per halo slot of Z: 9, 19, 27
    __shared__ float zt[HP_NGRP * 32 * ZS];
    // B operand as the lanes read it: wl[j][hq][t][n][e] = weight of channel 16j + 4hq + e in column 16t + n (one 16-byte
    // read per lane, j and t: the four k-steps a loaded float4 feeds)
    __shared__ __attribute__((aligned(16))) float wl[16 * NT * 64];
    __shared__ __attribute__((aligned(16))) float ssl[4 * HC_CIN];       // scale[64], shift[64], then 128 zeroes for the slots outside the image
    __shared__ float bl[HC_MAXCO];                       // (a global read in the nine-tap sums would wait for the sixteen loads just issued)
    const int per_img = tiles_x * tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, hq = lane >> 4;
    for (int i = threadIdx.x; i < 16 * NT * 64; i += 512) {
        const int e = i & 3, nn = (i >> 2) & 15, t = (i >> 6) % NT, jh = i / (64 * NT);
        const int c = 16 * (jh >> 2) + 4 * (jh & 3) + e, col = 16 * t + nn;
        const bool used = col < 9 * COUT;
        const int off = used ? col / COUT : 0, co = used ? col - off * COUT : 0;
        wl[i] = used ? w[((int64_t)(co_base + co) * HC_CIN + c) * 9 + off] : 0.0f;
    }
    if (AFF && threadIdx.x < 4 * HC_CIN) ssl[threadIdx.x] = threadIdx.x < 2 * HC_CIN ? in_ss[threadIdx.x] : 0.0f;
    if (threadIdx.x < COUT) bl[threadIdx.x] = bias ? bias[co_base + threadIdx.x] : 0.0f;
    __syncthreads();
    // the lane's four halo slots (group u = 0, 1 of the wave, half s = 0, 1 of the group): constant over the tiles
    int loff[2][2], hrc[2][2];                          // element offset from the tile's origin; (row - 1) << 16 | (column - 1) & 0xffff
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int hp = (wave + 8 * u) * 32 + 16 * s + n;
            const int hr = hp / HM_HW, hc = hp - hr * HM_HW;
            hrc[u][s] = hp < HP_NPIX ? (int)((unsigned)(hr - 1) << 16) | ((hc - 1) & 0xffff) : (int)0xC0000000u;     // (slots 510, 511: never inside)
            loff[u][s] = (int)(((int64_t)(hr - 1) * W + hc - 1) * xs) + 4 * hq;
        }
    f4 ld[2][8];
    unsigned okm[2] = {0u, 0u};
    // (every lane always loads - from the tensor's first bytes when its slot lies outside the image - and the zero is put in
    // when the value is used: with the loads inside per-lane branches the compiler cannot count them and waits for ALL
    // outstanding loads, the ones just issued for the next tile included, before the second group's products)
#define HQ_LOAD(T, LIVE, U) {                                                                                         \
        const int tb_ = (T) / per_img, trem_ = (T) - tb_ * per_img;                                                   \
        const int ty0_ = (trem_ / tiles_x) * HP_TR, tx0_ = (trem_ % tiles_x) * HM_TW;                                 \
        const float* base_ = x + (((int64_t)tb_ * H + ty0_) * W + tx0_) * xs;                                         \
        okm[U] = 0u;                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                               \
            const bool ok = (LIVE) && (unsigned)(ty0_ + (hrc[U][s] >> 16)) < (unsigned)H && (unsigned)(tx0_ + (short)hrc[U][s]) < (unsigned)W; \
            okm[U] |= ok ? (1u << s) : 0u;                                                                            \
            const float* p_ = ok ? base_ + loff[U][s] : x + 4 * hq;                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) ld[U][4 * s + j] = HQ_FETCH_(p_ + 16 * j);                  \
        } }
#define HQ_FETCH_(P) (*reinterpret_cast<const f4*>(P))
#define HQ_B_(J, T_) (reinterpret_cast<const f4*>(wl)[wo + ((J) * 4 * NT + (T_)) * 16])
    // products of group U (2 x 16 pixels) and its Z rows: D register v of lane (n, hq) = pixel 4 * hq + v, column n
#define HQ_MMA(U) {                                                                                                   \
        f4 acc[2][NT], bc[NT];                                                                                        \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[s][t] = f4{0.f, 0.f, 0.f, 0.f}; \
        int wo = hq * NT * 16 + n, so = hq;            /* in 16-byte units */                                          \
        asm volatile("" : "+v"(wo), "+v"(so));        /* (the operand rows are read per tile, not held in 64 registers) */ \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) bc[t] = HQ_B_(j, t);                                       \
            f4 v[2] = {ld[U][j], ld[U][4 + j]};                                                                       \
            if (AFF) {       /* input = relu(x * scale + shift); a slot outside the image reads the zero scale / shift row: relu(x * 0 + 0) */ \
                _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                       \
                    const int sq = ((okm[U] >> s) & 1u) ? so : so + 32;                                               \
                    const f4 sc = reinterpret_cast<const f4*>(ssl)[sq + 4 * j], sf = reinterpret_cast<const f4*>(ssl)[sq + 16 + 4 * j]; \
                    const f2 lo = __builtin_elementwise_fma(f2{v[s][0], v[s][1]}, f2{sc[0], sc[1]}, f2{sf[0], sf[1]}); \
                    const f2 hi = __builtin_elementwise_fma(f2{v[s][2], v[s][3]}, f2{sc[2], sc[3]}, f2{sf[2], sf[3]}); \
                    v[s] = f4{fmaxf(lo[0], 0.0f), fmaxf(lo[1], 0.0f), fmaxf(hi[0], 0.0f), fmaxf(hi[1], 0.0f)};       \
                }                                                                                                     \
            } else {                                                                                                  \
                _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                       \
                    const bool ok = (okm[U] >> s) & 1u;                                                               \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) v[s][e] = ok ? v[s][e] : 0.0f;                      \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                             \
                _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                      \
                    HQ_MFMA_(acc[0][t], v[0][e], bc[t][e]) HQ_MFMA_(acc[1][t], v[1][e], bc[t][e])                     \
                }                                                                                                     \
        }                                                                                                             \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int t = 0; t < NT; ++t)                  \
            if (16 * t + n < 9 * COUT) {                                                                              \
                float* zg = zt + ((wave + 8 * (U)) * 32 + 16 * s + 4 * hq) * ZS + 16 * t + n;                         \
                _Pragma("unroll") for (int v_ = 0; v_ < 4; ++v_) zg[v_ * ZS] = acc[s][t][v_];                         \
            } }
#define HQ_MFMA_(ACC, A, Bv) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A, Bv, ACC, 0, 0, 0);
    // Tile walk: workgroup b runs on XCD b % 8 (round-robin dispatch). Each XCD takes one contiguous eighth of the tiles and its
    // workgroups walk it side by side, so that the tiles above / below / beside a tile are read through the same L2 at about
    // the same time and the halo rows (2 of 15, 2 of 34 columns) come from there instead of from memory a second time.
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, wgs = ((int)gridDim.x + 7 - xcd) >> 3;
    const int chunk = (n_tiles + 7) >> 3;
    const int t_end = min(n_tiles, (xcd + 1) * chunk);
    int tile = xcd * chunk + wg;
    if (tile >= t_end) return;
    HQ_LOAD(tile, true, 0)
    HQ_LOAD(tile, true, 1)
    for (; tile < t_end; tile += wgs) {
        const int b = tile / per_img;
        const int rem = tile - b * per_img;
        const int y0 = (rem / tiles_x) * HP_TR, x0 = (rem % tiles_x) * HM_TW;
        const int nxt = tile + wgs;
        HQ_MMA(0)
        HQ_LOAD(nxt, nxt < t_end, 0)                      // a whole tile ahead (past the end: dummy loads, so that the count holds)
        HQ_MMA(1)
        HQ_LOAD(nxt, nxt < t_end, 1)
        __syncthreads();
        if (threadIdx.x < HP_TR * HM_TW) {               // one thread per output pixel, all its channels
            const int ty = threadIdx.x / HM_TW, tx = threadIdx.x - ty * HM_TW;
            const int oy = y0 + ty, ox = x0 + tx;
            if (oy < H && ox < W) {
                float s[COUT];
#pragma unroll
                for (int co = 0; co < COUT; ++co) s[co] = bl[co];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float* z = zt + ((ty + ky) * HM_HW + tx + kx) * ZS + (ky * 3 + kx) * COUT;
#pragma unroll
                        for (int co = 0; co < COUT; ++co) s[co] += z[co];
                    }
                float* yo = y + (((int64_t)b * cout_total + co_base) * H + oy) * W + ox;
#pragma unroll
                for (int co = 0; co < COUT; ++co) yo[(int64_t)co * H * W] = s[co];
            }
        }
        __syncthreads();                                  // Z is written again
    }
#undef HQ_LOAD
#undef HQ_FETCH_
#undef HQ_MMA
#undef HQ_MFMA_
#undef HQ_B_
}

// dW[co][ci][off] = sum_p x[p+off][ci] * dy[co][p]; dbias[co] = sum_p dy[co][p], on the matrix
// cores with the same role swap: per tile, D[ci][n] += sum_q x[q][ci] * G[q][n] over the halo
// pixels q, where G[q][off*COUT + co] = dy[co][q - off] if q - off is an output pixel of the
// tile (zero otherwise) - a [64 x halo] x [halo x 32] product, K = pixels. Lane (m, h) feeds
// x[q_h][m] / x[q_h][32 + m] (one coalesced 128 B row segment per half wave, each input element
// read exactly once per tile, no LDS staging) and G[q_h][m] (read from the tile's dy, staged in
// LDS with a zero border so the shifted read needs no branch). Persistent 512-thread
// workgroups (4 per CU: the loads of one hide behind the MFMAs of the others - 60 VGPRs with the
// K loop unrolled by 11; measured 0.136 ms at 2 per CU, 0.120 ms at 4) walk the tiles; the 8 waves split K (44 halo pixels each) and keep their partial D
// in accumulators across tiles; one fixed-order fold per workgroup at the end, then
// headconv_wgrad_final_kernel adds the workgroups (no atomics).
#define HW_PR (HM_HR + 2)
#define HW_PC (HM_HW + 2)
#define HW_KW ((HM_NGRP * 32) / 8)      // halo pixels per wave (44)
#ifndef HW_UNROLL
#define HW_UNROLL 11
#endif
#ifndef HW_MINW
#define HW_MINW 8
#endif

template <int COUT>
__global__ __launch_bounds__(512, HW_MINW) void headconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int B, int H, int W, int tiles_x, int tiles_y,
                                                            int64_t n_tiles, int cout_total, int co_base,
                                                            const float* __restrict__ in_ss, int64_t xs,
                                                            float* __restrict__ partials) {
    typedef float acc16 __attribute__((ext_vector_type(16)));
    __shared__ float gds[2][COUT * HW_PR * HW_PC];
    __shared__ float red[HC_CIN * 33];
    __shared__ float bred[8][HC_MAXCO];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const bool used = m < 9 * COUT;
    const int off = used ? m / COUT : 0, co = used ? m - off * COUT : 0;
    const int ky = off / 3, kx = off - ky * 3;
    const int gbase = co * HW_PR * HW_PC + (2 - ky) * HW_PC + (2 - kx);
    const float usedf = used ? 1.0f : 0.0f;
    // optional input affine + ReLU (the lane's two channels are fixed)
    const float sc0 = in_ss ? in_ss[m] : 1.0f, sc1 = in_ss ? in_ss[32 + m] : 1.0f;
    const float sf0 = in_ss ? in_ss[HC_CIN + m] : 0.0f, sf1 = in_ss ? in_ss[HC_CIN + 32 + m] : 0.0f;
    const float lo = in_ss ? 0.0f : -INFINITY;
    for (int i = threadIdx.x; i < 2 * COUT * HW_PR * HW_PC; i += 512) (&gds[0][0])[i] = 0.0f;    // borders stay zero
    // (the zeroes must have landed before anyone stages the first tile into the same words: without this barrier a wave that
    // was held up in the loop above could wipe values another wave had already staged - seen once in ~2000 steps, and only with
    // a second stream's kernels sharing the CU: one tile's dy partly zeroed, 3e-3 of the branch's weight gradient)
    __syncthreads();
    acc16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    float bs = 0.0f;                     // this thread's share of sum(dy) for channel (threadIdx.x >> 8) (+2 on odd passes)
    float bs2 = 0.0f;
    const int per_img = tiles_x * tiles_y;
    int it = 0;
    for (int64_t ti = blockIdx.x; ti < n_tiles; ti += gridDim.x, ++it) {
        const int b = (int)(ti / per_img);
        const int rem = (int)(ti - (int64_t)b * per_img);
        const int y0 = (rem / tiles_x) * HM_TR, x0 = (rem % tiles_x) * HM_TW;
        float* g = gds[it & 1];
        // stage dy of the tile (zero outside the image): thread -> (channel, pixel), at most 2 passes
#pragma unroll
        for (int pass = 0; pass < (COUT + 1) / 2; ++pass) {
            const int i = threadIdx.x + 512 * pass;
            const int ci = i >> 8, pid = i & 255;
            if (ci < COUT) {
                const int ty = pid >> 5, tx = pid & 31;
                const int oy = y0 + ty, ox = x0 + tx;
                const float v = (oy < H && ox < W) ? dy[(((int64_t)b * cout_total + co_base + ci) * H + oy) * W + ox] : 0.0f;
                g[ci * HW_PR * HW_PC + (ty + 2) * HW_PC + tx + 2] = v;
                if (pass == 0) bs += v; else bs2 += v;
            }
        }
        __syncthreads();      // tile `it` staged; every wave is done with tile it-1, so the other buffer is free
#pragma unroll HW_UNROLL
        for (int s = 0; s < HW_KW / 2; ++s) {
            const int q = HW_KW * wave + 2 * s + h;
            const bool inh = q < HM_NPIX;
            const int hr = inh ? q / HM_HW : 0, hx = inh ? q - hr * HM_HW : 0;
            const int iy = y0 + hr - 1, ix = x0 + hx - 1;
            const bool ok = inh && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float* xp = x + (((int64_t)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * xs + m;
            const float mk = ok ? 1.0f : 0.0f;
            const float x0_ = xp[0], x1_ = xp[32];
            const float a0 = fmaxf(fmaf(x0_, sc0, sf0), lo) * mk, a1 = fmaxf(fmaf(x1_, sc1, sf1), lo) * mk;
            const float gv = g[gbase + hr * HW_PC + hx] * usedf;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, gv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, gv, acc1, 0, 0, 0);
        }
    }
    // fixed-order fold of the 8 waves' partial D[ci][n] (register v of lane l: row (v/4)*8 + (l/32)*4 + v%4, column l%32)
    for (int wv = 0; wv < 8; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = (v >> 2) * 8 + h * 4 + (v & 3);
                float* r0 = red + row * 33 + m;
                float* r1 = red + (32 + row) * 33 + m;
                *r0 = (wv ? *r0 : 0.0f) + acc0[v];
                *r1 = (wv ? *r1 : 0.0f) + acc1[v];
            }
        }
    }
    // bias sums: threads 0..255 hold channel 0 (pass 0) and 2 (pass 1), threads 256..511 channels 1 and 3
    {
        const float t0 = wave_sum(bs), t1 = wave_sum(bs2);
        if (lane == 0) { bred[wave][0] = t0; bred[wave][1] = t1; }
    }
    __syncthreads();
    const int row_len = cout_total * HC_CIN * 9 + cout_total;
    float* out = partials + (int64_t)blockIdx.x * row_len;
    for (int i = threadIdx.x; i < 9 * COUT * HC_CIN; i += 512) {
        const int ci = i & 63, n = i >> 6;
        const int o = n / COUT, c = n - o * COUT;
        out[((int64_t)(co_base + c) * HC_CIN + ci) * 9 + o] = red[ci * 33 + n];
    }
    if (threadIdx.x < COUT) {
        const int c = threadIdx.x;                       // channel c: waves (c&1)*4 .. +3, pass c>>1
        const int w0 = (c & 1) * 4, ps = c >> 1;
        out[cout_total * HC_CIN * 9 + co_base + c] = (bred[w0][ps] + bred[w0 + 1][ps]) + (bred[w0 + 2][ps] + bred[w0 + 3][ps]);
    }
}

// Round 3: the weight gradient over INPUT tiles. The kernel above walks the halo pixels of an output tile (every input element
// read 1.33 times) and feeds the matrix cores from 4-byte loads behind ~25 address instructions per pixel pair; with neither
// its loads nor its products it still took 55 of its 100 us (tools_dev/exp_headconv.py). Here a workgroup owns 8 x 32 INPUT
// pixels - disjoint tiles, every element of x read exactly once - and the shifted operand is taken from the tile of dy with a
// one-pixel border instead (a few KB): dW[co][ci][off] = sum_q x[q][ci] * dy[co][q - (off - 1)] over the tile's own pixels q.
// v_mfma_f32_16x16x4_f32 with A[i][k] = x[pixel q0 + k][channel 4i + e] for e = 0..3: lane (i, k) loads ONE float4 (its
// pixel's channels 4i .. 4i+3; a 16-lane group reads the pixel's 256 contiguous bytes, four pixels per instruction) and
// feeds its four floats to four products whose D rows are the channels 4i + e; B[k][n] = dy of column n = off * COUT + co at
// pixel q0 + k shifted by the tap, one LDS read with an immediate offset. A wave owns one row of the tile (8 steps of 4
// pixels); every float4 is requested a whole tile ahead and every lane always loads (counted waits, see the forward kernel).
#define HG_TR 8
#define HG_TW 32
#define HG_PR (HG_TR + 2)
#define HG_PC (HG_TW + 2)
#define HG_GP (HG_PR * HG_PC)                             // 340 staged dy values per channel

template <int COUT, bool AFF>
__global__ __launch_bounds__(512, 4) void headconv_wgrad16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              int B, int H, int W, int tiles_x, int tiles_y, int n_tiles,
                                                              int cout_total, int co_base, const float* __restrict__ in_ss,
                                                              int64_t xs, float* __restrict__ partials) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NT = COUT == 1 ? 1 : 2;                // 16-column tiles of the result
    constexpr int GN = (COUT * HG_GP + 511) / 512;       // staged dy values per thread
    __shared__ float gds[2][(COUT + 1) * HG_GP];         // (+ a plane of zeroes: the columns beyond 9 * COUT read it)
    __shared__ float red[HC_CIN * 33];
    __shared__ float bred[GN * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, hq = lane >> 4;
    const int per_img = tiles_x * tiles_y;
    // B operand: column n + 16t = (tap, channel); input pixel (row `wave`, column 4s + hq) meets dy at (row - ky + 1, column - kx + 1)
    int gb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = n + 16 * t;
        const bool used = col < 9 * COUT;
        const int off = used ? col / COUT : 4, co = used ? col - off * COUT : COUT;
        const int ky = off / 3, kx = off - ky * 3;
        gb[t] = co * HG_GP + (2 - ky + wave) * HG_PC + (2 - kx) + hq;
    }
    f4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (AFF) {                                            // (scalar reads: the pointer is only known to be 4-byte aligned)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = in_ss[4 * n + e]; sf[e] = in_ss[HC_CIN + 4 * n + e]; }
    }
    // the thread's staged dy values: (channel, row, column) of the bordered tile, fixed over the tiles
    int gq[GN];
#pragma unroll
    for (int e = 0; e < GN; ++e) {
        const int i = tid + 512 * e;
        const int c = i / HG_GP, q = i - c * HG_GP;
        const int hr = q / HG_PC, hx = q - hr * HG_PC;
        gq[e] = i < COUT * HG_GP ? (c << 16) | (hr << 8) | hx : -1;
    }
    for (int i = tid; i < 2 * (COUT + 1) * HG_GP; i += 512) (&gds[0][0])[i] = 0.0f;
    const int lo = (int)(((int64_t)wave * W + hq) * xs) + 4 * n;       // the lane's element offset from the tile's first pixel
    f4 ld[8], acc[4][NT];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[e][t] = f4{0.f, 0.f, 0.f, 0.f};
    float greg[GN], bsum[GN];
#pragma unroll
    for (int e = 0; e < GN; ++e) bsum[e] = 0.0f;
    unsigned okm = 0u;
#define HG_FETCH_(P) (*reinterpret_cast<const f4*>(P))
#define HG_COORDS(T) const int tb_ = (T) / per_img, trem_ = (T) - tb_ * per_img;                                      \
        const int ty0_ = (trem_ / tiles_x) * HG_TR, tx0_ = (trem_ % tiles_x) * HG_TW;
#define HG_XLOAD(T, LIVE, S) {                                                                                        \
        HG_COORDS(T)                                                                                                  \
        const bool ok = (LIVE) && ty0_ + wave < H && tx0_ + 4 * (S) + hq < W;                                         \
        okm = ok ? okm | (1u << (S)) : okm & ~(1u << (S));                                                            \
        const float* p_ = ok ? x + (((int64_t)tb_ * H + ty0_) * W + tx0_ + 4 * (S)) * xs + lo : x + 4 * n;           \
        ld[S] = HG_FETCH_(p_); }
#define HG_GLOAD(T, LIVE) {                                                                                           \
        HG_COORDS(T)                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < GN; ++e) {                                                              \
            const int c = gq[e] >> 16, hr = (gq[e] >> 8) & 255, hx = gq[e] & 255;                                     \
            const int iy = ty0_ + hr - 1, ix = tx0_ + hx - 1;                                                         \
            const bool ok = (LIVE) && gq[e] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;         \
            const float v = *(ok ? dy + (((int64_t)tb_ * cout_total + co_base + c) * H + iy) * W + ix : dy);          \
            greg[e] = ok ? v : 0.0f;                                                                                  \
        } }
    // (the select above is applied when greg is stored, after the loads that follow have been issued: the compiler keeps it there)
#define HG_GSTORE(BUF) {                                                                                              \
        _Pragma("unroll") for (int e = 0; e < GN; ++e) if (gq[e] >= 0) {                                              \
            (BUF)[tid + 512 * e] = greg[e];                                                                           \
            const int hr = (gq[e] >> 8) & 255, hx = gq[e] & 255;                                                      \
            if (hr >= 1 && hr <= HG_TR && hx >= 1 && hx <= HG_TW) bsum[e] += greg[e];     /* the tile's own pixels */  \
        } }
    int tile = blockIdx.x;
    const int stride = gridDim.x;
    __syncthreads();                                      // the zeroes have landed (see the race note in the kernel above)
    HG_GLOAD(tile, tile < n_tiles)
#pragma unroll
    for (int s = 0; s < 8; ++s) HG_XLOAD(tile, tile < n_tiles, s)
    HG_GSTORE(gds[0])
    for (int it = 0; tile < n_tiles; tile += stride, ++it) {
        __syncthreads();          // tile `it` staged; every wave is done with tile it-1, so the other buffer is free
        const int nxt = tile + stride;
        const bool live = nxt < n_tiles;
        HG_GLOAD(nxt, live)
        const float* g = gds[it & 1];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            f4 v = ld[s];
            if (AFF) {
                const f2 a = __builtin_elementwise_fma(f2{v[0], v[1]}, f2{sc[0], sc[1]}, f2{sf[0], sf[1]});
                const f2 c = __builtin_elementwise_fma(f2{v[2], v[3]}, f2{sc[2], sc[3]}, f2{sf[2], sf[3]});
                v = f4{fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(c[0], 0.0f), fmaxf(c[1], 0.0f)};
            }
            const bool ok = (okm >> s) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
            float bw[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bw[t] = g[gb[t] + 4 * s];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[e][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e], bw[t], acc[e][t], 0, 0, 0);
                }
            HG_XLOAD(nxt, live, s)
        }
        HG_GSTORE(gds[(it + 1) & 1])
    }
#undef HG_FETCH_
#undef HG_COORDS
#undef HG_XLOAD
#undef HG_GLOAD
#undef HG_GSTORE
    // fixed-order fold of the 8 waves' partial D (register v of lane (n, hq) in product e, tile t: channel 16hq + 4v + e, column 16t + n)
    for (int wv = 0; wv < 8; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float* r = red + (16 * hq + 4 * v + e) * 33 + 16 * t + n;
                        *r = (wv ? *r : 0.0f) + acc[e][t][v];
                    }
        }
    }
#pragma unroll
    for (int e = 0; e < GN; ++e) bred[tid + 512 * e] = bsum[e];
    __syncthreads();
    const int row_len = cout_total * HC_CIN * 9 + cout_total;
    float* out = partials + (int64_t)blockIdx.x * row_len;
    for (int i = tid; i < 9 * COUT * HC_CIN; i += 512) {
        const int ci = i & 63, nn = i >> 6;
        const int o = nn / COUT, c = nn - o * COUT;
        out[((int64_t)(co_base + c) * HC_CIN + ci) * 9 + o] = red[ci * 33 + nn];
    }
    if (tid < COUT) {                                     // dbias: the channel's staged slots in index order
        float sum = 0.0f;
        for (int i = tid * HG_GP; i < (tid + 1) * HG_GP; ++i) sum += bred[i];
        out[cout_total * HC_CIN * 9 + co_base + tid] = sum;
    }
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

#ifndef HC_WGRAD_BLOCKS
#define HC_WGRAD_BLOCKS 1024
#endif

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

static int headconv_stride(const char* fn, const void* x, int64_t xs) {
    GGA_REQUIRE(xs >= HC_CIN && xs % 4 == 0 && ((uintptr_t)x & 15) == 0,
                "%s: pixel stride %lld must be a multiple of 4 floats >= %d and the base 16-byte aligned", fn, (long long)xs, HC_CIN);
    return GGA_OK;
}

extern "C" int gga_head_conv3x3_fwd(const float* x, int64_t x_pixel_stride, const float* in_scale_shift, const float* weight,
                                    const float* bias, int B, int H, int W, int cin, int cout, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = headconv_check("gga_head_conv3x3_fwd", B, H, W, cin, cout)) return rc;
    GGA_REQUIRE(x && weight && y, "gga_head_conv3x3_fwd: null pointer argument");
    if (int rc = headconv_stride("gga_head_conv3x3_fwd", x, x_pixel_stride)) return rc;
    const int tx = (W + HM_TW - 1) / HM_TW, ty = (H + HP_TR - 1) / HP_TR;
    const int64_t n_tiles = (int64_t)B * tx * ty;
    GGA_REQUIRE(n_tiles < 2147483647ll, "gga_head_conv3x3_fwd: too many tiles");
    static const bool parked_env = getenv("GGA_HEADCONV_PARKED") && (atoi(getenv("GGA_HEADCONV_PARKED")) & 1) != 0;   // A/B: the round-2 kernel (1: forward, 2: weight gradient, 3: both)
    // (the round-3 kernel keeps per-lane 32-bit element offsets within a tile and 16-bit halo coordinates: larger maps take the other one)
    const bool parked = parked_env || !(16ll * W * x_pixel_stride < 2147483647ll && H < 16384 && W < 16384);
    const dim3 grid((unsigned)(n_tiles < 512 ? n_tiles : 512)), block(512);       // persistent: two workgroups per CU
#define HC_G(CO, BASE, AF) hipLaunchKernelGGL((headconv_fwd16_kernel<CO, AF>), grid, block, 0, stream, x, weight, bias, B, H, W, tx, ty, (int)n_tiles, cout, BASE, in_scale_shift, x_pixel_stride, y)
#define HC_F(CO, BASE) { if (parked) hipLaunchKernelGGL(headconv_fwd_kernel<CO>, grid, block, 0, stream, x, weight, bias, B, H, W, tx, ty, (int)n_tiles, cout, BASE, in_scale_shift, x_pixel_stride, y); \
                         else if (in_scale_shift) HC_G(CO, BASE, true); else HC_G(CO, BASE, false); }
    switch (cout) {
        case 1: HC_F(1, 0); break;
        case 2: HC_F(2, 0); break;
        case 3: HC_F(3, 0); break;
        default: HC_F(2, 0); HC_F(2, 2); break;      // 9*4 columns do not fit one 32-wide tile
    }
#undef HC_F
#undef HC_G
    GGA_CHECK_LAUNCH("headconv_fwd_kernel");
    return GGA_OK;
}

extern "C" int gga_head_conv3x3_wgrad(const float* x, int64_t x_pixel_stride, const float* in_scale_shift, const float* grad_y,
                                      int B, int H, int W, int cin, int cout, float* grad_weight, float* grad_bias,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = headconv_check("gga_head_conv3x3_wgrad", B, H, W, cin, cout)) return rc;
    if (int rc = headconv_stride("gga_head_conv3x3_wgrad", x, x_pixel_stride)) return rc;
    GGA_REQUIRE(x && grad_y && grad_weight && workspace, "gga_head_conv3x3_wgrad: null pointer argument");
    if (workspace_bytes < gga_head_conv3x3_workspace_bytes(cout)) {
        gga_set_error("gga_head_conv3x3_wgrad: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    static const bool parked_env = getenv("GGA_HEADCONV_PARKED") && (atoi(getenv("GGA_HEADCONV_PARKED")) & 2) != 0;   // A/B: the round-2 kernel
    const int tx = (W + HM_TW - 1) / HM_TW, ty = (H + HM_TR - 1) / HM_TR;          // (both kernels: tiles of 8 x 32)
    const int64_t n_tiles = (int64_t)B * tx * ty;
    // (the round-3 kernel keeps per-lane 32-bit element offsets within a tile and an int tile count: larger maps take the other one)
    const bool parked = parked_env || !(n_tiles < 2147483647ll && 8ll * W * x_pixel_stride < 2147483647ll);
    const int cap = parked ? HC_WGRAD_BLOCKS : 512;                               // persistent: two workgroups per CU
    const int nb = (int)(n_tiles < cap ? n_tiles : cap);
    float* partials = (float*)workspace;
#define HC_W(CO, BASE) { if (parked) hipLaunchKernelGGL(headconv_wgrad_kernel<CO>, dim3(nb), dim3(512), 0, stream, x, grad_y, B, H, W, tx, ty, n_tiles, cout, BASE, in_scale_shift, x_pixel_stride, partials); \
                         else if (in_scale_shift) hipLaunchKernelGGL((headconv_wgrad16_kernel<CO, true>), dim3(nb), dim3(512), 0, stream, x, grad_y, B, H, W, tx, ty, (int)n_tiles, cout, BASE, in_scale_shift, x_pixel_stride, partials); \
                         else hipLaunchKernelGGL((headconv_wgrad16_kernel<CO, false>), dim3(nb), dim3(512), 0, stream, x, grad_y, B, H, W, tx, ty, (int)n_tiles, cout, BASE, in_scale_shift, x_pixel_stride, partials); }
    switch (cout) {
        case 1: HC_W(1, 0); break;
        case 2: HC_W(2, 0); break;
        case 3: HC_W(3, 0); break;
        default: HC_W(2, 0); HC_W(2, 2); break;
    }
#undef HC_W
    GGA_CHECK_LAUNCH("headconv_wgrad_kernel");
    const int n_w = cout * HC_CIN * 9;
    hipLaunchKernelGGL(headconv_wgrad_final_kernel, dim3((n_w + cout + 3) / 4), dim3(256), 0, stream, partials, nb, n_w,
                       cout, grad_weight, grad_bias);
    GGA_CHECK_LAUNCH("headconv_wgrad_final_kernel");
    return GGA_OK;
}


// ------------------------------------------------------------------------------ tail of a head branch, backward
// Branch tail = BatchNorm(training) -> ReLU -> 3x3 conv to COUT <= 4 channels. Its backward w.r.t. the
// BatchNorm input x is  dx = gamma*invstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dh * [x*scale+shift > 0],
// where dh = backward-data of the output conv. dh has 64 channels but only 9*COUT <= 36 terms per
// element, all taken from the tiny grad_y: both passes over x (the two sums, then dx) rebuild it in
// registers instead of reading a 219 MB tensor that a separate backward-data kernel would have written -
// three passes over the activation (x, x, dx) where backward-data + reduce + apply took six.
//   tile: 8 x 32 pixels per step of a persistent 256-thread workgroup; grad_y of the tile + a one-pixel
//   border goes to LDS (zero outside the image, double buffered, requested a tile ahead); thread (cg = tid % 16, pg = tid / 16) owns channels
//   4cg..4cg+3 - its 36*COUT weights live in registers - and walks the pixels pg, pg + 16, ... of the tile,
//   reading x as one float4 (16 lanes = the 256 contiguous bytes of a pixel).
//   partial sums: [block][2][64] f64, the layout of bn_reduce_kernel, folded by gga_bn_bwd_finalize.
#define HT_TR 8
#define HT_TW 32
#define HT_HR (HT_TR + 2)
#define HT_HW (HT_TW + 2)
#define HT_MAX_BLOCKS 2048           // = BN_MAX_BLOCKS: the partials fit the BatchNorm workspace

template <int COUT, bool APPLY>
__global__ __launch_bounds__(256, COUT >= 4 ? 1 : 2) void headtail_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ ss,
                                                          const float* __restrict__ saved, const float* __restrict__ coef,
                                                          int B, int H, int W, int tiles_x, int tiles_y, int64_t n_tiles,
                                                          int64_t xs, int64_t dxs, double* __restrict__ partials,
                                                          float* __restrict__ dx, uint32_t* __restrict__ amax) {
    uint32_t am = 0;                                     // largest finite |dx| written (apply pass)
    __shared__ float gs[2][COUT * HT_HR * HT_HW];
    __shared__ double red[APPLY ? 1 : 256][8];
    const int tid = threadIdx.x, cg = tid & 15, pg = tid >> 4;
    // w[co][ci][tap] -> wr[tap][co][j] for the thread's four channels
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 wr[9][COUT][2];                                   // (register pairs: channels (0, 1) and (2, 3) of the thread)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[t][c][j >> 1][j & 1] = w[((int64_t)c * HC_CIN + 4 * cg + j) * 9 + t];
    float sc[4], sf[4], mean[4], inv[4], kk[4], mg[4], mgx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * cg + j;
        sc[j] = ss[c]; sf[j] = ss[HC_CIN + c]; mean[j] = saved[c]; inv[j] = saved[HC_CIN + c];
        kk[j] = APPLY ? coef[c] : 0.0f; mg[j] = APPLY ? coef[HC_CIN + c] : 0.0f; mgx[j] = APPLY ? coef[2 * HC_CIN + c] : 0.0f;
    }
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    const int per_img = tiles_x * tiles_y;
    // Everything a step needs is requested one step ahead (two waves per SIMD do not hide a memory round trip
    // by themselves): grad_y of the next tile while this one is computed, x of the next four pixels while these
    // four are.
    constexpr int GN = (COUT * HT_HR * HT_HW + 255) / 256;
    float greg[GN];
    float4 nxt[4];
    auto coords = [&](int64_t ti, int& b, int& y0, int& x0) {
        b = (int)(ti / per_img);
        const int rem = (int)(ti - (int64_t)b * per_img);
        y0 = (rem / tiles_x) * HT_TR; x0 = (rem % tiles_x) * HT_TW;
    };
    auto load_g = [&](int b, int y0, int x0) {
#pragma unroll
        for (int e = 0; e < GN; ++e) {
            const int i = tid + 256 * e;
            const int c = i / (HT_HR * HT_HW), q = i - c * (HT_HR * HT_HW);
            const int hr = q / HT_HW, hx = q - hr * HT_HW;
            const int iy = y0 + hr - 1, ix = x0 + hx - 1;
            greg[e] = (c < COUT && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? gy[(((int64_t)b * COUT + c) * H + iy) * W + ix] : 0.0f;
        }
    };
    auto store_g = [&](float* g) {
#pragma unroll
        for (int e = 0; e < GN; ++e) if (tid + 256 * e < COUT * HT_HR * HT_HW) g[tid + 256 * e] = greg[e];
    };
    auto load_x = [&](int b, int y0, int x0, int u0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = pg + 16 * (u0 + u);
            const int oy = y0 + (p >> 5), ox = x0 + (p & 31);
#ifndef HT_NO_NT
            // (round 5: the apply pass reads x for the last time and writes a gradient nobody reads before 3 GB of other branches'
            // gradients have been written - non-temporal, so that what the reduce pass left of x in the Infinity Cache stays there;
            // same-box A/B, profiles/r05_ab_ht_nt.txt: PointPillars step -0.10 / -0.16 ms, sparse config -0.15 / -0.09 ms)
            typedef float ht_v4f __attribute__((ext_vector_type(4)));
            if (APPLY) {
                ht_v4f t_ = {0.f, 0.f, 0.f, 0.f};
                if (oy < H && ox < W) t_ = __builtin_nontemporal_load(reinterpret_cast<const ht_v4f*>(x + (((int64_t)b * H + oy) * W + ox) * xs + 4 * cg));
                nxt[u] = make_float4(t_.x, t_.y, t_.z, t_.w);
            } else
#endif
            nxt[u] = (oy < H && ox < W) ? *reinterpret_cast<const float4*>(x + (((int64_t)b * H + oy) * W + ox) * xs + 4 * cg)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int64_t ti = blockIdx.x;
    int b = 0, y0 = 0, x0 = 0;
    if (ti < n_tiles) {
        coords(ti, b, y0, x0);
        load_g(b, y0, x0);
        load_x(b, y0, x0, 0);
        store_g(gs[0]);
    }
    __syncthreads();
    for (int it = 0; ti < n_tiles; ++it) {
        const int64_t tn = ti + gridDim.x;
        const bool more = tn < n_tiles;
        int bn = 0, y0n = 0, x0n = 0;
        if (more) { coords(tn, bn, y0n, x0n); load_g(bn, y0n, x0n); }
        const float* g = gs[it & 1];
        float f0[4] = {0, 0, 0, 0}, f1[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int u0 = 0; u0 < (HT_TR * HT_TW) / 16; u0 += 4) {
            float4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = nxt[u];
            if (u0 + 4 < (HT_TR * HT_TW) / 16) load_x(b, y0, x0, u0 + 4);
            else if (more) load_x(bn, y0n, x0n, 0);
            // dh[q][ci] = sum_tap sum_co gy[co][q - (tap - 1)] * w[co][ci][tap], for the thread's four channels: 36 * COUT multiply-adds
            // per pixel, which is what bounds the 2- and 3-channel forms (VALU, not memory). Two PIXELS per v_pk_fma_f32: the
            // thread's pixels 2m and 2m+1 of a step lie 16 columns apart in one row, their two dy values come as one register pair
            // (one ds_read2), and the weight is broadcast out of its pair by the instruction's op_sel - no copies, half the issues.
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int pa = pg + 16 * (u0 + 2 * m);              // pixel 2m; pixel 2m+1 = the same row, column + 16
                const int ty = pa >> 5, tx = pa & 31;
                f2 d[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int c = 0; c < COUT; ++c) {
                            const float* gp = g + c * HT_HR * HT_HW + (ty + 2 - ky) * HT_HW + tx + 2 - kx;
                            const f2 g2 = {gp[0], gp[16]};
                            const f2 w01 = wr[ky * 3 + kx][c][0], w23 = wr[ky * 3 + kx][c][1];
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(d[0]) : "v"(g2), "v"(w01));
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d[1]) : "v"(g2), "v"(w01));
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(d[2]) : "v"(g2), "v"(w23));
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d[3]) : "v"(g2), "v"(w23));
                        }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int u = 2 * m + half;
                    const int oy = y0 + ty, ox = x0 + tx + 16 * half;
                    if (oy >= H || ox >= W) continue;
                    const float dh[4] = {d[0][half], d[1][half], d[2][half], d[3][half]};
                    const float xa[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gj = fmaf(xa[j], sc[j], sf[j]) > 0.0f ? dh[j] : 0.0f;
                        const float xh = (xa[j] - mean[j]) * inv[j];
                        if (APPLY) o[j] = kk[j] * (gj - mg[j] - xh * mgx[j]);
                        else { f0[j] += gj; f1[j] += gj * xh; }
                    }
                    if (APPLY) {
#ifndef HT_NO_NT
                        { typedef float ht_v4f __attribute__((ext_vector_type(4))); const ht_v4f t_ = {o[0], o[1], o[2], o[3]};
                          __builtin_nontemporal_store(t_, reinterpret_cast<ht_v4f*>(dx + (((int64_t)b * H + oy) * W + ox) * dxs + 4 * cg)); }
#else
                        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + oy) * W + ox) * dxs + 4 * cg) = make_float4(o[0], o[1], o[2], o[3]);
#endif
                        if (amax) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) am = gga_amax_of(o[j], am);
                        }
                    }
                }
            }
        }
        if (!APPLY) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] += f0[j]; s1[j] += f1[j]; }      // 16 elements per f32 run
        }
        if (more) store_g(gs[(it + 1) & 1]);     // last read of that buffer: tile it-1, before the previous barrier
        __syncthreads();
        ti = tn; b = bn; y0 = y0n; x0 = x0n;
    }
    if (APPLY && amax) gga_amax_commit(am, amax);
    if (!APPLY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[tid][j] = s0[j]; red[tid][4 + j] = s1[j]; }
        __syncthreads();
        if (tid < 16) {                                  // fixed order over the 16 pixel groups
            double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = tid; t < 256; t += 16)
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] += red[t][q];
            double* out = partials + (int64_t)blockIdx.x * 2 * HC_CIN;
#pragma unroll
            for (int j = 0; j < 4; ++j) { out[4 * tid + j] = r[j]; out[HC_CIN + 4 * tid + j] = r[4 + j]; }
        }
    }
}

extern "C" int gga_head_tail_bwd(const float* grad_y, const float* x, int64_t x_pixel_stride, const float* scale_shift,
                                 const float* gamma, const float* saved, const float* weight, int B, int H, int W, int cin,
                                 int cout, float* grad_x, int64_t grad_x_pixel_stride, float* grad_gamma, float* grad_beta,
                                 uint32_t* amax_grad_x, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = headconv_check("gga_head_tail_bwd", B, H, W, cin, cout)) return rc;
    if (int rc = headconv_stride("gga_head_tail_bwd", x, x_pixel_stride)) return rc;
    if (int rc = headconv_stride("gga_head_tail_bwd", grad_x, grad_x_pixel_stride)) return rc;
    GGA_REQUIRE(grad_y && x && scale_shift && saved && weight && grad_x && workspace, "gga_head_tail_bwd: null pointer argument");
    const int64_t rows = (int64_t)B * H * W;
    if (workspace_bytes < gga_bn_relu_workspace_bytes(rows, HC_CIN)) {
        gga_set_error("gga_head_tail_bwd: workspace %zu B < required %zu B", workspace_bytes, gga_bn_relu_workspace_bytes(rows, HC_CIN));
        return GGA_ERR_WORKSPACE;
    }
    const int tx = (W + HT_TW - 1) / HT_TW, ty = (H + HT_TR - 1) / HT_TR;
    const int64_t n_tiles = (int64_t)B * tx * ty;
    const int nb = (int)(n_tiles < HT_MAX_BLOCKS ? n_tiles : HT_MAX_BLOCKS);     // (512 .. 2048 workgroups measured: no difference beyond noise)
    double* partials = (double*)workspace;
    float* coef = nullptr;
#define HT_GO(CO, AP) hipLaunchKernelGGL((headtail_bwd_kernel<CO, AP>), dim3(nb), dim3(256), 0, stream, grad_y, x, weight, scale_shift, saved, coef, B, H, W, tx, ty, n_tiles, x_pixel_stride, grad_x_pixel_stride, partials, grad_x, amax_grad_x)
#define HT_SW(AP) switch (cout) { case 1: HT_GO(1, AP); break; case 2: HT_GO(2, AP); break; case 3: HT_GO(3, AP); break; default: HT_GO(4, AP); break; }
    HT_SW(false)
    GGA_CHECK_LAUNCH("headtail_bwd_kernel<reduce>");
    if (int rc = gga_bn_bwd_finalize(partials, nb, HC_CIN, rows, gamma, saved, grad_gamma, grad_beta, workspace, &coef, stream)) return rc;
    HT_SW(true)
    GGA_CHECK_LAUNCH("headtail_bwd_kernel<apply>");
#undef HT_SW
#undef HT_GO
    return GGA_OK;
}
