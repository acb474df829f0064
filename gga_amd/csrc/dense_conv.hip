// a4 / a5: the 3x3 stride-1 convolutions of the BEV trunk and the head (SECOND blocks, shared convolution, first convolution
// of every head branch: mmdet3d/models/backbones/second.py:58-63, dense_heads/centerpoint_head.py:58-68) on the 16-bit matrix
// instructions through operand planes (conv_planes.h): forward / backward-data (dense_conv3x3_x9_kernel and its epilogues),
// the weight packing and the weight gradient (dense_wgrad3x3_x9_kernel).
#include <stdlib.h>

#include "dense_conv.h"
#include <type_traits>
#include <hip/hip_fp16.h>

// ------------------------------------------------------------------------------ dense 3x3 convolution
// The dense kernels below use SIX of the nine partial products: with truncated planes
// a = a0 + a1 + a2 (|a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|) the products a1*b2, a2*b1 and a2*b2 together
// are below 2^-23 |a*b| - one fp32 ulp of the product, what an unfused multiply-add loses anyway -
// and the measured error of a whole convolution against float64 does not move (9.7e-7 of the
// output range with six or nine terms; MIOpen's fp32 kernels: 1.0e-6 .. 1.5e-6): the fp32
// accumulation dominates. One third fewer MFMAs: forward 416 -> 345 us, weight gradient 539 -> 434 us.
// Compile with -DX9_NINE for all nine.
#define X9_SIX 1
// 3x3 / stride 1 / pad 1 convolution of a channels-last image on the same bf16x9 matrix path
// (SECOND block convolutions and the first convolution of every head branch: second.py:58-63,
// centerpoint_head.py:58-68 - 64 -> 64 channels at 248 x 216, where MIOpen's fp32 implicit GEMM
// runs at 100-118 TFLOP/s). Unlike the gather form above, the input is regular: a 256-thread
// workgroup owns 8 rows x 32 pixels x all output channels, wave w rows 2w and 2w+1 (two 32-pixel
// M tiles that share every weight fragment). Per 16-input-channel chunk the 10 x 34 pixel halo
// is fetched ONCE, split into the three bf16 planes on the way into LDS (48-byte pixel rows:
// 32 + 16 pad, conflict-free ds_read_b128) and then feeds all nine taps - lane (r, h) reads
// pixel (row + ky, r + kx), channels 8h .. 8h+7 - so there are no per-tap gathers and no per-use
// split. The weight stage of one (tap, chunk) goes through LDS double buffered (the 32-byte half
// rows of the packed layout of gga_sparse_pack_weight_split with kvol = 9); the next chunk's halo
// is requested from global memory before the taps of the current chunk run. 66 KB of LDS: two
// workgroups per CU.
#ifndef DC_PIPE_ON
#define DC_PIPE_ON 1
#endif
#ifndef DC_PIPE4_ON
#define DC_PIPE4_ON 1
#endif
#define DC_P4_MAX_TILES 256                       /* launches of at most this many tiles take the one-workgroup-per-CU form */

// NP = 3: three bf16 planes, six partial products (any fp32 input). NP = 2: two fp16 planes of the scaled operands, three
// partial products (see h2_split2); `amax` then points to {bits of max finite |x|, bits of max finite |w|}.
// MT: image rows (32-pixel M tiles) per wave; a workgroup has TR / MT waves. Shipped: MT = 2. (MT = 4 with 16-row tiles
// and four waves at 64 output channels - 0.5 instead of 0.67 LDS fragment reads per MFMA on two fp16 planes - needs 50
// spilled registers next to its 128 accumulators: 368 instead of 297 us per 64 -> 64 call incl. its absmax pass.
// Also measured on the two-plane 64-channel form, each within 1 % of the shipped 0.206 ms: three waves per SIMD (168
// registers, 18 spilled); a second fragment set read one tap ahead of the MFMAs; the next halo requested after stage 0's
// weight load instead of before it. LDS reads deliver 174 B/clk/CU with this access pattern
// (tools_dev/micro/lds_bw.hip); the kernel uses about half of that. A separate kernel that staged the weights of a whole
// kernel row per barrier (36 MFMAs and one barrier per row stage instead of 12 and one per tap, weights requested a full
// row stage ahead) measured 0.219 against 0.209 ms on the same box, alternating runs.)
// Several images of different sizes in ONE launch (gga_dense_conv3x3_levels: the tower convolutions of an FPN head share
// their weights over the levels, and all but the largest level are too small to fill the chip - 12 x 24 x 78 is 108
// tiles, 12 x 3 x 10 is 12): entry e owns the tiles [start[e], start[e + 1]) of the grid and brings its own input,
// output, size, absmax and (for output slices) weight operand. n = 0: the kernel's scalar arguments describe the one image.
#define DC_MAX_ENTRIES 16
struct DcLevels {
    int n;
    int start[DC_MAX_ENTRIES + 1];
    int H[DC_MAX_ENTRIES], W[DC_MAX_ENTRIES];
    const float* x[DC_MAX_ENTRIES];
    float* y[DC_MAX_ENTRIES];
    const uint16_t* w[DC_MAX_ENTRIES];
    const uint32_t* amax_x[DC_MAX_ENTRIES];
    const float* bias[DC_MAX_ENTRIES];       // per output channel of the entry, added in the epilogue; null: none
    double* stats[DC_MAX_ENTRIES];           // the entry's per-tile BatchNorm sums [tiles][2][cout]; null: none
    int transposed;                          // every entry walks its map transposed (tiles 32 pixels long along the image's H)
};

// (P4 form, round 3: 128 output channels in 8-row tiles on two fp16 planes with ONE workgroup per CU, so that its four waves, one
// per SIMD, have 512 registers each: room for the 128 accumulators AND two sets of the 12 fragments of a stage, see PIPE below.
// Measured against the two-workgroups-per-CU form of the same tile: launches of at most one tile per CU - the small FPN levels
// of the camera-only head, 62 x 54 maps - 69 against 83 us and 63 against 77; launches with more tiles than CUs 265 against 248
// and 217 against 202, where the second workgroup hides more than the pipelining wins. The launcher picks by tile count.)
template <int NT, int TR, int NP, int MT, int P4 = 0>
__global__ __launch_bounds__(TR / MT * 64, (P4 && NT == 4 && NP == 2) ? 1 : 2) void dense_conv3x3_x9_kernel(const float* __restrict__ X, const uint16_t* __restrict__ Wp,
                                                                 int B, int H, int W, int cin, int cout, int tiles_x,
                                                                 int tiles_y, float* __restrict__ Y, int ystride,
                                                                 int prow, int pcol, double* __restrict__ stats,
                                                                 const uint32_t* __restrict__ amax_x,
                                                                 const uint32_t* __restrict__ amax_w, DcBnBwd bn,
                                                                 DcLevels lv) {
    int tile = blockIdx.x;
    const float* bias = nullptr;
    if (lv.n) {                                          // which image this workgroup's tile belongs to (wave-uniform)
        int e = 0;
        while (e + 1 < lv.n && tile >= lv.start[e + 1]) ++e;
        tile -= lv.start[e];
        X = lv.x[e]; Y = lv.y[e]; Wp = lv.w[e]; amax_x = lv.amax_x[e]; bias = lv.bias[e]; stats = lv.stats[e];
        H = lv.H[e]; W = lv.W[e];
        prow = W; pcol = 1;
        if (lv.transposed) { prow = 1; pcol = W; const int t_ = H; H = W; W = t_; }      // tile space of the transposed walk
        tiles_x = (W + DC_TW - 1) / DC_TW; tiles_y = (H + TR - 1) / TR;
    }
    // H x W is the tile space (rows x 32-pixel columns); pixel (r, c) of it is pixel r*prow + c*pcol of
    // the image: (W, 1) for the image as stored, (1, image width) with H and W swapped for the
    // transposed walk (tiles 32 pixels long along the image's H), chosen by the caller per shape.
    // TR = 8: 256 threads own 8 rows x 32 pixels (64 output channels: two workgroups per CU; 128: one).
    // TR = 16 (128 output channels on maps with enough tiles): 512 threads own 16 rows - one workgroup per
    // CU but two waves per SIMD again (124 x 108: 324 instead of 379 us); on small maps the 16-row tiles
    // leave CUs idle (62 x 54: 523 instead of 366 us), so the launcher picks per shape.
    constexpr int NWAVES = TR / MT, THREADS = NWAVES * 64;
    constexpr int HP = (TR + 2) * DC_HW, NA = (HP * 4 + THREADS - 1) / THREADS;
    constexpr int CO = NT * 32;
    constexpr int BPL = CO * DC_ROWB, BSZ = NP * BPL, BPIECES = NP * CO * 2;
    constexpr int NB = (BPIECES + THREADS - 1) / THREADS;
    constexpr int APL = HP * DC_ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char As[NP * APL];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * BSZ];
    int sbx = 127, sbw = 127;
    if (NP == 2) { sbx = h2_scale_exp(*amax_x); sbw = h2_scale_exp(*amax_w); }
    const float xscale = h2_scale(sbx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int per_img = tiles_x * tiles_y;
    const int n_tiles = B * per_img;
    const int nchunks = cin / DC_CK;                  // 16-channel chunks
    mf_v16 acc[MT][NT];

    // halo piece e of this thread: pixel (tid + 256 e) / 4, channels 4 * ((tid + 256 e) % 4) .. +3 of the chunk
    float4 ra[NA];
    int aoff[NA];                              // float offset of the piece in its image, -1: outside (zeros)
    const float* Xb = X;
#define DC_TILE(T, TB, TY0, TX0) const int TB = (T) / per_img; const int TY0 = (((T) - TB * per_img) / tiles_x) * TR, TX0 = (((T) - TB * per_img) % tiles_x) * DC_TW;
#define DC_AOFF(TB, TY0, TX0) {                                                                                       \
        Xb = X + (int64_t)(TB) * H * W * cin;                                                                         \
        _Pragma("unroll") for (int e = 0; e < NA; ++e) {                                                           \
            const int f = tid + THREADS * e;                                                                              \
            const int hp = f >> 2, q = f & 3;                                                                         \
            const int hr = hp / DC_HW, hx = hp - hr * DC_HW;                                                          \
            const int iy = (TY0) + hr - 1, ix = (TX0) + hx - 1;                                                       \
            const bool ok = hp < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;                   \
            aoff[e] = ok ? (iy * prow + ix * pcol) * cin + q * 4 : -1;                                                          \
        } }
#define DC_LOAD_A(CH) _Pragma("unroll") for (int e = 0; e < NA; ++e) ra[e] = *reinterpret_cast<const float4*>(Xb + (aoff[e] >= 0 ? aoff[e] : 0) + (CH) * DC_CK);
#define DC_STORE_A()                                                                                                  \
    _Pragma("unroll") for (int e = 0; e < NA; ++e) {                                                               \
        const int f = tid + THREADS * e;                                                                                  \
        if (f < HP * 4) {                                                                                          \
            const float4 v = aoff[e] >= 0 ? ra[e] : make_float4(0.f, 0.f, 0.f, 0.f);                                   \
            unsigned char* dst = As + (f >> 2) * DC_ROWB + (f & 3) * 8;                                               \
            if (NP == 3) {                                                                                            \
                uint32_t lo1, lo2, lo3, hi1, hi2, hi3;                                                                \
                x9_split2(v.x, v.y, lo1, lo2, lo3); x9_split2(v.z, v.w, hi1, hi2, hi3);                               \
                *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                \
                *reinterpret_cast<uint2*>(dst + APL) = make_uint2(lo2, hi2);                                          \
                *reinterpret_cast<uint2*>(dst + (NP - 1) * APL) = make_uint2(lo3, hi3);                               \
            } else {                                                                                                  \
                uint32_t lo1, lo2, hi1, hi2;                                                                          \
                h2_split2s(v.x, v.y, xscale, lo1, lo2); h2_split2s(v.z, v.w, xscale, hi1, hi2);           \
                *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                \
                *reinterpret_cast<uint2*>(dst + APL) = make_uint2(lo2, hi2);                                          \
            }                                                                                                         \
        }                                                                                                             \
    }
    // weight stage (tap, 16-channel chunk c): piece f = (plane, column, 16-byte half) of the 32-byte half row.
    // Two register sets (named registers: an indexed array ends up in scratch): the weights of stage s + 3 are
    // requested at the start of stage s and written to LDS at the end of stage s + 1, so a load has two stages
    // (~1.5 us with two waves per SIMD) to come back from the L2 - with one stage the per-stage s_waitcnt was the
    // largest single loss of the kernel (ablation: 375 -> 303 us at 64 -> 64 without the loads).
    // (the 512-thread form is limited to 256 registers by its two waves per SIMD and keeps one set on three bf16 planes;
    // on two fp16 planes both sets fit: 253 -> 240 us at 128 -> 128, 16 x 124 x 108.)
    constexpr bool DEEP = (NP == 2 || !(NT == 4 && TR == 16)) && MT == 2;
    uint4 bq0, bq1, bq2, cq0, cq1, cq2;
    bq0 = bq1 = bq2 = cq0 = cq1 = cq2 = make_uint4(0, 0, 0, 0);
    // the packed stage (tap, 16-channel chunk) is contiguous and in LDS piece order (dense_pack_weight_kernel): a wave
    // load covers 1 KB of whole cache lines (the half-row layout of the sparse kernels touched 32 half-used lines per
    // load and kept the address unit busy for most of a stage)
#define DC_BLD(E, V) if ((E) < NB) { const int f = min(tid + THREADS * (E), BPIECES - 1); V = bsrc[f]; }
#define DC_LOAD_B(TAP, CH, V0, V1, V2) {                                                                              \
        const uint4* bsrc = reinterpret_cast<const uint4*>(Wp + ((int64_t)(TAP) * nchunks + (CH)) * (NP * CO * DC_CK)); \
        DC_BLD(0, V0) DC_BLD(1, V1) DC_BLD(2, V2) }
#define DC_BST(BUF, E, V) if ((E) < NB) { const int f = tid + THREADS * (E); if (f < BPIECES) *reinterpret_cast<uint4*>(Bs + (BUF) * BSZ + (f >> 1) * DC_ROWB + (f & 1) * 16) = V; }
#define DC_STORE_B(BUF, V0, V1, V2) { DC_BST(BUF, 0, V0) DC_BST(BUF, 1, V1) DC_BST(BUF, 2, V2) }
    static_assert(NB <= 3, "weight stage pieces per thread");

    // Stage (chunk, tap): fragments from LDS, 36 MFMAs, and meanwhile the weights of the stage
    // after next travel global -> registers -> LDS (three weight buffers; stage chunk*9 + tap lives
    // in buffer tap % 3 because 9 % 3 == 0); one barrier per stage. The nine taps are unrolled, so
    // tap offsets and buffer numbers are immediates.
    // fragments: the two M tiles' A planes, and the B planes of TWO N tiles at a time (with four N tiles all
    // twelve B fragments alive next to 128 accumulator registers do not fit 256 registers)
    // Round 3 (PIPE, the forms with two N tiles = 64 output channels): a second fragment set; stage s multiplies the set
    // that stage s - 1 read for it and reads the next stage's set between its own MFMAs (sched_group_barrier pins the
    // interleave: left alone the scheduler sinks every read to just before its use, which is what the round-2 attempt at
    // this measured). The weights of stage s + 1 are in LDS since the barrier before stage s (they are written a stage
    // early), the halo image is constant over a chunk; the first offset of a chunk reads its own fragments.
    constexpr bool PIPE = DC_PIPE_ON && NT == 2 && MT == 2 && NP == 2;       // (three planes: the second set spills)
    constexpr bool PIPE4 = P4 && NT == 4 && NP == 2;        // (MT = 2: 8-row tiles; MT = 4: 16-row tiles, four image rows per wave)
    mf_v8bf fa[MT][NP], fb[2][NP];
    mf_v8bf ga[MT][NP], gb[2][NP];
    mf_v8bf fb2[2][NP], gb2[2][NP];                      // PIPE4: the B fragments of N tiles 2 and 3
#define DC_READ_A_(FA, TAP) {                                                                                         \
        const unsigned char* Ap = As + ((MT * wave + (TAP) / 3) * DC_HW + r + (TAP) % 3) * DC_ROWB + h * 16;           \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int p = 0; p < NP; ++p)                 \
            FA[m][p] = *reinterpret_cast<const mf_v8bf*>(Ap + p * APL + m * DC_HW * DC_ROWB); }
#define DC_READ_B_(FB, TAP, T0) {                                                                                     \
        const unsigned char* Bp = Bs + ((TAP) % 3) * BSZ + r * DC_ROWB + h * 16 + (T0) * 32 * DC_ROWB;                \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int p = 0; p < NP; ++p)                  \
            FB[t][p] = *reinterpret_cast<const mf_v8bf*>(Bp + p * BPL + t * 32 * DC_ROWB); }
#define DC_READ_A(TAP) DC_READ_A_(fa, TAP)
#define DC_READ_B(TAP, T0) DC_READ_B_(fb, TAP, T0)
    // partial products smallest first; tiles innermost so consecutive MFMAs never share an accumulator
#define DC_MM1_(FA, FB, T0, PA, PB) _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[m][(T0) + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[m][PA], FB[t][PB], acc[m][(T0) + t], 0, 0, 0);
#define DC_MH1_(FA, FB, T0, PA, PB) _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[m][(T0) + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_v8h, FA[m][PA]), __builtin_bit_cast(mf_v8h, FB[t][PB]), acc[m][(T0) + t], 0, 0, 0);
#define DC_MMA3_(FA, FB, T0) DC_MM1_(FA, FB, T0, 0, NP - 1) DC_MM1_(FA, FB, T0, 1, 1) DC_MM1_(FA, FB, T0, NP - 1, 0) DC_MM1_(FA, FB, T0, 0, 1) DC_MM1_(FA, FB, T0, 1, 0) DC_MM1_(FA, FB, T0, 0, 0)
#define DC_MMA_(FA, FB, T0) { if (NP == 3) { DC_MMA3_(FA, FB, T0) } else { DC_MH1_(FA, FB, T0, 0, 1) DC_MH1_(FA, FB, T0, 1, 0) DC_MH1_(FA, FB, T0, 0, 0) } }
#define DC_MMA(T0) DC_MMA_(fa, fb, T0)
    // groups of (MFMAs, LDS reads) of a pipelined stage: 12 MFMAs and 8 reads on two planes, 24 and 12 on three
    constexpr int PG = NP == 2 ? 4 : 12, PG_M = NP == 2 ? 3 : 2, PG_R = NP == 2 ? 2 : 1;

    // Persistent workgroups: tiles blockIdx.x, blockIdx.x + gridDim.x, ... as one uninterrupted
    // stream of stages - the halo of the next tile's first chunk is requested during the last
    // chunk of the current tile, the weight stages wrap around, and the output stores of a tile
    // drain while the next tile computes.
    // Measured at [16,64,248,216] -> 64 (63 GFLOP): 0.416 ms = 152 TFLOP/s-equivalent (MIOpen fp32:
    // 0.62 ms forward, 0.54 ms backward-data). With the fragment reads, the halo staging, the
    // weight copies and the barriers compiled out the MFMA stream alone takes 0.363 ms, so the
    // kernel is within 15 % of what its MFMA issue pattern delivers here; reading the next tap's
    // fragments ahead of the MFMAs (two register sets), one tile per workgroup instead of persistent
    // ones, and dropping the per-stage barrier all measured 0.414-0.420 ms.
    if (tile >= n_tiles) return;
    {
        DC_TILE(tile, tb, ty0, tx0)
        DC_AOFF(tb, ty0, tx0)
    }
    DC_LOAD_A(0);
    DC_LOAD_B(0, 0, bq0, bq1, bq2);
    DC_LOAD_B(1, 0, cq0, cq1, cq2);
    DC_STORE_A();
    DC_STORE_B(0, bq0, bq1, bq2);
    DC_STORE_B(1, cq0, cq1, cq2);
    if (DEEP) { DC_LOAD_B(2, 0, bq0, bq1, bq2); }     // stage 2: written to LDS at the end of stage 0
    __syncthreads();
    bool first = true;
    for (; tile < n_tiles; tile += gridDim.x) {
        DC_TILE(tile, b, y0, x0)
        const bool more_tiles = tile + (int)gridDim.x < n_tiles;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.0f;
        // stage s = chunk * 9 + tap loads the weights of stage s + 3 into register set (s + 1) % 2 and writes those of
        // stage s + 2 from set s % 2 into LDS buffer (s + 2) % 3; chunks come in pairs so that the set of every
        // stage is fixed at compile time (nchunks is even: cin % 32 == 0)
#define DC_CHUNK_HEAD(CH)                                                                                             \
            if (!first) {                              /* every wave passed the barrier of the previous stage */      \
                DC_STORE_A();                                                                                         \
                __syncthreads();                                                                                      \
            }                                                                                                         \
            first = false;                                                                                            \
            if ((CH) + 1 < nchunks) { DC_LOAD_A((CH) + 1); }                                                          \
            else if (more_tiles) {                     /* first chunk of the next tile */                             \
                DC_TILE(tile + (int)gridDim.x, nb_, ny0, nx0)                                                         \
                DC_AOFF(nb_, ny0, nx0)                                                                                \
                DC_LOAD_A(0);                                                                                         \
            }
#define DC_STAGE(TAP, CH, L0, L1, L2, S0, S1, S2, CA, CB, XA, XB, CB2, XB2) {                                         \
            const bool last_chunk = (CH) + 1 >= nchunks;                                                              \
            const bool more3 = (TAP) + 3 < 9 || !last_chunk || more_tiles;     /* a stage three ahead exists */       \
            const bool more2 = (TAP) + 2 < 9 || !last_chunk || more_tiles;                                            \
            if (DEEP) {                                                                                               \
                if (more3) {                                                                                          \
                    if ((TAP) + 3 < 9) { DC_LOAD_B((TAP) + 3, (CH), L0, L1, L2); }                                    \
                    else { DC_LOAD_B((TAP) + 3 - 9, last_chunk ? 0 : (CH) + 1, L0, L1, L2); }                         \
                }                                                                                                     \
            } else if (more2) {       /* one register set: stage s + 2 requested now, written at the end of this stage */ \
                if ((TAP) + 2 < 9) { DC_LOAD_B((TAP) + 2, (CH), bq0, bq1, bq2); }                                     \
                else { DC_LOAD_B((TAP) + 2 - 9, last_chunk ? 0 : (CH) + 1, bq0, bq1, bq2); }                          \
            }                                                                                                         \
            if (PIPE4) {                                                                                              \
                if ((TAP) == 0) { DC_READ_A_(CA, 0); DC_READ_B_(CB, 0, 0); DC_READ_B_(CB2, 0, 2); }                    \
                if ((TAP) < 8) { DC_READ_A_(XA, (TAP) + 1); DC_READ_B_(XB, (TAP) + 1, 0); DC_READ_B_(XB2, (TAP) + 1, 2); } \
                DC_MMA_(CA, CB, 0)                                                                                    \
                DC_MMA_(CA, CB2, 2)                                                                                   \
                if ((TAP) < 8) {      /* MT * 12 MFMAs, 2 MT + 8 reads: 2 : 1 at MT = 2, 3 : 1 at MT = 4 */                \
                    _Pragma("unroll") for (int g_ = 0; g_ < 2 * MT + 8; ++g_) {                                       \
                        __builtin_amdgcn_sched_group_barrier(0x008, MT == 4 ? 3 : 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
                }                                                                                                     \
            } else if (PIPE) {                                                                                        \
                if ((TAP) == 0) { DC_READ_A_(CA, 0); DC_READ_B_(CB, 0, 0); }                                           \
                if ((TAP) < 8) { DC_READ_A_(XA, (TAP) + 1); DC_READ_B_(XB, (TAP) + 1, 0); }                            \
                DC_MMA_(CA, CB, 0)                                                                                    \
                if ((TAP) < 8) {                                                                                      \
                    _Pragma("unroll") for (int g_ = 0; g_ < PG; ++g_) {                                               \
                        __builtin_amdgcn_sched_group_barrier(0x008, PG_M, 0); __builtin_amdgcn_sched_group_barrier(0x100, PG_R, 0); } \
                }                                                                                                     \
            } else {                                                                                                  \
            DC_READ_A(TAP);                                                                                           \
            _Pragma("unroll") for (int t0 = 0; t0 < NT; t0 += 2) {                                                    \
                DC_READ_B(TAP, t0);                                                                                   \
                DC_MMA(t0)                                                                                            \
            } }                                                                                                       \
            if (more2) { if (DEEP) { DC_STORE_B(((TAP) + 2) % 3, S0, S1, S2); } else { DC_STORE_B(((TAP) + 2) % 3, bq0, bq1, bq2); } } \
            __syncthreads(); }
#define DC_EVEN(TAP, CH) DC_STAGE(TAP, CH, cq0, cq1, cq2, bq0, bq1, bq2, fa, fb, ga, gb, fb2, gb2)      /* even stage: load set 1, store set 0 */
#define DC_ODD(TAP, CH) DC_STAGE(TAP, CH, bq0, bq1, bq2, cq0, cq1, cq2, ga, gb, fa, fb, gb2, fb2)
        for (int ch = 0; ch < nchunks; ch += 2) {
            DC_CHUNK_HEAD(ch)
            DC_EVEN(0, ch) DC_ODD(1, ch) DC_EVEN(2, ch) DC_ODD(3, ch) DC_EVEN(4, ch) DC_ODD(5, ch) DC_EVEN(6, ch) DC_ODD(7, ch) DC_EVEN(8, ch)
            DC_CHUNK_HEAD(ch + 1)
            DC_ODD(0, ch + 1) DC_EVEN(1, ch + 1) DC_ODD(2, ch + 1) DC_EVEN(3, ch + 1) DC_ODD(4, ch + 1) DC_EVEN(5, ch + 1) DC_ODD(6, ch + 1) DC_EVEN(7, ch + 1) DC_ODD(8, ch + 1)
        }
#undef DC_CHUNK_HEAD
#undef DC_STAGE
#undef DC_EVEN
#undef DC_ODD
        {
            // back from the scaled operands (two exact powers of two), and the bias of the lane's output channels
            const float dx = NP == 2 ? h2_descale(sbx) : 1.0f, dw = NP == 2 ? h2_descale(sbw) : 1.0f;
            float bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = bias ? bias[t * 32 + r] : 0.0f;
            if (NP == 2 || bias) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[m][t][i] = acc[m][t][i] * dx * dw + bv[t];
            }
        }
        // D layout of 32x32x16: register v of lane l holds row (v/4)*8 + (l/32)*4 + v%4 (= pixel of the M tile's row), column l%32
        float s1[NT], s2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { s1[t] = 0.0f; s2[t] = 0.0f; }
        if (bn.y) {                                        // see DcBnBwd: ReLU mask and the BatchNorm backward sums
            float bsc[NT], bsh[NT], bmu[NT], biv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = t * 32 + r;
                bmu[t] = bn.mean[c]; biv[t] = bn.invstd[c];
                gga_bn_scale_shift(bn.gamma ? bn.gamma[c] : 1.0f, bn.beta ? bn.beta[c] : 0.0f, bmu[t], biv[t], bsc[t], bsh[t]);
            }
            // 32 values of y per lane are requested before the first of them is used (a load per store serialises on
            // the memory latency: + 110 .. 250 us per launch)
            constexpr int VB = 32 / NT;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = y0 + MT * wave + m;
                if (oy >= H) continue;
#pragma unroll
                for (int v0 = 0; v0 < 16; v0 += VB) {
                    float yv[VB][NT];
#pragma unroll
                    for (int j = 0; j < VB; ++j) {
                        const int ox = x0 + ((v0 + j) >> 2) * 8 + h * 4 + ((v0 + j) & 3);
                        const float* src = bn.y + ((int64_t)b * H * W + oy * prow + (ox < W ? ox : W - 1) * pcol) * bn.ystride;
#pragma unroll
                        for (int t = 0; t < NT; ++t) yv[j][t] = src[t * 32 + r];
                    }
#pragma unroll
                    for (int j = 0; j < VB; ++j) {
                        const int ox = x0 + ((v0 + j) >> 2) * 8 + h * 4 + ((v0 + j) & 3);
                        if (ox >= W) continue;
                        float* dst = Y + ((int64_t)b * H * W + oy * prow + ox * pcol) * ystride;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float g = fmaf(yv[j][t], bsc[t], bsh[t]) > 0.0f ? acc[m][t][v0 + j] : 0.0f;
                            dst[t * 32 + r] = g;
                            s1[t] += g; s2[t] += g * ((yv[j][t] - bmu[t]) * biv[t]);
                        }
                    }
                }
            }
        } else
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + MT * wave + m;
            if (oy >= H) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ox = x0 + (v >> 2) * 8 + h * 4 + (v & 3);
                if (ox >= W) continue;
                float* dst = Y + ((int64_t)b * H * W + oy * prow + ox * pcol) * ystride;       // ystride > cout: a channel slice of a wider tensor
#pragma unroll
                for (int t = 0; t < NT; ++t) dst[t * 32 + r] = acc[m][t][v];
            }
        }
        if (stats) {
            // per-channel sum and sum of squares of the tile's outputs (the batch statistics of the
            // BatchNorm that follows, so it need not read y again): lane sums over its pixels, the
            // two half waves and the four waves are folded through LDS, one f64 row pair per tile.
            if (!bn.y)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool rowok = y0 + MT * wave + m < H;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const bool ok = rowok && x0 + (v >> 2) * 8 + h * 4 + (v & 3) < W;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float a = ok ? acc[m][t][v] : 0.0f;
                        s1[t] += a; s2[t] += a * a;
                    }
                }
            }
            float* red = reinterpret_cast<float*>(As);       // free: the last stage ended with a barrier
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                s1[t] += __shfl_xor(s1[t], 32);
                s2[t] += __shfl_xor(s2[t], 32);
                if (h == 0) { red[(wave * 2 + 0) * CO + t * 32 + r] = s1[t]; red[(wave * 2 + 1) * CO + t * 32 + r] = s2[t]; }
            }
            __syncthreads();
            if (tid < 2 * CO) {
                const int which = tid / CO, c = tid - which * CO;
                if (c < cout) {
                    double a = 0.0;
#pragma unroll
                    for (int w_ = 0; w_ < NWAVES; ++w_) a += (double)red[(w_ * 2 + which) * CO + c];
                    stats[((int64_t)tile * 2 + which) * cout + c] = a;
                }
            }
            __syncthreads();                                  // red is the next tile's halo buffer
        }
    }
#undef DC_READ_A
#undef DC_READ_B
#undef DC_READ_A_
#undef DC_READ_B_
#undef DC_MM1_
#undef DC_MH1_
#undef DC_MMA3_
#undef DC_MMA_
#undef DC_MMA
#undef DC_LOAD_A
#undef DC_STORE_A
#undef DC_LOAD_B
#undef DC_STORE_B
#undef DC_BLD
#undef DC_BST
#undef DC_TILE
#undef DC_AOFF
}

// Packs a 3x3 convolution weight straight from the framework tensor (any strides, e.g. the
// channels-last memory of a [cout, cin, 3, 3] parameter) into the split layout of
// gga_sparse_pack_weight_split with kvol = 9; `backward` builds the operand of the backward-data
// convolution instead (taps reversed, channel roles swapped). One thread per (tap, chunk, col, ch).
__global__ __launch_bounds__(256) void dense_pack_weight_kernel(const float* __restrict__ W, int64_t s_co, int64_t s_ci,
                                                               int64_t s_ky, int64_t s_kx, int cin, int cout,
                                                               int backward, int nt, int64_t total, int np,
                                                               const uint32_t* __restrict__ amax_w,
                                                               uint16_t* __restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int n_in = backward ? cout : cin, n_out = backward ? cin : cout;      // channels of the convolution being run
    const int co = 32 * nt, nchunks = (n_in + MF_TK - 1) / MF_TK;
    const int ch = (int)(i & 31);
    const int col = (int)((i >> 5) % co);
    const int64_t stage = (i >> 5) / co;                                        // (tap, 32-channel chunk)
    const int tap = (int)(stage / nchunks), chunk = (int)(stage - (int64_t)tap * nchunks);
    const int c = chunk * MF_TK + ch;
    float v = 0.0f;
    if (c < n_in && col < n_out) {
        const int t = backward ? 8 - tap : tap;
        const int ky = t / 3, kx = t - ky * 3;
        const int wco = backward ? c : col, wci = backward ? col : c;
        v = W[wco * s_co + wci * s_ci + ky * s_ky + kx * s_kx];
    }
    // dense layout: [tap][16-channel chunk][plane][column][16 channels] - one contiguous block per kernel stage,
    // in the order the kernel's threads copy it to LDS
    const int64_t stage16 = (int64_t)tap * (2 * nchunks) + (c >> 4);
    uint16_t* dst = P + stage16 * (np * (int64_t)co * 16) + (int64_t)col * 16 + (c & 15);
    if (np == 3) {
        uint32_t p1, p2, p3;
        x9_split(v, p1, p2, p3);
        dst[0] = (uint16_t)p1; dst[(int64_t)co * 16] = (uint16_t)p2; dst[2 * (int64_t)co * 16] = (uint16_t)p3;
    } else {                                  // two fp16 planes of the scaled weight (h2_split2)
        uint32_t w0, w1;
        h2_split2(v * h2_scale(h2_scale_exp(*amax_w)), 0.0f, w0, w1);
        dst[0] = (uint16_t)(w0 & 0xFFFFu); dst[(int64_t)co * 16] = (uint16_t)(w1 & 0xFFFFu);
    }
}

extern "C" int gga_dense_conv3x3_pack(const float* weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                      int64_t stride_kx, int cin, int cout, int backward, void* packed, void* stream) {
    return gga_dense_conv3x3_pack_planes(weight, stride_co, stride_ci, stride_ky, stride_kx, cin, cout, backward, 3, nullptr,
                                         packed, stream);
}

extern "C" int gga_dense_conv3x3_pack_planes(const float* weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                             int64_t stride_kx, int cin, int cout, int backward, int planes,
                                             const uint32_t* amax_weight, void* packed, void* stream) {
    GGA_REQUIRE(weight && packed, "gga_dense_conv3x3_pack: null pointer argument");
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_weight), "gga_dense_conv3x3_pack: planes must be 3 (bf16) or 2 (fp16, with amax_weight)");
    const int n_in = backward ? cout : cin, n_out = backward ? cin : cout;
    GGA_REQUIRE(n_in >= 1 && n_out >= 1 && n_out <= 128, "gga_dense_conv3x3_pack: bad sizes (%d -> %d)", n_in, n_out);
    const int64_t total = (int64_t)(gga_sparse_split_weight_bytes(9, n_in, n_out) / (3 * sizeof(uint16_t)));
    hipLaunchKernelGGL(dense_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, stride_co, stride_ci, stride_ky, stride_kx, cin, cout, backward, mf_nt(n_out), total, planes,
                       amax_weight, (uint16_t*)packed);
    GGA_CHECK_LAUNCH("dense_pack_weight_kernel");
    return GGA_OK;
}

// rows per tile: 16 only for 128 output channels, when that still gives every CU a workgroup or two, and when the rounds
// the chip needs for the tiles do not eat what the 16-row form wins per tile. Both forms run one 128-column workgroup per CU
// (256 slots); a 16-row tile takes ~1.87 x the time of an 8-row tile (7 % better per pixel). 8 x 200 x 176 (the shipped config's
// second stage): 624 tiles = 3 rounds at 81 % against 1200 = 5 rounds at 94 % -> 8 rows (272 against 286 us, same box);
// 16 x 124 x 108: 2 against 4 rounds -> 16 rows (231 against 249 us); 16 x 248 x 216: 7 against 14 -> 16 rows (464 against 490).
int dc_tile_rows(int B, int H, int W, int cout, int planes) {
    if (dc_ws_enabled(planes)) return cout == 128 ? 8 : 16;          // dense_conv_ws.hip: four consumer waves x 128 accumulators
    if (cout != 128) return 8;
    static const int forced = getenv("GGA_DC_TILE_ROWS") ? atoi(getenv("GGA_DC_TILE_ROWS")) : 0;      // A/B switch: 8 or 16
    if (forced == 8 || forced == 16) return forced;
    const int64_t cols = (W + DC_TW - 1) / DC_TW;
    const int64_t t16 = (int64_t)B * cols * ((H + 15) / 16), t8 = (int64_t)B * cols * ((H + 7) / 8);
    if (t16 < 384) return 8;
    const int64_t r16 = (t16 + 255) / 256, r8 = (t8 + 255) / 256;
    return 1000 * r16 <= 535 * r8 ? 16 : 8;
}

extern "C" int64_t gga_dense_conv3x3_tiles_planes(int B, int H, int W, int cout, int planes) {   // H, W of the tile space (swapped when transposed)
    const int tr = dc_tile_rows(B, H, W, cout, planes);
    return (int64_t)B * ((W + DC_TW - 1) / DC_TW) * ((H + tr - 1) / tr);
}

// rows of `stats` (per slice) of a launch over n_slices 128-channel slices (1: gga_dense_conv3x3_bn_bwd): the lock-step kernel
// leaves one row per tile, the producer / consumer form one per workgroup
extern "C" int64_t gga_dense_conv3x3_stat_rows(int B, int H, int W, int cout, int planes, int n_slices) {
    const int64_t tiles = gga_dense_conv3x3_tiles_planes(B, H, W, cout, planes);
    if (dc_ws_enabled(planes)) return dc_ws_grid(tiles * (n_slices > 1 ? n_slices : 1));
    return tiles;
}

extern "C" int gga_dense_conv3x3_tile_rows(int B, int H, int W, int cout, int planes) { return dc_tile_rows(B, H, W, cout, planes); }

extern "C" int64_t gga_dense_conv3x3_tiles(int B, int H, int W, int cout) { return gga_dense_conv3x3_tiles_planes(B, H, W, cout, 3); }

// Whether the BatchNorm-backward epilogue (gga_dense_conv3x3_bn_bwd) is cheaper than the reduce pass it replaces. Lock-step forms
// (three planes; measured inside the PointPillars step, 16 frames): 64 output channels (two workgroups per CU, the other one's
// MFMAs cover the epilogue's loads) + 0 us per launch against 100 us of reduce pass; 128 channels in 16-row tiles + 25 .. 100 us
// against 55 .. 200; 128 channels in 8-row tiles (small maps, one workgroup per CU) + 33 us against 15: not there. Producer /
// consumer form (two planes): everywhere - same box, alternating runs of the bench: PointPillars step 34.65 / 34.67 ms with the
// epilogue on every launch against 34.90 / 34.92 with none and 35.06 / 34.89 with the lock-step rule; gga_kitti_config.py
// 54.98 / 55.11 against 54.88 / 55.07 and 54.76.
extern "C" int gga_dense_conv3x3_bn_bwd_pays_planes(int B, int H, int W, int cout, int planes) {
    static const int forced = getenv("GGA_DC_BN_BWD_PAYS") ? atoi(getenv("GGA_DC_BN_BWD_PAYS")) : -1;      // A/B switch: 0 or 1
    if (forced == 0 || forced == 1) return forced;
    if (dc_ws_enabled(planes)) return 1;
    return cout == 64 || dc_tile_rows(B, H, W, cout, planes) == 16;
}

extern "C" int gga_dense_conv3x3_bn_bwd_pays(int B, int H, int W, int cout) { return gga_dense_conv3x3_bn_bwd_pays_planes(B, H, W, cout, 3); }

extern "C" int gga_dense_conv3x3_slice(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                                       float* y, int64_t y_pixel_stride, int transposed, double* stats, void* stream_) {
    return gga_dense_conv3x3_planes(x, split_weight, B, H, W, cin, cout, y, y_pixel_stride, transposed, stats, 3, nullptr, nullptr,
                                    stream_);
}

extern "C" int gga_dense_conv3x3_planes(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                                        float* y, int64_t y_pixel_stride, int transposed, double* stats, int planes,
                                        const uint32_t* amax_x, const uint32_t* amax_weight, void* stream_) {
    return gga_dense_conv3x3_bn_bwd(x, split_weight, B, H, W, cin, cout, y, y_pixel_stride, transposed, stats, planes, amax_x,
                                    amax_weight, nullptr, 0, nullptr, nullptr, nullptr, nullptr, stream_);
}

extern "C" int gga_dense_conv3x3_bn_bwd(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                                        float* y, int64_t y_pixel_stride, int transposed, double* stats, int planes,
                                        const uint32_t* amax_x, const uint32_t* amax_weight, const float* bn_x,
                                        int64_t bn_x_pixel_stride, const float* bn_gamma, const float* bn_beta,
                                        const float* bn_mean, const float* bn_invstd, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(!bn_x || (stats && bn_mean && bn_invstd && bn_x_pixel_stride >= cout && bn_x_pixel_stride < 2147483647ll),
                "gga_dense_conv3x3_bn_bwd: the BatchNorm epilogue needs stats, the saved mean / invstd and a pixel stride >= cout");
    DcBnBwd bn;
    bn.y = bn_x; bn.gamma = bn_gamma; bn.beta = bn_beta; bn.mean = bn_mean; bn.invstd = bn_invstd; bn.ystride = (int)bn_x_pixel_stride;
    DcLevels lv;
    lv.n = 0;
    GGA_REQUIRE(x && split_weight && y, "gga_dense_conv3x3: null pointer argument");
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_x && amax_weight),
                "gga_dense_conv3x3: planes must be 3 (bf16) or 2 (fp16, with the operands' absmax bits)");
    GGA_REQUIRE(y_pixel_stride >= cout && y_pixel_stride < 2147483647ll, "gga_dense_conv3x3: y pixel stride %lld < cout",
                (long long)y_pixel_stride);
    GGA_REQUIRE(B >= 1 && H >= 1 && W >= 1 && cin >= 32 && cin % 32 == 0 && (cout == 64 || cout == 128) &&
                    (int64_t)H * W * cin < 2147483647ll,
                "gga_dense_conv3x3: need cin %% 32 == 0 and cout 64 or 128 (got %d -> %d)", cin, cout);
    const int prow = transposed ? 1 : W, pcol = transposed ? W : 1;
    if (transposed) { const int t = H; H = W; W = t; }          // tile space of the transposed walk
    // gga_dense_conv3x3_stat_rows / _tile_rows describe the producer / consumer grid whenever that form is enabled for `planes`: a
    // shape it cannot take must not fall back silently to the lock-step kernel (other tile size, other number of stats rows)
    GGA_REQUIRE(!dc_ws_enabled(planes) || cin <= DC_WS_MAX_CIN,
                "gga_dense_conv3x3: %d input channels on two planes (the producer / consumer form takes <= %d; use planes = 3 or GGA_DC_WS=0)",
                cin, DC_WS_MAX_CIN);
    if (dc_ws_enabled(planes)) {      // two fp16 planes: the producer / consumer form (dense_conv_ws.hip)
        hipEvent_t* tev = gga_timing_acquire(GGA_TIME_DENSE_CONV, GGA_TIMING_CONV_KEY(cin, cout, (int64_t)H * W));
        GGA_TIME_START(tev, stream);
        const int rc = dc_launch_ws(x, split_weight, B, H, W, cin, cout, y, (int)y_pixel_stride, prow, pcol, stats, amax_x, amax_weight, bn,
                                    nullptr, stream);
        GGA_TIME_STOP(tev, stream);
        return rc;
    }
    const int trows = dc_tile_rows(B, H, W, cout, planes);
    const int tx = (W + DC_TW - 1) / DC_TW, ty = (H + trows - 1) / trows;
    const int64_t n_tiles = (int64_t)B * tx * ty;
    GGA_REQUIRE(n_tiles < 2147483647ll, "gga_dense_conv3x3: too many tiles");
    // One tile per workgroup. The kernel also runs as persistent workgroups (grid < tiles, same speed
    // in isolation), but inside the train step a persistent grid starts while the previous kernel's
    // tail still occupies some CUs and the static tile split then leaves stragglers (one bench run in
    // two measured 89.7 instead of 73.8 ms per step); the hardware dispatcher balances one-tile workgroups.
    const bool pipe4 = DC_PIPE4_ON && planes == 2 && cout == 128 && trows == 8 && n_tiles <= DC_P4_MAX_TILES;
    // (16-row tiles as four waves x four image rows on the same one-workgroup-per-CU form - 256 accumulators next to two fragment
    // sets - need more than 512 registers: 123 spilled dwords, 339 against 241 us at 16 x 124 x 108; not instantiated)
    const dim3 grid((unsigned)n_tiles), block(trows * 32);
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_DENSE_CONV, GGA_TIMING_CONV_KEY(cin, cout, (int64_t)H * W));
    GGA_TIME_START(tev, stream);
#define DC_GO(NT_, TR_, NP_) hipLaunchKernelGGL((dense_conv3x3_x9_kernel<NT_, TR_, NP_, 2>), grid, block, 0, stream, x, (const uint16_t*)split_weight, B, H, W, cin, cout, tx, ty, y, (int)y_pixel_stride, prow, pcol, stats, amax_x, amax_weight, bn, lv)
    if (planes == 3) {
        if (cout == 64) DC_GO(2, 8, 3);
        else if (trows == 16) DC_GO(4, 16, 3);
        else DC_GO(4, 8, 3);
    } else {
        if (cout == 64) DC_GO(2, 8, 2);
        else if (trows == 16) DC_GO(4, 16, 2);
        else if (pipe4) hipLaunchKernelGGL((dense_conv3x3_x9_kernel<4, 8, 2, 2, 1>), grid, block, 0, stream, x, (const uint16_t*)split_weight, B, H, W, cin, cout, tx, ty, y, (int)y_pixel_stride, prow, pcol, stats, amax_x, amax_weight, bn, lv);
        else DC_GO(4, 8, 2);
    }
#undef DC_GO
    GGA_CHECK_LAUNCH("dense_conv3x3_x9_kernel");
    GGA_TIME_STOP(tev, stream);
    return GGA_OK;
}

extern "C" int gga_dense_conv3x3_levels(int n_entries, const float* const* x, const int32_t* heights, const int32_t* widths,
                                        const void* const* split_weight, int B, int cin, int cout, float* const* y,
                                        int64_t y_pixel_stride, int planes, const uint32_t* const* amax_x,
                                        const uint32_t* amax_weight, const float* const* bias, int tile_rows, int transposed,
                                        double* const* stats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(tile_rows == 8 || (tile_rows == 16 && cout == 128), "gga_dense_conv3x3_levels: tile_rows 8, or 16 with cout 128");
    GGA_REQUIRE(n_entries >= 1 && n_entries <= DC_MAX_ENTRIES, "gga_dense_conv3x3_levels: 1 .. %d entries (got %d)", DC_MAX_ENTRIES,
                n_entries);
    GGA_REQUIRE(x && heights && widths && split_weight && y, "gga_dense_conv3x3_levels: null pointer argument");
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_x && amax_weight),
                "gga_dense_conv3x3_levels: planes must be 3 (bf16) or 2 (fp16, with the operands' absmax bits)");
    GGA_REQUIRE(B >= 1 && cin >= 32 && cin % 32 == 0 && (cout == 64 || cout == 128) && y_pixel_stride >= cout &&
                    y_pixel_stride < 2147483647ll, "gga_dense_conv3x3_levels: need cin %% 32 == 0 and cout 64 or 128 (got %d -> %d)",
                cin, cout);
    static const int ws_slices = getenv("GGA_DC_WS_SLICES") ? atoi(getenv("GGA_DC_WS_SLICES")) : 1;       // A/B switch
    GGA_REQUIRE(!dc_ws_enabled(planes) || cin <= DC_WS_MAX_CIN,
                "gga_dense_conv3x3_levels: %d input channels on two planes (the producer / consumer form takes <= %d)", cin, DC_WS_MAX_CIN);
    if (ws_slices && dc_ws_enabled(planes) && cout == 128 && tile_rows == 8 && !bias) {
        // 128-channel slices of ONE convolution's output (same input, one absmax): the producer / consumer form walks them as one grid
        bool same = true;
        for (int e = 1; e < n_entries; ++e)
            same = same && x[e] == x[0] && heights[e] == heights[0] && widths[e] == widths[0] && amax_x[e] == amax_x[0];
        if (same) {
            GGA_REQUIRE(x[0] && heights[0] >= 1 && widths[0] >= 1 && (int64_t)heights[0] * widths[0] * cin < 2147483647ll && amax_x[0],
                        "gga_dense_conv3x3_levels: bad entry 0");
            DcSlices sl;
            sl.n = n_entries;
            for (int e = 0; e < n_entries; ++e) {
                GGA_REQUIRE(y[e] && split_weight[e], "gga_dense_conv3x3_levels: bad entry %d", e);
                sl.w[e] = (const uint16_t*)split_weight[e]; sl.y[e] = y[e]; sl.stats[e] = stats ? stats[e] : nullptr;
            }
            DcBnBwd nobn;
            nobn.y = nullptr; nobn.gamma = nobn.beta = nobn.mean = nobn.invstd = nullptr; nobn.ystride = 0;
            const int H = transposed ? widths[0] : heights[0], W = transposed ? heights[0] : widths[0];      // tile space
            return dc_launch_ws(x[0], split_weight[0], B, H, W, cin, cout, y[0], (int)y_pixel_stride, transposed ? 1 : widths[0],
                                transposed ? widths[0] : 1, stats ? stats[0] : nullptr, amax_x[0], amax_weight, nobn, &sl, stream);
        }
    }
    DcLevels lv;
    lv.n = n_entries;
    int64_t total = 0;
    for (int e = 0; e < n_entries; ++e) {
        GGA_REQUIRE(x[e] && y[e] && split_weight[e] && heights[e] >= 1 && widths[e] >= 1 &&
                        (int64_t)heights[e] * widths[e] * cin < 2147483647ll && (planes == 3 || amax_x[e]),
                    "gga_dense_conv3x3_levels: bad entry %d", e);
        lv.start[e] = (int)total;
        lv.H[e] = heights[e]; lv.W[e] = widths[e];
        lv.x[e] = x[e]; lv.y[e] = y[e]; lv.w[e] = (const uint16_t*)split_weight[e];
        lv.amax_x[e] = planes == 2 ? amax_x[e] : nullptr;
        lv.bias[e] = bias ? bias[e] : nullptr;
        lv.stats[e] = stats ? stats[e] : nullptr;
        const int th = transposed ? widths[e] : heights[e], tw = transposed ? heights[e] : widths[e];      // tile space
        total += (int64_t)B * ((tw + DC_TW - 1) / DC_TW) * ((th + tile_rows - 1) / tile_rows);
        GGA_REQUIRE(total < 2147483647ll, "gga_dense_conv3x3_levels: too many tiles");
    }
    lv.start[n_entries] = (int)total;
    lv.transposed = transposed ? 1 : 0;
    for (int e = n_entries + 1; e <= DC_MAX_ENTRIES; ++e) lv.start[e] = (int)total;
    DcBnBwd bn;
    bn.y = nullptr; bn.gamma = bn.beta = bn.mean = bn.invstd = nullptr; bn.ystride = 0;
    const dim3 grid((unsigned)total), block(tile_rows * 32);
#define DC_LV(NT_, TR_, NP_) hipLaunchKernelGGL((dense_conv3x3_x9_kernel<NT_, TR_, NP_, 2>), grid, block, 0, stream, x[0], (const uint16_t*)split_weight[0], B, heights[0], widths[0], cin, cout, 1, 1, y[0], (int)y_pixel_stride, widths[0], 1, (double*)nullptr, planes == 2 ? amax_x[0] : nullptr, amax_weight, bn, lv)
    if (planes == 3) { if (cout == 64) DC_LV(2, 8, 3); else if (tile_rows == 16) DC_LV(4, 16, 3); else DC_LV(4, 8, 3); }
    else {
        if (cout == 64) DC_LV(2, 8, 2);
        else if (tile_rows == 16) DC_LV(4, 16, 2);
        else if (DC_PIPE4_ON && total <= DC_P4_MAX_TILES)
            hipLaunchKernelGGL((dense_conv3x3_x9_kernel<4, 8, 2, 2, 1>), grid, block, 0, stream, x[0], (const uint16_t*)split_weight[0], B, heights[0], widths[0], cin, cout, 1, 1, y[0], (int)y_pixel_stride, widths[0], 1, (double*)nullptr, planes == 2 ? amax_x[0] : nullptr, amax_weight, bn, lv);
        else DC_LV(4, 8, 2);
    }
#undef DC_LV
    GGA_CHECK_LAUNCH("dense_conv3x3_x9_kernel (levels)");
    return GGA_OK;
}

extern "C" int gga_dense_conv3x3_stats(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                                       float* y, double* stats, void* stream) {
    return gga_dense_conv3x3_slice(x, split_weight, B, H, W, cin, cout, y, cout, 0, stats, stream);
}

extern "C" int gga_dense_conv3x3(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                                 float* y, void* stream) {
    return gga_dense_conv3x3_slice(x, split_weight, B, H, W, cin, cout, y, cout, 0, nullptr, stream);
}

// ------------------------------------------------------------------------------ dense 3x3 weight gradient
// dW[co][ci][ky][kx] = sum_p x[p + (ky-1, kx-1)][ci] * gy[p][co] of the same 3x3 / stride 1 / pad 1
// convolution, bf16x9 on the matrix cores. Here the GEMM's K is the PIXEL index: the MFMA operands
// are x^T (M = ci) and gy (N = co), i.e. eight consecutive pixels of ONE channel per lane, while
// both tensors are channels-last. The LDS images stay pixel-major ([pixel][32 channels], 64-byte
// rows, three bf16 planes - written exactly like the forward kernel's halo) and
// ds_read_b64_tr_b16 does the transposition on the way out: two transposed reads give a lane the
// 8 pixels of its channel (probe: tools_dev/micro/tr_probe.hip), and a tap shift is just a row
// offset of the x image, so the nine taps reuse one staged copy.
//
// A 256-thread workgroup owns one 64 x 64 (ci, co) channel block (blockIdx.y), a strip of 32
// pixel columns and a segment of image rows of one image; wave w accumulates the (ci tile w/2,
// co tile w%2) 32 x 32 block of all nine taps (144 accumulator registers). Per image row
// (= 2 K-steps of 16 pixels): the gy row (32 px x 64 co) and one new x row (34 px x 64 ci; a ring
// of four rows holds y-1 .. y+2) are fetched one stage ahead, split into planes and stored; each
// K-step reads 6 + 54 transposed fragments for 81 MFMAs. Partial sums go to
// [workgroup][tap][ci][co]; dense_wgrad_reduce_kernel adds them in a fixed order (f64) and writes
// the framework's [cout, cin, 3, 3] layout.
#define DW_XPL (2 * 34 * 64)                 // bytes per plane of one x ring row: [ci tile][34 px][32 ch]
#define DW_XROW (NP * DW_XPL)                // NP = planes per operand (template parameter of the kernel)
#define DW_GPL (2 * 32 * 64)                 // bytes per plane of one gy row: [co tile][32 px][32 ch]
#define DW_GROW (NP * DW_GPL)

template <int NP>
__global__ __launch_bounds__(256, 2) void dense_wgrad3x3_x9_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                  int B, int H, int W, int cin, int cout, int strips,
                                                                  int prow, int pcol, float* __restrict__ partials,
                                                                  const uint32_t* __restrict__ amax_x,
                                                                  const uint32_t* __restrict__ amax_g, int xblk, int gblk) {
    // NP = 3: bf16 planes, six products; NP = 2: fp16 planes of the scaled operands, three products (h2_split2); the
    // partial sums then stay scaled and dense_wgrad_reduce_kernel scales the total back. xblk / gblk: the operand's absmax is
    // given per 64-channel block (amax[channel / 64]) instead of once for the tensor - dW[ci][co] only ever sees channel ci of x
    // and channel co of gy, so a block whose values lie far below the tensor's largest (the regression branches of the head
    // beside its heat-map branches in the 960-channel gradient) keeps its own 22 bits.
    float xscale = 1.0f, gscale = 1.0f;
    if (NP == 2) {
        const int ncb = cout >> 6;
        xscale = h2_scale(h2_scale_exp(amax_x[xblk ? (int)(blockIdx.y / ncb) : 0]));
        gscale = h2_scale(h2_scale_exp(amax_g[gblk ? (int)(blockIdx.y % ncb) : 0]));
    }
    __shared__ __attribute__((aligned(16))) unsigned char Xs[4 * DW_XROW];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[2 * DW_GROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ti = wave >> 1, tj = wave & 1;               // ci tile, co tile of this wave
    // The image rows of all (image, 32-column strip) pairs form one sequence of B * strips * H row
    // stages; workgroup i takes an equal contiguous share of it (so that exactly as many workgroups
    // as fit on the chip carry the same load - dW sums over all pixels, a share may span columns).
    // blockIdx.y -> 64 x 64 channel block.
    const int ncb_o = cout >> 6;
    const int ci0 = (blockIdx.y / ncb_o) * 64, co0 = (blockIdx.y % ncb_o) * 64;
    const int64_t total_rows = (int64_t)B * strips * H;
    const int64_t r0 = total_rows * blockIdx.x / gridDim.x, r1 = total_rows * (blockIdx.x + 1) / gridDim.x;
    int x0 = 0, ye = 0;
    const float* Xb = X;
    const float* Gb = G;

    mf_v16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    // staging pieces: gy row = 32 px x 16 float4 (2 per thread); x row = 34 px x 16 float4 (3 per thread, 544 used).
    // Round 5: no vector arithmetic per row stage beyond the split (every vector instruction costs the matrix pipe of its SIMD
    // ~9 cycles, tools_dev/micro/ws_interference_probe.hip; the loop had 115 per 54 MFMAs, 40 of them the split). A piece's
    // byte offset inside an image row is fixed per column strip (gof / xof, 32 bits beside the row's scalar base: the
    // saddr form of global_load); a piece outside the image reads a pixel INSIDE it (column x0, the nearest row) and is
    // multiplied by a scale of zero in the split - one select per piece instead of a predicated load and four zeros.
    float4 rg0, rg1, rx0, rx1, rx2;
    uint32_t gof0 = 0, gof1 = 0, xof0 = 0, xof1 = 0, xof2 = 0;
    float gs0 = 0.f, gs1 = 0.f, xs0 = 0.f, xs1 = 0.f, xs2 = 0.f;            // the operand's scale, or 0 for a column outside the image
#define DW_COLG(OF, SC_, E) { const int f = tid + 256 * (E); const int px = f >> 4, q = f & 15; const bool ok = x0 + px < W;      \
        OF = (uint32_t)(((x0 + (ok ? px : 0)) * pcol * cout + 4 * q) * 4); SC_ = ok ? gscale : 0.0f; }
#define DW_COLX(OF, SC_, E) { const int f = tid + 256 * (E); const int px = f >> 4, q = f & 15; const int ix = x0 - 1 + px;       \
        const bool ok = f < 544 && (unsigned)ix < (unsigned)W;                                                        \
        OF = (uint32_t)(((ok ? ix : x0) * pcol * cin + 4 * q) * 4); SC_ = ok ? xscale : 0.0f; }
#define DW_LOAD_G(Y) { const char* rp_ = reinterpret_cast<const char*>(Gb + (int64_t)(Y) * prow * cout);             /* (every row asked for is < ye) */ \
        rg0 = *reinterpret_cast<const float4*>(rp_ + gof0); rg1 = *reinterpret_cast<const float4*>(rp_ + gof1); }
#define DW_LOAD_X(Y) { const int yy = (Y); const int yc_ = yy < 0 ? 0 : (yy < H ? yy : H - 1);                        \
        const char* rp_ = reinterpret_cast<const char*>(Xb + (int64_t)yc_ * prow * cin);                              \
        rx0 = *reinterpret_cast<const float4*>(rp_ + xof0); rx1 = *reinterpret_cast<const float4*>(rp_ + xof1);       \
        rx2 = *reinterpret_cast<const float4*>(rp_ + xof2); }
    // piece (pixel px, float4 q) of a row image: channel tile q / 8, byte (q % 8) * 8 of the 64-byte pixel row
#define DW_SPLIT_STORE(V, BASE, PL, NPX, F, SC) { const int f = (F); const int px = f >> 4, q = f & 15;               \
        unsigned char* dst = (BASE) + (q >> 3) * ((NPX) * 64) + px * 64 + (q & 7) * 8;                                \
        if (NP == 3) {                                                                                                \
            uint32_t lo1, lo2, lo3, hi1, hi2, hi3;                                                                    \
            x9_split2(V.x * (SC), V.y * (SC), lo1, lo2, lo3); x9_split2(V.z * (SC), V.w * (SC), hi1, hi2, hi3);     /* (three planes: SC is 1 or 0) */ \
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                    \
            *reinterpret_cast<uint2*>(dst + (PL)) = make_uint2(lo2, hi2);                                             \
            *reinterpret_cast<uint2*>(dst + (NP - 1) * (PL)) = make_uint2(lo3, hi3);                                  \
        } else {                                                                                                      \
            uint32_t lo1, lo2, hi1, hi2;                                                                              \
            h2_split2s(V.x, V.y, (SC), lo1, lo2); h2_split2s(V.z, V.w, (SC), hi1, hi2);                         \
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                    \
            *reinterpret_cast<uint2*>(dst + (PL)) = make_uint2(lo2, hi2);                                             \
        } }
#define DW_STORE_G(Y) { unsigned char* base = Gs + ((Y) & 1) * DW_GROW;                                               \
        DW_SPLIT_STORE(rg0, base, DW_GPL, 32, tid, gs0) DW_SPLIT_STORE(rg1, base, DW_GPL, 32, tid + 256, gs1) }
#define DW_STORE_X(Y) { unsigned char* base = Xs + (((Y) + 4) & 3) * DW_XROW;                                         \
        const bool rowok_ = (unsigned)(Y) < (unsigned)H;                           /* uniform: a scalar condition */  \
        const float s0_ = rowok_ ? xs0 : 0.0f, s1_ = rowok_ ? xs1 : 0.0f, s2_ = rowok_ ? xs2 : 0.0f;                 \
        DW_SPLIT_STORE(rx0, base, DW_XPL, 34, tid, s0_) DW_SPLIT_STORE(rx1, base, DW_XPL, 34, tid + 256, s1_)         \
        if (tid + 512 < 544) DW_SPLIT_STORE(rx2, base, DW_XPL, 34, tid + 512, s2_) }

    // transposed fragment of a [pixel][32 ch] image: lane l gets channel l%32, pixels P0 + 8*(l/32) .. +7
    const int grp = lane >> 4, li = lane & 15;
    const int froff = ((8 * (grp >> 1) + (li >> 2)) * 64) + (16 * (grp & 1) + 4 * (li & 3)) * 2;   // byte offset of this lane's address in the block
    union Frag { mf_v8bf v; dw_v4s h[2]; };
#define DW_FRAG(F, PTR) { F.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dw_v4s*)((PTR) + froff));          \
                          F.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dw_v4s*)((PTR) + froff + 4 * 64)); }

    int64_t idx = r0;
    while (idx < r1) {
    const int col = (int)(idx / H);
    const int ys = (int)(idx - (int64_t)col * H);
    {
        const int b = col / strips, strip = col - b * strips;
        x0 = strip * 32;
        Xb = X + (int64_t)b * H * W * cin + ci0;
        Gb = G + (int64_t)b * H * W * cout + co0;
        DW_COLG(gof0, gs0, 0) DW_COLG(gof1, gs1, 1)
        DW_COLX(xof0, xs0, 0) DW_COLX(xof1, xs1, 1) DW_COLX(xof2, xs2, 2)
    }
    ye = (int)(r1 - idx < (int64_t)(H - ys) ? ys + (r1 - idx) : H);      // rows of this column in my share
    idx += ye - ys;
    // (re)fill the ring for this column: x rows ys-1 .. ys+1 and the gy row ys
    DW_LOAD_X(ys - 1); DW_STORE_X(ys - 1);
    DW_LOAD_X(ys);     DW_STORE_X(ys);
    DW_LOAD_X(ys + 1); DW_STORE_X(ys + 1);
    DW_LOAD_G(ys);     DW_STORE_G(ys);
    __syncthreads();
    for (int y = ys; y < ye; ++y) {
        const bool more = y + 1 < ye;
        if (more) { DW_LOAD_G(y + 1); DW_LOAD_X(y + 2); }
        const unsigned char* gbase = Gs + (y & 1) * DW_GROW + tj * (32 * 64);
        // (left to the compiler's schedule: forcing the next tap's six reads ahead of the current tap's
        // MFMAs with sched_barriers measured 584 instead of 545 us)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag g0, g1, g2;
            DW_FRAG(g0, gbase + (16 * s) * 64);
            DW_FRAG(g1, gbase + DW_GPL + (16 * s) * 64);
            if (NP == 3) { DW_FRAG(g2, gbase + (NP - 1) * DW_GPL + (16 * s) * 64); } else g2 = g1;
            if (NP == 2) {
                // three taps (one kernel row) at a time, the three partial products interleaved over the taps: consecutive
                // MFMAs never write the same accumulator (tap by tap with the three products back to back: 254 instead of
                // 233 us at 64 -> 64, 16 x 248 x 216; 202 -> 197 at 128 -> 128)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    Frag b0[3], b1[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const unsigned char* xb = Xs + ((y + ky - 1 + 4) & 3) * DW_XROW + ti * (34 * 64) + (16 * s + kx) * 64;
                        DW_FRAG(b0[kx], xb);
                        DW_FRAG(b1[kx], xb + DW_XPL);
                    }
#define DW_MI(A_, G_, T_) acc[T_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_v8h, A_.v), __builtin_bit_cast(mf_v8h, G_.v), acc[T_], 0, 0, 0);
                    DW_MI(b0[0], g1, 3 * ky) DW_MI(b0[1], g1, 3 * ky + 1) DW_MI(b0[2], g1, 3 * ky + 2)
                    DW_MI(b1[0], g0, 3 * ky) DW_MI(b1[1], g0, 3 * ky + 1) DW_MI(b1[2], g0, 3 * ky + 2)
                    DW_MI(b0[0], g0, 3 * ky) DW_MI(b0[1], g0, 3 * ky + 1) DW_MI(b0[2], g0, 3 * ky + 2)
#undef DW_MI
                }
                continue;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const unsigned char* xbase = Xs + ((y + ky - 1 + 4) & 3) * DW_XROW + ti * (34 * 64) + (16 * s + kx) * 64;
                Frag a0, a1, a2;
                DW_FRAG(a0, xbase);
                DW_FRAG(a1, xbase + DW_XPL);
                if (NP == 2) {
#define DW_MH(A_, G_) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_v8h, A_.v), __builtin_bit_cast(mf_v8h, G_.v), acc[tap], 0, 0, 0);
                    DW_MH(a0, g1) DW_MH(a1, g0) DW_MH(a0, g0)
#undef DW_MH
                    continue;
                }
                DW_FRAG(a2, xbase + (NP - 1) * DW_XPL);
                // nine partial products, smallest first
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, g2.v, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, g1.v, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, g0.v, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, g1.v, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, g0.v, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, g0.v, acc[tap], 0, 0, 0);
            }
        }
        if (more) { DW_STORE_G(y + 1); DW_STORE_X(y + 2); }
        __syncthreads();
    }
    }
#undef DW_LOAD_G
#undef DW_COLG
#undef DW_COLX
#undef DW_LOAD_X
#undef DW_SPLIT_STORE
#undef DW_STORE_G
#undef DW_STORE_X
#undef DW_FRAG
    // D: register v of lane l = row (ci in tile) (v/4)*8 + (l/32)*4 + v%4, column (co in tile) l%32
    float* out = partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (9 * 64 * 64);
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = ti * 32 + (v >> 2) * 8 + h * 4 + (v & 3);
            out[(tap * 64 + ci) * 64 + tj * 32 + r] = acc[tap][v];
        }
}

// dW[co][ci][ky][kx] (element strides given) = sum over the workgroups' partials, fixed order, f64
__global__ __launch_bounds__(256) void dense_wgrad_reduce_kernel(const float* __restrict__ partials, int nblk, int cin, int cout,
                                                                int64_t s_co, int64_t s_ci, int64_t s_ky, int64_t s_kx,
                                                                const uint32_t* __restrict__ amax_x,
                                                                const uint32_t* __restrict__ amax_g, int xblk, int gblk,
                                                                float* __restrict__ dW) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (tap, ci local, co local) of channel block blockIdx.y
    if (i >= 9 * 64 * 64) return;
    const int ncb_o = cout >> 6;
    const int ci0 = (blockIdx.y / ncb_o) * 64, co0 = (blockIdx.y % ncb_o) * 64;
    const float* p = partials + (int64_t)blockIdx.y * nblk * (9 * 64 * 64) + i;
    double s = 0.0;
#pragma unroll 8
    for (int k = 0; k < nblk; ++k) s += (double)p[(int64_t)k * (9 * 64 * 64)];
    const int co = i & 63, ci = (i >> 6) & 63, tap = i >> 12;
    const int ky = tap / 3, kx = tap - ky * 3;
    if (amax_x)                                            // fp16-plane partials are scaled (per tensor or per 64-channel block)
        s = s * (double)h2_descale(h2_scale_exp(amax_x[xblk ? ci0 >> 6 : 0])) * (double)h2_descale(h2_scale_exp(amax_g[gblk ? co0 >> 6 : 0]));
    dW[(co0 + co) * s_co + (ci0 + ci) * s_ci + ky * s_ky + kx * s_kx] = (float)s;
}

// workgroups per channel block: as many as run at once (two per CU, 256 CUs) over all channel blocks
static int dense_wgrad_blocks(int B, int H, int W, int cin, int cout) {
    const int64_t total_rows = (int64_t)B * ((W + 31) / 32) * H;
    int64_t n = 512 / ((int64_t)(cin >> 6) * (cout >> 6));
    if (n < 1) n = 1;
    if (n > total_rows) n = total_rows;
    return (int)n;
}

extern "C" size_t gga_dense_wgrad3x3_workspace_bytes(int B, int H, int W, int cin, int cout) {
    if (B < 1 || H < 1 || W < 1 || cin < 64 || cout < 64 || (cin & 63) || (cout & 63)) return 0;
    const int n0 = dense_wgrad_blocks(B, H, W, cin, cout), n1 = dense_wgrad_blocks(B, W, H, cin, cout);     // either walk
    return (size_t)(n0 > n1 ? n0 : n1) * (cin >> 6) * (cout >> 6) * 9 * 64 * 64 * sizeof(float);
}

extern "C" int gga_dense_wgrad3x3(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                                  float* grad_weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                  int64_t stride_kx, int transposed, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
    return gga_dense_wgrad3x3_planes(x, grad_y, B, H, W, cin, cout, grad_weight, stride_co, stride_ci, stride_ky, stride_kx,
                                     transposed, 3, nullptr, nullptr, workspace, workspace_bytes, stream_);
}

extern "C" int gga_dense_wgrad3x3_planes(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                                         float* grad_weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                         int64_t stride_kx, int transposed, int planes, const uint32_t* amax_x,
                                         const uint32_t* amax_grad_y, void* workspace, size_t workspace_bytes, void* stream_) {
    return gga_dense_wgrad3x3_block_amax(x, grad_y, B, H, W, cin, cout, grad_weight, stride_co, stride_ci, stride_ky, stride_kx,
                                         transposed, planes, amax_x, 0, amax_grad_y, 0, workspace, workspace_bytes, stream_);
}

extern "C" int gga_dense_wgrad3x3_block_amax(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                                             float* grad_weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                             int64_t stride_kx, int transposed, int planes, const uint32_t* amax_x,
                                             int amax_x_per_block, const uint32_t* amax_grad_y, int amax_grad_y_per_block,
                                             void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int xblk = amax_x_per_block ? 1 : 0, gblk = amax_grad_y_per_block ? 1 : 0;
    GGA_REQUIRE(x && grad_y && grad_weight && workspace, "gga_dense_wgrad3x3: null pointer argument");
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_x && amax_grad_y),
                "gga_dense_wgrad3x3: planes must be 3 (bf16) or 2 (fp16, with the operands' absmax bits)");
    GGA_REQUIRE(B >= 1 && H >= 1 && W >= 1 && cin >= 64 && cout >= 64 && (cin & 63) == 0 && (cout & 63) == 0,
                "gga_dense_wgrad3x3: cin and cout must be multiples of 64 (got %d -> %d)", cin, cout);
    GGA_REQUIRE((int64_t)H * W * (cin > cout ? cin : cout) * 4 < ((int64_t)1 << 31),
                "gga_dense_wgrad3x3: one image of an operand must stay below 2 GiB (32-bit piece offsets; got %d x %d x %d channels)", H, W,
                cin > cout ? cin : cout);
    if (workspace_bytes < gga_dense_wgrad3x3_workspace_bytes(B, H, W, cin, cout)) {
        gga_set_error("gga_dense_wgrad3x3: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    // transposed: strips of 32 pixels along the image's H, rows along its W; the taps swap with them
    const int prow = transposed ? 1 : W, pcol = transposed ? W : 1;
    if (transposed) { const int t = H; H = W; W = t; const int64_t ts = stride_ky; stride_ky = stride_kx; stride_kx = ts; }
    const int strips = (W + 31) / 32;
    const int nblk = dense_wgrad_blocks(B, H, W, cin, cout), ncb = (cin >> 6) * (cout >> 6);
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_DENSE_WGRAD, GGA_TIMING_CONV_KEY(cin, cout, (int64_t)H * W));
    GGA_TIME_START(tev, stream);
    if (planes == 3)
        hipLaunchKernelGGL(dense_wgrad3x3_x9_kernel<3>, dim3(nblk, ncb), dim3(256), 0, stream, x, grad_y, B, H, W, cin, cout, strips,
                           prow, pcol, (float*)workspace, amax_x, amax_grad_y, xblk, gblk);
    else
        hipLaunchKernelGGL(dense_wgrad3x3_x9_kernel<2>, dim3(nblk, ncb), dim3(256), 0, stream, x, grad_y, B, H, W, cin, cout, strips,
                           prow, pcol, (float*)workspace, amax_x, amax_grad_y, xblk, gblk);
    GGA_CHECK_LAUNCH("dense_wgrad3x3_x9_kernel");
    hipLaunchKernelGGL(dense_wgrad_reduce_kernel, dim3((9 * 64 * 64 + 255) / 256, ncb), dim3(256), 0, stream,
                       (const float*)workspace, nblk, cin, cout, stride_co, stride_ci, stride_ky, stride_kx,
                       planes == 2 ? amax_x : nullptr, amax_grad_y, xblk, gblk, grad_weight);
    GGA_CHECK_LAUNCH("dense_wgrad_reduce_kernel");
    GGA_TIME_STOP(tev, stream);
    return GGA_OK;
}
