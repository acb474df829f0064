// The two kernel templates of dense_conv_ws.hip (see the comment there). One instantiation per translation unit
// (dense_conv_ws_i*.hip, round 6): the four instantiations in ONE unit took 5 min 12 s to compile on one core - a cold
// `make -j4` waited for that file alone; apart they build side by side.
#pragma once
#include <stdlib.h>

#include "dense_conv.h"

// The consumers' stage barrier: the bare instruction, NOT __syncthreads(). The compiler puts s_waitcnt lgkmcnt(0) in front of the
// latter (and in front of an inline-asm barrier with a "memory" clobber) - every fragment read in flight must land before the
// wave may even arrive, at every stage, and the last read of a stage is issued two products before its barrier: ~100 cycles of
// a 768-cycle stage. The protocol does not need it: a consumer's reads are of data the producers completed before the PREVIOUS
// barrier, and the buffers they come from (weight ring, halo image) are not written again until at least one more whole stage has
// passed; the reads are waited for where their values are used (counted lgkmcnt, the compiler's). What must not happen is a read
// of the NEXT stage's data moving above the barrier: the lane offsets every fragment address is built from go through the asm
// as read-write operands, so those reads depend on it. The producers keep __syncthreads(): their LDS writes must be complete
// when they arrive. -DWS_CONSUMER_SYNC: the old form.
#ifdef WS_CONSUMER_SYNC
#define WS_CONSUMER_BARRIER(OFF_A, OFF_B) __syncthreads();
#else
#define WS_CONSUMER_BARRIER(OFF_A, OFF_B) asm volatile("s_barrier" : "+v"(OFF_A), "+v"(OFF_B));
#endif
#define DC_WS_DEFAULT_MFMA 32                 // consumer waves' matrix instruction unless GGA_DC_WS_MFMA says otherwise (16: 16x16x32)

template <int NT, int MT>
__global__ __launch_bounds__(512, 2) void dense_conv3x3_ws_kernel(const float* __restrict__ X, const uint16_t* __restrict__ Wp, int B,
                                                                   int H, int W, int cin, int cout, int tiles_x, int tiles_y,
                                                                   float* __restrict__ Y, int ystride, int prow, int pcol,
                                                                   double* __restrict__ stats, const uint32_t* __restrict__ amax_x,
                                                                   const uint32_t* __restrict__ amax_w, DcBnBwd bn,
                                                                   const float* __restrict__ zero_page, DcSlices sl) {
    constexpr int TR = 4 * MT, HP = (TR + 2) * DC_HW, CO = NT * 32;
    constexpr int APL = HP * DC_ROWB, ASZ = 2 * APL;                       // one plane / both planes of a halo image
    constexpr int BPL = CO * DC_ROWB, BSZ = 2 * BPL;                       // one plane / both planes of a weight stage
    constexpr int BPIECES = 2 * CO * 2, NB = BPIECES / 256;                // 16-byte pieces of a weight stage, per producer lane
    constexpr int NA = (HP * 4 + 255) / 256;                               // float4 pieces of a halo chunk per producer lane
    static_assert(NB == 1 || NB == 2, "weight stage pieces per producer lane");
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * ASZ];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * BSZ];
    __shared__ float red[4 * 2 * CO];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int r = lane & 31, h = lane >> 5;
    // sl.n > 1: the launch computes sl.n 128-channel slices of one convolution's output (same input, one weight operand, output
    // columns and statistics per slice); tile t of the grid is tile t % img_tiles of slice t / img_tiles
    const int per_img = tiles_x * tiles_y, img_tiles = B * per_img, n_tiles = img_tiles * (sl.n > 1 ? sl.n : 1), nchunks = cin / DC_CK;
    const int sbx = h2_scale_exp(*amax_x), sbw = h2_scale_exp(*amax_w);
    const float xscale = h2_scale(sbx);
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // ---- producer state: pieces f = ptid + 256 e of a halo chunk = pixel f / 4, channels 4 (f % 4) .. + 3
    const int ptid = tid - 256;
    typedef float ws_v2f __attribute__((ext_vector_type(2)));
    float4 ra[NA];
    const float* pcur[NA];                     // piece e of the current / next tile: its 16-channel chunk 0 in the image, or the zero page
    const float* pnxt[NA];                     // for pixels outside the image (no select on the loaded values, no flag to carry)
    uint4 bq0, bq1, cq0, cq1;
    bq0 = bq1 = cq0 = cq1 = make_uint4(0, 0, 0, 0);
    const uint16_t* Wcur = sl.n > 1 ? sl.w[tile / img_tiles] : Wp;      // weight operand of the current / next tile's slice
    const uint16_t* Wnxt = Wcur;
#define WS_AOFF(P, T_) {                                                                                              \
        const int it_ = (T_) % img_tiles;                                                                             \
        const int tb_ = it_ / per_img, rem_ = it_ - tb_ * per_img;                                                    \
        const int ty0_ = (rem_ / tiles_x) * TR, tx0_ = (rem_ % tiles_x) * DC_TW;                                      \
        const float* xb_ = X + (int64_t)tb_ * H * W * cin;                                                            \
        _Pragma("unroll") for (int e = 0; e < NA; ++e) {                                                             \
            const int f = ptid + 256 * e;                                                                             \
            const int hp = f >> 2, q = f & 3;                                                                         \
            const int hr = hp / DC_HW, hx = hp - hr * DC_HW;                                                          \
            const int iy = ty0_ + hr - 1, ix = tx0_ + hx - 1;                                                         \
            const bool ok = hp < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;                     \
            P[e] = ok ? xb_ + ((iy * prow + ix * pcol) * cin + q * 4) : zero_page;                                 \
        } }
#define WS_LOAD_PIECE(E, P, CH) ra[E] = *reinterpret_cast<const float4*>(P[E] + (CH) * DC_CK);
#define WS_STORE_PIECE(E, BUF) {                                                                                      \
        const int f = ptid + 256 * (E);                                                                               \
        if (f < HP * 4) {                                                                                             \
            unsigned char* dst = As + (BUF) * ASZ + (f >> 2) * DC_ROWB + (f & 3) * 8;                                 \
            uint32_t lo1, lo2, hi1, hi2;                                                                              \
            h2_split2u(ra[E].x, ra[E].y, xscale, lo1, lo2); h2_split2u(ra[E].z, ra[E].w, xscale, hi1, hi2);            \
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                    \
            *reinterpret_cast<uint2*>(dst + APL) = make_uint2(lo2, hi2);                                              \
        } }
    // weight stage (tap, 16-channel chunk): contiguous and in LDS piece order in the packed operand (piece f = (plane, column, half))
#define WS_BLD(WP, TAP, CH, V0, V1) {                                                                                     \
        const uint4* bsrc = reinterpret_cast<const uint4*>((WP) + ((int64_t)(TAP) * nchunks + (CH)) * (2 * CO * DC_CK)); \
        V0 = bsrc[ptid]; if (NB > 1) V1 = bsrc[ptid + 256]; }
#define WS_BST(BUF, V0, V1) {                                                                                         \
        *reinterpret_cast<uint4*>(Bs + (BUF) * BSZ + (ptid >> 1) * DC_ROWB + (ptid & 1) * 16) = V0;                   \
        if (NB > 1) { const int f_ = ptid + 256; *reinterpret_cast<uint4*>(Bs + (BUF) * BSZ + (f_ >> 1) * DC_ROWB + (f_ & 1) * 16) = V1; } }

    // ---- consumer state
    mf_v16 acc[MT][NT];
    mf_v8h fa[MT][2], fb[NT][2], ga[MT][2], gb[NT][2];
    int cons_a = r * DC_ROWB + h * 16, cons_b = r * DC_ROWB + h * 16;      // the lane's offset in a halo row block / a weight stage (see WS_CONSUMER_BARRIER)
#define WS_READ_A(FA, TAP, HB) {                                                                                      \
        const unsigned char* Ap = As + (HB) * ASZ + ((MT * wave + (TAP) / 3) * DC_HW + (TAP) % 3) * DC_ROWB + cons_a;  \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int p = 0; p < 2; ++p)                  \
            FA[m][p] = *reinterpret_cast<const mf_v8h*>(Ap + p * APL + m * DC_HW * DC_ROWB); }
#define WS_READ_B(FB, BUF) {                                                                                          \
        const unsigned char* Bp = Bs + (BUF) * BSZ + cons_b;                                                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int p = 0; p < 2; ++p)                  \
            FB[t][p] = *reinterpret_cast<const mf_v8h*>(Bp + p * BPL + t * 32 * DC_ROWB); }
    // partial products smallest first; tiles innermost so consecutive MFMAs never share an accumulator
#define WS_MM1(FA, FB, PA, PB) _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[m][PA], FB[t][PB], acc[m][t], 0, 0, 0);
#define WS_MMA(FA, FB) { WS_MM1(FA, FB, 0, 1) WS_MM1(FA, FB, 1, 0) WS_MM1(FA, FB, 0, 0) }
    constexpr int N_MFMA = 3 * MT * NT, N_READ = 2 * MT + 2 * NT;          // per stage: 24 and 12
    static_assert(N_MFMA == 2 * N_READ, "two MFMAs per fragment read");

    // ---- prologue: the first tile's first halo chunk and weight stages 0 and 1 in LDS, stage 2 and halo chunk 1 in registers
    if (!consumer) {
        WS_AOFF(pcur, tile)
#pragma unroll
        for (int e = 0; e < NA; ++e) { WS_LOAD_PIECE(e, pcur, 0) }
        WS_BLD(Wcur, 0, 0, bq0, bq1)
        WS_BLD(Wcur, 1, 0, cq0, cq1)
#pragma unroll
        for (int e = 0; e < NA; ++e) { WS_STORE_PIECE(e, 0) }
        WS_BST(0, bq0, bq1)
        WS_BST(1, cq0, cq1)
        WS_BLD(Wcur, 2, 0, bq0, bq1)
#pragma unroll
        for (int e = 0; e < NA; ++e) { WS_LOAD_PIECE(e, pcur, 1) }
    }
    __syncthreads();

    // Stage s = chunk * 9 + tap, one barrier per stage, the same number of barriers on both paths. Consumers: MFMAs of the set
    // read during stage s - 1, reads of stage s + 1's set. Producers: request the weights of stage s + 3 (register set
    // (s + 1) % 2), handle the halo pieces scheduled on this tap - piece e of the NEXT chunk's image (loaded a chunk ago) is
    // split and written into halo image (chunk + 1) % 2, and the same piece of the chunk after it is requested -, then write the
    // weights of stage s + 2 (requested during stage s - 1, set s % 2) into LDS buffer (s + 2) % 3. Chunks come in pairs so that
    // register sets and halo images are compile-time. (Two loops, not one loop with a branch per stage: in one loop the
    // register allocator keeps both roles' state alive in every wave - 1300 spilled registers.)
    if (!consumer) {
        for (; tile < n_tiles; tile += gridDim.x) {
            if (tile + (int)gridDim.x < n_tiles) {
                WS_AOFF(pnxt, tile + (int)gridDim.x)
                if (sl.n > 1) Wnxt = sl.w[(tile + (int)gridDim.x) / img_tiles];
            } else {
#pragma unroll
                for (int e = 0; e < NA; ++e) pnxt[e] = zero_page;
            }
#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) {                                                             \
                if ((TAP) + 3 < 9) { WS_BLD(Wcur, (TAP) + 3, (CH), L0, L1) }                                          \
                else if ((CH) + 1 < nchunks) { WS_BLD(Wcur, (TAP) + 3 - 9, (CH) + 1, L0, L1) }                        \
                else { WS_BLD(Wnxt, (TAP) + 3 - 9, 0, L0, L1) }                                                       \
                if ((TAP) < 8) {                                                                                      \
                    _Pragma("unroll") for (int e = 0; e < NA; ++e) if ((e * 8) / NA == (TAP)) {                       \
                        WS_STORE_PIECE(e, 1 - (HB))                                                                   \
                        if (INCUR2) { WS_LOAD_PIECE(e, pcur, (CH) + 2) }                                              \
                        else { WS_LOAD_PIECE(e, pnxt, (CH) + 2 - nchunks) }                                           \
                    }                                                                                                 \
                }                                                                                                     \
                WS_BST(((TAP) + 2) % 3, S0, S1)                                                                       \
                __syncthreads(); }
#define WS_EVEN(TAP, CH, HB, INCUR2) WS_STAGE_P(TAP, CH, HB, INCUR2, cq0, cq1, bq0, bq1)      /* even stage: request into set 1, write set 0 */
#define WS_ODD(TAP, CH, HB, INCUR2) WS_STAGE_P(TAP, CH, HB, INCUR2, bq0, bq1, cq0, cq1)
            // (the last pair of chunks, whose pieces two chunks ahead are the next tile's, is its own copy of the code: one loop with
            // the tile as a run-time choice selects between two 64-bit pointers per piece with vector instructions)
#define WS_PAIR(CH, INCUR2) {                                                                                         \
                WS_EVEN(0, CH, 0, INCUR2) WS_ODD(1, CH, 0, INCUR2) WS_EVEN(2, CH, 0, INCUR2) WS_ODD(3, CH, 0, INCUR2) WS_EVEN(4, CH, 0, INCUR2) WS_ODD(5, CH, 0, INCUR2) WS_EVEN(6, CH, 0, INCUR2) WS_ODD(7, CH, 0, INCUR2) WS_EVEN(8, CH, 0, INCUR2) \
                WS_ODD(0, (CH) + 1, 1, INCUR2) WS_EVEN(1, (CH) + 1, 1, INCUR2) WS_ODD(2, (CH) + 1, 1, INCUR2) WS_EVEN(3, (CH) + 1, 1, INCUR2) WS_ODD(4, (CH) + 1, 1, INCUR2) WS_EVEN(5, (CH) + 1, 1, INCUR2) WS_ODD(6, (CH) + 1, 1, INCUR2) WS_EVEN(7, (CH) + 1, 1, INCUR2) WS_ODD(8, (CH) + 1, 1, INCUR2) }
            int ch = 0;
            for (; ch + 2 < nchunks; ch += 2) WS_PAIR(ch, 1)
            WS_PAIR(ch, 0)
#undef WS_PAIR
#undef WS_EVEN
#undef WS_ODD
#undef WS_STAGE_P
#pragma unroll
            for (int e = 0; e < NA; ++e) pcur[e] = pnxt[e];
            Wcur = Wnxt;
            if (stats) { __syncthreads(); __syncthreads(); }               // the consumers' statistics fold
        }
        return;
    }

    // statistics: one f64 row pair per WORKGROUP and slice (row blockIdx.x of the slice's `stats`: the sums of all the tiles the
    // workgroup walked there; the BatchNorm's fold reads at most gridDim.x rows instead of one per tile - 3472 at the head's first
    // convolutions). Thread tid < 2 CO owns one (sum | sum of squares, channel) of the row.
    double run_sum = 0.0;
    int run_slice = tile / img_tiles;
    if (stats && sl.n > 1 && tid < 2 * CO && tid % CO < cout)      // rows of slices this workgroup never visits stay zero
        for (int s_ = 0; s_ < sl.n; ++s_) sl.stats[s_][((int64_t)blockIdx.x * 2 + tid / CO) * cout + tid % CO] = 0.0;
#define WS_FLUSH_STATS() {                                                                                            \
        if (tid < 2 * CO && tid % CO < cout)                                                                          \
            (sl.n > 1 ? sl.stats[run_slice] : stats)[((int64_t)blockIdx.x * 2 + tid / CO) * cout + tid % CO] = run_sum; \
        run_sum = 0.0; }
    for (; tile < n_tiles; tile += gridDim.x) {
        const int slice = tile / img_tiles, itile = tile - slice * img_tiles;
        if (stats && slice != run_slice) { WS_FLUSH_STATS() run_slice = slice; }
        const int tb = itile / per_img, trem = itile - tb * per_img;
        const int y0 = (trem / tiles_x) * TR, x0 = (trem % tiles_x) * DC_TW;
        float* __restrict__ Ys = sl.n > 1 ? sl.y[slice] : Y;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.0f;
        WS_READ_A(fa, 0, 0)
        WS_READ_B(fb, 0)
#define WS_STAGE_C(TAP, CH, HB, LASTABLE, CA, CB, XA, XB) {      /* (a tile's last stage reads a set nobody uses: no branch in the stream) */ \
            WS_READ_A(XA, ((TAP) + 1) % 9, (TAP) == 8 ? 1 - (HB) : (HB))                                              \
            WS_READ_B(XB, ((TAP) + 1) % 3)                                                                            \
            WS_MMA(CA, CB)                                                                                            \
            _Pragma("unroll") for (int g_ = 0; g_ < N_READ; ++g_) {                                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
            WS_CONSUMER_BARRIER(cons_a, cons_b) }
#define WS_EVEN(TAP, CH, HB, LASTABLE) WS_STAGE_C(TAP, CH, HB, LASTABLE, fa, fb, ga, gb)
#define WS_ODD(TAP, CH, HB, LASTABLE) WS_STAGE_C(TAP, CH, HB, LASTABLE, ga, gb, fa, fb)
        for (int ch = 0; ch < nchunks; ch += 2) {
            WS_EVEN(0, ch, 0, 0) WS_ODD(1, ch, 0, 0) WS_EVEN(2, ch, 0, 0) WS_ODD(3, ch, 0, 0) WS_EVEN(4, ch, 0, 0) WS_ODD(5, ch, 0, 0) WS_EVEN(6, ch, 0, 0) WS_ODD(7, ch, 0, 0) WS_EVEN(8, ch, 0, 0)
            WS_ODD(0, ch + 1, 1, 1) WS_EVEN(1, ch + 1, 1, 1) WS_ODD(2, ch + 1, 1, 1) WS_EVEN(3, ch + 1, 1, 1) WS_ODD(4, ch + 1, 1, 1) WS_EVEN(5, ch + 1, 1, 1) WS_ODD(6, ch + 1, 1, 1) WS_EVEN(7, ch + 1, 1, 1) WS_ODD(8, ch + 1, 1, 1)
        }
#undef WS_EVEN
#undef WS_ODD
#undef WS_STAGE_C
        // ---- epilogue (consumers): back from the scaled operands (two exact powers of two), then as dense_conv3x3_x9_kernel
        {
            const float dx = h2_descale(sbx), dw = h2_descale(sbw);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][t][i] = acc[m][t][i] * dx * dw;
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
            constexpr int VB = 32 / NT;                    // values of y requested before the first of them is used
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
                        const float* src = bn.y + ((int64_t)tb * H * W + oy * prow + (ox < W ? ox : W - 1) * pcol) * bn.ystride;
#pragma unroll
                        for (int t = 0; t < NT; ++t) yv[j][t] = src[t * 32 + r];
                    }
#pragma unroll
                    for (int j = 0; j < VB; ++j) {
                        const int ox = x0 + ((v0 + j) >> 2) * 8 + h * 4 + ((v0 + j) & 3);
                        if (ox >= W) continue;
                        float* dst = Ys + ((int64_t)tb * H * W + oy * prow + ox * pcol) * ystride;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float g = fmaf(yv[j][t], bsc[t], bsh[t]) > 0.0f ? acc[m][t][v0 + j] : 0.0f;
                            dst[t * 32 + r] = g;
                            s1[t] += g; s2[t] += g * ((yv[j][t] - bmu[t]) * biv[t]);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = y0 + MT * wave + m;
                if (oy >= H) continue;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int ox = x0 + (v >> 2) * 8 + h * 4 + (v & 3);
                    if (ox >= W) continue;
                    float* dst = Ys + ((int64_t)tb * H * W + oy * prow + ox * pcol) * ystride;       // ystride > cout: a channel slice of a wider tensor
#pragma unroll
                    for (int t = 0; t < NT; ++t) dst[t * 32 + r] = acc[m][t][v];
                }
            }
        }
        if (stats) {
            // per-channel sum and sum of squares of the tile's outputs (the batch statistics of the BatchNorm that follows), or the
            // BatchNorm-backward sums collected above: lane sums over its pixels, the two half waves and the four consumer waves are
            // folded through LDS, one f64 row pair per tile
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
                    for (int w_ = 0; w_ < 4; ++w_) a += (double)red[(w_ * 2 + which) * CO + c];
                    run_sum += a;
                }
            }
            __syncthreads();                                  // red is the next tile's
        }
    }
    if (stats) { WS_FLUSH_STATS() }
#undef WS_FLUSH_STATS
#undef WS_READ_A
#undef WS_READ_B
#undef WS_MM1
#undef WS_MMA
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same producer / consumer kernel with v_mfma_f32_16x16x32_f16 in the consumer waves (round 5). Why: the 32x32x16 form is
// power-managed (busy x clock pinned near 1.0-1.1 GHz-equivalents, EXPERIMENTS.md 6d), and under that regime the chip holds a
// higher clock on the 16x16x32 shape (MI355X_MICROARCH.md 'DVFS give-back' item 7). tools_dev/micro/ws_shape_probe.hip - this
// kernel's consumer loop alone, random fp16 operands in LDS, same output tile per wave (128 accumulator registers), same LDS
// bytes per FLOP - measured 1523 against 1374 TFLOP/s issued (2.32 against 1.98 GHz in-kernel) on this part.
//
// What changes. A matrix instruction now takes K = 32: a "double stage" is two consecutive (tap, 16-channel chunk) stages of
// the old stream - lane groups 0, 1 (lane / 16) carry the 16 channels of the first stage, groups 2, 3 those of the second, each
// from its own tap position of the halo image and its own weight stage buffer, so the LDS images, the packed weight operand and
// the producers' pieces stay exactly as they are. 9 taps x 4 chunks = 36 stages = 18 double stages make a "quad" (taps pair up
// as (0,1) (2,3) (4,5) (6,7) (8 | next chunk's 0) (1,2) ... (7,8)): everything about a double stage - halo image, tap offsets,
// weight slots - is a compile-time function of its index in the quad, hence cin % 64 == 0 (else the 32x32x16 form runs).
// A wave's tile is M16 x N16 tiles of 16 x 16 (2 MT x 2 NT): D register v of lane (c = lane % 16, g = lane / 16) = pixel
// 16 (tile % 2) + 4 g + v of image row tile / 2, output column 16 nt + c.
// Registers: the fragments of a K = 32 step are 24 x 4 = 96 registers beside the 128 accumulators - a second set for the
// next double stage (the 32x32x16 form's scheme) does not fit in 256, so ONE set refilled in place as its pieces die (see the
// consumers' loop; a first version that read all 24 fragments at the start of each double stage was 4.5 % SLOWER than the
// 32x32x16 form: four waves' 96 KB of reads in one burst behind every barrier). The producers write the weights of double stage
// d + 1 (slot pair (d + 1) % 3) and the next chunk's halo image while double stage d multiplies.
template <int NT, int MT>
__global__ __launch_bounds__(512, 2) void dense_conv3x3_ws16_kernel(const float* __restrict__ X, const uint16_t* __restrict__ Wp, int B,
                                                                     int H, int W, int cin, int cout, int tiles_x, int tiles_y,
                                                                     float* __restrict__ Y, int ystride, int prow, int pcol,
                                                                     double* __restrict__ stats, const uint32_t* __restrict__ amax_x,
                                                                     const uint32_t* __restrict__ amax_w, DcBnBwd bn,
                                                                     const float* __restrict__ zero_page, DcSlices sl) {
    constexpr int TR = 4 * MT, HP = (TR + 2) * DC_HW, CO = NT * 32;
    constexpr int APL = HP * DC_ROWB, ASZ = 2 * APL;
    constexpr int BPL = CO * DC_ROWB, BSZ = 2 * BPL;
    constexpr int BPIECES = 2 * CO * 2, NB = BPIECES / 256;
    constexpr int NA = (HP * 4 + 255) / 256;
    constexpr int M16 = 2 * MT, N16 = 2 * NT;
    static_assert(NB == 1 || NB == 2, "weight stage pieces per producer lane");
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * ASZ];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[6 * BSZ];      // three slot pairs: double stage d in pair d % 3
    __shared__ float red[4 * 2 * CO];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int per_img = tiles_x * tiles_y, img_tiles = B * per_img, n_tiles = img_tiles * (sl.n > 1 ? sl.n : 1), nchunks = cin / DC_CK;
    const int nquads = nchunks >> 2;
    const int sbx = h2_scale_exp(*amax_x), sbw = h2_scale_exp(*amax_w);
    const float xscale = h2_scale(sbx);
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // ---- producers (pieces and weight stages as in dense_conv3x3_ws_kernel)
    const int ptid = tid - 256;
    typedef float ws_v2f __attribute__((ext_vector_type(2)));
    float4 ra[NA];
    const float* pcur[NA];
    const float* pnxt[NA];
    uint4 wq[2][2][2];                           // [double stage parity][stage of it][piece]: the weights of the next two double stages
#pragma unroll
    for (int i_ = 0; i_ < 8; ++i_) wq[i_ >> 2][(i_ >> 1) & 1][i_ & 1] = make_uint4(0, 0, 0, 0);
    const uint16_t* Wcur = sl.n > 1 ? sl.w[tile / img_tiles] : Wp;
    const uint16_t* Wnxt = Wcur;
    if (!consumer) {
        WS_AOFF(pcur, tile)
#pragma unroll
        for (int e = 0; e < NA; ++e) { WS_LOAD_PIECE(e, pcur, 0) }
        WS_BLD(Wcur, 0, 0, wq[0][0][0], wq[0][0][1])
        WS_BLD(Wcur, 1, 0, wq[0][1][0], wq[0][1][1])
#pragma unroll
        for (int e = 0; e < NA; ++e) { WS_STORE_PIECE(e, 0) }
        WS_BST(0, wq[0][0][0], wq[0][0][1])
        WS_BST(1, wq[0][1][0], wq[0][1][1])
        WS_BLD(Wcur, 2, 0, wq[1][0][0], wq[1][0][1])          // double stage 1 -> set 1, double stage 2 -> set 0
        WS_BLD(Wcur, 3, 0, wq[1][1][0], wq[1][1][1])
        WS_BLD(Wcur, 4, 0, wq[0][0][0], wq[0][0][1])
        WS_BLD(Wcur, 5, 0, wq[0][1][0], wq[0][1][1])
#pragma unroll
        for (int e = 0; e < NA; ++e) { WS_LOAD_PIECE(e, pcur, 1) }
    }
    __syncthreads();

    // Double stage k of a quad (stages 2k, 2k + 1; stage j = chunk j / 9 of the quad, tap j % 9), one barrier each, the same number
    // on both paths. Producers: write the weights of double stage k + 1 (requested during k - 2) into slot pair (k + 1) % 3 (three
    // pairs: the consumers' reads of a pair are not waited for at the barrier, WS_CONSUMER_BARRIER, so a pair rests for a whole
    // double stage between its last reader and its next writer), request those of k + 3 into the register set just freed,
    // handle the halo pieces of this double stage (schedule: below), barrier.
    if (!consumer) {
        for (; tile < n_tiles; tile += gridDim.x) {
            if (tile + (int)gridDim.x < n_tiles) {
                WS_AOFF(pnxt, tile + (int)gridDim.x)
                if (sl.n > 1) Wnxt = sl.w[(tile + (int)gridDim.x) / img_tiles];
            } else {
#pragma unroll
                for (int e = 0; e < NA; ++e) pnxt[e] = zero_page;
            }
            for (int q = 0; q < nquads; ++q) {
                const int c0 = 4 * q;
                const bool last = q + 1 == nquads;
                const uint16_t* wnext = last ? Wnxt : Wcur;          // the operand stages past the quad's end come from
                const int cnext = last ? 0 : c0 + 4;
#pragma unroll
                for (int k = 0; k < 18; ++k) {
                    WS_BST(2 * ((k + 1) % 3) + 0, wq[(k + 1) & 1][0][0], wq[(k + 1) & 1][0][1])
                    WS_BST(2 * ((k + 1) % 3) + 1, wq[(k + 1) & 1][1][0], wq[(k + 1) & 1][1][1])
#pragma unroll
                    for (int s_ = 0; s_ < 2; ++s_) {
                        const int j = 2 * k + 6 + s_;                  // a stage of double stage k + 3
                        if (j < 36) { WS_BLD(Wcur, j % 9, c0 + j / 9, wq[(k + 1) & 1][s_][0], wq[(k + 1) & 1][s_][1]) }
                        else { WS_BLD(wnext, (j - 36) % 9, cnext + (j - 36) / 9, wq[(k + 1) & 1][s_][0], wq[(k + 1) & 1][s_][1]) }
                    }
                    // halo pieces: chunk c0 + grp + 1 is written during three double stages - not the first one after the image's last
                    // reader (tap 8 of the chunk before: its reads were issued before that barrier but are not waited for there),
                    // done before the barrier in front of its first reader
                    const int grp = (k >= 1 && k <= 3) ? 0 : ((k >= 6 && k <= 8) ? 1 : ((k >= 10 && k <= 12) ? 2 : (k >= 15 ? 3 : -1)));
                    if (grp >= 0) {
                        const int k0 = grp == 0 ? 1 : (grp == 1 ? 6 : (grp == 2 ? 10 : 15));
#pragma unroll
                        for (int e = 0; e < NA; ++e)
                            if ((e * 3) / NA == k - k0) {
                                WS_STORE_PIECE(e, (grp + 1) & 1)       // chunk c0 + grp + 1 -> image (grp + 1) % 2
                                const int cl = grp + 2;                // then request the same piece of chunk c0 + grp + 2
                                if (cl < 4) { WS_LOAD_PIECE(e, pcur, c0 + cl) }
                                else if (!last) { WS_LOAD_PIECE(e, pcur, c0 + cl) }
                                else { WS_LOAD_PIECE(e, pnxt, cl - 4) }
                            }
                    }
                    __syncthreads();
                }
            }
#pragma unroll
            for (int e = 0; e < NA; ++e) pcur[e] = pnxt[e];
            Wcur = Wnxt;
            if (stats) { __syncthreads(); __syncthreads(); }
        }
        return;
    }

    // ---- consumers
    typedef float ws_v4f __attribute__((ext_vector_type(4)));
    ws_v4f acc[M16][N16];
    mf_v8h fa[M16][2], fb[N16][2];
    const int c16 = lane & 15, g = lane >> 4, sel = g >> 1;
    int a_lane = ((MT * wave) * DC_HW + c16) * DC_ROWB + (g & 1) * 16;       // (not const: see WS_CONSUMER_BARRIER)
    int b_lane = c16 * DC_ROWB + (g & 1) * 16 + sel * BSZ;
    double run_sum = 0.0;
    int run_slice = tile / img_tiles;
    if (stats && sl.n > 1 && tid < 2 * CO && tid % CO < cout)
        for (int s_ = 0; s_ < sl.n; ++s_) sl.stats[s_][((int64_t)blockIdx.x * 2 + tid / CO) * cout + tid % CO] = 0.0;
#define WS_FLUSH_STATS() {                                                                                            \
        if (tid < 2 * CO && tid % CO < cout)                                                                          \
            (sl.n > 1 ? sl.stats[run_slice] : stats)[((int64_t)blockIdx.x * 2 + tid / CO) * cout + tid % CO] = run_sum; \
        run_sum = 0.0; }
    for (; tile < n_tiles; tile += gridDim.x) {
        const int slice = tile / img_tiles, itile = tile - slice * img_tiles;
        if (stats && slice != run_slice) { WS_FLUSH_STATS() run_slice = slice; }
        const int tb = itile / per_img, trem = itile - tb * per_img;
        const int y0 = (trem / tiles_x) * TR, x0 = (trem % tiles_x) * DC_TW;
        float* __restrict__ Ys = sl.n > 1 ? sl.y[slice] : Y;
#pragma unroll
        for (int m = 0; m < M16; ++m)
#pragma unroll
            for (int t = 0; t < N16; ++t) acc[m][t] = ws_v4f{0.f, 0.f, 0.f, 0.f};
        // Fragment addresses of double stage K_ of a quad: lane groups 0, 1 read stage 2 K_ (its halo image and tap position, its
        // weight slot), groups 2, 3 stage 2 K_ + 1
#define WS16_ADDR(K_)                                                                                                     \
        const int ja_ = 2 * (K_), jb_ = 2 * (K_) + 1;                                                                       \
        const int off_a_ = ((ja_ / 9) & 1) * ASZ + (((ja_ % 9) / 3) * DC_HW + (ja_ % 9) % 3) * DC_ROWB;                     \
        const int off_b_ = ((jb_ / 9) & 1) * ASZ + (((jb_ % 9) / 3) * DC_HW + (jb_ % 9) % 3) * DC_ROWB;                     \
        const unsigned char* Ap = As + a_lane + (sel ? off_b_ : off_a_);                                                    \
        const unsigned char* Bp = Bs + b_lane + 2 * ((K_) % 3) * BSZ;
#define WS16_LD_A(M_, P_) fa[M_][P_] = *reinterpret_cast<const mf_v8h*>(Ap + (P_) * APL + (((M_) >> 1) * DC_HW + ((M_) & 1) * 16) * DC_ROWB);
#define WS16_LD_B(T_, P_) fb[T_][P_] = *reinterpret_cast<const mf_v8h*>(Bp + (P_) * BPL + (T_) * 16 * DC_ROWB);
#define WS16_MFMA(M_, T_, PA, PB) acc[M_][T_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[M_][PA], fb[T_][PB], acc[M_][T_], 0, 0, 0);
        {   // the tile's first double stage: all 24 fragments at once (its data was written during the previous tile's last one)
            WS16_ADDR(0)
#pragma unroll
            for (int m = 0; m < M16; ++m) { WS16_LD_A(m, 1) }
#pragma unroll
            for (int t = 0; t < N16; ++t) { WS16_LD_B(t, 0) }
#pragma unroll
            for (int m = 0; m < M16; ++m) { WS16_LD_A(m, 0) }
#pragma unroll
            for (int t = 0; t < N16; ++t) { WS16_LD_B(t, 1) }
        }
        // One fragment set, refilled IN PLACE as its pieces die (a second set does not fit beside 128 accumulators): double
        // stage k multiplies plane 1 of the pixels x plane 0 of the weights, then plane 0 x plane 1 - after which both plane-1
        // sets are dead; the stage barrier makes the producers' data of k + 1 visible; the plane-1 sets of k + 1 are requested
        // into the dead registers and travel while plane 0 x plane 0 of k runs column tile by column tile, each tile's weight
        // fragment re-requested for k + 1 as soon as its four products are issued; the pixels' plane 0 follows at the end and
        // arrives during the first product group of k + 1, which does not need it. Per accumulator the products still arrive
        // cross terms first, the large one last.
        for (int q = 0; q < nquads; ++q) {
#pragma unroll
            for (int k = 0; k < 18; ++k) {
#pragma unroll
                for (int t = 0; t < N16; ++t)
#pragma unroll
                    for (int m = 0; m < M16; ++m) { WS16_MFMA(m, t, 1, 0) }
#pragma unroll
                for (int t = 0; t < N16; ++t)
#pragma unroll
                    for (int m = 0; m < M16; ++m) { WS16_MFMA(m, t, 0, 1) }
                WS_CONSUMER_BARRIER(a_lane, b_lane)
                // (the tile's last double stage requests the next tile's first like any other - no branch in the stream, the
                // registers are not carried through the epilogue: the next tile starts by reading its set again)
                // The order of this barrier-to-barrier region, pinned (left alone the scheduler sinks every read to just before its
                // first use and waits there): the 12 plane-1 reads, (4 products, 1 read) x N16, the M16 pixel reads, then the
                // next double stage's first two product groups.
                __builtin_amdgcn_sched_group_barrier(0x100, M16 + N16, 0);
#pragma unroll
                for (int t = 0; t < N16; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, M16, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, M16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * M16 * N16, 0);
                WS16_ADDR((k + 1) % 18)
#pragma unroll
                for (int m = 0; m < M16; ++m) { WS16_LD_A(m, 1) }
#pragma unroll
                for (int t = 0; t < N16; ++t) { WS16_LD_B(t, 1) }
#pragma unroll
                for (int t = 0; t < N16; ++t) {
#pragma unroll
                    for (int m = 0; m < M16; ++m) { WS16_MFMA(m, t, 0, 0) }
                    WS16_LD_B(t, 0)
                }
#pragma unroll
                for (int m = 0; m < M16; ++m) { WS16_LD_A(m, 0) }
            }
        }
#undef WS16_ADDR
#undef WS16_LD_A
#undef WS16_LD_B
#undef WS16_MFMA
        // ---- epilogue: as dense_conv3x3_ws_kernel, in the 16 x 16 tiles' register layout
        {
            const float dx = h2_descale(sbx), dw = h2_descale(sbw);
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int t = 0; t < N16; ++t) acc[m][t] = acc[m][t] * dx * dw;
        }
        float s1[N16], s2[N16];
#pragma unroll
        for (int t = 0; t < N16; ++t) { s1[t] = 0.0f; s2[t] = 0.0f; }
        if (bn.y) {
            float bsc[N16], bsh[N16], bmu[N16], biv[N16];
#pragma unroll
            for (int t = 0; t < N16; ++t) {
                const int c = t * 16 + c16;
                bmu[t] = bn.mean[c]; biv[t] = bn.invstd[c];
                gga_bn_scale_shift(bn.gamma ? bn.gamma[c] : 1.0f, bn.beta ? bn.beta[c] : 0.0f, bmu[t], biv[t], bsc[t], bsh[t]);
            }
#pragma unroll
            for (int m = 0; m < M16; ++m) {
                const int oy = y0 + MT * wave + (m >> 1);
                if (oy >= H) continue;
                float yv[4][N16];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ox = x0 + (m & 1) * 16 + 4 * g + v;
                    const float* src = bn.y + ((int64_t)tb * H * W + oy * prow + (ox < W ? ox : W - 1) * pcol) * bn.ystride;
#pragma unroll
                    for (int t = 0; t < N16; ++t) yv[v][t] = src[t * 16 + c16];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ox = x0 + (m & 1) * 16 + 4 * g + v;
                    if (ox >= W) continue;
                    float* dst = Ys + ((int64_t)tb * H * W + oy * prow + ox * pcol) * ystride;
#pragma unroll
                    for (int t = 0; t < N16; ++t) {
                        const float gv = fmaf(yv[v][t], bsc[t], bsh[t]) > 0.0f ? acc[m][t][v] : 0.0f;
                        dst[t * 16 + c16] = gv;
                        s1[t] += gv; s2[t] += gv * ((yv[v][t] - bmu[t]) * biv[t]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < M16; ++m) {
                const int oy = y0 + MT * wave + (m >> 1);
                if (oy >= H) continue;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ox = x0 + (m & 1) * 16 + 4 * g + v;
                    if (ox >= W) continue;
                    float* dst = Ys + ((int64_t)tb * H * W + oy * prow + ox * pcol) * ystride;
#pragma unroll
                    for (int t = 0; t < N16; ++t) dst[t * 16 + c16] = acc[m][t][v];
                }
            }
        }
        if (stats) {
            if (!bn.y)
#pragma unroll
            for (int m = 0; m < M16; ++m) {
                const bool rowok = y0 + MT * wave + (m >> 1) < H;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool ok = rowok && x0 + (m & 1) * 16 + 4 * g + v < W;
#pragma unroll
                    for (int t = 0; t < N16; ++t) {
                        const float a = ok ? acc[m][t][v] : 0.0f;
                        s1[t] += a; s2[t] += a * a;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < N16; ++t) {
                s1[t] += __shfl_xor(s1[t], 16); s1[t] += __shfl_xor(s1[t], 32);
                s2[t] += __shfl_xor(s2[t], 16); s2[t] += __shfl_xor(s2[t], 32);
                if (g == 0) { red[(wave * 2 + 0) * CO + t * 16 + c16] = s1[t]; red[(wave * 2 + 1) * CO + t * 16 + c16] = s2[t]; }
            }
            __syncthreads();
            if (tid < 2 * CO) {
                const int which = tid / CO, c = tid - which * CO;
                if (c < cout) {
                    double a = 0.0;
#pragma unroll
                    for (int w_ = 0; w_ < 4; ++w_) a += (double)red[(w_ * 2 + which) * CO + c];
                    run_sum += a;
                }
            }
            __syncthreads();
        }
    }
    if (stats) { WS_FLUSH_STATS() }
#undef WS_FLUSH_STATS
}
#undef WS_AOFF
#undef WS_LOAD_PIECE
#undef WS_STORE_PIECE
#undef WS_BLD
#undef WS_BST

