// a3': the sparse convolution products on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32) - the generation of round 1,
// kept as the path for widths the split-plane kernels do not take (input or output channels not a multiple of 4, e.g. the
// 5-feature inputs of other voxel encoders) and as the exact-fp32 cross-check of the parity tests: forward / backward-data
// (gga_sparse_conv_apply, weights packed by gga_sparse_pack_weight) and the weight gradient with float atomics
// (gga_sparse_conv_wgrad). Rule books: sparse_index.hip.
#include <stdlib.h>

#include "gga_common.h"
#include <type_traits>
#include <hip/hip_fp16.h>
#include "conv_planes.h"

// ------------------------------------------------------------------------------ convolution
// Y[r, :] = sum_k X[map[kk][r], :] @ W[k]   (kk = K-1-k when `flip`) on v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD).
//
// A 256-thread workgroup owns 128 output rows (taken through `perm`, which orders rows by their
// neighbour bit mask so that a tile's rows use the same kernel offsets) x NT*32 output channels;
// wave w owns rows 32w..32w+31 and all NT column tiles (NT*16 accumulator registers). Offsets
// whose bit is clear in the OR of the tile's row masks are skipped without touching memory, and a
// wave skips the MFMAs of offsets none of its own 32 rows uses.
//
// Work is a flat sequence of (offset, 32-input-channel chunk) stages. Per stage the gathered
// input rows [128 x 32] and the weight slice [32 x NT*32] sit in LDS in *fragment order*: the
// 32x32x2 A operand of lane (h = lane/32, m = lane%32) at k-step s is A[m][2s+h], so row m keeps
// its even channels in floats 0..15 and its odd channels in 16..31 and a lane fetches four
// k-steps with one ds_read_b128; the weights are packed the same way on the host side of the
// ABI (gga_sparse_pack_weight), 16*NT contiguous floats per lane and stage, so staging them is a
// straight 16-byte copy. Row strides of 36 / 16*NT+4 floats keep the b128 reads conflict-free.
// The global loads of stage i+1 (and the rule-book entries of the offset after it) are issued
// before the MFMAs of stage i and land in LDS after them.
#define MF_TM 128
#define MF_ASTR 36


// packed[k][chunk][lane = h*32+n][g][t][j] = W[k][chunk*32 + 2*(4g+j) + h][t*32 + n]
__global__ __launch_bounds__(256) void sp_pack_weight_kernel(const float* __restrict__ W, int kvol, int cin, int cout,
                                                            int nt, int transpose, int64_t total,
                                                            float* __restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int per_lane = 16 * nt, per_stage = 64 * per_lane;
    const int nchunks = (cin + MF_TK - 1) / MF_TK;
    const int64_t stage = i / per_stage;
    int r = (int)(i - stage * per_stage);
    const int k = (int)(stage / nchunks), ch = (int)(stage - (int64_t)k * nchunks);
    const int lane = r / per_lane; r -= lane * per_lane;
    const int g = r / (4 * nt); r -= g * 4 * nt;
    const int t = r >> 2, j = r & 3;
    const int c = ch * MF_TK + 2 * (4 * g + j) + (lane >> 5), o = t * 32 + (lane & 31);
    float v = 0.0f;
    if (c < cin && o < cout)
        v = transpose ? W[((int64_t)k * cout + o) * cin + c] : W[((int64_t)k * cin + c) * cout + o];
    P[i] = v;
}

extern "C" size_t gga_sparse_packed_weight_bytes(int kvol, int cin, int cout) {
    if (kvol < 1 || cin < 1 || cout < 1 || cout > 128) return 0;
    return (size_t)kvol * ((cin + MF_TK - 1) / MF_TK) * 64 * 16 * mf_nt(cout) * sizeof(float);
}

extern "C" int gga_sparse_pack_weight(const float* weight, int kvol, int cin, int cout, int transpose, float* packed,
                                      void* stream) {
    GGA_REQUIRE(weight && packed, "gga_sparse_pack_weight: null pointer argument");
    GGA_REQUIRE(kvol >= 1 && cin >= 1 && cout >= 1 && cout <= 128, "gga_sparse_pack_weight: bad sizes (kvol=%d cin=%d cout=%d; cout <= 128)",
                kvol, cin, cout);
    const int64_t total = (int64_t)(gga_sparse_packed_weight_bytes(kvol, cin, cout) / sizeof(float));
    hipLaunchKernelGGL(sp_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, kvol, cin, cout, mf_nt(cout), transpose, total, packed);
    GGA_CHECK_LAUNCH("sp_pack_weight_kernel");
    return GGA_OK;
}

template <int NT, bool VEC>
__global__ __launch_bounds__(256) void sp_conv_mfma_kernel(const float* __restrict__ X, const int32_t* __restrict__ map,
                                                          const float* __restrict__ Wp,
                                                          const int32_t* __restrict__ perm,
                                                          const uint32_t* __restrict__ rowmask, int64_t n_rows,
                                                          int kvol, int cin, int cout, int flip,
                                                          float* __restrict__ Y) {
    constexpr int BL = 16 * NT;            // packed weight floats per lane and stage
    constexpr int BSTR = BL + 4;           // LDS stride of a lane's block
    constexpr int ASZ = MF_TM * MF_ASTR, BSZ = 64 * BSTR;
    __shared__ __attribute__((aligned(16))) float As[2 * ASZ];      // double buffered: one barrier per stage
    __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];
    __shared__ int prow[MF_TM];
    __shared__ uint32_t wmask_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // mask-sorted order puts the rows with the most neighbours last: start those tiles first
    const int64_t r0 = (int64_t)(gridDim.x - 1 - blockIdx.x) * MF_TM;
    if (tid < 4) wmask_s[tid] = 0;
    __syncthreads();
    if (tid < MF_TM) {
        const int64_t r = r0 + tid;
        int pr = -1;
        if (r < n_rows) pr = perm ? perm[r] : (int)r;
        prow[tid] = pr;
        uint32_t m = 0;
        if (pr >= 0) m = (rowmask && kvol <= 32) ? rowmask[pr] : 0xFFFFFFFFu;
        if (m) atomicOr(&wmask_s[tid >> 5], m);
    }
    __syncthreads();
    // wave-uniform: keep them in scalar registers so the offset scan below is scalar code
    const uint32_t wmask = __builtin_amdgcn_readfirstlane(wmask_s[wave]);
    const uint32_t tmask = __builtin_amdgcn_readfirstlane(wmask_s[0] | wmask_s[1] | wmask_s[2] | wmask_s[3]);
    mf_v16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const int nchunks = (cin + MF_TK - 1) / MF_TK;
    // staging roles. A: thread (ar = tid/4, aq = tid%4) loads float4 #aq and #aq+4 of the 32-channel
    // chunk for rows ar and ar+64. B: NT float4 of the packed stage, consecutive across threads.
    const int ar = tid >> 2, aq = tid & 3;
    const int p0 = prow[ar], p1 = prow[ar + 64];
    auto enabled = [&](int k) { const int kk = flip ? (kvol - 1 - k) : k; return kvol > 32 || ((tmask >> kk) & 1u); };
    auto next_enabled = [&](int k) { while (k < kvol && !enabled(k)) ++k; return k; };
    auto load_idx = [&](int k, int& i0, int& i1) {
        const int kk = flip ? (kvol - 1 - k) : k;
        const int32_t* mk = map + (int64_t)kk * n_rows;
        i0 = mk[p0 >= 0 ? p0 : 0];        // rows past n_rows gather something valid; they are never written
        i1 = mk[p1 >= 0 ? p1 : 0];
    };
    float4 a0[2], a1[2];
    float4 bq0, bq1, bq2, bq3;             // named (not an array): keeps them in registers across the MFMA phase
    bq0 = bq1 = bq2 = bq3 = make_float4(0.f, 0.f, 0.f, 0.f);
    // loads are unconditional (absent neighbours / channels past cin read row 0 / channel 0 and
    // are zeroed when they are written to LDS), so nothing waits on them before the MFMAs
    auto load_stage = [&](int k, int ch, int i0, int i1) {
        const int c0 = ch * MF_TK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cc = c0 + (aq + 4 * h) * 4;
            if (VEC) {
                const int co = cc < cin ? cc : 0;
                a0[h] = *reinterpret_cast<const float4*>(X + (int64_t)(i0 >= 0 ? i0 : 0) * cin + co);
                a1[h] = *reinterpret_cast<const float4*>(X + (int64_t)(i1 >= 0 ? i1 : 0) * cin + co);
            } else {                       // channel count not a multiple of 4: scalar gathers
                float t0[4], t1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = cc + e < cin ? cc + e : 0;
                    t0[e] = X[(int64_t)(i0 >= 0 ? i0 : 0) * cin + co];
                    t1[e] = X[(int64_t)(i1 >= 0 ? i1 : 0) * cin + co];
                }
                a0[h] = make_float4(t0[0], t0[1], t0[2], t0[3]);
                a1[h] = make_float4(t1[0], t1[1], t1[2], t1[3]);
            }
        }
        const float4* src = reinterpret_cast<const float4*>(Wp + ((int64_t)k * nchunks + ch) * (64 * BL));
        bq0 = src[tid];
        if (NT > 1) bq1 = src[tid + 256];
        if (NT > 2) { bq2 = src[tid + 512]; bq3 = src[tid + 768]; }
    };
    auto store_stage = [&](int buf, int ch, int i0, int i1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = aq + 4 * h;      // channels 4q..4q+3 -> k-steps 2q, 2q+1 of halves 0 (x, z) and 1 (y, w)
            const int cc = ch * MF_TK + 4 * q;
            const bool v0 = i0 >= 0, v1 = i1 >= 0;
            const bool cx = cc < cin, cy = cc + 1 < cin, cz = cc + 2 < cin, cw = cc + 3 < cin;
            float* d0 = As + buf * ASZ + ar * MF_ASTR + 2 * q;
            float* d1 = d0 + 64 * MF_ASTR;
            *reinterpret_cast<float2*>(d0) = make_float2(v0 && cx ? a0[h].x : 0.f, v0 && cz ? a0[h].z : 0.f);
            *reinterpret_cast<float2*>(d0 + 16) = make_float2(v0 && cy ? a0[h].y : 0.f, v0 && cw ? a0[h].w : 0.f);
            *reinterpret_cast<float2*>(d1) = make_float2(v1 && cx ? a1[h].x : 0.f, v1 && cz ? a1[h].z : 0.f);
            *reinterpret_cast<float2*>(d1 + 16) = make_float2(v1 && cy ? a1[h].y : 0.f, v1 && cw ? a1[h].w : 0.f);
        }
        // float4 #f of the stage belongs to lane block f / (4*NT), piece f % (4*NT)
#define MF_BST(E, V) { const int f = tid + 256 * (E); *reinterpret_cast<float4*>(Bs + buf * BSZ + (f / (4 * NT)) * BSTR + (f % (4 * NT)) * 4) = V; }
        MF_BST(0, bq0);
        if (NT > 1) MF_BST(1, bq1);
        if (NT > 2) { MF_BST(2, bq2); MF_BST(3, bq3); }
#undef MF_BST
    };

    // Stage bookkeeping (all wave-uniform): (k, ch) is being multiplied out of LDS buffer `buf`,
    // (k1, ch1) sits in the staging registers (loaded one iteration ago with rule-book entries
    // ia0/ia1), (k2, ch2) is fetched during this iteration. Within an iteration the staging work
    // is placed between the four MFMA groups so its VALU / LDS / VMEM instructions issue in the
    // shadow of the matrix pipe instead of in a separate phase.
    int k = next_enabled(0), ch = 0;
    if (k < kvol) {
        int ia0, ia1, in0n, in1n;
        load_idx(k, ia0, ia1);
        int knext = next_enabled(k + 1);                 // first enabled offset after the one being loaded
        load_idx(knext < kvol ? knext : k, in0n, in1n);
        load_stage(k, 0, ia0, ia1);
        store_stage(0, 0, ia0, ia1);
        int k1 = k, ch1 = 1;
        if (ch1 == nchunks) { ch1 = 0; k1 = knext; }
        auto fetch_next = [&](int kq, int chq) {         // issue the loads of stage (kq, chq); entering a new offset rotates the rule-book registers
            const bool valid = kq < kvol;
            const bool adv = valid && chq == 0;
            ia0 = adv ? in0n : ia0;
            ia1 = adv ? in1n : ia1;
            if (adv) knext = next_enabled(kq + 1);
            load_idx(knext < kvol ? knext : k, in0n, in1n);
            load_stage(valid ? kq : k, valid ? chq : ch, ia0, ia1);
        };
        fetch_next(k1, ch1);
        __syncthreads();
        int buf = 0;
        while (true) {
            const float* Ap = As + buf * ASZ + (wave * 32 + (lane & 31)) * MF_ASTR + (lane >> 5) * 16;
            const float* Bp = Bs + buf * BSZ + lane * BSTR;
            int k2 = k1, ch2 = ch1 + 1;
            if (ch2 == nchunks) { ch2 = 0; k2 = knext; }
            const int kk = flip ? (kvol - 1 - k) : k;
            const bool mm = kvol > 32 || ((wmask >> kk) & 1u);
#define MF_READ(G, S)                                                                                                \
            fa[S] = *reinterpret_cast<const float4*>(Ap + 4 * (G));                                                   \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) fb[S][t] = *reinterpret_cast<const float4*>(Bp + ((G) * NT + t) * 4);
#define MF_MMA(S)                                                                                                    \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S].x, fb[S][t].x, acc[t], 0, 0, 0); \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S].y, fb[S][t].y, acc[t], 0, 0, 0); \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S].z, fb[S][t].z, acc[t], 0, 0, 0); \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S].w, fb[S][t].w, acc[t], 0, 0, 0);
            if (mm) {
                float4 fa[2], fb[2][NT];
                MF_READ(0, 0);
                MF_READ(1, 1);
                MF_MMA(0);
                __builtin_amdgcn_sched_barrier(0);
                store_stage(buf ^ 1, ch1, ia0, ia1);     // buf^1 was last read before the previous barrier
                MF_READ(2, 0);
                __builtin_amdgcn_sched_barrier(0);
                MF_MMA(1);
                __builtin_amdgcn_sched_barrier(0);
                fetch_next(k2, ch2);
                MF_READ(3, 1);
                __builtin_amdgcn_sched_barrier(0);
                MF_MMA(0);
                MF_MMA(1);
            } else {                                     // none of this wave's rows uses the offset
                store_stage(buf ^ 1, ch1, ia0, ia1);
                fetch_next(k2, ch2);
            }
#undef MF_READ
#undef MF_MMA
            if (k1 >= kvol) break;
            __syncthreads();
            buf ^= 1;
            k = k1; ch = ch1; k1 = k2; ch1 = ch2;
        }
    }
    // D layout of 32x32x2: register v of lane l holds row (v/4)*8 + (l/32)*4 + v%4, column l%32
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int lr = wave * 32 + (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
        const int pr = prow[lr];
        if (pr < 0) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = t * 32 + (lane & 31);
            if (o < cout) Y[(int64_t)pr * cout + o] = acc[t][v];
        }
    }
}

extern "C" int gga_sparse_conv_apply(const float* x, const int32_t* map, const float* packed_weight, const int32_t* perm,
                                     const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                                     float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && map && packed_weight && y, "gga_sparse_conv_apply: null pointer argument");
    GGA_REQUIRE(n_rows >= 1 && kvol >= 1 && cin >= 1 && cout >= 1 && cout <= 128,
                "gga_sparse_conv_apply: bad sizes (rows=%lld kvol=%d cin=%d cout=%d; cout <= 128)", (long long)n_rows,
                kvol, cin, cout);
    const dim3 grid((unsigned)((n_rows + MF_TM - 1) / MF_TM)), block(256);
#define MF_LAUNCH(NT, VEC) hipLaunchKernelGGL((sp_conv_mfma_kernel<NT, VEC>), grid, block, 0, stream, x, map, packed_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y)
    if ((cin & 3) == 0) {
        switch (mf_nt(cout)) {
            case 1: MF_LAUNCH(1, true); break;
            case 2: MF_LAUNCH(2, true); break;
            default: MF_LAUNCH(4, true); break;
        }
    } else {
        switch (mf_nt(cout)) {
            case 1: MF_LAUNCH(1, false); break;
            case 2: MF_LAUNCH(2, false); break;
            default: MF_LAUNCH(4, false); break;
        }
    }
#undef MF_LAUNCH
    GGA_CHECK_LAUNCH("sp_conv_mfma_kernel");
    return GGA_OK;
}


// ------------------------------------------------------------------------------ weight gradient
// dW[k] (CI x CO) = Xp^T (CI x pairs) * Gp (pairs x CO) over the valid (input row, output row)
// pairs of offset k, on v_mfma_f32_32x32x2_f32. grid = (2048-row chunks, kvol): a workgroup
// compacts the chunk's valid pairs of its offset into LDS, then walks them 32 at a time: the
// gathered X rows and the G rows of the next 32 pairs are fetched into registers before the
// MFMAs of the current ones and written to the other LDS buffer after them (one barrier per
// stage). The NI x NJ 32x32 tiles of dW[k] are dealt to the 4 waves (tile = wave*TPW + t), so
// the waves of a row of tiles share the X fragment; one atomicAdd per weight and chunk.
#define SP_WCHUNK 2048
template <int NI, int NJ, bool VEC>
__global__ __launch_bounds__(256) void sp_conv_wgrad_mfma_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                const int32_t* __restrict__ map, int64_t n_rows,
                                                                int cin, int cout, float* __restrict__ dW) {
    constexpr int CI = NI * 32, CO = NJ * 32;
    constexpr int TILES = NI * NJ;
    constexpr int TPW = (TILES + 3) / 4;             // tiles per wave
    constexpr int XSZ = 32 * CI, GSZ = 32 * CO;
    __shared__ __attribute__((aligned(16))) float Xs[2 * XSZ];
    __shared__ __attribute__((aligned(16))) float Gs[2 * GSZ];
    __shared__ int pin[SP_WCHUNK];       // compacted valid pairs of the chunk: input row
    __shared__ uint16_t pout[SP_WCHUNK]; //                                      output row (chunk-local)
    __shared__ int npairs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * SP_WCHUNK;
    if (tid == 0) npairs = 0;
    __syncthreads();
    for (int t = tid; t < SP_WCHUNK; t += 256) {
        const int64_t r = r0 + t;
        const int v = r < n_rows ? map[(int64_t)k * n_rows + r] : -1;
        if (v >= 0) { const int p = atomicAdd(&npairs, 1); pin[p] = v; pout[p] = (uint16_t)t; }
    }
    __syncthreads();
    const int np = npairs;
    if (np == 0) return;
    mf_v16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    // staging registers: NI float4 of X and NJ float4 of G per thread and stage. Loads are
    // unconditional (pairs past np re-read pair 0, channels past cin/cout re-read channel 0)
    // and zeroed when they are written to LDS.
    float4 xr[NI], gr[NJ];
#define WG_LOAD(P0)                                                                                                  \
    _Pragma("unroll") for (int e = 0; e < NI; ++e) {                                                                 \
        const int t = tid + 256 * e, pp = t / (CI / 4), q = (t - pp * (CI / 4)) * 4;                                 \
        const int pi = (P0) + pp < np ? (P0) + pp : 0;                                                               \
        const float* src = X + (int64_t)pin[pi] * cin;                                                               \
        if (VEC) xr[e] = *reinterpret_cast<const float4*>(src + (q < cin ? q : 0));                                  \
        else xr[e] = make_float4(src[q < cin ? q : 0], src[q + 1 < cin ? q + 1 : 0], src[q + 2 < cin ? q + 2 : 0],    \
                                 src[q + 3 < cin ? q + 3 : 0]);                                                      \
    }                                                                                                                \
    _Pragma("unroll") for (int e = 0; e < NJ; ++e) {                                                                 \
        const int t = tid + 256 * e, pp = t / (CO / 4), q = (t - pp * (CO / 4)) * 4;                                 \
        const int pi = (P0) + pp < np ? (P0) + pp : 0;                                                               \
        const float* src = G + (r0 + pout[pi]) * cout;                                                               \
        if (VEC) gr[e] = *reinterpret_cast<const float4*>(src + (q < cout ? q : 0));                                 \
        else gr[e] = make_float4(src[q < cout ? q : 0], src[q + 1 < cout ? q + 1 : 0], src[q + 2 < cout ? q + 2 : 0], \
                                 src[q + 3 < cout ? q + 3 : 0]);                                                     \
    }
#define WG_STORE(BUF, P0)                                                                                            \
    _Pragma("unroll") for (int e = 0; e < NI; ++e) {                                                                 \
        const int t = tid + 256 * e, pp = t / (CI / 4), q = (t - pp * (CI / 4)) * 4;                                 \
        const bool ok = (P0) + pp < np;                                                                              \
        const float4 v = make_float4(ok && q < cin ? xr[e].x : 0.f, ok && q + 1 < cin ? xr[e].y : 0.f,               \
                                     ok && q + 2 < cin ? xr[e].z : 0.f, ok && q + 3 < cin ? xr[e].w : 0.f);          \
        *reinterpret_cast<float4*>(Xs + (BUF) * XSZ + pp * CI + q) = v;                                              \
    }                                                                                                                \
    _Pragma("unroll") for (int e = 0; e < NJ; ++e) {                                                                 \
        const int t = tid + 256 * e, pp = t / (CO / 4), q = (t - pp * (CO / 4)) * 4;                                 \
        const bool ok = (P0) + pp < np;                                                                              \
        const float4 v = make_float4(ok && q < cout ? gr[e].x : 0.f, ok && q + 1 < cout ? gr[e].y : 0.f,             \
                                     ok && q + 2 < cout ? gr[e].z : 0.f, ok && q + 3 < cout ? gr[e].w : 0.f);        \
        *reinterpret_cast<float4*>(Gs + (BUF) * GSZ + pp * CO + q) = v;                                              \
    }
    WG_LOAD(0);
    WG_STORE(0, 0);
    __syncthreads();
    int buf = 0;
    // a wave's TPW tiles sit in one row of tiles: i0 = tile0 / NJ, columns j0 .. j0+TPW-1, so one
    // X fragment serves all of them (all wave-uniform -> scalar address math)
    static_assert(NJ % TPW == 0, "tiles of a wave must share their tile row");
    const int tile0 = __builtin_amdgcn_readfirstlane(wave) * TPW;
    const bool wactive = tile0 < TILES;
    const int i0 = tile0 / NJ, j0 = tile0 - i0 * NJ;
    const int m = lane & 31, h = lane >> 5;
    for (int p0 = 0; p0 < np; p0 += 32) {
        WG_LOAD(p0 + 32);
        if (wactive) {
            const float* xb = Xs + buf * XSZ + h * CI + i0 * 32 + m;
            const float* gb = Gs + buf * GSZ + h * CO + j0 * 32 + m;
            float fa[2], fb[2][TPW];
#define WG_READ(S2, S)                                                                                               \
            fa[S] = xb[2 * (S2) * CI];                                                                               \
            _Pragma("unroll") for (int t = 0; t < TPW; ++t) fb[S][t] = gb[2 * (S2) * CO + t * 32];
            WG_READ(0, 0);
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                if (s2 + 1 < 16) { WG_READ(s2 + 1, (s2 + 1) & 1); }
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s2 & 1], fb[s2 & 1][t], acc[t], 0, 0, 0);
            }
#undef WG_READ
        }
        if (p0 + 32 >= np) break;
        buf ^= 1;                            // last read before the previous barrier
        WG_STORE(buf, p0 + 32);
        __syncthreads();
    }
#undef WG_LOAD
#undef WG_STORE
    float* dWk = dW + (int64_t)k * cin * cout;
    if (!wactive) return;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int co = (j0 + t) * 32 + (lane & 31);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = i0 * 32 + (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
            if (ci < cin && co < cout && acc[t][v] != 0.0f) atomicAdd(&dWk[(int64_t)ci * cout + co], acc[t][v]);
        }
    }
}

extern "C" int gga_sparse_conv_wgrad(const float* x, const float* grad_out, const int32_t* map, int64_t n_rows,
                                     int kvol, int cin, int cout, float* grad_weight, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && grad_out && map && grad_weight, "gga_sparse_conv_wgrad: null pointer argument");
    GGA_REQUIRE(n_rows >= 1 && kvol >= 1 && cin >= 1 && cin <= 128 && cout >= 1 && cout <= 128,
                "gga_sparse_conv_wgrad: bad sizes (cin, cout <= 128)");
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_SPARSE_WGRAD, GGA_TIMING_CONV_KEY(cin, cout, 0));
    GGA_TIME_START(tev, stream);
    GGA_CHECK_HIP(hipMemsetAsync(grad_weight, 0, (size_t)kvol * cin * cout * sizeof(float), stream), "wgrad memset");
    const dim3 grid((unsigned)((n_rows + SP_WCHUNK - 1) / SP_WCHUNK), kvol), block(256);
    const int ni = (cin + 31) / 32, nj = (cout + 31) / 32;
    const bool vec = (cin & 3) == 0 && (cout & 3) == 0;
#define MW(NI, NJ) { if (vec) hipLaunchKernelGGL((sp_conv_wgrad_mfma_kernel<NI, NJ, true>), grid, block, 0, stream, x, grad_out, map, n_rows, cin, cout, grad_weight); \
                     else hipLaunchKernelGGL((sp_conv_wgrad_mfma_kernel<NI, NJ, false>), grid, block, 0, stream, x, grad_out, map, n_rows, cin, cout, grad_weight); }
    if (ni == 1 && nj == 1) MW(1, 1)
    else if (ni == 1 && nj == 2) MW(1, 2)
    else if (ni == 2 && nj == 2) MW(2, 2)
    else if (ni == 2 && nj == 4) MW(2, 4)
    else if (ni == 4 && nj == 4) MW(4, 4)
    else if (ni <= 2 && nj <= 2) MW(2, 2)
    else MW(4, 4)
#undef MW
    GGA_CHECK_LAUNCH("sp_conv_wgrad_mfma_kernel");
    GGA_TIME_STOP(tev, stream);
    return GGA_OK;
}
