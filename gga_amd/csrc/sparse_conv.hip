// a3': sparse 3D convolution products for the SECOND-style SparseEncoder (gfx950), on the 16-bit matrix instructions through
// operand planes (conv_planes.h): the gather-GEMM forward / backward-data kernel (sp_conv_x9_kernel: also the stride-2,
// transposed, 1x1 and pillar-restricted convolutions of the BEV trunk through arithmetic rule books), its halo form for the
// submanifold convolutions of densely populated levels (sp_conv_halo_kernel), the deterministic weight gradient
// (sp_conv_wgrad_x9_kernel), the weight packing and the absmax pass. Index structures and rule books: sparse_index.hip;
// fp32-MFMA generation for odd widths: sparse_conv_f32.hip; the 3x3 stride-1 dense kernels: dense_conv.hip.
// Reference: mmdet3d/models/middle_encoders/sparse_encoder.py:107-214, mmdet3d/ops/sparse_block.py:82-199 (layers of the
// un-vendored mmcv / spconv wheels).
// Experiments that were measured and not shipped (an LDS-DMA ring form of the gather-GEMM, ablation builds of every kernel,
// in-kernel cycle accounting) are in the history up to commit a966a5a; EXPERIMENTS.md 6c has their numbers.
#include <stdlib.h>

#include "gga_common.h"
#include <type_traits>
#include <hip/hip_fp16.h>
#include "conv_planes.h"

// largest finite |x| of a [rows, width] matrix (row stride in floats), as float bits, by atomicMax into *out (zeroed first)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t rows, int width4, int64_t row_stride,
                                                    uint32_t* __restrict__ out) {
    const int64_t n4 = rows * width4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t m = 0;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                      // eight independent 16-byte loads in flight per thread
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                const int64_t rrow = row_stride == (int64_t)width4 * 4 ? 0 : i / width4;
                v[u] = *reinterpret_cast<const float4*>(x + rrow * row_stride + (i - rrow * width4) * 4);
            } else v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t q[4] = {__float_as_uint(v[u].x) & 0x7FFFFFFFu, __float_as_uint(v[u].y) & 0x7FFFFFFFu,
                                   __float_as_uint(v[u].z) & 0x7FFFFFFFu, __float_as_uint(v[u].w) & 0x7FFFFFFFu};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (q[j] < 0x7F800000u && q[j] > m) m = q[j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    // one atomic per workgroup, and only if it can still raise the result: thousands of same-address atomics
    // serialise in the L2 (measured: 16 k of them cost 150 us)
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
    }
}

extern "C" int gga_absmax_bits(const float* x, int64_t rows, int width, int64_t row_stride, uint32_t* out_bits, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && out_bits && rows >= 0 && width >= 4 && width % 4 == 0 && row_stride >= width && row_stride % 4 == 0 &&
                    ((uintptr_t)x & 15) == 0, "gga_absmax_bits: need a 16-byte aligned matrix with width and row stride %% 4 == 0");
    GGA_CHECK_HIP(hipMemsetAsync(out_bits, 0, sizeof(uint32_t), stream), "gga_absmax_bits: memset");
    if (rows == 0) return GGA_OK;
    const int64_t n4 = rows * (width / 4);
    int64_t nb = (n4 + 256 * 8 - 1) / (256 * 8);
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, rows, width / 4, row_stride, out_bits);
    GGA_CHECK_LAUNCH("absmax_kernel");
    return GGA_OK;
}

// packed[k][chunk][plane][col < CO][32 ch] bf16 = plane of W[k][chunk*32 + ch][col], CO = 32 * nt (quarters swizzled, below)
__global__ __launch_bounds__(256) void sp_pack_weight_split_kernel(const float* __restrict__ W, int kvol, int cin, int cout,
                                                                  int nt, int transpose, int64_t total, int np,
                                                                  const uint32_t* __restrict__ amax_w,
                                                                  uint16_t* __restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one (k, chunk, col, ch) per thread
    if (i >= total) return;
    const int co = 32 * nt, nchunks = (cin + MF_TK - 1) / MF_TK;
    const int ch = (int)(i & 31);
    const int col = (int)((i >> 5) % co);
    const int64_t stage = (i >> 5) / co;
    const int k = (int)(stage / nchunks), chunk = (int)(stage - (int64_t)k * nchunks);
    const int c = chunk * MF_TK + ch;
    float v = 0.0f;
    if (c < cin && col < cout)
        v = transpose ? W[((int64_t)k * cout + col) * cin + c] : W[((int64_t)k * cin + c) * cout + col];
    // the four 16-byte quarters of a column's 64-byte row are stored XOR-swizzled by (col >> 2) & 3: an LDS image that is a
    // plain copy of a packed stage (what an LDS-DMA produces) is then read conflict-free by ds_read_b128 without padding
    uint16_t* dst = P + stage * (np * (int64_t)co * 32) + (int64_t)col * 32 + ((((ch >> 3) ^ ((col >> 2) & 3)) << 3) | (ch & 7));
    if (np == 3) {
        uint32_t p1, p2, p3;
        x9_split(v, p1, p2, p3);
        dst[0] = (uint16_t)p1; dst[(int64_t)co * 32] = (uint16_t)p2; dst[2 * (int64_t)co * 32] = (uint16_t)p3;
    } else {                                  // two fp16 planes of the scaled weight (h2_split2)
        uint32_t w0, w1;
        h2_split2(v * h2_scale(h2_scale_exp(*amax_w)), 0.0f, w0, w1);
        dst[0] = (uint16_t)(w0 & 0xFFFFu); dst[(int64_t)co * 32] = (uint16_t)(w1 & 0xFFFFu);
    }
}

extern "C" size_t gga_sparse_split_weight_bytes(int kvol, int cin, int cout) {
    if (kvol < 1 || cin < 1 || cout < 1 || cout > 128) return 0;
    return (size_t)kvol * ((cin + MF_TK - 1) / MF_TK) * 3 * 32 * mf_nt(cout) * 32 * sizeof(uint16_t);
}

extern "C" int gga_sparse_pack_weight_split(const float* weight, int kvol, int cin, int cout, int transpose, void* packed,
                                            void* stream) {
    return gga_sparse_pack_weight_planes(weight, kvol, cin, cout, transpose, 3, nullptr, packed, stream);
}

extern "C" int gga_sparse_pack_weight_planes(const float* weight, int kvol, int cin, int cout, int transpose, int planes,
                                             const uint32_t* amax_weight, void* packed, void* stream) {
    GGA_REQUIRE(weight && packed, "gga_sparse_pack_weight_split: null pointer argument");
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_weight), "gga_sparse_pack_weight_split: planes must be 3 (bf16) or 2 (fp16, with amax_weight)");
    GGA_REQUIRE(kvol >= 1 && cin >= 1 && cout >= 1 && cout <= 128,
                "gga_sparse_pack_weight_split: bad sizes (kvol=%d cin=%d cout=%d; cout <= 128)", kvol, cin, cout);
    const int64_t total = (int64_t)(gga_sparse_split_weight_bytes(kvol, cin, cout) / (3 * sizeof(uint16_t)));
    hipLaunchKernelGGL(sp_pack_weight_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, kvol, cin, cout, mf_nt(cout), transpose, total, planes, amax_weight, (uint16_t*)packed);
    GGA_CHECK_LAUNCH("sp_pack_weight_split_kernel");
    return GGA_OK;
}

// Tile: 128 output rows x CO columns per 256-thread workgroup; wave w owns rows 32w..32w+31.
// The A operand never touches LDS: lane (r = lane%32, h = lane/32) of a k-step needs the 8
// channels 16s + 8h .. +7 of ITS OWN row, which are 32 contiguous bytes of the gathered fp32 row -
// so every lane fetches the 16 floats of its row straight from global memory (one stage ahead),
// splits them into the three planes in registers and feeds them to the MFMAs. Only the weight
// stage (24 KB for CO = 128, shared by the 4 waves) goes through LDS, double buffered: one
// barrier per stage, and a wave's gathers depend on nobody else. Two workgroups per CU.
// In-kernel timestamps (wall_clock64 per phase and wave) put a stage at ~3.3 us: 1.9 us are the
// two waves of a SIMD sharing the matrix pipe at full rate (2 x 72 MFMAs x 32 cycles), the rest is
// address processing of the gathers, the weight copy and the barrier, during which the pipe
// idles - 57 % busy. A-through-LDS forms, 256-row tiles, rows requested two stages ahead, and
// alternating gather-first / MFMA-first roles for the two waves of a SIMD all measured the same
// 2.4-2.5 ms; a third wave per SIMD does not fit the registers (spills: 3.6 ms).
#define X9_NW 4
#define X9_TM (32 * X9_NW)

// Backward-data launches whose result is the gradient of z = relu(bn(y)) take that BatchNorm's backward reduce pass into
// their epilogue, like the dense kernel (DcBnBwd): the rows are masked by the ReLU recomputed from y (row = output row,
// row stride ystride floats) before they are stored, and `stats` receives the sums of g and g * xhat.
struct SpBnBwd {
    const float* y;
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* invstd;
    int64_t ystride;
};

// Epilogue of the gather-GEMM kernels: rescale (two-plane arithmetic), store the rows through the row -> output row table,
// the optional BatchNorm-backward masking (SpBnBwd) and the per-channel sums of the tile (`stats`, row `tile`).
// D layout of 32x32x16: register v of lane l holds row (v/4)*8 + (l/32)*4 + v%4, column l%32; the row -> output row table
// goes through LDS (each lane knows only its own row). `scratch`: LDS no wave reads any more, >= max(NW * 128, NW * 8 * CO) bytes.
// RB: 32-row blocks per wave (accumulators accs[rb], output rows prs[rb]); block rb of wave w is row block w * RB + rb of the tile.
template <int NT, int NP, int NW, int RB>
__device__ __forceinline__ void x9_epilogue_rb(mf_v16 (&accs)[RB][NT], const int (&prs)[RB], const int wave, const int r, const int h,
                                               const int tid, const int cout, float* __restrict__ Y, const int64_t ys,
                                               double* __restrict__ stats, const int64_t tile, const SpBnBwd& bn,
                                               unsigned char* scratch, const int sbx, const int sbw) {
    constexpr int CO = NT * 32;
    if (NP == 2) {
        const float dx = h2_descale(sbx), dw = h2_descale(sbw);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accs[rb][t][i] = accs[rb][t][i] * dx * dw;
    }
    __syncthreads();
    int* prow = reinterpret_cast<int*>(scratch);
    if (h == 0)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) prow[(wave * RB + rb) * 32 + r] = prs[rb];
    __syncthreads();
    float s1[NT], s2[NT];                                  // per-column sums of the lane's 16 rows (stats)
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.0f; s2[t] = 0.0f; }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
    mf_v16 (&acc)[NT] = accs[rb];
    const int vw = wave * RB + rb;
    if (bn.y) {                                            // see SpBnBwd
        float bsc[NT], bsh[NT], bmu[NT], biv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = t * 32 + r < cout ? t * 32 + r : 0;
            bmu[t] = bn.mean[o]; biv[t] = bn.invstd[o];
            gga_bn_scale_shift(bn.gamma ? bn.gamma[o] : 1.0f, bn.beta ? bn.beta[o] : 0.0f, bmu[t], biv[t], bsc[t], bsh[t]);
        }
        constexpr int VB = 32 / NT < 16 ? 32 / NT : 16;    // 32 values of y requested before the first is used
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += VB) {
            float yv[VB][NT];
            int pos[VB];
#pragma unroll
            for (int j = 0; j < VB; ++j) {
                const int v = v0 + j;
                pos[j] = prow[vw * 32 + (v >> 2) * 8 + h * 4 + (v & 3)];
                const float* src = bn.y + (int64_t)(pos[j] >= 0 ? pos[j] : 0) * bn.ystride;
#pragma unroll
                for (int t = 0; t < NT; ++t) yv[j][t] = src[t * 32 + r < cout ? t * 32 + r : 0];
            }
#pragma unroll
            for (int j = 0; j < VB; ++j) {
                if (pos[j] < 0) continue;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int o = t * 32 + r;
                    if (o >= cout) continue;
                    const float g = fmaf(yv[j][t], bsc[t], bsh[t]) > 0.0f ? acc[t][v0 + j] : 0.0f;
                    Y[(int64_t)pos[j] * ys + o] = g;
                    s1[t] += g; s2[t] += g * ((yv[j][t] - bmu[t]) * biv[t]);
                }
            }
        }
    } else
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int lr = vw * 32 + (v >> 2) * 8 + h * 4 + (v & 3);
        const int po = prow[lr];
        if (po < 0) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = t * 32 + r;
            if (o < cout) Y[(int64_t)po * ys + o] = acc[t][v];
            s1[t] += acc[t][v]; s2[t] += acc[t][v] * acc[t][v];
        }
    }
    }
    if (stats) {
        // per-channel sum and sum of squares of the workgroup's rows (the batch statistics of the BatchNorm that follows,
        // as in the dense kernel): [workgroup][2][cout] f64
        __syncthreads();                                  // prow (in scratch) has been read by every wave
        float* red = reinterpret_cast<float*>(scratch);
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
                for (int w_ = 0; w_ < NW; ++w_) a += (double)red[(w_ * 2 + which) * CO + c];
                stats[(tile * 2 + which) * cout + c] = a;
            }
        }
    }
}

template <int NT, int NP, int NW>
__device__ __forceinline__ void x9_epilogue(mf_v16 (&acc)[NT], const int pr, const int wave, const int r, const int h,
                                            const int tid, const int cout, float* __restrict__ Y, const int64_t ys,
                                            double* __restrict__ stats, const int64_t tile, const SpBnBwd& bn,
                                            unsigned char* scratch, const int sbx, const int sbw) {
    const int prs[1] = {pr};
    x9_epilogue_rb<NT, NP, NW, 1>(reinterpret_cast<mf_v16 (&)[1][NT]>(acc), prs, wave, r, h, tid, cout, Y, ys, stats, tile, bn, scratch, sbx, sbw);
}

// (two waves per SIMD in the launch bounds: with the 512-register budget of ONE wave per SIMD the compiler gives the MFMAs
// accumulator-file destinations and then copies all 64 accumulators to and from the vector file around every stage -
// 128 v_accvgpr_read + 192 v_accvgpr_write in the stage loop of the 128-column form; with <= 256 registers it keeps them in place)
template <int NT, bool VEC, int NP>
__global__ __launch_bounds__(64 * X9_NW, 2) void sp_conv_x9_kernel(const float* __restrict__ X, const int32_t* __restrict__ map,
                                                        const uint16_t* __restrict__ Wp,
                                                        const int32_t* __restrict__ perm,
                                                        const uint32_t* __restrict__ rowmask, int64_t n_rows,
                                                        int kvol, int cin, int cout, int flip,
                                                        float* __restrict__ Y, int64_t ys,
                                                        const uint32_t* __restrict__ amax_x,
                                                        const uint32_t* __restrict__ amax_w, double* __restrict__ stats,
                                                        SpBnBwd bn, int tile_order, int64_t n_tiles, int offset_sums) {
    // NP = 3: bf16 planes, six partial products; NP = 2: fp16 planes of the scaled operands, three (h2_split2)
    // offset_sums (round 6, the default): the matrix instructions of ONE offset (cin / 16 k-steps x 3 or 6 products) run as
    // their own chain from zero and the offset's partial result is added to the row's sum once - the summation order of the
    // reference's gather -> GEMM -> scatter-add per offset (spconv / mmcv: one sgemm of K = cin per offset, 27 additions),
    // with chains of 3 cin / 16 roundings instead of ONE chain of 27 x 3 cin / 16: the accumulator's rounding noise
    // ~ 0.4 u sqrt(n / 2) for a chain of n additions falls from 7 u (cin 128) to 0.4 u sqrt((3 cin / 16 + 27) / 2) = 2 u
    // (EXPERIMENTS.md 6f). Costs cout / 4 v_pk_add_f32 per offset beside cin / 16 x 3 x cout / 32 matrix instructions.
    constexpr int CO = NT * 32;
    constexpr int BPL = CO * X9_ROWB;                     // bytes per B plane
    constexpr int BSZ = NP * BPL;                         // bytes per B buffer
    constexpr int BPIECES = NP * CO * 4;                  // 16-byte pieces of a packed weight stage
    int sbx = 127, sbw = 127;
    if (NP == 2) { sbx = h2_scale_exp(*amax_x); sbw = h2_scale_exp(*amax_w); }
    const float xscale = h2_scale(sbx);
    constexpr int NW = X9_NW, THREADS = 64 * NW;
    constexpr int NB = (BPIECES + THREADS - 1) / THREADS;   // pieces per thread
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * BSZ];
    __shared__ uint32_t wmask_s[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // tile of this workgroup. tile_order 0: tiles in reverse (the mask-sorted order puts the rows with most neighbours
    // last: those start first). 1: XCD-major - workgroups are dealt to the 8 XCDs round robin, so workgroup b takes tile
    // (b % 8) * ceil(tiles / 8) + b / 8: every XCD walks a contiguous range of tiles in order and the rows its tiles gather
    // (neighbours of a spatially ordered row range, see sparse.py) stay in that XCD's L2. The grid is rounded up to 8 * ceil.
    int64_t tile = (int64_t)gridDim.x - 1 - blockIdx.x;
    if (tile_order == 1) {
        const int64_t per = (n_tiles + 7) / 8;
        tile = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((int64_t)(blockIdx.x >> 3) >= per || tile >= n_tiles) return;
    }
    const int64_t r0 = tile * (32 * NW);
    if (tid < NW) wmask_s[tid] = 0;
    __syncthreads();
    const int64_t myrow = r0 + wave * 32 + r;
    const int pr = myrow < n_rows ? (perm ? perm[myrow] : (int)myrow) : -1;
    {
        uint32_t m = 0;
        if (pr >= 0) m = (rowmask && kvol <= 32) ? rowmask[pr] : 0xFFFFFFFFu;
        if (m && h == 0) atomicOr(&wmask_s[wave], m);
    }
    __syncthreads();
    const uint32_t wmask = __builtin_amdgcn_readfirstlane(wmask_s[wave]);
    uint32_t tm = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tm |= wmask_s[w];
    const uint32_t tmask = __builtin_amdgcn_readfirstlane(tm);
    mf_v16 acc[NT], total[NT];                            // the running offset's chain; the sum of the finished offsets
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[t][i] = 0.0f; total[t][i] = 0.0f; }

    const int nchunks = (cin + MF_TK - 1) / MF_TK;
    auto enabled = [&](int k) { const int kk = flip ? (kvol - 1 - k) : k; return kvol > 32 || ((tmask >> kk) & 1u); };
    auto next_enabled = [&](int k) { while (k < kvol && !enabled(k)) ++k; return k; };
    auto load_idx = [&](int k) {
        const int kk = flip ? (kvol - 1 - k) : k;
        return map[(int64_t)kk * n_rows + (pr >= 0 ? pr : 0)];   // rows past n_rows gather something valid; never written
    };
    // raw[s][j]: floats 16s + 8h + 4j .. +3 of the lane's row in the 32-channel chunk
    float4 rn00, rn01, rn10, rn11;                        // next stage, in flight
    float4 rc00, rc01, rc10, rc11;                        // current stage
    uint4 bq0, bq1, bq2, bq3, bq4, bq5;
    bq0 = bq1 = bq2 = bq3 = bq4 = bq5 = make_uint4(0, 0, 0, 0);
    rn00 = rn01 = rn10 = rn11 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld4 = [&](const float* row, int c) -> float4 {
        if (VEC) return *reinterpret_cast<const float4*>(row + (c < cin ? c : 0));
        return make_float4(row[c < cin ? c : 0], row[c + 1 < cin ? c + 1 : 0], row[c + 2 < cin ? c + 2 : 0],
                           row[c + 3 < cin ? c + 3 : 0]);
    };
    // loads are unconditional: an absent neighbour reads row 0 and is zeroed when it is split
    auto load_a = [&](int ch, int i0) {
        const float* row = X + (int64_t)(i0 >= 0 ? i0 : 0) * cin;
        const int c = ch * MF_TK + 8 * h;
        rn00 = ld4(row, c); rn01 = ld4(row, c + 4); rn10 = ld4(row, c + 16); rn11 = ld4(row, c + 20);
    };
    auto load_b = [&](int k, int ch) {
        const uint4* src = reinterpret_cast<const uint4*>(Wp + ((int64_t)k * nchunks + ch) * (NP * CO * 32));
        const int last = BPIECES - 1;
#define X9_BLD(E, V) if ((E) < NB) V = src[min(tid + THREADS * (E), last)];
        X9_BLD(0, bq0) X9_BLD(1, bq1) X9_BLD(2, bq2) X9_BLD(3, bq3) X9_BLD(4, bq4) X9_BLD(5, bq5)
#undef X9_BLD
    };
    auto store_b = [&](int buf) {
        // piece f of the packed stage: plane f / (CO*4), column (f / 4) % CO, quarter f % 4
#define X9_BST(E, V) if ((E) < NB) { const int f = tid + THREADS * (E); if (f < BPIECES) *reinterpret_cast<uint4*>(Bs + buf * BSZ + (f / (CO * 4)) * BPL + ((f >> 2) % CO) * X9_ROWB + (((f & 3) ^ ((f >> 4) & 3)) * 16)) = V; }
        X9_BST(0, bq0) X9_BST(1, bq1) X9_BST(2, bq2) X9_BST(3, bq3) X9_BST(4, bq4) X9_BST(5, bq5)
#undef X9_BST
    };
    // 8 floats -> one 8 x bf16 fragment per plane
    union Frag { mf_v8bf v; uint32_t u[4]; };
    auto split8 = [&](const float4& lo, const float4& hi, bool ok, int c, Frag& f1, Frag& f2, Frag& f3) {
        if (VEC && NP == 2) {
            // (cin % 4 == 0: a float4 is inside the channels or outside; an absent neighbour or a float4 beyond cin is zeroed by
            // its SCALE - two selects per fragment instead of a compare and a select per element: 46 of the stage's 83 vector
            // instructions, each of which costs the matrix pipe ~9 cycles)
            const float slo = ok && c < cin ? xscale : 0.0f, shi = ok && c + 4 < cin ? xscale : 0.0f;
            h2_split2s(lo.x, lo.y, slo, f1.u[0], f2.u[0]); h2_split2s(lo.z, lo.w, slo, f1.u[1], f2.u[1]);
            h2_split2s(hi.x, hi.y, shi, f1.u[2], f2.u[2]); h2_split2s(hi.z, hi.w, shi, f1.u[3], f2.u[3]);
            return;
        }
        const float e[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = ok && c + 2 * j < cin ? e[2 * j] : 0.f, y = ok && c + 2 * j + 1 < cin ? e[2 * j + 1] : 0.f;
            if (NP == 3) x9_split2(x, y, f1.u[j], f2.u[j], f3.u[j]);
            else h2_split2s(x, y, xscale, f1.u[j], f2.u[j]);
        }
    };

    int k = next_enabled(0), ch = 0;
    if (k < kvol) {
        int ia = load_idx(k);
        int knext = next_enabled(k + 1);
        int ian = load_idx(knext < kvol ? knext : k);
        load_a(0, ia);
        load_b(k, 0);
        store_b(0);
        rc00 = rn00; rc01 = rn01; rc10 = rn10; rc11 = rn11;
        int ic = ia;                                     // rule-book entry the current stage was loaded with
        int k1 = k, ch1 = 1;
        if (ch1 == nchunks) { ch1 = 0; k1 = knext; }
        __syncthreads();
        int buf = 0;
        while (true) {
            // request stage (k1, ch1): rows into the rn registers, weights into bq
            const bool valid1 = k1 < kvol;
            const bool adv = valid1 && ch1 == 0;
            ia = adv ? ian : ia;
            if (adv) knext = next_enabled(k1 + 1);
            ian = load_idx(knext < kvol ? knext : k);
            load_a(valid1 ? ch1 : ch, ia);
            load_b(valid1 ? k1 : k, valid1 ? ch1 : ch);
            const int kk = flip ? (kvol - 1 - k) : k;
            if (kvol > 32 || ((wmask >> kk) & 1u)) {
                Frag a0[3], a1[3];
                const int c = ch * MF_TK + 8 * h;
                split8(rc00, rc01, ic >= 0, c, a0[0], a0[1], a0[2]);
                split8(rc10, rc11, ic >= 0, c + 16, a1[0], a1[1], a1[2]);
                const unsigned char* Bp = Bs + buf * BSZ + r * X9_ROWB + h * 16;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const Frag* a = s ? a1 : a0;
                    mf_v8bf b[NT][NP];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int p = 0; p < NP; ++p) b[t][p] = *reinterpret_cast<const mf_v8bf*>(Bp + p * BPL + t * 32 * X9_ROWB + s * 32);
                    // the nine partial products, smallest first; the column tiles are the inner loop so
                    // that consecutive MFMAs never wait for each other's accumulator
#define X9_MM(PA, PB) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA].v, b[t][PB], acc[t], 0, 0, 0);
#define X9_MH(PA, PB) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_v8h, a[PA].v), __builtin_bit_cast(mf_v8h, b[t][PB]), acc[t], 0, 0, 0);
                    if (NP == 2) { X9_MH(0, 1) X9_MH(1, 0) X9_MH(0, 0) }
                    else {
                    X9_MM(0, NP - 1) X9_MM(1, 1) X9_MM(NP - 1, 0) X9_MM(0, 1) X9_MM(1, 0) X9_MM(0, 0)       // six terms, see the dense kernels
                    }
#undef X9_MH
#undef X9_MM
                }
                if (offset_sums && ch == nchunks - 1) {   // the offset's chain is complete: one addition per element, a fresh chain
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        total[t] += acc[t];
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
                    }
                }
            }
            if (!valid1) break;
            store_b(buf ^ 1);                            // last read before the previous barrier
            rc00 = rn00; rc01 = rn01; rc10 = rn10; rc11 = rn11;
            ic = ia;
            __syncthreads();
            buf ^= 1;
            k = k1; ch = ch1;
            ++ch1;
            if (ch1 == nchunks) { ch1 = 0; k1 = knext; }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] += total[t];      // (offset_sums: acc is zero here; otherwise total is)
    x9_epilogue<NT, NP, NW>(acc, pr, wave, r, h, tid, cout, Y, ys, stats, tile, bn, Bs, sbx, sbw);
}

// ------------------------------------------------------------------------------ the same product, halo form (SubM)
// What bounds the two forms above at the 128-channel level of the shipped config (EXPERIMENTS.md 6c) is not the matrix pipe:
//   bytes   sp_conv_x9_kernel moves 9.2 GB L2 -> CU per launch (every row 18 x its 512 bytes, a 16 KB weight stage per 128
//           rows and stage): 28 GB/s per CU, what 64 KB in flight per CU get from beyond the L2;
//   issue   every gathered fp32 element is split into its two fp16 planes by the lane that feeds it to the MFMA - ~12 vector
//           instructions per pair of elements, 27 times per element: ~900 issue cycles per wave and stage next to 768 cycles
//           of matrix work.
// This form removes both. The rows are tiled in SPATIAL order (256 consecutive rows of a Z-ordered level, sparse.py::_Halo),
// so the 27 x 256 neighbours of a tile are only ~1.4 x 256 DISTINCT rows (its halo). Per 32-channel chunk the halo is
// fetched ONCE into an LDS image (512 rows x 128 B, LDS-DMA, whole 128-byte lines), split ONCE, in place, into the two fp16
// planes (the same 128 bytes per row), and all 27 offsets read their A fragments from the image - ready to use - through a
// per-tile local rule book (u16 [kvol][256] = position in the halo list, 0xFFFF = no neighbour; in LDS for the whole tile).
//   stages  chunk-outer: stage s = (chunk s / kvol, offset s % kvol); weights by LDS-DMA into a ring of 3 slots, 2 stages
//           ahead (the packed stage is its own LDS image, sp_pack_weight_split_kernel), counted s_waitcnt vmcnt(N) behind a
//           raw s_barrier as in sp_conv_ring_kernel;
//   image   row L at byte L * 128 = [plane 0: 32 ch fp16 | plane 1]; its eight 16-byte segments (plane p, channels 8g..8g+7 =
//           segment 4p + g) XOR-ed with (L >> 1) & 7: 16 lanes reading one segment of rows L .. L+15 hit 16 bank groups.
//           Row 512 is all zero: what a lane without a neighbour reads. The wave that requested 8 rows splits them (the 8
//           lanes of a row read its fp32 segments in one instruction and write the planes with the next ones);
//   switch  the image is single: at a chunk boundary the waves request the next chunk's halo after the barrier, wait, split
//           and meet again (the weight ring keeps running ahead meanwhile) - ~3 us per chunk next to ~20 us of its stages;
//   spill   halo positions >= 512 (a tile whose neighbourhood is wider than the image) are read from global memory and split
//           by the lane that needs them, through the tile's halo list - slow, correct, rare;
//   skip    a wave whose 32 rows have no neighbour at an offset skips that stage's MFMAs (ballot of its lanes).
// Two fp16 planes only (the three-plane arithmetic stays on sp_conv_x9_kernel). D layout and epilogue (BatchNorm sums /
// BatchNorm-backward masking): those of sp_conv_x9_kernel; a row's sum runs over the same partial products, chunk-outer
// instead of offset-outer, so the two forms differ by fp32 summation order only.
#define XH_TM 256
#define XH_HCAP 512
#define XH_KMAX 27
template <int NT>
__global__ __launch_bounds__(256) void sp_conv_halo_kernel(const float* __restrict__ X, const uint16_t* __restrict__ Wp,
                                                          const int32_t* __restrict__ tperm, const int32_t* __restrict__ hcount,
                                                          int hcap, const int32_t* __restrict__ hlist_all,
                                                          const uint16_t* __restrict__ lmap, int64_t n_tiles, int kvol, int cin, int cout, int flip,
                                                          float* __restrict__ Y, int64_t ys, const uint32_t* __restrict__ amax_x,
                                                          const uint32_t* __restrict__ amax_w, double* __restrict__ stats,
                                                          SpBnBwd bn, int64_t stats_rows) {
    constexpr int CO = NT * 32, NW = 4, RB = 2, THREADS = 64 * NW, TM = XH_TM, D = 3, R = D + 1, HCAP = XH_HCAP, NP = 2;
    constexpr int A_IMG = (HCAP + 1) * 128;               // bytes of the halo image (+ the zero row)
    constexpr int B_PL = CO * 64, B_SLOT = NP * B_PL;     // bytes per plane / per packed weight stage
    constexpr int BPIECES = B_SLOT / 16;
    constexpr int NBP = BPIECES / THREADS;
    static_assert(BPIECES % THREADS == 0, "every thread moves NBP pieces of a weight stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // ALL of the kernel's LDS (one object)
    unsigned char* const Bring = smem + A_IMG;
    uint16_t* const lm = reinterpret_cast<uint16_t*>(Bring + R * B_SLOT);      // [XH_KMAX][TM]
    int* const hl = reinterpret_cast<int*>(lm + XH_KMAX * TM);                 // [HCAP]
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int sbx = h2_scale_exp(*amax_x), sbw = h2_scale_exp(*amax_w);
    const float xscale = h2_scale(sbx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t tile = blockIdx.x;
    int pr[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) pr[rb] = tperm[tile * TM + (wave * RB + rb) * 32 + r];
    const int32_t* const hlist = hlist_all + tile * hcap;
    const int hn = hcount[tile];                          // (submanifold maps: >= 1, a row is its own centre neighbour)
    // prologue (ordinary loads, nothing else in flight): the tile's local rule book and halo list into LDS, the zero row
    {
        const uint4* src = reinterpret_cast<const uint4*>(lmap + tile * (int64_t)kvol * TM);
        const int pieces = kvol * TM / 8;
        for (int i = tid; i < pieces; i += THREADS) reinterpret_cast<uint4*>(lm)[i] = src[i];
        for (int i = tid; i < HCAP; i += THREADS) hl[i] = hn > 0 ? hlist[i < hn ? i : hn - 1] : 0;     // (a tile without any entry: row 0, never read)
        if (tid < 32) reinterpret_cast<uint32_t*>(smem + HCAP * 128)[tid] = 0u;
    }
    __syncthreads();
    const int nchunks = cin / MF_TK;
    const int S = nchunks * kvol;
    mf_v16 acc[RB][NT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][t][i] = 0.0f;

    auto dma16 = [&](const void* src, uint32_t dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    // halo DMA q of chunk c: rows 8q .. 8q+7 of the image, 8 lanes per row; lane -> position lane & 7 of its row, which
    // receives fp32 segment (lane & 7) ^ ((row >> 1) & 7)
    const int nq = ((hn < HCAP ? hn : HCAP) + 7) >> 3;      // instructions that carry halo rows (the rest of the image is never read)
    auto issue_a = [&](int c, int q) {
        const int j = 8 * q + (lane >> 3);
        dma16(X + (int64_t)hl[j] * cin + c * MF_TK + 4 * ((lane & 7) ^ ((j >> 1) & 7)), __builtin_amdgcn_readfirstlane(lds0 + q * 1024));
    };
    // in place: fp32 segment f (channels 4f .. 4f+3) of row j -> halves (f & 1) of plane segments f >> 1 and 4 + (f >> 1)
    auto split_rows = [&](int q) {
        const int j = 8 * q + (lane >> 3), sw = (j >> 1) & 7, f = (lane & 7) ^ sw;
        unsigned char* row = smem + j * 128;
        const float4 v = *reinterpret_cast<const float4*>(row + ((lane & 7) << 4));
        uint2 p0, p1;
        h2_split2s(v.x, v.y, xscale, p0.x, p1.x);
        h2_split2s(v.z, v.w, xscale, p0.y, p1.y);
        *reinterpret_cast<uint2*>(row + ((((f >> 1)) ^ sw) << 4) + ((f & 1) << 3)) = p0;
        *reinterpret_cast<uint2*>(row + (((4 + (f >> 1)) ^ sw) << 4) + ((f & 1) << 3)) = p1;
    };
    constexpr int nbp = NBP;                              // a wave's weight DMAs per stage (pieces tid + THREADS e)
    int tb = 0, cb = 0, kb = 0;                            // next weight stage to request: number, chunk, offset
    const unsigned char* bsrc = nullptr;
    uint32_t b_dst = 0;
    auto issue_b_begin = [&]() {
        bsrc = reinterpret_cast<const unsigned char*>(Wp) + ((int64_t)kb * nchunks + cb) * B_SLOT + (int64_t)tid * 16;
        b_dst = __builtin_amdgcn_readfirstlane(lds0 + A_IMG + (tb & (R - 1)) * B_SLOT + wave * 1024);
        ++tb;
        if (++kb == kvol) { kb = 0; ++cb; }
    };
    auto issue_b_piece = [&](int e) { dma16(bsrc + e * (THREADS * 16), __builtin_amdgcn_readfirstlane(b_dst + e * (THREADS * 16))); };
    auto issue_b = [&]() {
        issue_b_begin();
#pragma unroll
        for (int e = 0; e < NBP; ++e) issue_b_piece(e);
    };
    static_assert(R == 4, "ring slot = stage & 3");
    auto wait_vm = [&](int n) {
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        }
    };
    static_assert(NBP == 2 || NBP == 4, "wait_vm covers 0 or NBP outstanding DMAs");
    union Frag { mf_v8h v; uint32_t u[4]; uint4 q; };
    struct Half { Frag a[RB][2]; mf_v8h b[NT][NP]; };     // operands of one 16-channel k-step: A row blocks x planes, B tiles x planes
    const int swz = (r >> 2) & 3;
    auto load_b = [&](Half& f, int s, int sk) {
        const unsigned char* Bp = Bring + (s & (R - 1)) * B_SLOT + r * 64 + (((sk * 2 + h) ^ swz) * 16);
#pragma unroll
        for (int p = NP - 1; p >= 0; --p)                 // plane 1 first: the first products are a0 x b1
#pragma unroll
            for (int t = 0; t < NT; ++t) f.b[t][p] = *reinterpret_cast<const mf_v8h*>(Bp + p * B_PL + t * 32 * 64);
    };
#define XH_MH(PA, PB) _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[rb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[rb][PA].v, f.b[t][PB], acc[rb][t], 0, 0, 0);
    auto mma = [&](const Half& f) { XH_MH(0, 1) XH_MH(1, 0) XH_MH(0, 0) };
    auto mma_head = [&](const Half& f) { XH_MH(0, 1) XH_MH(1, 0) };
#undef XH_MH
    // the last product (a0 x b0) of a k-step in NBP parts: dealt between the weight DMAs of the next stage
    static_assert(RB * NT % NBP == 0, "whole parts");
    auto mma_tail = [&](const Half& f, int e) {
#pragma unroll
        for (int i = e * (RB * NT / NBP); i < (e + 1) * (RB * NT / NBP); ++i)
            acc[i / NT][i % NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i / NT][0].v, f.b[i % NT][0], acc[i / NT][i % NT], 0, 0, 0);
    };
    auto entry = [&](int k, int rb) { return (int)lm[(flip ? (kvol - 1 - k) : k) * TM + (wave * RB + rb) * 32 + r]; };

    // Stage s reads weight slot s & 3. At its top every wave waits for ITS share of the weights of stage s + 1 (only those of
    // stage s + 2 stay in flight), so behind the barrier stages s and s + 1 are complete in LDS and slot (s - 1) & 3 is free
    // for stage s + 3. One wave per SIMD has nobody to hide its LDS latency behind, so the fragments of a k-step are requested
    // one k-step ahead (the second half of stage s fetches the first half of stage s + 1) and the rule-book entries one stage
    // ahead, and the stage body is straight-line code: a request inside a branch makes the compiler drain the LDS queue at
    // the join. That is why the lanes whose neighbour lies beyond the image (FAR) get a loop of their own - taken by a tile
    // only if its halo is longer than the image - and why no wave skips an offset its rows do not use.
    for (int t = 0; t < D && t < S; ++t) issue_b();
    auto stages = [&](auto far_tag) {
        constexpr bool FAR = decltype(far_tag)::value;
        // A fragments of k-step sk for the lane's neighbour at image position L (0xFFFF: none -> the zero row)
        auto load_a = [&](Frag (&fa)[2], int L, int sk, int c) {
            const bool far = FAR && L != 0xFFFF && L >= HCAP;
            const int Lc = L < HCAP ? L : HCAP;
            const unsigned char* Ap = smem + Lc * 128;
            const int sw = (Lc >> 1) & 7;
#pragma unroll
            for (int p = 0; p < 2; ++p) fa[p].q = *reinterpret_cast<const uint4*>(Ap + (((4 * p + 2 * sk + h) ^ sw) << 4));
            if (FAR && __builtin_amdgcn_ballot_w64(far) != 0) {   // beyond the image: from global memory through the halo list
                if (far) {
                    const float* row = X + (int64_t)hlist[L] * cin + c * MF_TK + 8 * h + 16 * sk;
                    const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 4);
                    h2_split2s(lo.x, lo.y, xscale, fa[0].u[0], fa[1].u[0]);
                    h2_split2s(lo.z, lo.w, xscale, fa[0].u[1], fa[1].u[1]);
                    h2_split2s(hi.x, hi.y, xscale, fa[0].u[2], fa[1].u[2]);
                    h2_split2s(hi.z, hi.w, xscale, fa[0].u[3], fa[1].u[3]);
                }
            }
        };
        auto load_as = [&](Half& f, const int (&Ls)[RB], int sk, int c) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) load_a(f.a[rb], Ls[rb], sk, c);
        };
        Half f0, f1;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) f1.a[rb][0].q = make_uint4(0, 0, 0, 0);       // (the tail of "stage -1" adds 0 x 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) f1.b[t][0] = __builtin_bit_cast(mf_v8h, make_uint4(0, 0, 0, 0));
        int c = 0, k = 0, L[RB], Ln[RB];
        for (int s = 0; s < S; ++s) {
            wait_vm(s + 2 < S ? nbp : 0);
            __builtin_amdgcn_s_barrier();
            {                                             // the weights of stage s + 3, under the last MFMAs of stage s - 1
                const bool more = k != 0 && tb < S;       // (at a chunk boundary they follow the image)
                if (more) issue_b_begin();
#pragma unroll
                for (int e = 0; e < NBP; ++e) {
                    __builtin_amdgcn_sched_barrier(0);
                    mma_tail(f1, e);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) issue_b_piece(e);
                }
            }
            if (k == 0) {                                 // chunk boundary: everyone is done with the image
                for (int q = wave; q < nq; q += NW) issue_a(c, q);
                const bool more = tb < S;
                if (more) issue_b();
                wait_vm(more ? nbp : 0);                  // in order: the image (and stage s + 2's weights) have landed
                for (int q = wave; q < nq; q += NW) split_rows(q);
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) L[rb] = entry(0, rb);
                load_as(f0, L, 0, c); load_b(f0, s, 0);
            }
            const int kn = k + 1 < kvol ? k + 1 : k;      // (at the end of a chunk: fetched for nothing, the boundary reloads)
            // one scheduling region per k-step: the MFMAs of the fragments at hand with the LDS reads of the next ones dealt
            // in between (2 MFMAs, 1 read); left alone the scheduler sinks the reads to just before their use
            constexpr int NRD = 2 * RB + NT * NP;         // ds_read_b128 of one k-step's fragments
            constexpr int NM = RB * 3 * NT;               // MFMAs of one k-step
            constexpr int NG = NT == 4 ? 12 : 4;          // groups of NM / NG MFMAs and NRD / NG reads
            static_assert(NM % NG == 0 && NRD % NG == 0, "whole groups");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) Ln[rb] = entry(kn, rb);
            load_as(f1, L, 1, c); load_b(f1, s, 1);
            mma(f0);
            __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);
#pragma unroll
            for (int i = 0; i < NG; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, NM / NG, 0); __builtin_amdgcn_sched_group_barrier(0x100, NRD / NG, 0); }
            __builtin_amdgcn_sched_barrier(0);
            load_as(f0, Ln, 0, c); load_b(f0, s + 1, 0);
            mma_head(f1);                                 // (its last third follows the next barrier)
            constexpr int NH = NM * 2 / 3, NG2 = NT == 4 ? 4 : 4;
            static_assert(NH % NG2 == 0 && NRD % NG2 == 0, "whole groups");
#pragma unroll
            for (int i = 0; i < NG2; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, NRD / NG2, 0); __builtin_amdgcn_sched_group_barrier(0x008, NH / NG2, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) L[rb] = Ln[rb];
            if (++k == kvol) { k = 0; ++c; }
        }
#pragma unroll
        for (int e = 0; e < NBP; ++e) mma_tail(f1, e);
    };
    if (hn > HCAP) stages(std::true_type()); else stages(std::false_type());
    // stats rows of the 128-row tiling this tile does not write (the caller sized `stats` for gga_sparse_conv_apply_tiles)
    if (stats) {
        const int64_t extra = n_tiles + tile;
        if (extra < stats_rows && tid < 2 * cout) stats[extra * 2 * cout + tid] = 0.0;
    }
    x9_epilogue_rb<NT, NP, NW, RB>(acc, pr, wave, r, h, tid, cout, Y, ys, stats, tile, bn, smem, sbx, sbw);
}

extern "C" int64_t gga_sparse_halo_tile_rows(void) { return XH_TM; }

// The tiling sp_conv_halo_kernel walks: one workgroup per tile of 256 rows collects the distinct input rows its kvol x 256
// rule-book entries name in an LDS hash set (open addressing, 8192 slots >= 27 x 256 entries), numbers them in the order of
// their row index (a counting rank over the tile's list while it has <= 1024 entries; longer lists keep the order of arrival)
// and rewrites every entry as its number. Sorted, the tiling is reproducible and a chunk's image is requested in address
// order; the convolution's time does not depend on it measurably (510 k rows x 128 -> 128, same box: 0.90 - 0.93 of the
// default kernel's time sorted, in arrival order and with torch-sorted lists alike; hash-slot order: 0.93 - 0.95).
#define XB_SLOTS 8192
#define XB_SORT 1024
__global__ __launch_bounds__(XH_TM) void sp_halo_build_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ tperm,
                                                             int64_t n_rows, int kvol, int hcap, int32_t* __restrict__ hlist_all,
                                                             int32_t* __restrict__ hcount, uint16_t* __restrict__ lmap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xb_smem[];
    int* const keys = reinterpret_cast<int*>(xb_smem);                              // [XB_SLOTS]
    int* const lst = keys + XB_SLOTS;                                               // [XB_SORT] the first keys, in arrival order
    uint16_t* const rank = reinterpret_cast<uint16_t*>(lst + XB_SORT);              // [XB_SLOTS] arrival number of the slot's key
    uint16_t* const slot = rank + XB_SLOTS;                                         // [XH_KMAX][XH_TM]
    uint16_t* const remap = slot + XH_KMAX * XH_TM;                                 // [XB_SORT] arrival number -> sorted number
    __shared__ int count;
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    for (int i = tid; i < XB_SLOTS; i += XH_TM) keys[i] = -1;
    if (tid == 0) count = 0;
    __syncthreads();
    const int row = tperm[tile * XH_TM + tid];
    for (int k = 0; k < kvol; ++k) {
        const int idx = row >= 0 ? nbr[(int64_t)k * n_rows + row] : -1;
        int at = 0xFFFF;
        if (idx >= 0) {
            at = (int)(((uint32_t)idx * 0x9E3779B1u) >> 19);          // 13 bits
            while (true) {
                const int old = atomicCAS(&keys[at], -1, idx);
                if (old == -1) {                                       // this thread brought the row in: it numbers it
                    const int nr = atomicAdd(&count, 1);
                    rank[at] = (uint16_t)nr;
                    if (nr < XB_SORT) lst[nr] = idx;
                    break;
                }
                if (old == idx) break;
                at = (at + 1) & (XB_SLOTS - 1);
            }
        }
        slot[k * XH_TM + tid] = (uint16_t)at;
    }
    __syncthreads();
    const int n = count;
    const bool sorted = n <= XB_SORT;
    if (sorted)
        for (int e = tid; e < n; e += XH_TM) {
            const int key = lst[e];
            int r = 0;
            for (int j = 0; j < n; ++j) r += lst[j] < key ? 1 : 0;    // (distinct keys: a permutation of 0 .. n-1)
            remap[e] = (uint16_t)r;
        }
    __syncthreads();
    if (tid == 0) hcount[tile] = n;
    int32_t* const hlist = hlist_all + tile * hcap;
    for (int i = tid; i < XB_SLOTS; i += XH_TM) {
        const int key = keys[i];
        if (key >= 0) {
            const int nr = sorted ? remap[rank[i]] : rank[i];
            if (nr < hcap) hlist[nr] = key;
        }
    }
    uint16_t* const lm = lmap + tile * (int64_t)kvol * XH_TM;
    for (int k = 0; k < kvol; ++k) {
        const int at = slot[k * XH_TM + tid];
        lm[k * XH_TM + tid] = at == 0xFFFF ? (uint16_t)0xFFFF : (sorted ? remap[rank[at]] : rank[at]);
    }
}

extern "C" int gga_sparse_halo_build(const int32_t* nbr, const int32_t* tile_rows, int64_t n_rows, int64_t n_tiles, int kvol,
                                     int halo_capacity, int32_t* halo_rows, int32_t* halo_counts, uint16_t* local_map,
                                     void* stream) {
    GGA_REQUIRE(nbr && tile_rows && halo_rows && halo_counts && local_map, "gga_sparse_halo_build: null pointer argument");
    GGA_REQUIRE(n_rows >= 1 && n_tiles == (n_rows + XH_TM - 1) / XH_TM && kvol >= 1 && kvol <= XH_KMAX && halo_capacity >= kvol * XH_TM,
                "gga_sparse_halo_build: bad sizes (rows=%lld tiles=%lld kvol=%d capacity=%d; kvol <= 27, capacity >= kvol * 256)",
                (long long)n_rows, (long long)n_tiles, kvol, halo_capacity);
    constexpr size_t lds = XB_SLOTS * 4 + XB_SORT * 4 + XB_SLOTS * 2 + XH_KMAX * XH_TM * 2 + XB_SORT * 2;
    static bool once = false;
    if (!once) { GGA_CHECK_HIP(hipFuncSetAttribute((const void*)sp_halo_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "sp_halo_build_kernel: LDS size"); once = true; }
    hipLaunchKernelGGL(sp_halo_build_kernel, dim3((unsigned)n_tiles), dim3(XH_TM), lds, (hipStream_t)stream, nbr, tile_rows, n_rows, kvol,
                       halo_capacity, halo_rows, halo_counts, local_map);
    GGA_CHECK_LAUNCH("sp_halo_build_kernel");
    return GGA_OK;
}

extern "C" int gga_sparse_conv_apply_halo(const float* x, const void* split_weight, const int32_t* tile_rows,
                                          const int32_t* halo_counts, int halo_capacity, const int32_t* halo_rows,
                                          const uint16_t* local_map, int64_t n_rows, int64_t n_tiles, int kvol, int cin, int cout, int flip, float* y,
                                          int64_t y_row_stride, int planes, const uint32_t* amax_x, const uint32_t* amax_weight,
                                          double* stats, const float* bn_x, int64_t bn_x_row_stride, const float* bn_gamma,
                                          const float* bn_beta, const float* bn_mean, const float* bn_invstd, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && split_weight && tile_rows && halo_counts && halo_rows && local_map && y && halo_capacity >= 1, "gga_sparse_conv_apply_halo: null pointer argument");
    GGA_REQUIRE(planes == 2 && amax_x && amax_weight, "gga_sparse_conv_apply_halo: two fp16 planes only (planes == 2, with the operands' absmax bits)");
    GGA_REQUIRE(n_rows >= 1 && n_tiles == (n_rows + XH_TM - 1) / XH_TM && kvol >= 8 && kvol <= XH_KMAX && cin >= 32 && cin % MF_TK == 0 &&
                    (cout == 64 || cout == 128) && y_row_stride >= cout,
                "gga_sparse_conv_apply_halo: bad sizes (rows=%lld tiles=%lld kvol=%d cin=%d cout=%d; 8 <= kvol <= 27, cin %% 32 == 0, cout 64 or 128)",
                (long long)n_rows, (long long)n_tiles, kvol, cin, cout);
    GGA_REQUIRE(!bn_x || (stats && bn_mean && bn_invstd && bn_x_row_stride >= cout),
                "gga_sparse_conv_apply_halo: the BatchNorm epilogue needs stats, the saved mean / invstd and a row stride >= cout");
    SpBnBwd bn;
    bn.y = bn_x; bn.gamma = bn_gamma; bn.beta = bn_beta; bn.mean = bn_mean; bn.invstd = bn_invstd; bn.ystride = bn_x_row_stride;
    const int64_t stats_rows = (n_rows + X9_TM - 1) / X9_TM;
    const dim3 grid((unsigned)n_tiles), block(256);
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_SPARSE_CONV, GGA_TIMING_CONV_KEY(cin, cout, kvol));     // (the default kernel's launches: (cin, cout, 0))
    GGA_TIME_START(tev, stream);
#define XH_LAUNCH(NT) { \
        constexpr size_t lds = (size_t)(XH_HCAP + 1) * 128 + (size_t)4 * 2 * NT * 32 * 64 + XH_KMAX * XH_TM * 2 + XH_HCAP * 4; \
        static_assert(lds <= 160 * 1024, "LDS of sp_conv_halo_kernel"); \
        static bool once = false; \
        if (!once) { GGA_CHECK_HIP(hipFuncSetAttribute((const void*)sp_conv_halo_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "sp_conv_halo_kernel: LDS size"); once = true; } \
        hipLaunchKernelGGL((sp_conv_halo_kernel<NT>), grid, block, lds, stream, x, (const uint16_t*)split_weight, tile_rows, halo_counts, halo_capacity, halo_rows, local_map, n_tiles, kvol, cin, cout, flip, y, y_row_stride, amax_x, amax_weight, stats, bn, stats_rows); }
    if (cout == 128) XH_LAUNCH(4) else XH_LAUNCH(2)
#undef XH_LAUNCH
    GGA_CHECK_LAUNCH("sp_conv_halo_kernel");
    GGA_TIME_STOP(tev, stream);
    return GGA_OK;
}

extern "C" int gga_sparse_conv_apply_split_strided(const float* x, const int32_t* map, const void* split_weight,
                                                   const int32_t* perm, const uint32_t* rowmask, int64_t n_rows, int kvol,
                                                   int cin, int cout, int flip, float* y, int64_t y_row_stride, void* stream_) {
    return gga_sparse_conv_apply_planes(x, map, split_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y, y_row_stride, 3,
                                        nullptr, nullptr, stream_);
}

extern "C" int gga_sparse_conv_apply_planes(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                            const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                                            float* y, int64_t y_row_stride, int planes, const uint32_t* amax_x,
                                            const uint32_t* amax_weight, void* stream_) {
    return gga_sparse_conv_apply_stats(x, map, split_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y, y_row_stride, planes,
                                       amax_x, amax_weight, nullptr, stream_);
}

extern "C" int64_t gga_sparse_conv_apply_tiles(int64_t n_rows) { return (n_rows + X9_TM - 1) / X9_TM; }

extern "C" int gga_sparse_conv_apply_stats(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                           const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                                           float* y, int64_t y_row_stride, int planes, const uint32_t* amax_x,
                                           const uint32_t* amax_weight, double* stats, void* stream_) {
    return gga_sparse_conv_apply_bn_bwd(x, map, split_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y, y_row_stride, planes,
                                        amax_x, amax_weight, stats, nullptr, 0, nullptr, nullptr, nullptr, nullptr, stream_);
}

extern "C" int gga_sparse_conv_apply_bn_bwd(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                            const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                                            float* y, int64_t y_row_stride, int planes, const uint32_t* amax_x,
                                            const uint32_t* amax_weight, double* stats, const float* bn_x,
                                            int64_t bn_x_row_stride, const float* bn_gamma, const float* bn_beta,
                                            const float* bn_mean, const float* bn_invstd, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(!bn_x || (stats && bn_mean && bn_invstd && bn_x_row_stride >= cout),
                "gga_sparse_conv_apply_bn_bwd: the BatchNorm epilogue needs stats, the saved mean / invstd and a row stride >= cout");
    SpBnBwd bn;
    bn.y = bn_x; bn.gamma = bn_gamma; bn.beta = bn_beta; bn.mean = bn_mean; bn.invstd = bn_invstd; bn.ystride = bn_x_row_stride;
    GGA_REQUIRE(x && map && split_weight && y, "gga_sparse_conv_apply_split: null pointer argument");
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_x && amax_weight),
                "gga_sparse_conv_apply_split: planes must be 3 (bf16) or 2 (fp16, with the operands' absmax bits)");
    GGA_REQUIRE(n_rows >= 1 && kvol >= 1 && cin >= 1 && cout >= 1 && cout <= 128 && y_row_stride >= cout,
                "gga_sparse_conv_apply_split: bad sizes (rows=%lld kvol=%d cin=%d cout=%d row stride %lld; cout <= 128)",
                (long long)n_rows, kvol, cin, cout, (long long)y_row_stride);
    const int64_t n_tiles = (n_rows + X9_TM - 1) / X9_TM;
    static const int tile_order = getenv("GGA_SP_TILE_ORDER") ? atoi(getenv("GGA_SP_TILE_ORDER")) : 0;
    static const int offset_sums = getenv("GGA_SP_OFFSET_SUMS") ? atoi(getenv("GGA_SP_OFFSET_SUMS")) : 1;      // 0: one chain over all offsets (rounds 1-5)
    const dim3 grid((unsigned)(tile_order == 1 ? 8 * ((n_tiles + 7) / 8) : n_tiles)), block(64 * X9_NW);
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_SPARSE_CONV, GGA_TIMING_CONV_KEY(cin, cout, 0));
    GGA_TIME_START(tev, stream);
#define X9_LAUNCH(NT, VEC) { if (planes == 3) hipLaunchKernelGGL((sp_conv_x9_kernel<NT, VEC, 3>), grid, block, 0, stream, x, map, (const uint16_t*)split_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y, y_row_stride, amax_x, amax_weight, stats, bn, tile_order, n_tiles, offset_sums); \
                             else hipLaunchKernelGGL((sp_conv_x9_kernel<NT, VEC, 2>), grid, block, 0, stream, x, map, (const uint16_t*)split_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y, y_row_stride, amax_x, amax_weight, stats, bn, tile_order, n_tiles, offset_sums); }
    if ((cin & 3) == 0) {
        switch (mf_nt(cout)) {
            case 1: X9_LAUNCH(1, true); break;
            case 2: X9_LAUNCH(2, true); break;
            default: X9_LAUNCH(4, true); break;
        }
    } else {
        switch (mf_nt(cout)) {
            case 1: X9_LAUNCH(1, false); break;
            case 2: X9_LAUNCH(2, false); break;
            default: X9_LAUNCH(4, false); break;
        }
    }
#undef X9_LAUNCH
    GGA_CHECK_LAUNCH("sp_conv_x9_kernel");
    GGA_TIME_STOP(tev, stream);
    return GGA_OK;
}

extern "C" int gga_sparse_conv_apply_split(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                           const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                                           float* y, void* stream_) {
    return gga_sparse_conv_apply_split_strided(x, map, split_weight, perm, rowmask, n_rows, kvol, cin, cout, flip, y, cout, stream_);
}

// ------------------------------------------------------------------------------ weight gradient, bf16 planes
// The same dW[k] = Xp^T Gp on the bf16 matrix cores (six of the nine partial products, as in the
// forward kernel), deterministic: no atomics anywhere.
//   * grid = (row chunks, kvol); a workgroup walks its chunk of output rows in sub-chunks of
//     SPW_SUB rows: the valid (input row, output row) pairs of its offset are compacted into LDS IN
//     ROW ORDER (ballot prefix, not an atomic counter), then consumed 32 pairs (two K-steps of 16) per
//     stage. The GEMM's K is the pair index, and both operands are row-major [pair][channel] in
//     memory, i.e. K-major - the MFMA wants 8 consecutive K of ONE channel per lane. As in
//     dense_wgrad3x3_x9_kernel the LDS images stay pair-major ([channel tile][pair][32 ch] bf16 per
//     plane, 64-byte rows, written with the same split-and-store) and ds_read_b64_tr_b16 transposes
//     on the way out.
//   * the NI x NJ 32x32 tiles of dW[k] are dealt to the four waves by tile rows (a wave's tiles
//     share the X fragment). With fewer than four tile groups (<= 32 or 64 channels) two waves share
//     a group and take one K-step of the stage each; their accumulators are added through LDS.
//   * every workgroup writes its partial dW[k] to workspace[chunk][k]; sp_wgrad_reduce_kernel sums
//     the chunks in a fixed order in f64.
#define SPW_SUB 2048
// Measured on the 510 k-row 128 -> 128 level (round 3, tools_dev/ab_wgrad.sh): a third stage of gathered rows in flight
// (-D2=3, 96 instead of 64 KB per CU) changes nothing, rows in spatial order 3 %, the offsets of a chunk on one XCD
// 4 %, and without the plane split of the staged rows (-DSPW_ABL_NOSPLIT) the kernel is 16 % faster: a stage is 12 MFMAs per
// wave behind ~200 vector instructions of split, masking and address arithmetic for its 16 pairs - issue-bound, like the
// forward kernel's in-register split; the operands would have to arrive as planes to remove it.
#define SPW_SPLIT4(V, SC, lo1, lo2, hi1, hi2) { h2_split2s(V.x, V.y, (SC), lo1, lo2); h2_split2s(V.z, V.w, (SC), hi1, hi2); }
template <int NI, int NJ, bool VEC, int NP>
__global__ __launch_bounds__(256, 2) void sp_conv_wgrad_x9_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                 const int32_t* __restrict__ map, int64_t n_rows, int kvol,
                                                                 int64_t rows_per_chunk, int cin, int cout, int64_t xs,
                                                                 int64_t gs, float* __restrict__ partials,
                                                                 const uint32_t* __restrict__ amax_x,
                                                                 const uint32_t* __restrict__ amax_g, int xcd_major) {
    // NP = 3: bf16 planes, six products; NP = 2: fp16 planes of the scaled operands, three (partials stay scaled)
    float xscale = 1.0f, gscale = 1.0f;
    if (NP == 2) { xscale = h2_scale(h2_scale_exp(*amax_x)); gscale = h2_scale(h2_scale_exp(*amax_g)); }
    constexpr int CI = NI * 32, CO = NJ * 32;
    constexpr int TILES = NI * NJ;
    constexpr int TPW = TILES >= 4 ? TILES / 4 : 1;          // tiles per wave
    constexpr int NG = TILES / TPW;                          // tile groups (1, 2 or 4)
    constexpr int KS = 4 / NG;                               // waves sharing a group, one K-step each (1 or 2; 4 groups -> 1)
    constexpr int KSTEPS = (NI + NJ >= 6) ? 1 : 2;           // K-steps of 16 pairs per stage: wide shapes stage 16 pairs
    constexpr int PAIRS = 16 * KSTEPS;                       //   (60 KB of LDS at 128 x 128: two workgroups per CU)
    constexpr int LX = PAIRS * NI / 32 > 0 ? PAIRS * NI / 32 : 1, LG = PAIRS * NJ / 32 > 0 ? PAIRS * NJ / 32 : 1;
    static_assert(NJ % TPW == 0, "tiles of a wave must share their tile row");
    static_assert(NG == 4 || NG == 2 || NG == 1, "tile groups");
    static_assert(KS == 1 || KSTEPS == 2, "a shared tile group needs two K-steps per stage");
    constexpr int XPL = NI * PAIRS * 64, GPL = NJ * PAIRS * 64;   // bytes per plane of one stage image [ch tile][pair][32 ch]
    constexpr int XSZ = NP * XPL, GSZ = NP * GPL;
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2 * XSZ];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[2 * GSZ];
    // compacted valid pairs of the sub-chunk as BYTE OFFSETS (round 5: multiplied once per pair here, not once per staged piece -
    // a pair's row is fetched by CI / 4 + CO / 4 threads): the input row's inside X, the output row's inside the sub-chunk of G
    // (32 bits: a kernel that meets an X row beyond 4 GiB traps). Entries np .. the next multiple of PAIRS point at row 0 of X /
    // of the sub-chunk: a stage's tail pieces read valid memory and are multiplied by a scale of zero in the split.
    __shared__ uint32_t pin[SPW_SUB];
    __shared__ uint32_t pout[SPW_SUB];
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup -> (offset, row chunk). The kvol workgroups of a chunk read the same G rows (and, on spatially ordered levels,
    // X rows from the same neighbourhood): they must meet in ONE L2. Workgroups are dealt to the 8 XCDs round robin, so
    // workgroup b runs on XCD b % 8: chunk c goes to XCD c % 8 and its offsets follow each other there. (xcd_major: this
    // mapping, grid.x = 8 * ceil(chunks / 8) * kvol; otherwise the plain (offset, chunk) grid of rounds 1-2, where the
    // offsets of a chunk were spread over all eight L2s and every one of them fetched the chunk: 7.3 GB per launch at
    // 510 k rows x 128 -> 128 against 0.58 GB algorithmic.)
    int k = blockIdx.x;
    int64_t chunk = blockIdx.y;
    if (xcd_major) {
        const int q = blockIdx.x >> 3;
        k = q % kvol;
        chunk = (int64_t)(q / kvol) * 8 + (blockIdx.x & 7);
        if (chunk * rows_per_chunk >= n_rows) return;
    }
    const int64_t c0 = chunk * rows_per_chunk;
    const int64_t c1 = c0 + rows_per_chunk < n_rows ? c0 + rows_per_chunk : n_rows;

    // wave -> (tile group, K-step share)
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int grp_w = KS == 1 ? wu : (NG == 2 ? (wu >> 1) : 0);
    const int ks_w = KS == 1 ? 0 : (NG == 2 ? (wu & 1) : wu);     // NG == 1: waves 0, 1 take a K-step each, 2 and 3 only stage
    const bool wactive = KS == 1 || NG == 2 || wu < 2;
    const int tile0 = grp_w * TPW;
    const int i0 = tile0 / NJ, j0 = tile0 - i0 * NJ;

    mf_v16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const int fgrp = lane >> 4, li = lane & 15;
    const int froff = ((8 * (fgrp >> 1) + (li >> 2)) * 64) + (16 * (fgrp & 1) + 4 * (li & 3)) * 2;
    union Frag { mf_v8bf v; dw_v4s h[2]; };
#define SW_FRAG(F, PTR) { F.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dw_v4s*)((PTR) + froff));          \
                          F.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dw_v4s*)((PTR) + froff + 4 * 64)); }

    float4 xa[LX], ga[LG], xb[LX], gb_[LG];              // two stages of gathered rows in flight
    for (int64_t r0 = c0; r0 < c1; r0 += SPW_SUB) {
        // ---- ordered compaction of the sub-chunk's valid pairs: thread t owns rows 8t .. 8t+7
        __syncthreads();                                   // previous sub-chunk's readers of pin / pout / images are done
        int mv[8];
        int cnt = 0;
        {
            const int64_t rb = r0 + 8 * tid;
            const int32_t* mp = map + (int64_t)k * n_rows + rb;
            if (rb + 8 <= c1 && ((((int64_t)k * n_rows + rb) & 3) == 0)) {
                const int4 a = *reinterpret_cast<const int4*>(mp), b = *reinterpret_cast<const int4*>(mp + 4);
                mv[0] = a.x; mv[1] = a.y; mv[2] = a.z; mv[3] = a.w; mv[4] = b.x; mv[5] = b.y; mv[6] = b.z; mv[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) mv[j] = rb + j < c1 ? mp[j] : -1;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt += mv[j] >= 0;
        }
        int incl = cnt;                                    // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int off = incl - cnt, np = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wtot[w]; off += w < wave ? c : 0; np += c; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (mv[j] >= 0) {
                const uint64_t xo = (uint64_t)(uint32_t)mv[j] * (uint64_t)xs * 4u;
                if (xo >> 32) __builtin_trap();
                pin[off] = (uint32_t)xo; pout[off] = (uint32_t)((8 * tid + j) * (int)gs * 4); ++off;
            }
        if (tid < PAIRS && np + tid < SPW_SUB) { pin[np + tid] = 0u; pout[np + tid] = 0u; }
        __syncthreads();
        if (np == 0) continue;
        const char* Gsub = reinterpret_cast<const char*>(G + r0 * gs);

#define SW_LOAD(P0, xr, gr)                                                                                          \
        _Pragma("unroll") for (int e = 0; e < LX; ++e) {                                                             \
            const int t = tid + 256 * e, pp = t / (CI / 4), q = (t - pp * (CI / 4)) * 4;                             \
            const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(X) + pin[(P0) + pp]);    \
            if (VEC) xr[e] = *reinterpret_cast<const float4*>(src + (q < cin ? q : 0));                              \
            else xr[e] = make_float4(src[q < cin ? q : 0], src[q + 1 < cin ? q + 1 : 0], src[q + 2 < cin ? q + 2 : 0], \
                                     src[q + 3 < cin ? q + 3 : 0]);                                                  \
        }                                                                                                            \
        _Pragma("unroll") for (int e = 0; e < LG; ++e) {                                                             \
            const int t = tid + 256 * e, pp = t / (CO / 4), q = (t - pp * (CO / 4)) * 4;                             \
            const float* src = reinterpret_cast<const float*>(Gsub + pout[(P0) + pp]);                               \
            if (VEC) gr[e] = *reinterpret_cast<const float4*>(src + (q < cout ? q : 0));                             \
            else gr[e] = make_float4(src[q < cout ? q : 0], src[q + 1 < cout ? q + 1 : 0], src[q + 2 < cout ? q + 2 : 0], \
                                     src[q + 3 < cout ? q + 3 : 0]);                                                 \
        }
        // float4 q4 (channels 4*q4 .. +3) of pair pp -> channel tile q4 / 8, byte (q4 % 8) * 8 of the pair's 64-byte row
#define SW_SPLIT_STORE(V, BASE, PL, PP, Q4, SC) {                                                                    \
        unsigned char* dst = (BASE) + ((Q4) >> 3) * (PAIRS * 64) + (PP) * 64 + ((Q4) & 7) * 8;                       \
        if (NP == 3) {                                                                                               \
            uint32_t lo1, lo2, lo3, hi1, hi2, hi3;                                                                   \
            x9_split2(V.x * (SC), V.y * (SC), lo1, lo2, lo3); x9_split2(V.z * (SC), V.w * (SC), hi1, hi2, hi3);    /* (three planes: SC is 1 or 0) */ \
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                   \
            *reinterpret_cast<uint2*>(dst + (PL)) = make_uint2(lo2, hi2);                                            \
            *reinterpret_cast<uint2*>(dst + (NP - 1) * (PL)) = make_uint2(lo3, hi3);                                 \
        } else {                                                                                                     \
            uint32_t lo1, lo2, hi1, hi2;                                                                             \
            SPW_SPLIT4(V, SC, lo1, lo2, hi1, hi2)                                                                    \
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo1, hi1);                                                   \
            *reinterpret_cast<uint2*>(dst + (PL)) = make_uint2(lo2, hi2);                                            \
        } }
    // VEC: a piece outside the stage's pairs or the operand's channels is zeroed by its SCALE (one select per piece; the
    // values it read are some valid row's); otherwise element by element as before
#define SW_STORE(BUF, P0, xr, gr)                                                                                    \
        _Pragma("unroll") for (int e = 0; e < LX; ++e) {                                                             \
            const int t = tid + 256 * e, pp = t / (CI / 4), q = (t - pp * (CI / 4)) * 4;                             \
            const bool ok = (P0) + pp < np && pp < PAIRS;                                                            \
            if (VEC) {                                                                                               \
                const float sc_ = ok && q < cin ? xscale : 0.0f;                                                     \
                if (pp < PAIRS) SW_SPLIT_STORE(xr[e], Xs + (BUF) * XSZ, XPL, pp, q >> 2, sc_)                        \
            } else {                                                                                                 \
                const float4 v = make_float4(ok && q < cin ? xr[e].x : 0.f, ok && q + 1 < cin ? xr[e].y : 0.f,       \
                                             ok && q + 2 < cin ? xr[e].z : 0.f, ok && q + 3 < cin ? xr[e].w : 0.f);  \
                if (pp < PAIRS) SW_SPLIT_STORE(v, Xs + (BUF) * XSZ, XPL, pp, q >> 2, xscale)                         \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int e = 0; e < LG; ++e) {                                                             \
            const int t = tid + 256 * e, pp = t / (CO / 4), q = (t - pp * (CO / 4)) * 4;                             \
            const bool ok = (P0) + pp < np && pp < PAIRS;                                                            \
            if (VEC) {                                                                                               \
                const float sc_ = ok && q < cout ? gscale : 0.0f;                                                    \
                if (pp < PAIRS) SW_SPLIT_STORE(gr[e], Gs + (BUF) * GSZ, GPL, pp, q >> 2, sc_)                        \
            } else {                                                                                                 \
                const float4 v = make_float4(ok && q < cout ? gr[e].x : 0.f, ok && q + 1 < cout ? gr[e].y : 0.f,     \
                                             ok && q + 2 < cout ? gr[e].z : 0.f, ok && q + 3 < cout ? gr[e].w : 0.f); \
                if (pp < PAIRS) SW_SPLIT_STORE(v, Gs + (BUF) * GSZ, GPL, pp, q >> 2, gscale)                         \
            }                                                                                                        \
        }
#define SW_MM(PA, PB) _Pragma("unroll") for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA.v, PB[t].v, acc[t], 0, 0, 0);
#define SW_MH(PA, PB) _Pragma("unroll") for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_v8h, PA.v), __builtin_bit_cast(mf_v8h, PB[t].v), acc[t], 0, 0, 0);
#define SW_COMPUTE(BUF)                                                                                              \
        if (wactive) {                                                                                               \
            const unsigned char* xbase = Xs + (BUF) * XSZ + i0 * (PAIRS * 64);                                       \
            const unsigned char* gbase = Gs + (BUF) * GSZ + j0 * (PAIRS * 64);                                       \
            _Pragma("unroll") for (int s = 0; s < KSTEPS; ++s) {                                                     \
                if (KS == 2 && s != ks_w) continue;     /* this K-step belongs to the other wave of the group */     \
                Frag a0, a1, a2;                                                                                     \
                SW_FRAG(a0, xbase + (16 * s) * 64);                                                                  \
                SW_FRAG(a1, xbase + XPL + (16 * s) * 64);                                                            \
                if (NP == 3) { SW_FRAG(a2, xbase + (NP - 1) * XPL + (16 * s) * 64); } else a2 = a1;                  \
                Frag g0[TPW], g1[TPW], g2[TPW];                                                                      \
                _Pragma("unroll") for (int t = 0; t < TPW; ++t) {                                                    \
                    SW_FRAG(g0[t], gbase + t * (PAIRS * 64) + (16 * s) * 64);                                        \
                    SW_FRAG(g1[t], gbase + GPL + t * (PAIRS * 64) + (16 * s) * 64);                                  \
                    if (NP == 3) { SW_FRAG(g2[t], gbase + (NP - 1) * GPL + t * (PAIRS * 64) + (16 * s) * 64); } else g2[t] = g1[t]; \
                }                                                                                                    \
                /* six partial products, smallest first; tiles are the inner loop so consecutive MFMAs never wait */ \
                /* for each other's accumulator */                                                                   \
                if (NP == 3) { SW_MM(a0, g2) SW_MM(a1, g1) SW_MM(a2, g0) SW_MM(a0, g1) SW_MM(a1, g0) SW_MM(a0, g0) }    \
                else { SW_MH(a0, g1) SW_MH(a1, g0) SW_MH(a0, g0) }                                                   \
            }                                                                                                        \
        }
        // The gathers are latency-bound (512-byte rows at random): a stage's loads are issued TWO stages
        // ahead (register sets a / b alternate), the LDS images are double buffered, one barrier per stage.
        SW_LOAD(0, xa, ga);
        SW_STORE(0, 0, xa, ga);
        if (PAIRS < np) { SW_LOAD(PAIRS, xb, gb_); }
        __syncthreads();
        int p0 = 0;                                        // (the image a stage multiplies alternates 0, 1: literal in each half)
        while (true) {
            if (p0 + 2 * PAIRS < np) { SW_LOAD(p0 + 2 * PAIRS, xa, ga); }
            SW_COMPUTE(0)
            if (p0 + PAIRS >= np) break;
            SW_STORE(1, p0 + PAIRS, xb, gb_);              // image 1: last read before the previous barrier
            __syncthreads();
            p0 += PAIRS;
            if (p0 + 2 * PAIRS < np) { SW_LOAD(p0 + 2 * PAIRS, xb, gb_); }
            SW_COMPUTE(1)
            if (p0 + PAIRS >= np) break;
            SW_STORE(0, p0 + PAIRS, xa, ga);
            __syncthreads();
            p0 += PAIRS;
        }
    }
#undef SW_COMPUTE
#undef SW_MM
#undef SW_MH
#undef SW_LOAD
#undef SW_SPLIT_STORE
#undef SW_STORE
#undef SW_FRAG
    // two waves per tile group: add the second K-step's accumulators through LDS (fixed order)
    if (KS == 2) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(pin);                // [group][16][64] floats, 8 KB
        if (wactive && ks_w == 1)
#pragma unroll
            for (int v = 0; v < 16; ++v) red[(grp_w * 16 + v) * 64 + lane] = acc[0][v];
        __syncthreads();
        if (wactive && ks_w == 0)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[0][v] += red[(grp_w * 16 + v) * 64 + lane];
    }
    if (!wactive || ks_w != 0) return;
    float* out = partials + (chunk * kvol + k) * (CI * CO);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int co = (j0 + t) * 32 + (lane & 31);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = i0 * 32 + (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
            out[ci * CO + co] = acc[t][v];
        }
    }
}

// dW[k][ci][co] = sum over the row chunks' partials [chunk][k][CI][CO], fixed order, f64
__global__ __launch_bounds__(256) void sp_wgrad_reduce_kernel(const float* __restrict__ partials, int nchunks, int kvol,
                                                             int cin, int cout, int CI, int CO, const uint32_t* __restrict__ amax_x,
                                                             const uint32_t* __restrict__ amax_g, float* __restrict__ dW) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (k, ci, co)
    if (i >= (int64_t)kvol * cin * cout) return;
    const int co = (int)(i % cout), ci = (int)((i / cout) % cin), k = (int)(i / ((int64_t)cin * cout));
    const float* p = partials + ((int64_t)k * CI + ci) * CO + co;
    double s = 0.0;
#pragma unroll 8
    for (int c = 0; c < nchunks; ++c) s += (double)p[(int64_t)c * kvol * CI * CO];
    if (amax_x) s = s * (double)h2_descale(h2_scale_exp(*amax_x)) * (double)h2_descale(h2_scale_exp(*amax_g));     // fp16-plane partials are scaled
    dW[i] = (float)s;
}

// Rows per chunk. Offset is the fastest grid dimension, so the kvol workgroups of one chunk run
// together and share its G rows and their X neighbours in the L2s: small chunks keep that working set
// (2 x rows x C x 4 B) inside them, at the price of one partial dW per chunk in the workspace.
static inline int spw_rows_per_chunk() { return SPW_SUB; }    // measured: 2048 / 4096 / 8192 rows -> 1.83 / 1.89 / 2.48 ms at 510 k x 128
static inline int spw_chunks(int64_t n_rows, int kvol) {
    (void)kvol;
    const int64_t rpc = spw_rows_per_chunk();
    const int64_t c = (n_rows + rpc - 1) / rpc;
    return (int)(c < 1 ? 1 : c);
}
static inline int spw_pad32(int c) { return (c + 31) / 32 * 32; }
static inline void spw_tiles(int cin, int cout, int& ni, int& nj) {
    ni = (cin + 31) / 32; nj = (cout + 31) / 32;
    // instantiated shapes: (1,1) (1,2) (2,2) (2,4) (4,4) and their transposes' covers
    if (ni == 3) ni = 4;
    if (nj == 3) nj = 4;
    if (ni == 2 && nj == 1) nj = 2;
    if (ni == 4 && nj < 4) nj = 4;
    if (ni == 1 && nj == 4) ni = 2;
}

extern "C" size_t gga_sparse_conv_wgrad_workspace_bytes(int64_t n_rows, int kvol, int cin, int cout) {
    if (n_rows < 1 || kvol < 1 || cin < 1 || cout < 1 || cin > 128 || cout > 128) return 0;
    int ni, nj;
    spw_tiles(cin, cout, ni, nj);
    return (size_t)spw_chunks(n_rows, kvol) * kvol * (ni * 32) * (nj * 32) * sizeof(float);
}

extern "C" int gga_sparse_conv_wgrad_split(const float* x, const float* grad_out, const int32_t* map, int64_t n_rows,
                                           int kvol, int cin, int cout, float* grad_weight, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    return gga_sparse_conv_wgrad_split_strided(x, cin, grad_out, cout, map, n_rows, kvol, cin, cout, grad_weight, workspace,
                                               workspace_bytes, stream_);
}

extern "C" int gga_sparse_conv_wgrad_split_strided(const float* x, int64_t x_row_stride, const float* grad_out,
                                                   int64_t grad_out_row_stride, const int32_t* map, int64_t n_rows, int kvol,
                                                   int cin, int cout, float* grad_weight, void* workspace,
                                                   size_t workspace_bytes, void* stream_) {
    return gga_sparse_conv_wgrad_planes(x, x_row_stride, grad_out, grad_out_row_stride, map, n_rows, kvol, cin, cout, grad_weight, 3,
                                        nullptr, nullptr, workspace, workspace_bytes, stream_);
}

extern "C" int gga_sparse_conv_wgrad_planes(const float* x, int64_t x_row_stride, const float* grad_out,
                                            int64_t grad_out_row_stride, const int32_t* map, int64_t n_rows, int kvol, int cin,
                                            int cout, float* grad_weight, int planes, const uint32_t* amax_x,
                                            const uint32_t* amax_grad_out, void* workspace, size_t workspace_bytes,
                                            void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(planes == 3 || (planes == 2 && amax_x && amax_grad_out),
                "gga_sparse_conv_wgrad_split: planes must be 3 (bf16) or 2 (fp16, with the operands' absmax bits)");
    GGA_REQUIRE(x && grad_out && map && grad_weight && workspace, "gga_sparse_conv_wgrad_split: null pointer argument");
    GGA_REQUIRE(n_rows >= 1 && kvol >= 1 && cin >= 1 && cin <= 128 && cout >= 1 && cout <= 128 && x_row_stride >= cin &&
                grad_out_row_stride >= cout, "gga_sparse_conv_wgrad_split: bad sizes (cin, cout <= 128; row strides >= widths)");
    if (workspace_bytes < gga_sparse_conv_wgrad_workspace_bytes(n_rows, kvol, cin, cout)) {
        gga_set_error("gga_sparse_conv_wgrad_split: workspace %zu B < required %zu B", workspace_bytes,
                      gga_sparse_conv_wgrad_workspace_bytes(n_rows, kvol, cin, cout));
        return GGA_ERR_WORKSPACE;
    }
    int ni, nj;
    spw_tiles(cin, cout, ni, nj);
    const int nchunks = spw_chunks(n_rows, kvol);
    const int64_t rpc = spw_rows_per_chunk();                   // whole sub-chunks
    static const int plain_grid = getenv("GGA_SP_WGRAD_PLAIN_GRID") ? atoi(getenv("GGA_SP_WGRAD_PLAIN_GRID")) : 0;     // A/B switch
    const dim3 grid = plain_grid ? dim3(kvol, (unsigned)nchunks) : dim3((unsigned)(8 * ((nchunks + 7) / 8) * kvol), 1), block(256);
    const bool vec = (cin & 3) == 0 && (cout & 3) == 0 && (x_row_stride & 3) == 0 && (grad_out_row_stride & 3) == 0 &&
                     ((uintptr_t)x & 15) == 0 && ((uintptr_t)grad_out & 15) == 0;
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_SPARSE_WGRAD, GGA_TIMING_CONV_KEY(cin, cout, 0));
    GGA_TIME_START(tev, stream);
#define SW_ARGS grid, block, 0, stream, x, grad_out, map, n_rows, kvol, rpc, cin, cout, x_row_stride, grad_out_row_stride, (float*)workspace, amax_x, amax_grad_out, plain_grid ? 0 : 1
#define SW(NI, NJ) { if (planes == 3) { if (vec) hipLaunchKernelGGL((sp_conv_wgrad_x9_kernel<NI, NJ, true, 3>), SW_ARGS); else hipLaunchKernelGGL((sp_conv_wgrad_x9_kernel<NI, NJ, false, 3>), SW_ARGS); } \
                     else { if (vec) hipLaunchKernelGGL((sp_conv_wgrad_x9_kernel<NI, NJ, true, 2>), SW_ARGS); else hipLaunchKernelGGL((sp_conv_wgrad_x9_kernel<NI, NJ, false, 2>), SW_ARGS); } }
    if (ni == 1 && nj == 1) SW(1, 1)
    else if (ni == 1 && nj == 2) SW(1, 2)
    else if (ni == 2 && nj == 2) SW(2, 2)
    else if (ni == 2 && nj == 4) SW(2, 4)
    else SW(4, 4)
#undef SW
#undef SW_ARGS
    GGA_CHECK_LAUNCH("sp_conv_wgrad_x9_kernel");
    const int64_t total = (int64_t)kvol * cin * cout;
    hipLaunchKernelGGL(sp_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const float*)workspace, nchunks, kvol, cin, cout, ni * 32, nj * 32, planes == 2 ? amax_x : nullptr,
                       amax_grad_out, grad_weight);
    GGA_CHECK_LAUNCH("sp_wgrad_reduce_kernel");
    GGA_TIME_STOP(tev, stream);
    return GGA_OK;
}
