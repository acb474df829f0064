// Y[out_row(p, tap)] = X[p] * W[tap]  over the rows of a channels-last image: the 1x1 convolutions (one tap, out_row = p) and the
// kernel = stride transposed convolutions (s * s taps, input pixel (b, i, j) -> output pixel (b, i s + a, j s + c) for tap
// a s + c) of SECONDFPN (necks/second_fpn.py:52-69), forward, and - with the transposed weight - the backward-data of the 1x1
// form. These ran on the gather-GEMM kernel (sp_conv_x9_kernel) through an arithmetic rule book: a map read and a row gather
// per (output row, tap) for what is a plain streaming matrix product. Here the rows are contiguous and nothing is gathered:
//   * a workgroup (grid.y = tap) copies the tap's whole packed weight (gga_sparse_pack_weight_planes layout: 32-channel stages,
//     two fp16 planes, quarters swizzled) into LDS ONCE and keeps it for all the row blocks it walks (persistent);
//   * a wave owns 32 rows at a time: lane (r, h) loads the channels 32 c + 16 k + 8 h .. + 7 of its row straight into the A
//     fragment's lane layout (as sp_conv_x9_kernel does), splits them into the two planes in registers, and multiplies with B
//     fragments read from the LDS image (v_mfma_f32_32x32x16_f16, three partial products); no barrier after the weight copy;
//   * the output rows are stored from the accumulator layout (a lane holds 16 rows of its column: two 128-byte row segments
//     per store instruction); the per-channel sums for the BatchNorm that follows are accumulated per wave over all its
//     blocks (f64) and written once per workgroup: [workgroups][2][cout] instead of one row per 128-row tile.
// HBM-bound by construction: 857 k rows x (64 in + 128 out) x 4 B = 657 MB for the 1x1 convolution of the PointPillars neck.
// Two-plane arithmetic only (three bf16 planes stay on the gather kernel).
#include "gga_common.h"
#include "conv_planes.h"

template <int NT, int NCH>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const float* __restrict__ X, int64_t xs, int64_t n_rows,
                                                       const uint16_t* __restrict__ Wp, int cout, float* __restrict__ Y,
                                                       int64_t ys, int H, int W, int s, const uint32_t* __restrict__ amax_x,
                                                       const uint32_t* __restrict__ amax_w, double* __restrict__ stats) {
    constexpr int CO = NT * 32;
    constexpr int STAGE = 2 * CO * 64;                     // bytes of one 32-channel stage: [plane][col][32 ch] f16
    extern __shared__ __attribute__((aligned(16))) unsigned char Bs[];      // [NCH][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tap = blockIdx.y;
    {   // the tap's packed weight: NCH consecutive stages of the operand
        const uint4* src = reinterpret_cast<const uint4*>(Wp + (int64_t)tap * NCH * (STAGE / 2));
        uint4* dst = reinterpret_cast<uint4*>(Bs);
        for (int i = tid; i < NCH * STAGE / 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int sbx = h2_scale_exp(*amax_x), sbw = h2_scale_exp(*amax_w);
    const float xscale = h2_scale(sbx);
    const float descale = h2_descale(sbx) * h2_descale(sbw);
    const int swz = (r >> 2) & 3;
    const int ta = s ? tap / s : 0, tc = s ? tap - ta * s : 0;
    double st1[NT], st2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { st1[t] = 0.0; st2[t] = 0.0; }
    const int64_t n_blocks = (n_rows + 31) / 32;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blocks; blk += (int64_t)gridDim.x * 4) {
        const int64_t row = blk * 32 + r;
        const float* xp = X + (row < n_rows ? row : n_rows - 1) * xs + 8 * h;
        float4 v[NCH][4];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                v[c][2 * k] = *reinterpret_cast<const float4*>(xp + c * 32 + k * 16);
                v[c][2 * k + 1] = *reinterpret_cast<const float4*>(xp + c * 32 + k * 16 + 4);
            }
        mf_v16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                union { uint32_t w[4]; mf_v8h f; } a0, a1;
                const float4 lo = v[c][2 * k], hi = v[c][2 * k + 1];
                h2_split2(lo.x * xscale, lo.y * xscale, a0.w[0], a1.w[0]);
                h2_split2(lo.z * xscale, lo.w * xscale, a0.w[1], a1.w[1]);
                h2_split2(hi.x * xscale, hi.y * xscale, a0.w[2], a1.w[2]);
                h2_split2(hi.z * xscale, hi.w * xscale, a0.w[3], a1.w[3]);
                const unsigned char* Bp = Bs + c * STAGE + r * 64 + (((k * 2 + h) ^ swz) * 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const mf_v8h b0 = *reinterpret_cast<const mf_v8h*>(Bp + t * 32 * 64);
                    const mf_v8h b1 = *reinterpret_cast<const mf_v8h*>(Bp + CO * 64 + t * 32 * 64);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.f, b1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.f, b0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.f, b0, acc[t], 0, 0, 0);
                }
            }
        // rows of this lane's accumulator registers: (v / 4) * 8 + h * 4 + v % 4 of the block
        float s1[NT], s2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { s1[t] = 0.0f; s2[t] = 0.0f; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t p = blk * 32 + (i >> 2) * 8 + h * 4 + (i & 3);
            if (p >= n_rows) continue;
            int64_t o = p;
            if (s) {                                           // transposed convolution: input pixel -> its output pixel for this tap
                const int64_t hw = (int64_t)H * W;
                const int64_t b = p / hw;
                const int rem = (int)(p - b * hw);
                const int yi = rem / W, xj = rem - yi * W;
                o = ((b * H + yi) * s + ta) * ((int64_t)W * s) + (int64_t)xj * s + tc;
            }
            float* yp = Y + o * ys;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 32 + r;
                const float val = acc[t][i] * descale;
                if (col < cout) yp[col] = val;
                s1[t] += val; s2[t] += val * val;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) { st1[t] += (double)s1[t]; st2[t] += (double)s2[t]; }
    }
    if (stats) {      // [workgroup = blockIdx.y * gridDim.x + blockIdx.x][2][cout]
        __syncthreads();                                       // (the LDS image is done with: reuse it for the reduction)
        double* red = reinterpret_cast<double*>(Bs);           // [4 waves][2][CO]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double a = st1[t] + __shfl_xor(st1[t], 32), b = st2[t] + __shfl_xor(st2[t], 32);
            if (h == 0) { red[(wave * 2 + 0) * CO + t * 32 + r] = a; red[(wave * 2 + 1) * CO + t * 32 + r] = b; }
        }
        __syncthreads();
        if (tid < 2 * CO) {
            const int which = tid / CO, c = tid - which * CO;
            if (c < cout) {
                const double a = red[(0 * 2 + which) * CO + c] + red[(1 * 2 + which) * CO + c] + red[(2 * 2 + which) * CO + c] + red[(3 * 2 + which) * CO + c];
                stats[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + which) * cout + c] = a;
            }
        }
    }
}

// workgroups per tap: as many as keep the chip full at the occupancy the weight image allows
static int rows_gemm_groups(int64_t n_rows, int cin, int cout) {
    const int64_t lds = (int64_t)(cin / 32) * 2 * (cout <= 32 ? 32 : (cout <= 64 ? 64 : 128)) * 64;
    int per_cu = (int)(160 * 1024 / (lds > 0 ? lds : 1));
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t g = 256 * per_cu;
    const int64_t need = (n_rows + 127) / 128;
    return (int)(g < need ? g : (need < 1 ? 1 : need));
}

extern "C" int64_t gga_rows_gemm_workgroups(int64_t n_rows, int cin, int cout, int taps) {
    return (int64_t)rows_gemm_groups(n_rows, cin, cout) * (taps < 1 ? 1 : taps);
}

extern "C" int gga_rows_gemm(const float* x, int64_t x_row_stride, int64_t n_rows, int cin, const void* split_weight, int cout,
                             float* y, int64_t y_row_stride, int taps, int in_h, int in_w, int stride,
                             const uint32_t* amax_x, const uint32_t* amax_weight, double* stats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && split_weight && y && amax_x && amax_weight, "gga_rows_gemm: null pointer argument (two-plane arithmetic: absmax slots needed)");
    GGA_REQUIRE(n_rows >= 1 && cout >= 1 && cout <= 128 && (cin == 64 || cin == 128 || cin == 256) && x_row_stride >= cin &&
                    (x_row_stride & 3) == 0 && ((uintptr_t)x & 15) == 0 && y_row_stride >= cout,
                "gga_rows_gemm: cin must be 64 / 128 / 256, cout <= 128, rows 16-byte aligned (got %d -> %d)", cin, cout);
    GGA_REQUIRE((stride == 0 && taps == 1) || (stride >= 2 && taps == stride * stride && in_h >= 1 && in_w >= 1 &&
                                               n_rows % ((int64_t)in_h * in_w) == 0),
                "gga_rows_gemm: taps / stride / image size do not describe a 1x1 or a kernel = stride transposed convolution");
    const int groups = rows_gemm_groups(n_rows, cin, cout);
    const dim3 grid(groups, taps), block(256);
    const int nt = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4), nch = cin / 32;
    const size_t lds = (size_t)nch * 2 * nt * 32 * 64;
#define RG_GO(NT_, NCH_) { \
        static bool once = false; \
        if (!once) { GGA_CHECK_HIP(hipFuncSetAttribute((const void*)rows_gemm_kernel<NT_, NCH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "rows_gemm: LDS size"); once = true; } \
        hipLaunchKernelGGL((rows_gemm_kernel<NT_, NCH_>), grid, block, lds, stream, x, x_row_stride, n_rows, (const uint16_t*)split_weight, cout, y, y_row_stride, in_h, in_w, stride, amax_x, amax_weight, stats); }
#define RG_NT(NCH_) { if (nt == 1) RG_GO(1, NCH_) else if (nt == 2) RG_GO(2, NCH_) else RG_GO(4, NCH_) }
    if (nch == 2) RG_NT(2) else if (nch == 4) RG_NT(4) else RG_NT(8)
#undef RG_NT
#undef RG_GO
    GGA_CHECK_LAUNCH("rows_gemm_kernel");
    return GGA_OK;
}
