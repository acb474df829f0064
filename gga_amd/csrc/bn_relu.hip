// Fused BatchNorm (+ residual add) (+ ReLU) over a [rows, C] row-major tensor, forward and backward
// (gfx950). [rows, C] is both the sparse feature matrix of SparseEncoder ([N, C]) and the memory of
// a channels-last [B, C, H, W] activation of the BEV trunk ([B*H*W, C]).
//
// Reference: the BN + ReLU pairs of mmdet3d/models/backbones/second.py:46-63, necks/
// second_fpn.py:66-69, dense_heads/centerpoint_head.py (ConvModule), ops/sparse_block.py:117-134
// (norm1+relu, norm2 + identity + relu) and sparse_block.py:186-196 run as separate cuDNN / ATen
// kernels: stats, normalise, ReLU, (add), and in backward ReLU-grad, reduce, elementwise — 13
// passes over the activation per layer. Here:
//   fwd  pass 1  per-channel sum / sum of squares (f32 per thread over a short run, f64 across)
//        pass 2  y = relu(x*scale + shift (+ residual)); the ReLU sign is kept as 1 bit/element
//   bwd  pass 1  sum g, sum g*xhat with g = dy * sign bit           (reads dy, x, bits)
//        pass 2  dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)), d_residual = g
// 8 passes, all 16 B/lane coalesced, fixed-order (deterministic) reductions.
// A thread always owns the same 4 channels: its float4 index advances by a multiple of C/4.
#include "gga_common.h"

#define BN_MAX_BLOCKS 2048
#define BN_U 2            // independent 16 B loads in flight per thread and tensor (round 5: 2 instead of 4 - twice the workgroups; sparse config -0.2 ms per step, PointPillars +-0, profiles/r05_ab_bn_geometry.txt)

struct BnGeom {
    int64_t n4;      // rows * C / 4
    int c4;          // C / 4 (a power of two, see bn_check)
    int sh;          // log2(c4)
    int64_t ys4;     // row stride (in float4) of the strided operand: y in forward, grad_y in backward; c4 = dense
};

// float4 index of element e (dense index over [rows, c4]) in the strided operand
__device__ __forceinline__ int64_t bn_strided(const BnGeom& g, int64_t e, int cg) { return (e >> g.sh) * g.ys4 + cg; }

static BnGeom bn_geom(int64_t rows, int channels, int64_t row_stride) {
    BnGeom g;
    g.n4 = rows * channels / 4;
    g.c4 = channels / 4;
    g.sh = 0;
    while ((1 << g.sh) < g.c4) ++g.sh;
    g.ys4 = row_stride / 4;
    return g;
}

static int bn_grid(int64_t n4) {
    int64_t b = (n4 + 256 * BN_U - 1) / (256 * BN_U);
    if (b < 1) b = 1;
    return (int)(b > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : b);
}

// partials layout: [block][stat][C]  (stat 0/1)
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       const unsigned long long* __restrict__ bits,
                                                       const float* __restrict__ saved, BnGeom g, int relu,
                                                       double* __restrict__ partials) {
    // forward : a = x                -> sums of x and x^2
    // backward: a = dy, b = x, bits  -> sums of g and g*xhat
    const int tid = threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t e0 = (int64_t)blockIdx.x * 256 + tid;
    const int cg = (int)(e0 % g.c4);                  // fixed channel group of this thread
    float mean[4] = {0, 0, 0, 0}, inv[4] = {1, 1, 1, 1};
    float msc[4] = {0, 0, 0, 0}, msh[4] = {0, 0, 0, 0};      // relu == 2: the ReLU mask is recomputed from x
    if (BWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mean[j] = saved[cg * 4 + j]; inv[j] = saved[g.c4 * 4 + cg * 4 + j]; }
        if (relu == 2) {
            const float* ss = reinterpret_cast<const float*>(bits);
#pragma unroll
            for (int j = 0; j < 4; ++j) { msc[j] = ss[cg * 4 + j]; msh[j] = ss[g.c4 * 4 + cg * 4 + j]; }
        }
    }
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float f0[4] = {0, 0, 0, 0}, f1[4] = {0, 0, 0, 0};
    int run = 0;
    for (int64_t eb = e0; eb < g.n4; eb += stride * BN_U) {
        float4 av[BN_U], bv[BN_U];
        unsigned long long wv[BN_U][4];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {                 // issue every load of the batch first
            const int64_t e = eb + u * stride;
            av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bv[u] = av[u];
            if (e < g.n4) {
                av[u] = BWD ? a[bn_strided(g, e, cg)] : a[e];
                if (BWD) {
                    bv[u] = b[e];
                    if (relu == 1) {
                        const unsigned long long* w = bits + (e >> 6) * 4;   // [e / 64][component]
#pragma unroll
                        for (int j = 0; j < 4; ++j) wv[u][j] = w[j];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int64_t e = eb + u * stride;
            if (e >= g.n4) continue;
            float va[4] = { av[u].x, av[u].y, av[u].z, av[u].w };
            if (BWD) {
                const float xa[4] = { bv[u].x, bv[u].y, bv[u].z, bv[u].w };
                if (relu == 1) {
                    const int l = (int)(e & 63);
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (!((wv[u][j] >> l) & 1ull)) va[j] = 0.0f;
                } else if (relu == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (!(fmaf(xa[j], msc[j], msh[j]) > 0.0f)) va[j] = 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { f0[j] += va[j]; f1[j] += va[j] * ((xa[j] - mean[j]) * inv[j]); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { f0[j] += va[j]; f1[j] += va[j] * va[j]; }
            }
        }
        if (++run == 8) {                               // flush the short f32 run (32 rows) into f64
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] += f0[j]; s1[j] += f1[j]; f0[j] = 0; f1[j] = 0; }
            run = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s0[j] += f0[j]; s1[j] += f1[j]; }
    __shared__ double sh[256][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[tid][j] = s0[j]; sh[tid][4 + j] = s1[j]; }
    __syncthreads();
    // threads with the same channel group are tid = cg0 + k*c4 (256 % c4 == 0 when c4 <= 256)
    const int cg0 = (int)(((int64_t)blockIdx.x * 256) % g.c4);
    const int per = g.c4 <= 256 ? g.c4 : 256;
    if (tid < per) {
        double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = tid; t < 256; t += per)
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] += sh[t][q];
        const int grp = (cg0 + tid) % g.c4;
        double* out = partials + (int64_t)blockIdx.x * 2 * g.c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[grp * 4 + j] = r[j]; out[g.c4 * 4 + grp * 4 + j] = r[4 + j]; }
    }
}

// Sum the two statistics of 8 neighbouring channels over all block partials. Thread t of the
// 1024-thread block reads channel (t & 7) of partial rows t>>3, t>>3 + 128, ...: 8 lanes share one
// 64 B line and all loads of a thread are independent and issued together, so the [nblocks][2][C]
// array is read once with one memory round trip (one wavefront per channel with 1 KB-strided lanes
// fetched it 8 times over in 32 dependent trips: 16 us per layer, 70 layers a step). Row groups
// fold with three shuffles inside a wavefront and in fixed order across the 16 wavefronts; lanes
// 0..7 return the totals.
// (Cw, off: the rows are [2][Cw] wide and this BatchNorm's channels are columns off .. off + C of them - the statistics of a
// convolution launch that computed several BatchNorms' inputs side by side; Cw = C, off = 0 otherwise)
__device__ __forceinline__ void bn_fold_partials(const double* __restrict__ partials, int nblocks, int C, int c,
                                                 bool ok, double& s0, double& s1, int Cw = 0, int off = 0) {
    if (Cw == 0) Cw = C;
    __shared__ double red[2][16][8];
    const int t = threadIdx.x, rg = t >> 3;
    double a0 = 0.0, a1 = 0.0;
    if (ok) {
#pragma unroll 16
        for (int b = rg; b < nblocks; b += 128) {
            a0 += partials[(int64_t)b * 2 * Cw + off + c];
            a1 += partials[(int64_t)b * 2 * Cw + Cw + off + c];
        }
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
        a0 += __shfl_xor(a0, m);
        a1 += __shfl_xor(a1, m);
    }
    if ((t & 63) < 8) { red[0][t >> 6][t & 7] = a0; red[1][t >> 6][t & 7] = a1; }
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (t < 8)
        for (int w = 0; w < 16; ++w) { s0 += red[0][w][t]; s1 += red[1][w][t]; }
}

// one 1024-thread block per 8 channels
__global__ __launch_bounds__(1024) void bn_fwd_final_kernel(const double* __restrict__ partials, int nblocks, int C,
                                                           double rows, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           int training, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ saved,
                                                           float* __restrict__ scale_shift, int Cw = 0, int off = 0) {
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    const bool ok = c < C;
    float mean, invstd;
    if (training) {
        double s, ss;
        bn_fold_partials(partials, nblocks, C, c, ok, s, ss, Cw, off);
        if (threadIdx.x >= 8 || !ok) return;
        const double mu = s / rows;
        double var = ss / rows - mu * mu;
        var = var > 0.0 ? var : 0.0;
        mean = (float)mu;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        const double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    } else {
        if (threadIdx.x >= 8 || !ok) return;
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    saved[c] = mean;
    saved[C + c] = invstd;
    float sc, sh;
    gga_bn_scale_shift(gamma ? gamma[c] : 1.0f, beta ? beta[c] : 0.0f, mean, invstd, sc, sh);
    scale_shift[c] = sc;
    scale_shift[C + c] = sh;
}

// Last-use streaming reads: the conv output x and the incoming gradient are not read again in this pass, so a non-temporal load
// leaves the L2 / Infinity Cache lines to the tensor being written, which the next kernel reads. Round 5, same-box A/B
// (tools_dev/ab_lib.sh, profiles/r05_ab_bn_nt.txt): PointPillars step -0.1 ms, sparse config -0.3..0.4 ms. -DBN_NO_NT_LOADS: plain loads.
typedef float bn_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bn_ld_stream(const float4* p) {
#ifndef BN_NO_NT_LOADS
    const bn_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bn_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ res,
                                                      const float* __restrict__ scale_shift, BnGeom g, int relu,
                                                      float4* __restrict__ y, unsigned long long* __restrict__ bits,
                                                      uint32_t* __restrict__ amax) {
    uint32_t am = 0;                                     // largest finite |y| written (for the consumer's fp16 scale)
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(e0 % g.c4);
    const int C = g.c4 * 4;
    const float4 sc = *reinterpret_cast<const float4*>(scale_shift + cg * 4);
    const float4 sh = *reinterpret_cast<const float4*>(scale_shift + C + cg * 4);
    const int64_t n4_round = (g.n4 + 63) & ~63ll;       // whole waves iterate together (ballot)
    for (int64_t eb = e0; eb < n4_round; eb += stride * BN_U) {
        float4 xv[BN_U], rv[BN_U];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int64_t e = eb + u * stride;
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            rv[u] = xv[u];
            if (e < g.n4) { xv[u] = bn_ld_stream(x + e); if (res) rv[u] = res[e]; }
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int64_t e = eb + u * stride;
            if (e >= n4_round) continue;                 // wave-uniform: e and n4_round are multiples of 64 apart
            const bool live = e < g.n4;
            float4 v;
            v.x = xv[u].x * sc.x + sh.x + rv[u].x; v.y = xv[u].y * sc.y + sh.y + rv[u].y;
            v.z = xv[u].z * sc.z + sh.z + rv[u].z; v.w = xv[u].w * sc.w + sh.w + rv[u].w;
            if (relu) {
                const unsigned long long b0 = __ballot(live && v.x > 0.0f), b1 = __ballot(live && v.y > 0.0f);
                const unsigned long long b2 = __ballot(live && v.z > 0.0f), b3 = __ballot(live && v.w > 0.0f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* w = bits + (e >> 6) * 4;
                    w[0] = b0; w[1] = b1; w[2] = b2; w[3] = b3;
                }
                v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            }
            if (live) {
                y[bn_strided(g, e, cg)] = v;
                if (amax) { am = gga_amax_of(v.x, am); am = gga_amax_of(v.y, am); am = gga_amax_of(v.z, am); am = gga_amax_of(v.w, am); }
            }
        }
    }
    if (amax) gga_amax_commit(am, amax);
}

__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const double* __restrict__ partials, int nblocks, int C,
                                                           double rows, const float* __restrict__ gamma,
                                                           const float* __restrict__ saved, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ coef, int training) {
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    const bool ok = c < C;
    double s, sx;
    bn_fold_partials(partials, nblocks, C, c, ok, s, sx);
    if (threadIdx.x >= 8 || !ok) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)sx;
    const float k = (gamma ? gamma[c] : 1.0f) * saved[C + c];
    coef[c] = k;                              // gamma * invstd
    // evaluation-mode statistics (running mean / variance, e.g. a frozen backbone) do not depend on x: dx = gamma*invstd*g
    coef[C + c] = training ? (float)(s / rows) : 0.0f;          // mean(g)
    coef[2 * C + c] = training ? (float)(sx / rows) : 0.0f;     // mean(g * xhat)
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                          const unsigned long long* __restrict__ bits,
                                                          const float* __restrict__ saved, const float* __restrict__ coef,
                                                          BnGeom g, int relu, float4* __restrict__ dx,
                                                          float4* __restrict__ dres, uint32_t* __restrict__ amax) {
    uint32_t am = 0;                                     // largest finite |dx| written
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(e0 % g.c4);
    const int C = g.c4 * 4;
    float mean[4], inv[4], k[4], mg[4], mgx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = saved[cg * 4 + j]; inv[j] = saved[C + cg * 4 + j];
        k[j] = coef[cg * 4 + j]; mg[j] = coef[C + cg * 4 + j]; mgx[j] = coef[2 * C + cg * 4 + j];
    }
    float msc[4] = {0, 0, 0, 0}, msh[4] = {0, 0, 0, 0};
    if (relu == 2) {
        const float* ss = reinterpret_cast<const float*>(bits);
#pragma unroll
        for (int j = 0; j < 4; ++j) { msc[j] = ss[cg * 4 + j]; msh[j] = ss[C + cg * 4 + j]; }
    }
    for (int64_t eb = e0; eb < g.n4; eb += stride * BN_U) {
        float4 gv[BN_U], xv[BN_U];
        unsigned long long wv[BN_U][4];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int64_t e = eb + u * stride;
            if (e < g.n4) {
                gv[u] = bn_ld_stream(dy + bn_strided(g, e, cg)); xv[u] = bn_ld_stream(x + e);
                if (relu == 1) {
                    const unsigned long long* w = bits + (e >> 6) * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) wv[u][j] = w[j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int64_t e = eb + u * stride;
            if (e >= g.n4) continue;
            float ga[4] = { gv[u].x, gv[u].y, gv[u].z, gv[u].w };
            const float xa[4] = { xv[u].x, xv[u].y, xv[u].z, xv[u].w };
            if (relu == 1) {
                const int l = (int)(e & 63);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!((wv[u][j] >> l) & 1ull)) ga[j] = 0.0f;
            } else if (relu == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!(fmaf(xa[j], msc[j], msh[j]) > 0.0f)) ga[j] = 0.0f;
            }
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = k[j] * (ga[j] - mg[j] - ((xa[j] - mean[j]) * inv[j]) * mgx[j]);
            dx[e] = make_float4(o[0], o[1], o[2], o[3]);
            if (amax) {
#pragma unroll
                for (int j = 0; j < 4; ++j) am = gga_amax_of(o[j], am);
            }
            if (dres) dres[e] = make_float4(ga[0], ga[1], ga[2], ga[3]);
        }
    }
    if (amax) gga_amax_commit(am, amax);
}

extern "C" size_t gga_bn_relu_workspace_bytes(int64_t rows, int channels) {
    (void)rows;
    return (size_t)BN_MAX_BLOCKS * 2 * channels * sizeof(double) + 3 * (size_t)channels * sizeof(float) + 256;
}
extern "C" size_t gga_bn_relu_mask_bytes(int64_t rows, int channels) {
    const int64_t n4 = rows * channels / 4;
    return (size_t)((n4 + 63) / 64) * 4 * sizeof(unsigned long long);
}

static int bn_check(const char* fn, int64_t rows, int C) {
    GGA_REQUIRE(rows >= 1 && C >= 4 && C % 4 == 0, "%s: need rows >= 1 and channels %% 4 == 0 (rows=%lld C=%d)", fn,
                (long long)rows, C);
    const int c4 = C / 4;
    GGA_REQUIRE(c4 <= 256 && 256 % c4 == 0, "%s: channels/4 (%d) must divide 256", fn, c4);
    return GGA_OK;
}

static int bn_fwd_impl(const float* x, const float* residual, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                       float momentum, int training, int relu, float* y, int64_t y_row_stride,
                       void* mask_bits, float* saved, const double* given_partials, int n_given, void* workspace,
                       size_t workspace_bytes, void* stream_, uint32_t* amax_y = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = bn_check("gga_bn_relu_fwd", rows, channels)) return rc;
    GGA_REQUIRE(y_row_stride >= channels && y_row_stride % 4 == 0 && ((uintptr_t)y & 15) == 0,
                "gga_bn_relu_fwd: y row stride %lld must be a multiple of 4 floats >= channels, y 16 B aligned",
                (long long)y_row_stride);
    GGA_REQUIRE(x && y && saved && workspace && running_mean && running_var && (!relu || mask_bits),
                "gga_bn_relu_fwd: null pointer argument");
    if (workspace_bytes < gga_bn_relu_workspace_bytes(rows, channels)) {
        gga_set_error("gga_bn_relu_fwd: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    const BnGeom g = bn_geom(rows, channels, y_row_stride);
    const int nb = bn_grid(g.n4);
    const double* partials = given_partials ? given_partials : (const double*)workspace;
    float* scale_shift = (float*)((char*)workspace + (size_t)BN_MAX_BLOCKS * 2 * channels * sizeof(double));
    if (training && !given_partials) {
        hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(nb), dim3(256), 0, stream, (const float4*)x, (const float4*)nullptr,
                           (const unsigned long long*)nullptr, (const float*)nullptr, g, 0, (double*)workspace);
        GGA_CHECK_LAUNCH("bn_reduce_kernel<fwd>");
    }
    hipLaunchKernelGGL(bn_fwd_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, partials,
                       given_partials ? n_given : nb, channels,
                       (double)rows, gamma, beta, eps, momentum, training, running_mean, running_var, saved,
                       scale_shift);
    GGA_CHECK_LAUNCH("bn_fwd_final_kernel");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(nb), dim3(256), 0, stream, (const float4*)x, (const float4*)residual,
                       scale_shift, g, relu, (float4*)y, (unsigned long long*)mask_bits, amax_y);
    GGA_CHECK_LAUNCH("bn_apply_kernel");
    return GGA_OK;
}

extern "C" int gga_bn_relu_fwd_ex(const float* x, const float* residual, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                                  float momentum, int training, int relu, float* y, int64_t y_row_stride, void* mask_bits,
                                  float* saved, const double* partials, int n_partials, uint32_t* amax_y, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    return bn_fwd_impl(x, residual, gamma, beta, running_mean, running_var, rows, channels, eps, momentum,
                       partials ? 1 : training, relu, y, y_row_stride, mask_bits, saved, partials, n_partials, workspace,
                       workspace_bytes, stream_, amax_y);
}

extern "C" int gga_bn_stats_partials_cols(const float* gamma, const float* beta, float* running_mean, float* running_var,
                                          int64_t rows, int channels, float eps, float momentum, float* saved,
                                          float* scale_shift, const double* partials, int n_partials, int partials_width,
                                          int column_offset, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = bn_check("gga_bn_stats_partials", rows, channels)) return rc;
    GGA_REQUIRE(saved && scale_shift && running_mean && running_var && partials && n_partials >= 1,
                "gga_bn_stats_partials: null pointer argument");
    GGA_REQUIRE(column_offset >= 0 && partials_width >= column_offset + channels,
                "gga_bn_stats_partials_cols: columns %d .. %d of rows %d wide", column_offset, column_offset + channels, partials_width);
    hipLaunchKernelGGL(bn_fwd_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, partials, n_partials, channels,
                       (double)rows, gamma, beta, eps, momentum, 1, running_mean, running_var, saved, scale_shift, partials_width,
                       column_offset);
    GGA_CHECK_LAUNCH("bn_fwd_final_kernel");
    return GGA_OK;
}

extern "C" int gga_bn_stats_partials(const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int64_t rows, int channels, float eps, float momentum, float* saved,
                                     float* scale_shift, const double* partials, int n_partials, void* stream_) {
    return gga_bn_stats_partials_cols(gamma, beta, running_mean, running_var, rows, channels, eps, momentum, saved, scale_shift,
                                      partials, n_partials, channels, 0, stream_);
}

extern "C" int gga_bn_stats(const float* x, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, int64_t rows, int channels, float eps, float momentum, int training,
                            float* saved, float* scale_shift, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = bn_check("gga_bn_stats", rows, channels)) return rc;
    GGA_REQUIRE(x && saved && scale_shift && workspace && running_mean && running_var, "gga_bn_stats: null pointer argument");
    if (workspace_bytes < gga_bn_relu_workspace_bytes(rows, channels)) {
        gga_set_error("gga_bn_stats: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    const BnGeom g = bn_geom(rows, channels, channels);
    const int nb = bn_grid(g.n4);
    double* partials = (double*)workspace;
    if (training) {
        hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(nb), dim3(256), 0, stream, (const float4*)x, (const float4*)nullptr,
                           (const unsigned long long*)nullptr, (const float*)nullptr, g, 0, partials);
        GGA_CHECK_LAUNCH("bn_reduce_kernel<fwd>");
    }
    hipLaunchKernelGGL(bn_fwd_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, partials, nb, channels,
                       (double)rows, gamma, beta, eps, momentum, training, running_mean, running_var, saved,
                       scale_shift);
    GGA_CHECK_LAUNCH("bn_fwd_final_kernel");
    return GGA_OK;
}

extern "C" int gga_bn_relu_fwd_strided(const float* x, const float* residual, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                                       float momentum, int training, int relu, float* y, int64_t y_row_stride,
                                       void* mask_bits, float* saved, void* workspace, size_t workspace_bytes,
                                       void* stream_) {
    return bn_fwd_impl(x, residual, gamma, beta, running_mean, running_var, rows, channels, eps, momentum, training, relu,
                       y, y_row_stride, mask_bits, saved, nullptr, 0, workspace, workspace_bytes, stream_);
}

extern "C" int gga_bn_relu_fwd_partials(const float* x, const float* residual, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                                        float momentum, int relu, float* y, int64_t y_row_stride, void* mask_bits,
                                        float* saved, const double* partials, int n_partials, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
    GGA_REQUIRE(partials && n_partials >= 1, "gga_bn_relu_fwd_partials: no partial sums given");
    return bn_fwd_impl(x, residual, gamma, beta, running_mean, running_var, rows, channels, eps, momentum, 1, relu, y,
                       y_row_stride, mask_bits, saved, partials, n_partials, workspace, workspace_bytes, stream_);
}

extern "C" int gga_bn_relu_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                               float momentum, int training, int relu, float* y, void* mask_bits, float* saved,
                               void* workspace, size_t workspace_bytes, void* stream_) {
    return gga_bn_relu_fwd_strided(x, residual, gamma, beta, running_mean, running_var, rows, channels, eps, momentum,
                                   training, relu, y, channels, mask_bits, saved, workspace, workspace_bytes, stream_);
}

// Library-internal: fold [nblocks][2][C] partial sums (at the head of `workspace`) into grad_gamma / grad_beta and
// the coefficient table the apply passes read (returned in *coef, inside the workspace).
int gga_bn_bwd_finalize(const double* partials, int nblocks, int channels, int64_t rows, const float* gamma,
                        const float* saved, float* grad_gamma, float* grad_beta, void* workspace, float** coef,
                        hipStream_t stream) {
    *coef = (float*)((char*)workspace + (size_t)BN_MAX_BLOCKS * 2 * channels * sizeof(double));
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, partials, nblocks, channels,
                       (double)rows, gamma, saved, grad_gamma, grad_beta, *coef, 1);
    GGA_CHECK_LAUNCH("bn_bwd_final_kernel");
    return GGA_OK;
}

extern "C" int gga_bn_relu_bwd_strided(const float* grad_y, int64_t grad_y_row_stride, const float* x,
                                       const void* mask_bits, const float* gamma, const float* saved, int64_t rows,
                                       int channels, int relu, float* grad_x, float* grad_residual, float* grad_gamma,
                                       float* grad_beta, void* workspace, size_t workspace_bytes, void* stream_) {
    return gga_bn_relu_bwd_ex(grad_y, grad_y_row_stride, x, mask_bits, gamma, saved, rows, channels, relu, 1, grad_x, grad_residual,
                              grad_gamma, grad_beta, nullptr, workspace, workspace_bytes, stream_);
}

extern "C" int gga_bn_relu_bwd_ex(const float* grad_y, int64_t grad_y_row_stride, const float* x, const void* mask_bits,
                                  const float* gamma, const float* saved, int64_t rows, int channels, int relu, int training,
                                  float* grad_x,
                                  float* grad_residual, float* grad_gamma, float* grad_beta, uint32_t* amax_grad_x,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = bn_check("gga_bn_relu_bwd", rows, channels)) return rc;
    GGA_REQUIRE(grad_y_row_stride >= channels && grad_y_row_stride % 4 == 0 && ((uintptr_t)grad_y & 15) == 0,
                "gga_bn_relu_bwd: grad_y row stride %lld must be a multiple of 4 floats >= channels, grad_y 16 B aligned",
                (long long)grad_y_row_stride);
    GGA_REQUIRE(grad_y && x && saved && grad_x && workspace && (!relu || mask_bits),
                "gga_bn_relu_bwd: null pointer argument");
    if (workspace_bytes < gga_bn_relu_workspace_bytes(rows, channels)) {
        gga_set_error("gga_bn_relu_bwd: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    const BnGeom g = bn_geom(rows, channels, grad_y_row_stride);
    const int nb = bn_grid(g.n4);
    double* partials = (double*)workspace;
    float* coef = (float*)((char*)workspace + (size_t)BN_MAX_BLOCKS * 2 * channels * sizeof(double));
    hipLaunchKernelGGL(bn_reduce_kernel<true>, dim3(nb), dim3(256), 0, stream, (const float4*)grad_y, (const float4*)x,
                       (const unsigned long long*)mask_bits, saved, g, relu, partials);
    GGA_CHECK_LAUNCH("bn_reduce_kernel<bwd>");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, partials, nb, channels,
                       (double)rows, gamma, saved, grad_gamma, grad_beta, coef, training);
    GGA_CHECK_LAUNCH("bn_bwd_final_kernel");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb), dim3(256), 0, stream, (const float4*)grad_y, (const float4*)x,
                       (const unsigned long long*)mask_bits, saved, coef, g, relu, (float4*)grad_x,
                       (float4*)grad_residual, amax_grad_x);
    GGA_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return GGA_OK;
}

// per-channel sums of a [rows, C] matrix (the bias gradient of a convolution: sum of the output gradient over batch and
// pixels; torch's reduction over the non-channel dimensions of a channels-last tensor reads at 1.6 TB/s)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const double* __restrict__ partials, int nblocks, int C,
                                                           float* __restrict__ out) {
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    const bool ok = c < C;
    double s, ss;
    bn_fold_partials(partials, nblocks, C, c, ok, s, ss);
    if (threadIdx.x >= 8 || !ok) return;
    out[c] = (float)s;
}

extern "C" int gga_column_sums(const float* x, int64_t rows, int channels, float* sums, void* workspace, size_t workspace_bytes,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = bn_check("gga_column_sums", rows, channels)) return rc;
    GGA_REQUIRE(x && sums && workspace && ((uintptr_t)x & 15) == 0, "gga_column_sums: null or misaligned pointer argument");
    if (workspace_bytes < gga_bn_relu_workspace_bytes(rows, channels)) {
        gga_set_error("gga_column_sums: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    const BnGeom g = bn_geom(rows, channels, channels);
    const int nb = bn_grid(g.n4);
    hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(nb), dim3(256), 0, stream, (const float4*)x, (const float4*)nullptr,
                       (const unsigned long long*)nullptr, (const float*)nullptr, g, 0, (double*)workspace);
    GGA_CHECK_LAUNCH("bn_reduce_kernel<fwd>");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, (const double*)workspace, nb, channels,
                       sums);
    GGA_CHECK_LAUNCH("colsum_final_kernel");
    return GGA_OK;
}

extern "C" int gga_bn_relu_bwd_partials(const float* grad_masked, int64_t grad_row_stride, const float* x, const float* gamma,
                                        const float* saved, int64_t rows, int channels, int training, const double* partials,
                                        int n_partials, float* grad_x, float* grad_gamma, float* grad_beta,
                                        uint32_t* amax_grad_x, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = bn_check("gga_bn_relu_bwd_partials", rows, channels)) return rc;
    GGA_REQUIRE(grad_row_stride >= channels && grad_row_stride % 4 == 0 && ((uintptr_t)grad_masked & 15) == 0,
                "gga_bn_relu_bwd_partials: grad row stride %lld must be a multiple of 4 floats >= channels, grad 16 B aligned",
                (long long)grad_row_stride);
    GGA_REQUIRE(grad_masked && x && saved && grad_x && workspace && partials && n_partials >= 1,
                "gga_bn_relu_bwd_partials: null pointer argument");
    if (workspace_bytes < gga_bn_relu_workspace_bytes(rows, channels)) {
        gga_set_error("gga_bn_relu_bwd_partials: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    const BnGeom g = bn_geom(rows, channels, grad_row_stride);
    const int nb = bn_grid(g.n4);
    float* coef = (float*)((char*)workspace + (size_t)BN_MAX_BLOCKS * 2 * channels * sizeof(double));
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((channels + 7) / 8), dim3(1024), 0, stream, partials, n_partials, channels,
                       (double)rows, gamma, saved, grad_gamma, grad_beta, coef, training);
    GGA_CHECK_LAUNCH("bn_bwd_final_kernel");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb), dim3(256), 0, stream, (const float4*)grad_masked, (const float4*)x,
                       (const unsigned long long*)nullptr, saved, coef, g, 0, (float4*)grad_x, (float4*)nullptr, amax_grad_x);
    GGA_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return GGA_OK;
}

extern "C" int gga_bn_relu_bwd(const float* grad_y, const float* x, const void* mask_bits, const float* gamma,
                               const float* saved, int64_t rows, int channels, int relu, float* grad_x,
                               float* grad_residual, float* grad_gamma, float* grad_beta, void* workspace,
                               size_t workspace_bytes, void* stream_) {
    return gga_bn_relu_bwd_strided(grad_y, channels, x, mask_bits, gamma, saved, rows, channels, relu, grad_x,
                                   grad_residual, grad_gamma, grad_beta, workspace, workspace_bytes, stream_);
}
