// Fused GroupNorm (+ ReLU) over a channels-last [B, H*W, C] activation, forward and backward (gfx950).
//
// Reference: the GN + ReLU of every ConvModule in the PGD / FCOS3D head towers and branches
// (mmdet3d/models/dense_heads/anchor_free_mono3d_head.py:160-250 with norm_cfg = dict(type='GN', num_groups=32,
// requires_grad=True), configs/_base_/models/pgd.py); the framework's GroupNorm kernels work on NCHW memory, so a
// channels-last trunk pays two layout copies per layer and direction around them (measured: 45 ms of the 390 ms PGD
// step). Same structure as bn_relu.hip with a sample dimension: statistics per (sample, group of C/G channels).
//   fwd  pass 1  per (sample, channel) sum / sum of squares -> per (sample, group) mean, rstd -> per (sample, channel)
//                scale = gamma * rstd, shift = beta - mean * scale
//        pass 2  y = relu(x * scale + shift)
//   bwd  pass 1  per (sample, channel) sum g, sum g * xhat with g = dy * [x * scale + shift > 0] (mask recomputed)
//                -> per (sample, group) A = sum gamma g, B = sum gamma g xhat; d_gamma / d_beta = sums over samples
//        pass 2  dx = gamma * rstd * g - rstd * (A + xhat * B) / m,  m = elements per (sample, group)
// A thread owns the same 4 channels throughout (C/4 divides 256); a group is a whole number of such quads.
#include "gga_common.h"

#define GN_CHUNKS 64          // row chunks per sample (partial sums per (sample, chunk))
#define GN_U 4

struct GnGeom {
    int B, C, c4, sh, G, gs;      // c4 = C / 4 (power of two), sh = log2(c4), gs = channels per group
    int64_t rows;                 // H * W rows per sample
};

// partials: [B][GN_CHUNKS][2][C] f64.  BWD: a = dy, b = x
template <bool BWD>
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       const float* __restrict__ stat,       // [B][G][2] mean, rstd
                                                       const float* __restrict__ scale_shift, // [B][2][C]
                                                       GnGeom g, int relu, double* __restrict__ partials) {
    const int tid = threadIdx.x, smp = blockIdx.y, chunk = blockIdx.x;
    const int cg = tid & (g.c4 - 1), r0 = tid >> g.sh, rstep = 256 >> g.sh;
    const int64_t rb = g.rows * chunk / GN_CHUNKS, re = g.rows * (chunk + 1) / GN_CHUNKS;
    const float4* ap = a + (int64_t)smp * g.rows * g.c4;
    const float4* bp = BWD ? b + (int64_t)smp * g.rows * g.c4 : nullptr;
    float mean = 0.f, rstd = 1.f, sc[4] = {0, 0, 0, 0}, sf[4] = {0, 0, 0, 0};
    if (BWD) {
        const int grp = (4 * cg) / g.gs;
        mean = stat[((int64_t)smp * g.G + grp) * 2];
        rstd = stat[((int64_t)smp * g.G + grp) * 2 + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = scale_shift[((int64_t)smp * 2) * g.C + 4 * cg + j]; sf[j] = scale_shift[((int64_t)smp * 2 + 1) * g.C + 4 * cg + j]; }
    }
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float f0[4] = {0, 0, 0, 0}, f1[4] = {0, 0, 0, 0};
    int run = 0;
    for (int64_t rbase = rb + r0; rbase < re; rbase += (int64_t)rstep * GN_U) {
        float4 av[GN_U], bv[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int64_t rr = rbase + (int64_t)u * rstep;
            av[u] = make_float4(0.f, 0.f, 0.f, 0.f); bv[u] = av[u];
            if (rr < re) { av[u] = ap[rr * g.c4 + cg]; if (BWD) bv[u] = bp[rr * g.c4 + cg]; }
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            if (rbase + (int64_t)u * rstep >= re) continue;
            float va[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
            if (BWD) {
                const float xa[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (relu && !(fmaf(xa[j], sc[j], sf[j]) > 0.0f)) va[j] = 0.0f;
                    f0[j] += va[j]; f1[j] += va[j] * ((xa[j] - mean) * rstd);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { f0[j] += va[j]; f1[j] += va[j] * va[j]; }
            }
        }
        if (++run == 8) {
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
    if (tid < g.c4) {                                   // threads tid, tid + c4, ... share the channel quad: fixed order
        double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = tid; t < 256; t += g.c4)
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] += sh[t][q];
        double* out = partials + ((int64_t)smp * GN_CHUNKS + chunk) * 2 * g.C;
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[4 * tid + j] = r[j]; out[g.C + 4 * tid + j] = r[4 + j]; }
    }
}

// one block per sample, thread = channel: fold the chunks, combine the channels of a group
__global__ __launch_bounds__(1024) void gn_fwd_final_kernel(const double* __restrict__ partials, GnGeom g, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ stat, float* __restrict__ scale_shift) {
    __shared__ double cs[1024], cq[1024];
    const int smp = blockIdx.x, c = threadIdx.x;
    double s = 0.0, q = 0.0;
    if (c < g.C)
        for (int k = 0; k < GN_CHUNKS; ++k) {
            const double* p = partials + ((int64_t)smp * GN_CHUNKS + k) * 2 * g.C;
            s += p[c]; q += p[g.C + c];
        }
    cs[c] = s; cq[c] = q;
    __syncthreads();
    if (c >= g.C) return;
    const int grp = c / g.gs;
    double gs_ = 0.0, gq = 0.0;
    for (int j = grp * g.gs; j < (grp + 1) * g.gs; ++j) { gs_ += cs[j]; gq += cq[j]; }
    const double m = (double)g.rows * g.gs;
    const double mu = gs_ / m;
    double var = gq / m - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (c == grp * g.gs) { stat[((int64_t)smp * g.G + grp) * 2] = mean; stat[((int64_t)smp * g.G + grp) * 2 + 1] = rstd; }
    const float scl = (gamma ? gamma[c] : 1.0f) * rstd;
    scale_shift[((int64_t)smp * 2) * g.C + c] = scl;
    scale_shift[((int64_t)smp * 2 + 1) * g.C + c] = (beta ? beta[c] : 0.0f) - mean * scl;
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float4* __restrict__ x, const float* __restrict__ scale_shift,
                                                      GnGeom g, int relu, float4* __restrict__ y, uint32_t* __restrict__ amax) {
    const int tid = threadIdx.x, smp = blockIdx.y;
    const int cg = tid & (g.c4 - 1);
    const float4 sc = *reinterpret_cast<const float4*>(scale_shift + ((int64_t)smp * 2) * g.C + 4 * cg);
    const float4 sf = *reinterpret_cast<const float4*>(scale_shift + ((int64_t)smp * 2 + 1) * g.C + 4 * cg);
    const int64_t n4 = g.rows * g.c4, base = (int64_t)smp * n4;
    uint32_t am = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + tid; e < n4; e += (int64_t)gridDim.x * 256) {     // stride multiple of c4: cg fixed
        const float4 xv = x[base + e];
        float4 v;
        v.x = fmaf(xv.x, sc.x, sf.x); v.y = fmaf(xv.y, sc.y, sf.y); v.z = fmaf(xv.z, sc.z, sf.z); v.w = fmaf(xv.w, sc.w, sf.w);
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        y[base + e] = v;
        if (amax) { am = gga_amax_of(v.x, am); am = gga_amax_of(v.y, am); am = gga_amax_of(v.z, am); am = gga_amax_of(v.w, am); }
    }
    if (amax) gga_amax_commit(am, amax);
}

// per sample: A = sum gamma g, B = sum gamma g xhat per group -> coef[b][3][C] = (gamma * rstd, rstd * A / m, rstd * B / m)
__global__ __launch_bounds__(1024) void gn_bwd_final_kernel(const double* __restrict__ partials, GnGeom g,
                                                           const float* __restrict__ gamma, const float* __restrict__ stat,
                                                           float* __restrict__ coef, double* __restrict__ per_sample) {
    __shared__ double cs[1024], cq[1024];
    const int smp = blockIdx.x, c = threadIdx.x;
    double s = 0.0, q = 0.0;
    if (c < g.C)
        for (int k = 0; k < GN_CHUNKS; ++k) {
            const double* p = partials + ((int64_t)smp * GN_CHUNKS + k) * 2 * g.C;
            s += p[c]; q += p[g.C + c];
        }
    const double gm = c < g.C ? (gamma ? (double)gamma[c] : 1.0) : 0.0;
    cs[c] = gm * s; cq[c] = gm * q;
    if (c < g.C) { per_sample[((int64_t)smp * 2) * g.C + c] = s; per_sample[((int64_t)smp * 2 + 1) * g.C + c] = q; }
    __syncthreads();
    if (c >= g.C) return;
    const int grp = c / g.gs;
    double A = 0.0, Bq = 0.0;
    for (int j = grp * g.gs; j < (grp + 1) * g.gs; ++j) { A += cs[j]; Bq += cq[j]; }
    const double m = (double)g.rows * g.gs;
    const float rstd = stat[((int64_t)smp * g.G + grp) * 2 + 1];
    coef[((int64_t)smp * 3) * g.C + c] = (float)gm * rstd;
    coef[((int64_t)smp * 3 + 1) * g.C + c] = (float)(rstd * A / m);
    coef[((int64_t)smp * 3 + 2) * g.C + c] = (float)(rstd * Bq / m);
}

__global__ __launch_bounds__(256) void gn_param_grad_kernel(const double* __restrict__ per_sample, int B, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < B; ++b) { s += per_sample[((int64_t)b * 2) * C + c]; q += per_sample[((int64_t)b * 2 + 1) * C + c]; }
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                          const float* __restrict__ stat, const float* __restrict__ scale_shift,
                                                          const float* __restrict__ coef, GnGeom g, int relu,
                                                          float4* __restrict__ dx, uint32_t* __restrict__ amax) {
    const int tid = threadIdx.x, smp = blockIdx.y;
    const int cg = tid & (g.c4 - 1);
    const int grp = (4 * cg) / g.gs;
    const float mean = stat[((int64_t)smp * g.G + grp) * 2], rstd = stat[((int64_t)smp * g.G + grp) * 2 + 1];
    float sc[4], sf[4], k[4], c1[4], c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * cg + j;
        sc[j] = scale_shift[((int64_t)smp * 2) * g.C + c]; sf[j] = scale_shift[((int64_t)smp * 2 + 1) * g.C + c];
        k[j] = coef[((int64_t)smp * 3) * g.C + c]; c1[j] = coef[((int64_t)smp * 3 + 1) * g.C + c]; c2[j] = coef[((int64_t)smp * 3 + 2) * g.C + c];
    }
    const int64_t n4 = g.rows * g.c4, base = (int64_t)smp * n4;
    uint32_t am = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + tid; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 gv = dy[base + e], xv = x[base + e];
        float ga[4] = {gv.x, gv.y, gv.z, gv.w};
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (relu && !(fmaf(xa[j], sc[j], sf[j]) > 0.0f)) ga[j] = 0.0f;
            o[j] = k[j] * ga[j] - c1[j] - ((xa[j] - mean) * rstd) * c2[j];
            if (amax) am = gga_amax_of(o[j], am);
        }
        dx[base + e] = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (amax) gga_amax_commit(am, amax);
}

static int gn_check(const char* fn, int B, int64_t rows, int C, int G) {
    GGA_REQUIRE(B >= 1 && rows >= 1 && C >= 4 && C <= 1024 && G >= 1 && C % G == 0 && (C / G) % 4 == 0,
                "%s: need channels <= 1024 and channels per group %% 4 == 0 (B=%d rows=%lld C=%d groups=%d)", fn, B, (long long)rows, C, G);
    const int c4 = C / 4;
    GGA_REQUIRE(c4 <= 256 && 256 % c4 == 0, "%s: channels/4 (%d) must divide 256", fn, c4);
    return GGA_OK;
}
static GnGeom gn_geom(int B, int64_t rows, int C, int G) {
    GnGeom g;
    g.B = B; g.C = C; g.c4 = C / 4; g.G = G; g.gs = C / G; g.rows = rows;
    g.sh = 0;
    while ((1 << g.sh) < g.c4) ++g.sh;
    return g;
}
static int gn_apply_grid(const GnGeom& g) {
    int64_t b = (g.rows * g.c4 + 256 * 8 - 1) / (256 * 8);
    b = b < 1 ? 1 : (b > 512 ? 512 : b);
    return (int)b;                                        // x 256 threads: a multiple of c4, so a thread keeps its channel quad
}

// workspace: partials [B][GN_CHUNKS][2][C] f64 + per-sample sums [B][2][C] f64 + coef [B][3][C] f32
extern "C" size_t gga_gn_relu_workspace_bytes(int B, int channels) {
    return (size_t)B * GN_CHUNKS * 2 * channels * sizeof(double) + (size_t)B * 2 * channels * sizeof(double) +
           (size_t)B * 3 * channels * sizeof(float) + 256;
}

extern "C" int gga_gn_relu_fwd(const float* x, const float* gamma, const float* beta, int B, int64_t rows_per_sample, int channels,
                               int groups, float eps, int relu, float* y, float* stat, float* scale_shift, uint32_t* amax_y,
                               void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = gn_check("gga_gn_relu_fwd", B, rows_per_sample, channels, groups)) return rc;
    GGA_REQUIRE(x && y && stat && scale_shift && workspace && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
                "gga_gn_relu_fwd: null or misaligned pointer argument");
    if (workspace_bytes < gga_gn_relu_workspace_bytes(B, channels)) { gga_set_error("gga_gn_relu_fwd: workspace too small"); return GGA_ERR_WORKSPACE; }
    const GnGeom g = gn_geom(B, rows_per_sample, channels, groups);
    double* partials = (double*)workspace;
    hipLaunchKernelGGL(gn_reduce_kernel<false>, dim3(GN_CHUNKS, B), dim3(256), 0, stream, (const float4*)x, (const float4*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, g, 0, partials);
    GGA_CHECK_LAUNCH("gn_reduce_kernel<fwd>");
    hipLaunchKernelGGL(gn_fwd_final_kernel, dim3(B), dim3(1024), 0, stream, partials, g, eps, gamma, beta, stat, scale_shift);
    GGA_CHECK_LAUNCH("gn_fwd_final_kernel");
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_apply_grid(g), B), dim3(256), 0, stream, (const float4*)x, scale_shift, g, relu,
                       (float4*)y, amax_y);
    GGA_CHECK_LAUNCH("gn_apply_kernel");
    return GGA_OK;
}

extern "C" int gga_gn_relu_bwd(const float* grad_y, const float* x, const float* gamma, const float* stat, const float* scale_shift,
                               int B, int64_t rows_per_sample, int channels, int groups, int relu, float* grad_x, float* grad_gamma,
                               float* grad_beta, uint32_t* amax_grad_x, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = gn_check("gga_gn_relu_bwd", B, rows_per_sample, channels, groups)) return rc;
    GGA_REQUIRE(grad_y && x && stat && scale_shift && grad_x && workspace, "gga_gn_relu_bwd: null pointer argument");
    if (workspace_bytes < gga_gn_relu_workspace_bytes(B, channels)) { gga_set_error("gga_gn_relu_bwd: workspace too small"); return GGA_ERR_WORKSPACE; }
    const GnGeom g = gn_geom(B, rows_per_sample, channels, groups);
    double* partials = (double*)workspace;
    double* per_sample = partials + (size_t)B * GN_CHUNKS * 2 * channels;
    float* coef = (float*)(per_sample + (size_t)B * 2 * channels);
    hipLaunchKernelGGL(gn_reduce_kernel<true>, dim3(GN_CHUNKS, B), dim3(256), 0, stream, (const float4*)grad_y, (const float4*)x, stat,
                       scale_shift, g, relu, partials);
    GGA_CHECK_LAUNCH("gn_reduce_kernel<bwd>");
    hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(B), dim3(1024), 0, stream, partials, g, gamma, stat, coef, per_sample);
    GGA_CHECK_LAUNCH("gn_bwd_final_kernel");
    if (grad_gamma || grad_beta) {
        hipLaunchKernelGGL(gn_param_grad_kernel, dim3((channels + 255) / 256), dim3(256), 0, stream, per_sample, B, channels, grad_gamma,
                           grad_beta);
        GGA_CHECK_LAUNCH("gn_param_grad_kernel");
    }
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gn_apply_grid(g), B), dim3(256), 0, stream, (const float4*)grad_y, (const float4*)x, stat,
                       scale_shift, coef, g, relu, (float4*)grad_x, amax_grad_x);
    GGA_CHECK_LAUNCH("gn_bwd_apply_kernel");
    return GGA_OK;
}
