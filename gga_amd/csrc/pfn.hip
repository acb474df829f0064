// a2': fused PillarFeatureNet (single PFNLayer, legacy=True, mode='max') for gfx950.
//
// Reference: mmdet3d/models/voxel_encoders/pillar_encoder.py:93-159 (decorate: offsets to the
// cluster mean and to the pillar centre, legacy in-place view), voxel_encoders/utils.py:145-182
// (Linear(10->64, no bias) + BatchNorm1d over ALL M*P rows incl. zero padding + ReLU + max over
// the P points). Eager PyTorch materialises [M,P,10] and [M,P,64] (2 GB at M=256000) several
// times and runs an 8M-row GEMM; here:
//
//   pass 1  pfn_moments_kernel   one thread per pillar: S1 = sum f (10) and S2 = sum f f^T (55
//                                unique) over the VALID points only (padding rows are zero).
//                                Because z = W f is linear, the BatchNorm batch statistics of
//                                every output channel follow exactly from these moments:
//                                mean_c = w_c.S1/R, E[z^2]_c = w_c^T S2 w_c / R  (R = M*P).
//           pfn_stats_kernel     fixed-order reduction of the block partials in f64, per-channel
//                                mean / invstd / scale / shift, running-stat update.
//   pass 2  pfn_apply_kernel     one wavefront per pillar, lane = output channel: each lane keeps
//                                its 10 weights in registers, the pillar's points are read through
//                                wave-uniform (scalar) loads, y = relu(z*scale+shift), running
//                                max (+ the padding row's relu(shift) when n < P), one coalesced
//                                256 B store per pillar; the arg-max point index is kept (u8).
//   backward pfn_bwd_kernel      lane = channel again: per channel accumulates A = sum g,
//                                Bx = sum g*xhat, G[10] = sum g*f(argmax) over the pillars;
//           pfn_bwd_final_kernel closes the BatchNorm backward analytically with S1/S2:
//                                dW_c = gamma*invstd*(G - A/R*S1 - Bx/R*invstd*(S2 w_c - mean*S1)).
// HBM traffic: the voxel buffer is read twice (only the lines holding valid points), the
// [M,64] output written once; nothing of size M*P*64 ever exists.
#include "gga_common.h"

// rows the caller marks valid: min(m, *num_valid) when a device count is given (capacity-sized
// buffers of the sync-free voxelizer), m otherwise
__device__ __forceinline__ int64_t pfn_valid_rows(int64_t m, const int32_t* __restrict__ num_valid) {
    if (!num_valid) return m;
    const int64_t v = *num_valid;
    return v < m ? (v < 0 ? 0 : v) : m;
}

#define PFN_C 64
#define PFN_F 10
#define PFN_NM 65            // 10 first moments + 55 second moments
#define PFN_SAVED 238        // S1[10] S2[100] mean[64] invstd[64]  (doubles); saved[238] = rows normalised over

struct PfnGeom {
    float vx, vy, vz, xo, yo, zo;
};

// decorated feature vector of one point (pillar_encoder.py:106-150, legacy=True):
// (x-cx, y-cy, z-cz, r, x-mx, y-my, z-mz, x-cx, y-cy, z-cz)
__device__ __forceinline__ void pfn_decorate(const float4 p, const float mean[3], const float cen[3], float f[PFN_F]) {
    f[0] = p.x - cen[0]; f[1] = p.y - cen[1]; f[2] = p.z - cen[2]; f[3] = p.w;
    f[4] = p.x - mean[0]; f[5] = p.y - mean[1]; f[6] = p.z - mean[2];
    f[7] = f[0]; f[8] = f[1]; f[9] = f[2];
}

__device__ __forceinline__ void pfn_pillar_consts(const float4* __restrict__ pts, int n, const int4 co,
                                                  const PfnGeom g, float mean[3], float cen[3]) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int p = 0; p < n; ++p) { const float4 q = pts[p]; sx += q.x; sy += q.y; sz += q.z; }
    const float fn = (float)n;
    mean[0] = sx / fn; mean[1] = sy / fn; mean[2] = sz / fn;       // sum(dim=1) / num_points
    // coors[:,3]*vx + x_offset as TWO rounded f32 ops like the reference: a 1-ulp difference of the
    // ~70 m centre is amplified by gamma*invstd ~ 30-60 downstream. HIP's __fmul_rn/__fadd_rn are
    // plain * and + and hipcc fuses them under its default -ffp-contract=fast, so the product is
    // passed through an empty asm to keep it a separately rounded value.
    float tx = (float)co.w * g.vx, ty = (float)co.z * g.vy, tz = (float)co.y * g.vz;
    asm volatile("" : "+v"(tx), "+v"(ty), "+v"(tz));
    cen[0] = tx + g.xo; cen[1] = ty + g.yo; cen[2] = tz + g.zo;
}

__global__ __launch_bounds__(256) void pfn_moments_kernel(const float4* __restrict__ voxels,
                                                         const int32_t* __restrict__ num_points,
                                                         const int4* __restrict__ coors, int64_t m,
                                                         const int32_t* __restrict__ num_valid, int P,
                                                         PfnGeom g, double* __restrict__ partials) {
    m = pfn_valid_rows(m, num_valid);
    double acc[PFN_NM];
#pragma unroll
    for (int i = 0; i < PFN_NM; ++i) acc[i] = 0.0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < m; v += (int64_t)gridDim.x * 256) {
        int n = num_points[v];
        n = n < P ? n : P;
        if (n <= 0) continue;
        const float4* pts = voxels + v * P;
        float mean[3], cen[3];
        pfn_pillar_consts(pts, n, coors[v], g, mean, cen);
        float loc[PFN_NM];
#pragma unroll
        for (int i = 0; i < PFN_NM; ++i) loc[i] = 0.f;
        for (int p = 0; p < n; ++p) {
            float f[PFN_F];
            pfn_decorate(pts[p], mean, cen, f);
            int k = PFN_F;
#pragma unroll
            for (int a = 0; a < PFN_F; ++a) {
                loc[a] += f[a];
#pragma unroll
                for (int b = a; b < PFN_F; ++b) loc[k++] += f[a] * f[b];
            }
        }
#pragma unroll
        for (int i = 0; i < PFN_NM; ++i) acc[i] += (double)loc[i];
    }
    __shared__ double sh[4][PFN_NM];
#pragma unroll
    for (int i = 0; i < PFN_NM; ++i) {
        const double s = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < PFN_NM)
        partials[(int64_t)blockIdx.x * PFN_NM + threadIdx.x] =
            (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// one block of 64 threads (lane = channel)
__global__ __launch_bounds__(64) void pfn_stats_kernel(const double* __restrict__ partials, int nblocks,
                                                      int64_t m, const int32_t* __restrict__ num_valid, int P,
                                                      const float* __restrict__ weight,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float momentum, int training,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      double* __restrict__ saved, float* __restrict__ scale_shift) {
    __shared__ double S[PFN_NM];
    const int c = threadIdx.x;
    double rows = (double)pfn_valid_rows(m, num_valid) * (double)P;
    rows = rows > 0.0 ? rows : 1.0;
    if (c == 0) saved[PFN_SAVED] = rows;
    if (training) {
        for (int i = c; i < PFN_NM; i += 64) {
            double s = 0.0;
#pragma unroll 16
            for (int b = 0; b < nblocks; ++b) s += partials[(int64_t)b * PFN_NM + i];   // fixed order, loads batched
            S[i] = s;
        }
        __syncthreads();
        // unpack the symmetric second moment
        double S1[PFN_F], S2[PFN_F][PFN_F];
        int k = PFN_F;
        for (int a = 0; a < PFN_F; ++a) {
            S1[a] = S[a];
            for (int b = a; b < PFN_F; ++b) { S2[a][b] = S[k]; S2[b][a] = S[k]; ++k; }
        }
        double w[PFN_F];
        for (int a = 0; a < PFN_F; ++a) w[a] = (double)weight[c * PFN_F + a];
        double m1 = 0.0, m2 = 0.0;
        for (int a = 0; a < PFN_F; ++a) {
            m1 += w[a] * S1[a];
            double t = 0.0;
            for (int b = 0; b < PFN_F; ++b) t += S2[a][b] * w[b];
            m2 += w[a] * t;
        }
        const double mean = m1 / rows;
        double var = m2 / rows - mean * mean;          // biased, as BatchNorm normalises with
        var = var > 0.0 ? var : 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        if (c == 0)
            for (int a = 0; a < PFN_F; ++a) {
                saved[a] = S1[a];
                for (int b = 0; b < PFN_F; ++b) saved[PFN_F + a * PFN_F + b] = S2[a][b];
            }
        saved[110 + c] = mean;
        saved[174 + c] = invstd;
        const float sc = gamma[c] * (float)invstd;
        scale_shift[c] = sc;
        scale_shift[PFN_C + c] = beta[c] - (float)mean * sc;
        // running stats: momentum update with the UNBIASED variance (torch BatchNorm)
        const double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    } else {
        const float invstd = 1.0f / sqrtf(running_var[c] + eps);
        const float sc = gamma[c] * invstd;
        scale_shift[c] = sc;
        scale_shift[PFN_C + c] = beta[c] - running_mean[c] * sc;
    }
}

// one wavefront per pillar, lane = channel
__global__ __launch_bounds__(256) void pfn_apply_kernel(const float4* __restrict__ voxels,
                                                       const int32_t* __restrict__ num_points,
                                                       const int4* __restrict__ coors, int64_t m,
                                                       const int32_t* __restrict__ num_valid, int P, PfnGeom g,
                                                       const float* __restrict__ weight,
                                                       const float* __restrict__ scale_shift,
                                                       float* __restrict__ out, uint8_t* __restrict__ argmax) {
    const int lane = threadIdx.x & 63;
    const int64_t mv = pfn_valid_rows(m, num_valid);
    float w[PFN_F];
#pragma unroll
    for (int a = 0; a < PFN_F; ++a) w[a] = weight[lane * PFN_F + a];
    const float sc = scale_shift[lane], sh = scale_shift[PFN_C + lane];
    const float ypad = fmaxf(sh, 0.0f);               // a zero (padding) row after BN + ReLU
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t v = wave0; v < m; v += nwaves) {
        const int64_t vu = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff)) |
                           ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32);   // wave-uniform
        if (vu >= mv) {                                // capacity rows past the valid count: defined zeros
            out[vu * PFN_C + lane] = 0.0f;
            argmax[vu * PFN_C + lane] = 255;
            continue;
        }
        int n = num_points[vu];
        n = n < P ? n : P;
        const float4* pts = voxels + vu * P;
        float best = -INFINITY;
        int bi = 255;                                  // 255 = padding row
        if (n < P || n <= 0) { best = ypad; }
        if (n > 0) {
            float mean[3], cen[3];
            pfn_pillar_consts(pts, n, coors[vu], g, mean, cen);
            for (int p = 0; p < n; ++p) {
                float f[PFN_F];
                pfn_decorate(pts[p], mean, cen, f);
                float z = 0.0f;
#pragma unroll
                for (int a = 0; a < PFN_F; ++a) z += f[a] * w[a];
                const float y = fmaxf(z * sc + sh, 0.0f);
                if (y > best) { best = y; bi = p; }
            }
        }
        out[vu * PFN_C + lane] = best;
        argmax[vu * PFN_C + lane] = (uint8_t)bi;
    }
}

#define PFN_BW 12   // per-channel accumulators of the backward: A, Bx, G[10]

__global__ __launch_bounds__(256) void pfn_bwd_kernel(const float4* __restrict__ voxels,
                                                     const int32_t* __restrict__ num_points,
                                                     const int4* __restrict__ coors, int64_t m,
                                                     const int32_t* __restrict__ num_valid, int P, PfnGeom g,
                                                     const float* __restrict__ weight, const double* __restrict__ saved,
                                                     const float* __restrict__ out, const uint8_t* __restrict__ argmax,
                                                     const float* __restrict__ grad_out, float* __restrict__ partials) {
    const int lane = threadIdx.x & 63;
    float w[PFN_F];
#pragma unroll
    for (int a = 0; a < PFN_F; ++a) w[a] = weight[lane * PFN_F + a];
    const float mean_c = (float)saved[110 + lane], invstd_c = (float)saved[174 + lane];
    float acc[PFN_BW];
#pragma unroll
    for (int i = 0; i < PFN_BW; ++i) acc[i] = 0.0f;
    m = pfn_valid_rows(m, num_valid);
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t v = wave0; v < m; v += nwaves) {
        const int64_t vu = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff)) |
                           ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32);
        const float y = out[vu * PFN_C + lane];
        const float gy = y > 0.0f ? grad_out[vu * PFN_C + lane] : 0.0f;    // ReLU gate
        const int bi = argmax[vu * PFN_C + lane];
        int n = num_points[vu];
        n = n < P ? n : P;
        float f[PFN_F];
#pragma unroll
        for (int a = 0; a < PFN_F; ++a) f[a] = 0.0f;
        float z = 0.0f;
        if (n > 0) {                                    // uniform branch; the gather below is per lane
            const float4* pts = voxels + vu * P;
            float mean[3], cen[3];
            pfn_pillar_consts(pts, n, coors[vu], g, mean, cen);
            if (bi != 255) {
                pfn_decorate(pts[bi], mean, cen, f);
#pragma unroll
                for (int a = 0; a < PFN_F; ++a) z += f[a] * w[a];
            }
        }
        const float xhat = (z - mean_c) * invstd_c;
        acc[0] += gy;
        acc[1] += gy * xhat;
#pragma unroll
        for (int a = 0; a < PFN_F; ++a) acc[2 + a] += gy * f[a];
    }
    __shared__ float sh[4][PFN_BW][PFN_C];
#pragma unroll
    for (int i = 0; i < PFN_BW; ++i) sh[threadIdx.x >> 6][i][lane] = acc[i];
    __syncthreads();
    for (int t = threadIdx.x; t < PFN_BW * PFN_C; t += 256) {
        const int i = t / PFN_C, c = t - i * PFN_C;
        partials[(int64_t)blockIdx.x * PFN_BW * PFN_C + t] = (sh[0][i][c] + sh[1][i][c]) + (sh[2][i][c] + sh[3][i][c]);
    }
}

__global__ __launch_bounds__(768) void pfn_bwd_final_kernel(const float* __restrict__ partials, int nblocks,
                                                           const float* __restrict__ weight,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ saved,
                                                           float* __restrict__ grad_weight,
                                                           float* __restrict__ grad_gamma,
                                                           float* __restrict__ grad_beta) {
    // stage 1: thread t = i*64 + c sums its accumulator over the block partials (coalesced,
    // independent loads, fixed order)
    __shared__ double red[PFN_BW][PFN_C];
    {
        const int t = threadIdx.x;
        double s = 0.0;
#pragma unroll 16
        for (int b = 0; b < nblocks; ++b) s += (double)partials[(int64_t)b * PFN_BW * PFN_C + t];
        red[t / PFN_C][t % PFN_C] = s;
    }
    __syncthreads();
    if (threadIdx.x >= PFN_C) return;
    const int c = threadIdx.x;
    const double A = red[0][c], Bx = red[1][c];
    const double rows = saved[PFN_SAVED];
    const double mean = saved[110 + c], invstd = saved[174 + c];
    grad_beta[c] = (float)A;
    grad_gamma[c] = (float)Bx;
    const double k = (double)gamma[c] * invstd;
    for (int a = 0; a < PFN_F; ++a) {
        double s2w = 0.0;
        for (int b = 0; b < PFN_F; ++b) s2w += saved[PFN_F + a * PFN_F + b] * (double)weight[c * PFN_F + b];
        const double xf = invstd * (s2w - mean * saved[a]);                  // sum_rows xhat_row * f_row[a]
        grad_weight[c * PFN_F + a] = (float)(k * (red[2 + a][c] - A / rows * saved[a] - Bx / rows * xf));
    }
}

static int pfn_blocks(int64_t m) {
    int64_t b = (m + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}
static int pfn_wave_blocks(int64_t m) {
    int64_t b = (m + 3) / 4;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

extern "C" size_t gga_pfn_workspace_bytes(int64_t m) {
    const size_t a = (size_t)pfn_blocks(m) * PFN_NM * sizeof(double);
    const size_t b = (size_t)pfn_wave_blocks(m) * PFN_BW * PFN_C * sizeof(float);
    return gga_align_up(a > b ? a : b, 256) + 2 * PFN_C * sizeof(float);
}

static int pfn_check(const char* fn, const gga_pfn_params* prm, int64_t m, int P) {
    GGA_REQUIRE(prm, "%s: null params", fn);
    GGA_REQUIRE(m >= 1 && P >= 1 && P <= 254, "%s: bad sizes (m=%lld, max_points=%d; need 1..254)", fn, (long long)m, P);
    GGA_REQUIRE(prm->channels == PFN_C && prm->in_features == 4,
                "%s: the fused kernel is specialised for 4 point features -> %d channels (got %d -> %d)", fn, PFN_C,
                prm->in_features, prm->channels);
    return GGA_OK;
}

extern "C" int gga_pfn_fwd(const float* voxels, const int32_t* num_points, const int32_t* coors, int64_t m,
                           const int32_t* num_valid, int P, const gga_pfn_params* prm, const float* weight, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* out, uint8_t* argmax, double* saved,
                           void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = pfn_check("gga_pfn_fwd", prm, m, P)) return rc;
    GGA_REQUIRE(voxels && num_points && coors && weight && gamma && beta && running_mean && running_var && out &&
                    argmax && saved && workspace,
                "gga_pfn_fwd: null pointer argument");
    if (workspace_bytes < gga_pfn_workspace_bytes(m)) {
        gga_set_error("gga_pfn_fwd: workspace %zu B < required %zu B", workspace_bytes, gga_pfn_workspace_bytes(m));
        return GGA_ERR_WORKSPACE;
    }
    const PfnGeom g = { prm->voxel_size[0], prm->voxel_size[1], prm->voxel_size[2],
                        prm->offsets[0], prm->offsets[1], prm->offsets[2] };
    double* partials = (double*)workspace;
    float* scale_shift = (float*)((char*)workspace + gga_pfn_workspace_bytes(m) - 2 * PFN_C * sizeof(float));
    const int nb = pfn_blocks(m);
    if (prm->training) {
        hipLaunchKernelGGL(pfn_moments_kernel, dim3(nb), dim3(256), 0, stream, (const float4*)voxels, num_points,
                           (const int4*)coors, m, num_valid, P, g, partials);
        GGA_CHECK_LAUNCH("pfn_moments_kernel");
    }
    hipLaunchKernelGGL(pfn_stats_kernel, dim3(1), dim3(64), 0, stream, partials, nb, m, num_valid, P, weight,
                       gamma, beta, prm->eps, prm->momentum, prm->training, running_mean, running_var, saved,
                       scale_shift);
    GGA_CHECK_LAUNCH("pfn_stats_kernel");
    hipLaunchKernelGGL(pfn_apply_kernel, dim3(pfn_wave_blocks(m)), dim3(256), 0, stream, (const float4*)voxels,
                       num_points, (const int4*)coors, m, num_valid, P, g, weight, scale_shift, out, argmax);
    GGA_CHECK_LAUNCH("pfn_apply_kernel");
    return GGA_OK;
}

extern "C" int gga_pfn_bwd(const float* voxels, const int32_t* num_points, const int32_t* coors, int64_t m,
                           const int32_t* num_valid, int P, const gga_pfn_params* prm, const float* weight, const float* gamma, const float* out,
                           const uint8_t* argmax, const double* saved, const float* grad_out, float* grad_weight,
                           float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes,
                           void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = pfn_check("gga_pfn_bwd", prm, m, P)) return rc;
    GGA_REQUIRE(voxels && num_points && coors && weight && gamma && out && argmax && saved && grad_out &&
                    grad_weight && grad_gamma && grad_beta && workspace,
                "gga_pfn_bwd: null pointer argument");
    GGA_REQUIRE(prm->training, "gga_pfn_bwd: backward is defined for training-mode batch statistics");
    if (workspace_bytes < gga_pfn_workspace_bytes(m)) {
        gga_set_error("gga_pfn_bwd: workspace %zu B < required %zu B", workspace_bytes, gga_pfn_workspace_bytes(m));
        return GGA_ERR_WORKSPACE;
    }
    const PfnGeom g = { prm->voxel_size[0], prm->voxel_size[1], prm->voxel_size[2],
                        prm->offsets[0], prm->offsets[1], prm->offsets[2] };
    const int nb = pfn_wave_blocks(m);
    hipLaunchKernelGGL(pfn_bwd_kernel, dim3(nb), dim3(256), 0, stream, (const float4*)voxels, num_points,
                       (const int4*)coors, m, num_valid, P, g, weight, saved, out, argmax, grad_out, (float*)workspace);
    GGA_CHECK_LAUNCH("pfn_bwd_kernel");
    hipLaunchKernelGGL(pfn_bwd_final_kernel, dim3(1), dim3(PFN_BW * PFN_C), 0, stream, (const float*)workspace, nb,
                       weight, gamma, saved, grad_weight, grad_gamma, grad_beta);
    GGA_CHECK_LAUNCH("pfn_bwd_final_kernel");
    return GGA_OK;
}
