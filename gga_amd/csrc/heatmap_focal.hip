// a6/a7 heat-map target splat and a8 fused clip_sigmoid + gaussian focal loss (gfx950).
//
// Reference: mmdet3d/core/utils/gaussian.py:25-54 (draw_heatmap_gaussian, called per
// object from centerpoint_head_gga.py:576 with a numpy patch + H2D copy + torch.max),
// mmdet3d/models/utils/clip_sigmoid.py:16 and mmdet's GaussianFocalLoss with
// avg_factor = max(num_pos, 1) (centerpoint_head_gga.py:650-655, which host-syncs on
// num_pos.item()). Here: one launch splats every object of every frame/task with an
// order-independent atomic max, and the focal loss reads logits + target once, keeps
// num_pos on the device and reduces deterministically (fixed-order two-stage sum).
#include <float.h>

#include "gga_common.h"

__global__ __launch_bounds__(256) void heatmap_splat_kernel(float* __restrict__ heatmap, int H, int W,
                                                           const int32_t* __restrict__ objs,
                                                           const float* __restrict__ patch_table,
                                                           const int32_t* __restrict__ patch_offsets,
                                                           int max_radius) {
    const int4 o = reinterpret_cast<const int4*>(objs)[blockIdx.x];   // (map, cx, cy, radius)
    const int r = o.w;
    if (r < 0 || r > max_radius || o.y < 0 || o.y >= W || o.z < 0 || o.z >= H) return;
    const int d = 2 * r + 1;
    // clip the patch to the map exactly like gaussian.py:42-50
    const int left = min(o.y, r), right = min(W - o.y, r + 1);
    const int top = min(o.z, r), bottom = min(H - o.z, r + 1);
    const int pw = left + right, ph = top + bottom;
    const float* patch = patch_table + patch_offsets[r];
    int* hm = reinterpret_cast<int*>(heatmap + (int64_t)o.x * H * W);
    for (int t = threadIdx.x; t < pw * ph; t += 256) {
        const int yy = t / pw - top, xx = t - (t / pw) * pw - left;
        const float g = patch[(r + yy) * d + (r + xx)];
        // values are >= 0, so float order == int order: max is associative & commutative,
        // the result does not depend on the order objects arrive in.
        atomicMax(&hm[(int64_t)(o.z + yy) * W + (o.y + xx)], __float_as_int(g));
    }
}

extern "C" int gga_heatmap_splat(float* heatmap, int n_maps, int H, int W, const int32_t* objs, int n_obj,
                                 const float* patch_table, const int32_t* patch_offsets, int max_radius,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(heatmap && n_maps >= 1 && H >= 1 && W >= 1, "gga_heatmap_splat: bad heatmap arguments");
    GGA_REQUIRE(n_obj == 0 || (objs && patch_table && patch_offsets), "gga_heatmap_splat: null pointer argument");
    GGA_CHECK_HIP(hipMemsetAsync(heatmap, 0, (size_t)n_maps * H * W * sizeof(float), stream), "heatmap memset");
    if (n_obj > 0) {
        hipLaunchKernelGGL(heatmap_splat_kernel, dim3(n_obj), dim3(256), 0, stream, heatmap, H, W, objs, patch_table,
                           patch_offsets, max_radius);
        GGA_CHECK_LAUNCH("heatmap_splat_kernel");
    }
    return GGA_OK;
}

// ---------------------------------------------------------------------------------
#define FOCAL_MAX_BLOCKS 1024

struct FocalTerms { float loss; float dlogit; };

// loss_i and (optionally) d loss_i / d logit_i for one element, fp32 like the reference.
template <bool GRAD>
__device__ __forceinline__ FocalTerms focal_terms(float x, float t, float alpha, float gamma) {
    const float eps = 1e-12f, lo = 1e-4f, hi = 1.0f - 1e-4f;
    const float s = 1.0f / (1.0f + expf(-x));
    const float p = fminf(fmaxf(s, lo), hi);
    const float omp = 1.0f - p;
    const bool pos = (t == 1.0f);
    const float omt = 1.0f - t;
    float negw;
    if (gamma == 4.0f) { const float q = omt * omt; negw = q * q; } else negw = powf(omt, gamma);
    float pa = 1.0f, oa = 1.0f;                       // p^alpha, (1-p)^alpha
    if (alpha != 0.0f) { pa = powf(p, alpha); oa = powf(omp, alpha); }
    const float lp = logf(p + eps), lq = logf(omp + eps);
    FocalTerms r;
    r.loss = (pos ? -lp * oa : 0.0f) + (-lq * pa * negw);
    r.dlogit = 0.0f;
    if (GRAD) {
        float dp = 0.0f;
        if (pos) {
            dp += -oa / (p + eps);
            if (alpha != 0.0f) dp += lp * alpha * powf(omp, alpha - 1.0f);
        }
        if (negw != 0.0f) {
            dp += pa * negw / (omp + eps);
            if (alpha != 0.0f) dp += -lq * alpha * powf(p, alpha - 1.0f) * negw;
        }
        const float pass = (s >= lo && s <= hi) ? 1.0f : 0.0f;   // clamp backward
        r.dlogit = dp * pass * s * (1.0f - s);
    }
    return r;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits,
                                                       const float* __restrict__ target, int64_t n, float alpha,
                                                       float gamma, float* __restrict__ partials) {
    float acc = 0.0f;
    int npos = 0;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(logits)[i];
        const float4 t = reinterpret_cast<const float4*>(target)[i];
        acc += focal_terms<false>(x.x, t.x, alpha, gamma).loss; npos += (t.x == 1.0f);
        acc += focal_terms<false>(x.y, t.y, alpha, gamma).loss; npos += (t.y == 1.0f);
        acc += focal_terms<false>(x.z, t.z, alpha, gamma).loss; npos += (t.z == 1.0f);
        acc += focal_terms<false>(x.w, t.w, alpha, gamma).loss; npos += (t.w == 1.0f);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            acc += focal_terms<false>(logits[i], target[i], alpha, gamma).loss;
            npos += (target[i] == 1.0f);
        }
    }
    acc = wave_sum(acc);
    npos = wave_sum(npos);
    __shared__ float s_acc[4];
    __shared__ int s_pos[4];
    if ((threadIdx.x & 63) == 0) { s_acc[threadIdx.x >> 6] = acc; s_pos[threadIdx.x >> 6] = npos; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3]);
        partials[2 * blockIdx.x + 1] = (float)(s_pos[0] + s_pos[1] + s_pos[2] + s_pos[3]);
    }
}

__global__ __launch_bounds__(256) void focal_final_kernel(const float* __restrict__ partials, int nblocks,
                                                         float scale, float* __restrict__ out) {
    double acc = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { acc += partials[2 * i]; cnt += partials[2 * i + 1]; }
    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    __shared__ double s[8];
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = acc; s[4 + (threadIdx.x >> 6)] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (s[0] + s[1]) + (s[2] + s[3]);
        const double npos = (s[4] + s[5]) + (s[6] + s[7]);
        const float avg = (float)((npos > 1.0 ? npos : 1.0) + (double)FLT_EPSILON);   // mmdet weight_reduce_loss
        out[0] = ((float)tot / avg) * scale;
        out[1] = (float)npos;
    }
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits,
                                                       const float* __restrict__ target, int64_t n, float alpha,
                                                       float gamma, float scale, const float* __restrict__ fwd_out,
                                                       const float* __restrict__ grad_out,
                                                       float* __restrict__ grad_logits) {
    const float npos = fwd_out[1];
    const float avg = (float)((double)(npos > 1.0f ? npos : 1.0f) + (double)FLT_EPSILON);
    const float k = (*grad_out) * scale / avg;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(logits)[i];
        const float4 t = reinterpret_cast<const float4*>(target)[i];
        float4 g;
        g.x = k * focal_terms<true>(x.x, t.x, alpha, gamma).dlogit;
        g.y = k * focal_terms<true>(x.y, t.y, alpha, gamma).dlogit;
        g.z = k * focal_terms<true>(x.z, t.z, alpha, gamma).dlogit;
        g.w = k * focal_terms<true>(x.w, t.w, alpha, gamma).dlogit;
        reinterpret_cast<float4*>(grad_logits)[i] = g;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256)
            grad_logits[i] = k * focal_terms<true>(logits[i], target[i], alpha, gamma).dlogit;
}

static int focal_blocks(int64_t n) {
    int64_t b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > FOCAL_MAX_BLOCKS ? FOCAL_MAX_BLOCKS : b);
}

extern "C" size_t gga_focal_loss_workspace_bytes(int64_t n) { return (size_t)focal_blocks(n) * 2 * sizeof(float); }

extern "C" int gga_focal_loss_fwd(const float* logits, const float* target, int64_t n, float alpha, float gamma,
                                  float scale, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(logits && target && out && workspace && n > 0, "gga_focal_loss_fwd: null pointer or n <= 0");
    GGA_REQUIRE(((uintptr_t)logits & 15) == 0 && ((uintptr_t)target & 15) == 0,
                "gga_focal_loss_fwd: logits/target must be 16-byte aligned");
    if (workspace_bytes < gga_focal_loss_workspace_bytes(n)) {
        gga_set_error("gga_focal_loss_fwd: workspace %zu B < required %zu B", workspace_bytes,
                      gga_focal_loss_workspace_bytes(n));
        return GGA_ERR_WORKSPACE;
    }
    const int nb = focal_blocks(n);
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(nb), dim3(256), 0, stream, logits, target, n, alpha, gamma,
                       (float*)workspace);
    GGA_CHECK_LAUNCH("focal_fwd_kernel");
    hipLaunchKernelGGL(focal_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, nb, scale, out);
    GGA_CHECK_LAUNCH("focal_final_kernel");
    return GGA_OK;
}

extern "C" int gga_focal_loss_bwd(const float* logits, const float* target, int64_t n, float alpha, float gamma,
                                  float scale, const float* fwd_out, const float* grad_out, float* grad_logits,
                                  void* stream_) {
    GGA_REQUIRE(logits && target && fwd_out && grad_out && grad_logits && n > 0,
                "gga_focal_loss_bwd: null pointer or n <= 0");
    GGA_REQUIRE(((uintptr_t)logits & 15) == 0 && ((uintptr_t)target & 15) == 0 && ((uintptr_t)grad_logits & 15) == 0,
                "gga_focal_loss_bwd: buffers must be 16-byte aligned");
    int64_t b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream_, logits, target, n,
                       alpha, gamma, scale, fwd_out, grad_out, grad_logits);
    GGA_CHECK_LAUNCH("focal_bwd_kernel");
    return GGA_OK;
}
