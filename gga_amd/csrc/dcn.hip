// Modulated deformable convolution (DCNv2) for gfx950: the sampling half.
//
// Reference call site: the last tower convolution of the PGD / FCOS3D heads
// (mmdet3d/models/dense_heads/anchor_free_mono3d_head.py:187-211, `dcn_on_last_conv=True` in
// configs/_base_/models/pgd.py:47 -> mmcv `ModulatedDeformConv2dPack`, third-party: the op is
// restated from its published definition, Zhu et al. "Deformable ConvNets v2"):
//
//   col[p, k, c] = mask[p, k] * bilinear(x[.., c], p * stride - pad + k * dil + offset[p, k])
//   y[p, :]      = W[:, k, c] . col[p, k, c] + bias
//
// The GEMM half is a plain library GEMM ([pixels, K*C] x [K*C, Cout], hipBLASLt through the host
// framework); what is hand-written is the data movement around it:
//
//   dcn_im2col_kernel   channels-last x [B, H, W, C]: one wavefront per output pixel walks the K taps;
//                       the four neighbours of a sampling point are wave-uniform addresses, every lane
//                       loads its C/64 channels of each as one float4 (1 KB coalesced rows), blends and
//                       writes the column row - no LDS, no divergence.
//   dcn_col2im_kernel   backward of the sampling: per (pixel, tap) the lane's channels of grad_col give
//                       (a) four float4 atomic adds into grad_x (the scatter is data dependent, so it
//                       cannot be turned into a gather), (b) the partial sums of grad_offset (h, w) and
//                       grad_mask, closed with wavefront shuffles.
#include "gga_common.h"

struct DcnGeom {
    int B, H, W, C, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
};

__device__ __forceinline__ float4 f4_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_fma(float a, float4 v, float4 acc) {
    return make_float4(acc.x + a * v.x, acc.y + a * v.y, acc.z + a * v.z, acc.w + a * v.w);
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// sampling geometry of one (pixel, tap): corner validity and bilinear weights (mmcv dmcn_im2col_bilinear)
struct DcnTap {
    int hl, wl;                 // low corner
    float lh, lw;               // fractional parts
    bool inside, v1, v2, v3, v4;
};
__device__ __forceinline__ DcnTap dcn_tap(const DcnGeom& g, int ho, int wo, int i, int j, float off_h, float off_w) {
    DcnTap t;
    const float h = (float)(ho * g.sh - g.ph + i * g.dh) + off_h;
    const float w = (float)(wo * g.sw - g.pw + j * g.dw) + off_w;
    t.inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
    const float hf = floorf(h), wf = floorf(w);
    t.hl = (int)hf; t.wl = (int)wf;
    t.lh = h - hf; t.lw = w - wf;
    const int hh = t.hl + 1, wh = t.wl + 1;
    t.v1 = t.inside && t.hl >= 0 && t.wl >= 0;
    t.v2 = t.inside && t.hl >= 0 && wh <= g.W - 1;
    t.v3 = t.inside && hh <= g.H - 1 && t.wl >= 0;
    t.v4 = t.inside && hh <= g.H - 1 && wh <= g.W - 1;
    return t;
}

// grid: one wavefront per output pixel (4 per 256-thread workgroup); VPL = float4 pieces per lane (C = 256 * VPL)
template <int VPL>
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                        const float* __restrict__ mask, DcnGeom g,
                                                        float* __restrict__ col) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t npix = (int64_t)g.B * g.Ho * g.Wo;
    if (p >= npix) return;
    const int b = (int)(p / ((int64_t)g.Ho * g.Wo));
    const int rem = (int)(p - (int64_t)b * g.Ho * g.Wo);
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    const int K = g.kh * g.kw;
    const int64_t plane = (int64_t)g.Ho * g.Wo;
    const float* xb = x + (int64_t)b * g.H * g.W * g.C;
    const float* ob = offset + (int64_t)b * 2 * K * plane + rem;
    const float* mb = mask + (int64_t)b * K * plane + rem;
    float* cp = col + p * (int64_t)K * g.C;
    for (int k = 0; k < K; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        const float off_h = ob[(int64_t)(2 * k) * plane], off_w = ob[(int64_t)(2 * k + 1) * plane];
        const float m = mb[(int64_t)k * plane];
        const DcnTap t = dcn_tap(g, ho, wo, i, j, off_h, off_w);
        const float w1 = (1.f - t.lh) * (1.f - t.lw), w2 = (1.f - t.lh) * t.lw, w3 = t.lh * (1.f - t.lw), w4 = t.lh * t.lw;
        const float* r1 = xb + ((int64_t)t.hl * g.W + t.wl) * g.C;
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const int c = (lane + 64 * e) * 4;
            float4 acc = f4_zero();
            if (t.v1) acc = f4_fma(w1, f4_ld(r1 + c), acc);
            if (t.v2) acc = f4_fma(w2, f4_ld(r1 + g.C + c), acc);
            if (t.v3) acc = f4_fma(w3, f4_ld(r1 + (int64_t)g.W * g.C + c), acc);
            if (t.v4) acc = f4_fma(w4, f4_ld(r1 + (int64_t)(g.W + 1) * g.C + c), acc);
            *reinterpret_cast<float4*>(cp + (int64_t)k * g.C + c) = make_float4(acc.x * m, acc.y * m, acc.z * m, acc.w * m);
        }
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                        const float* __restrict__ mask, const float* __restrict__ gcol,
                                                        DcnGeom g, float* __restrict__ gx, float* __restrict__ goffset,
                                                        float* __restrict__ gmask) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t npix = (int64_t)g.B * g.Ho * g.Wo;
    if (p >= npix) return;
    const int b = (int)(p / ((int64_t)g.Ho * g.Wo));
    const int rem = (int)(p - (int64_t)b * g.Ho * g.Wo);
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    const int K = g.kh * g.kw;
    const int64_t plane = (int64_t)g.Ho * g.Wo;
    const float* xb = x + (int64_t)b * g.H * g.W * g.C;
    float* gxb = gx ? gx + (int64_t)b * g.H * g.W * g.C : nullptr;
    const float* ob = offset + (int64_t)b * 2 * K * plane + rem;
    const float* mb = mask + (int64_t)b * K * plane + rem;
    const float* gp = gcol + p * (int64_t)K * g.C;
    for (int k = 0; k < K; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        const float off_h = ob[(int64_t)(2 * k) * plane], off_w = ob[(int64_t)(2 * k + 1) * plane];
        const float m = mb[(int64_t)k * plane];
        const DcnTap t = dcn_tap(g, ho, wo, i, j, off_h, off_w);
        const float hh = 1.f - t.lh, hw = 1.f - t.lw;
        const float w1 = hh * hw, w2 = hh * t.lw, w3 = t.lh * hw, w4 = t.lh * t.lw;
        const int64_t o1 = ((int64_t)t.hl * g.W + t.wl) * g.C;
        float s_val = 0.f, s_dh = 0.f, s_dw = 0.f;            // sum_c gcol * {sample, d sample / dh, d sample / dw}
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const int c = (lane + 64 * e) * 4;
            const float4 gc = f4_ld(gp + (int64_t)k * g.C + c);
            const float4 a1 = t.v1 ? f4_ld(xb + o1 + c) : f4_zero();
            const float4 a2 = t.v2 ? f4_ld(xb + o1 + g.C + c) : f4_zero();
            const float4 a3 = t.v3 ? f4_ld(xb + o1 + (int64_t)g.W * g.C + c) : f4_zero();
            const float4 a4 = t.v4 ? f4_ld(xb + o1 + (int64_t)(g.W + 1) * g.C + c) : f4_zero();
            const float d1 = f4_dot(gc, a1), d2 = f4_dot(gc, a2), d3 = f4_dot(gc, a3), d4 = f4_dot(gc, a4);
            s_val += w1 * d1 + w2 * d2 + w3 * d3 + w4 * d4;
            s_dh += hw * (d3 - d1) + t.lw * (d4 - d2);        // d/dh: -hw v1 - lw v2 + hw v3 + lw v4
            s_dw += hh * (d2 - d1) + t.lh * (d4 - d3);        // d/dw: -hh v1 + hh v2 - lh v3 + lh v4
            if (gxb) {
                const float4 gm = make_float4(gc.x * m, gc.y * m, gc.z * m, gc.w * m);
#define DCN_ADD(OK, OFF, WGT) if (OK) { float* d = gxb + (OFF) + c; atomicAdd(d, (WGT) * gm.x); atomicAdd(d + 1, (WGT) * gm.y); \
                                        atomicAdd(d + 2, (WGT) * gm.z); atomicAdd(d + 3, (WGT) * gm.w); }
                DCN_ADD(t.v1, o1, w1) DCN_ADD(t.v2, o1 + g.C, w2) DCN_ADD(t.v3, o1 + (int64_t)g.W * g.C, w3)
                DCN_ADD(t.v4, o1 + (int64_t)(g.W + 1) * g.C, w4)
#undef DCN_ADD
            }
        }
        s_val = wave_sum(s_val); s_dh = wave_sum(s_dh); s_dw = wave_sum(s_dw);
        if (lane == 0) {
            gmask[(int64_t)b * K * plane + (int64_t)k * plane + rem] = s_val;
            goffset[(int64_t)b * 2 * K * plane + (int64_t)(2 * k) * plane + rem] = t.inside ? s_dh * m : 0.f;
            goffset[(int64_t)b * 2 * K * plane + (int64_t)(2 * k + 1) * plane + rem] = t.inside ? s_dw * m : 0.f;
        }
    }
}

static int dcn_geom(const char* fn, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                    int dw, DcnGeom* g) {
    GGA_REQUIRE(B >= 1 && H >= 1 && W >= 1 && kh >= 1 && kw >= 1 && sh >= 1 && sw >= 1 && dh >= 1 && dw >= 1 && ph >= 0 &&
                    pw >= 0, "%s: bad geometry", fn);
    GGA_REQUIRE(C >= 256 && C % 256 == 0 && C <= 1024, "%s: channels (%d) must be 256, 512, 768 or 1024", fn, C);
    g->B = B; g->H = H; g->W = W; g->C = C; g->kh = kh; g->kw = kw; g->sh = sh; g->sw = sw; g->ph = ph; g->pw = pw;
    g->dh = dh; g->dw = dw;
    g->Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g->Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    GGA_REQUIRE(g->Ho >= 1 && g->Wo >= 1, "%s: empty output", fn);
    return GGA_OK;
}

extern "C" int gga_dcn_im2col(const float* x, const float* offset, const float* mask, int B, int H, int W, int C, int kh,
                              int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, float* col,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && offset && mask && col, "gga_dcn_im2col: null pointer argument");
    DcnGeom g;
    if (int rc = dcn_geom("gga_dcn_im2col", B, H, W, C, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, &g)) return rc;
    const int64_t npix = (int64_t)B * g.Ho * g.Wo;
    const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL(dcn_im2col_kernel<1>, grid, block, 0, stream, x, offset, mask, g, col); break;
        case 2: hipLaunchKernelGGL(dcn_im2col_kernel<2>, grid, block, 0, stream, x, offset, mask, g, col); break;
        case 3: hipLaunchKernelGGL(dcn_im2col_kernel<3>, grid, block, 0, stream, x, offset, mask, g, col); break;
        default: hipLaunchKernelGGL(dcn_im2col_kernel<4>, grid, block, 0, stream, x, offset, mask, g, col); break;
    }
    GGA_CHECK_LAUNCH("dcn_im2col_kernel");
    return GGA_OK;
}

extern "C" int gga_dcn_col2im(const float* x, const float* offset, const float* mask, const float* grad_col, int B, int H,
                              int W, int C, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                              int dil_w, float* grad_x, float* grad_offset, float* grad_mask, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && offset && mask && grad_col && grad_offset && grad_mask, "gga_dcn_col2im: null pointer argument");
    DcnGeom g;
    if (int rc = dcn_geom("gga_dcn_col2im", B, H, W, C, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, &g)) return rc;
    if (grad_x) GGA_CHECK_HIP(hipMemsetAsync(grad_x, 0, (size_t)B * H * W * C * sizeof(float), stream), "dcn memset");
    const int64_t npix = (int64_t)B * g.Ho * g.Wo;
    const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL(dcn_col2im_kernel<1>, grid, block, 0, stream, x, offset, mask, grad_col, g, grad_x, grad_offset, grad_mask); break;
        case 2: hipLaunchKernelGGL(dcn_col2im_kernel<2>, grid, block, 0, stream, x, offset, mask, grad_col, g, grad_x, grad_offset, grad_mask); break;
        case 3: hipLaunchKernelGGL(dcn_col2im_kernel<3>, grid, block, 0, stream, x, offset, mask, grad_col, g, grad_x, grad_offset, grad_mask); break;
        default: hipLaunchKernelGGL(dcn_col2im_kernel<4>, grid, block, 0, stream, x, offset, mask, grad_col, g, grad_x, grad_offset, grad_mask); break;
    }
    GGA_CHECK_LAUNCH("dcn_col2im_kernel");
    return GGA_OK;
}
