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

// Backward of the sampling. The scatter into grad_x is data dependent (a sample's four corners are wherever
// its offset points), so it has to be an atomic accumulation - but not one global atomic per contribution:
// a 256-thread workgroup owns an 8 x 8 tile of output pixels and 64 channels at a time and accumulates into
// an LDS window of the input plane that covers the tile's receptive field plus DCN_R pixels of offset slack
// (lane = channel: conflict-free LDS atomics); the window is flushed once with coalesced global atomics, and
// only samples whose corners leave the window go to global memory directly. At 12 x 96 x 312 pixels x 256
// channels that is 16 k global atomics per tile and chunk instead of 147 k (first version: one thread-level
// global atomic per contribution, 14 ms per call; this one: see DESIGN.md).
// grad_offset / grad_mask: per (pixel, tap) sums over the channels, accumulated over the four channel chunks
// in LDS and written once.
// Measured in round 2 at 12 x 96 x 312 x 256 (tools_dev/bench_dcn.py, random offsets of std 0.5): 17.4 ms per call, of
// which 1.7 ms are everything but grad_x (loads, geometry, the three reductions). Three rewrites of the grad_x half left
// the time where it was: four pairs per wave with float4 lanes (a quarter of the load / geometry / reduction
// instructions), a 72-float window stride with the channels interleaved (no LDS bank conflicts among a wave's pairs), and
// writing the windows to a scratch image that a second kernel sums per input pixel (no global atomics at all). What is
// left in common is the 3.3 G LDS float atomics themselves (64 channels x 4 corners per pair and chunk).
#define DCN_TH 8
#define DCN_TW 8
#define DCN_R 2
#define DCN_CCH 64
#define DCN_U 4                  // (pixel, tap) pairs in flight per wave
#define DCN_MAXWIN 400           // window pixels held in LDS (x 64 channels x 4 B = 100 KB at most; 3x3/s1/d1: 16 x 16)

__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                        const float* __restrict__ mask, const float* __restrict__ gcol,
                                                        DcnGeom g, int tiles_x, int tiles_y, int win_h, int win_w,
                                                        float* __restrict__ gx, float* __restrict__ goffset,
                                                        float* __restrict__ gmask) {
    extern __shared__ __attribute__((aligned(16))) float dcn_lds[];
    const int K = g.kh * g.kw, NP = DCN_TH * DCN_TW * K;       // (pixel, tap) pairs of the tile
    float* win = dcn_lds;                                       // [win_h * win_w][64]
    float* sums = dcn_lds + (size_t)win_h * win_w * DCN_CCH;    // [NP][3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / (tiles_x * tiles_y);
    const int t = blockIdx.x - b * (tiles_x * tiles_y);
    const int ty0 = (t / tiles_x) * DCN_TH, tx0 = (t % tiles_x) * DCN_TW;
    const int wy0 = ty0 * g.sh - g.ph - DCN_R, wx0 = tx0 * g.sw - g.pw - DCN_R;    // window origin in the input plane
    const int64_t plane = (int64_t)g.Ho * g.Wo;
    const float* xb = x + (int64_t)b * g.H * g.W * g.C;
    float* gxb = gx ? gx + (int64_t)b * g.H * g.W * g.C : nullptr;
    float* om = sums + NP * 3;                                  // [NP][3]: offset_h, offset_w, mask of every pair (-inf offset: no pair)
    for (int pr = tid; pr < NP; pr += 256) {
        const int pl = pr / K, k = pr - pl * K;
        const int ho = ty0 + pl / DCN_TW, wo = tx0 + pl % DCN_TW;
        const bool ok = ho < g.Ho && wo < g.Wo;
        const int rem = ok ? ho * g.Wo + wo : 0;
        sums[pr * 3] = sums[pr * 3 + 1] = sums[pr * 3 + 2] = 0.f;
        om[pr * 3] = ok ? offset[((int64_t)b * 2 * K + 2 * k) * plane + rem] : -1e30f;       // far outside: tap.inside = false
        om[pr * 3 + 1] = ok ? offset[((int64_t)b * 2 * K + 2 * k + 1) * plane + rem] : -1e30f;
        om[pr * 3 + 2] = ok ? mask[((int64_t)b * K + k) * plane + rem] : 0.f;
    }
    for (int c0 = 0; c0 < g.C; c0 += DCN_CCH) {
        for (int i = tid; i < win_h * win_w * DCN_CCH; i += 256) win[i] = 0.f;
        __syncthreads();
        const int c = c0 + lane;
        // the gathers are a dependent chain (offset -> corner addresses -> loads): four pairs are in flight per wave
        for (int pr0 = wave; pr0 < NP; pr0 += 4 * DCN_U) {
            DcnTap tp[DCN_U];
            float gc[DCN_U], a1[DCN_U], a2[DCN_U], a3[DCN_U], a4[DCN_U], m[DCN_U];
            int64_t o1[DCN_U];
#pragma unroll
            for (int u = 0; u < DCN_U; ++u) {
                const int pr = pr0 + 4 * u;
                const bool live = pr < NP;
                const int prc = live ? pr : 0;
                const int pl = prc / K, k = prc - pl * K;
                const int ho = ty0 + pl / DCN_TW, wo = tx0 + pl % DCN_TW;
                tp[u] = dcn_tap(g, ho, wo, k / g.kw, k % g.kw, live ? om[prc * 3] : -1e30f, live ? om[prc * 3 + 1] : -1e30f);
                m[u] = om[prc * 3 + 2];
                const int rem = tp[u].inside ? ho * g.Wo + wo : 0;
                gc[u] = tp[u].inside ? gcol[((int64_t)b * plane + rem) * ((int64_t)K * g.C) + (int64_t)k * g.C + c] : 0.f;
                o1[u] = ((int64_t)tp[u].hl * g.W + tp[u].wl) * g.C + c;
                a1[u] = tp[u].v1 ? xb[o1[u]] : 0.f;
                a2[u] = tp[u].v2 ? xb[o1[u] + g.C] : 0.f;
                a3[u] = tp[u].v3 ? xb[o1[u] + (int64_t)g.W * g.C] : 0.f;
                a4[u] = tp[u].v4 ? xb[o1[u] + (int64_t)(g.W + 1) * g.C] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < DCN_U; ++u) {
                if (!tp[u].inside) continue;                     // wave-uniform
                const int pr = pr0 + 4 * u;
                const float hh = 1.f - tp[u].lh, hw = 1.f - tp[u].lw;
                const float w1 = hh * hw, w2 = hh * tp[u].lw, w3 = tp[u].lh * hw, w4 = tp[u].lh * tp[u].lw;
                float s_val = gc[u] * (w1 * a1[u] + w2 * a2[u] + w3 * a3[u] + w4 * a4[u]);
                float s_dh = gc[u] * (hw * (a3[u] - a1[u]) + tp[u].lw * (a4[u] - a2[u]));
                float s_dw = gc[u] * (hh * (a2[u] - a1[u]) + tp[u].lh * (a4[u] - a3[u]));
                s_val = wave_sum(s_val); s_dh = wave_sum(s_dh); s_dw = wave_sum(s_dw);
                if (lane == 0) { sums[pr * 3] += s_val; sums[pr * 3 + 1] += s_dh * m[u]; sums[pr * 3 + 2] += s_dw * m[u]; }   // one wave per pair
                if (gxb) {
                    const float gm = gc[u] * m[u];
                    const int ly = tp[u].hl - wy0, lx = tp[u].wl - wx0;             // window coordinates of the low corner
                    const bool in_win = ly >= 0 && lx >= 0 && ly + 1 < win_h && lx + 1 < win_w;
                    if (in_win) {
                        float* wp = win + ((size_t)ly * win_w + lx) * DCN_CCH + lane;
                        if (tp[u].v1) atomicAdd(wp, w1 * gm);
                        if (tp[u].v2) atomicAdd(wp + DCN_CCH, w2 * gm);
                        if (tp[u].v3) atomicAdd(wp + (size_t)win_w * DCN_CCH, w3 * gm);
                        if (tp[u].v4) atomicAdd(wp + (size_t)(win_w + 1) * DCN_CCH, w4 * gm);
                    } else {                                                     // far offset: straight to memory
                        if (tp[u].v1) atomicAdd(gxb + o1[u], w1 * gm);
                        if (tp[u].v2) atomicAdd(gxb + o1[u] + g.C, w2 * gm);
                        if (tp[u].v3) atomicAdd(gxb + o1[u] + (int64_t)g.W * g.C, w3 * gm);
                        if (tp[u].v4) atomicAdd(gxb + o1[u] + (int64_t)(g.W + 1) * g.C, w4 * gm);
                    }
                }
            }
        }
        __syncthreads();
        if (gxb) {      // flush the window: row of 64 channels per wave access
            for (int i = tid; i < win_h * win_w * DCN_CCH; i += 256) {
                const float v = win[i];
                if (v == 0.f) continue;
                const int px = i / DCN_CCH, c = i - px * DCN_CCH;
                const int iy = wy0 + px / win_w, ix = wx0 + px % win_w;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) atomicAdd(gxb + ((int64_t)iy * g.W + ix) * g.C + c0 + c, v);
            }
        }
        __syncthreads();
    }
    for (int pr = tid; pr < NP; pr += 256) {
        const int pl = pr / K, k = pr - pl * K;
        const int ho = ty0 + pl / DCN_TW, wo = tx0 + pl % DCN_TW;
        if (ho >= g.Ho || wo >= g.Wo) continue;
        const int rem = ho * g.Wo + wo;
        gmask[((int64_t)b * K + k) * plane + rem] = sums[pr * 3];
        goffset[((int64_t)b * 2 * K + 2 * k) * plane + rem] = sums[pr * 3 + 1];
        goffset[((int64_t)b * 2 * K + 2 * k + 1) * plane + rem] = sums[pr * 3 + 2];
    }
}

static int dcn_geom(const char* fn, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                    int dw, DcnGeom* g) {
    GGA_REQUIRE(B >= 1 && H >= 1 && W >= 1 && kh >= 1 && kw >= 1 && sh >= 1 && sw >= 1 && dh >= 1 && dw >= 1 && ph >= 0 &&
                    pw >= 0, "%s: bad geometry", fn);
    GGA_REQUIRE(C >= 256 && C % 256 == 0 && C <= 1024, "%s: channels (%d) must be 256, 512, 768 or 1024", fn, C);      // (col2im needs C % 64 == 0)
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
    const int tiles_x = (g.Wo + DCN_TW - 1) / DCN_TW, tiles_y = (g.Ho + DCN_TH - 1) / DCN_TH;
    // LDS window: receptive field of the tile + offset slack + the bilinear (+1) corner
    int win_h = (DCN_TH - 1) * stride_h + (kh - 1) * dil_h + 2 + 2 * DCN_R;
    int win_w = (DCN_TW - 1) * stride_w + (kw - 1) * dil_w + 2 + 2 * DCN_R;
    while (win_h * win_w > DCN_MAXWIN) { if (win_h > 4) win_h -= 1; if (win_w > 4 && win_h * win_w > DCN_MAXWIN) win_w -= 1; }   // smaller window: more direct atomics, same result
    const size_t lds = ((size_t)win_h * win_w * DCN_CCH + (size_t)DCN_TH * DCN_TW * kh * kw * 6) * sizeof(float);
    GGA_REQUIRE(lds <= 160 * 1024 && kh * kw <= 49, "gga_dcn_col2im: kernel %dx%d too large", kh, kw);
    GGA_CHECK_HIP(hipFuncSetAttribute((const void*)dcn_col2im_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                  "dcn col2im LDS size");
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3((unsigned)(B * tiles_x * tiles_y)), dim3(256), lds, stream, x, offset, mask, grad_col, g,
                       tiles_x, tiles_y, win_h, win_w, grad_x, grad_offset, grad_mask);
    GGA_CHECK_LAUNCH("dcn_col2im_kernel");
    return GGA_OK;
}
