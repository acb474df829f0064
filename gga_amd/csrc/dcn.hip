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
//   dcn_col2im_kernel   backward of the sampling: per (pixel, tap) the lane's channel of grad_col gives
//                       (a) four additions into grad_x, accumulated in wave-private LDS windows (the scatter
//                       is data dependent, so it cannot be turned into a gather), (b) the partial sums of
//                       grad_offset (h, w) and grad_mask, closed with wavefront shuffles.
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
                                                        float* __restrict__ col, uint32_t* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t npix = (int64_t)g.B * g.Ho * g.Wo;
    uint32_t am = 0;                                     // largest finite |col| written (for the consumer's fp16 scale)
    if (p < npix) {
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
            const float4 o = make_float4(acc.x * m, acc.y * m, acc.z * m, acc.w * m);
            *reinterpret_cast<float4*>(cp + (int64_t)k * g.C + c) = o;
            if (amax) { am = gga_amax_of(o.x, am); am = gga_amax_of(o.y, am); am = gga_amax_of(o.z, am); am = gga_amax_of(o.w, am); }
        }
    }
    }
    if (amax) gga_amax_commit(am, amax);
}

// Backward of the sampling. The scatter into grad_x is data dependent (a sample's four corners are wherever
// its offset points), so it is an accumulation - but LDS float atomics are the wrong tool for it on this part:
// ds_add_f32 retires about one LANE every three cycles (tools_dev/micro/lds_atomic.hip: 193 cycles per 64-lane add
// and CU, against 6.4 for a read-add-write by the wave that owns the addresses). The first versions of this kernel
// accumulated 64-channel windows shared by the workgroup's four waves with such atomics (17.4 ms per call at
// 12 x 96 x 312 x 256, 15.7 ms of it the atomics; three rewrites that kept them changed nothing).
// Now every wave OWNS what it accumulates into: a 256-thread workgroup takes a 4 x 8 tile of output pixels, wave w the
// channels 64 w .. 64 w + 63 of the current group of 256 (lane = channel), and walks all (pixel, tap) pairs of the
// tile; its private LDS window [window pixel][64] covers the tile's receptive field plus DCN_R pixels of offset slack
// and is updated with plain loads and stores (pairs in order, so two samples that hit the same cell cannot race). A
// window is flushed once with coalesced global atomics; only samples whose corners leave it go to memory directly.
// grad_offset / grad_mask: per (pixel, tap) sums over the channels - wave shuffles, then per-wave partial sums in LDS
// that are added at the end.
#define DCN_TH 4
#define DCN_TW 8
#define DCN_R 1
#define DCN_U 16                 // (pixel, tap) pairs in flight per wave: one wave per SIMD, so the memory latency is covered by depth
#define DCN_MAXWIN 128           // window pixels (x 256 channels x 4 B = 128 KB at most; 3x3/s1/d1: 9 x 13)

// grad_offset / grad_mask: one wavefront per output pixel walks the K taps like dcn_im2col_kernel does - every lane
// its C/64 channels as float4 - and closes the three sums over the channels with shuffles. No LDS, no atomics.
template <int VPL>
__global__ __launch_bounds__(256) void dcn_col2im_sums_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                             const float* __restrict__ mask, const float* __restrict__ gcol,
                                                             DcnGeom g, float* __restrict__ goffset, float* __restrict__ gmask) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t npix = (int64_t)g.B * g.Ho * g.Wo;
    if (p >= npix) return;
    const int64_t plane = (int64_t)g.Ho * g.Wo;
    const int b = (int)(p / plane);
    const int rem = (int)(p - (int64_t)b * plane);
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    const int K = g.kh * g.kw;
    const float* xb = x + (int64_t)b * g.H * g.W * g.C;
    const float* ob = offset + (int64_t)b * 2 * K * plane + rem;
    const float* mb = mask + (int64_t)b * K * plane + rem;
    const float* gp = gcol + p * (int64_t)K * g.C;
    for (int k = 0; k < K; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        const float off_h = ob[(int64_t)(2 * k) * plane], off_w = ob[(int64_t)(2 * k + 1) * plane];
        const float m = mb[(int64_t)k * plane];
        const DcnTap t = dcn_tap(g, ho, wo, i, j, off_h, off_w);
        float s_val = 0.f, s_dh = 0.f, s_dw = 0.f;
        if (t.inside) {                                          // wave-uniform
            const float hh = 1.f - t.lh, hw = 1.f - t.lw;
            const float w1 = hh * hw, w2 = hh * t.lw, w3 = t.lh * hw, w4 = t.lh * t.lw;
            const float* r1 = xb + ((int64_t)t.hl * g.W + t.wl) * g.C;
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                const int c = (lane + 64 * e) * 4;
                const float4 gc = f4_ld(gp + (int64_t)k * g.C + c);
                const float4 a1 = t.v1 ? f4_ld(r1 + c) : f4_zero();
                const float4 a2 = t.v2 ? f4_ld(r1 + g.C + c) : f4_zero();
                const float4 a3 = t.v3 ? f4_ld(r1 + (int64_t)g.W * g.C + c) : f4_zero();
                const float4 a4 = t.v4 ? f4_ld(r1 + (int64_t)(g.W + 1) * g.C + c) : f4_zero();
                s_val += f4_dot(gc, f4_fma(w1, a1, f4_fma(w2, a2, f4_fma(w3, a3, f4_fma(w4, a4, f4_zero())))));
                const float4 d31 = make_float4(a3.x - a1.x, a3.y - a1.y, a3.z - a1.z, a3.w - a1.w);
                const float4 d42 = make_float4(a4.x - a2.x, a4.y - a2.y, a4.z - a2.z, a4.w - a2.w);
                const float4 d21 = make_float4(a2.x - a1.x, a2.y - a1.y, a2.z - a1.z, a2.w - a1.w);
                const float4 d43 = make_float4(a4.x - a3.x, a4.y - a3.y, a4.z - a3.z, a4.w - a3.w);
                s_dh += f4_dot(gc, f4_fma(hw, d31, f4_fma(t.lw, d42, f4_zero())));
                s_dw += f4_dot(gc, f4_fma(hh, d21, f4_fma(t.lh, d43, f4_zero())));
            }
            s_val = wave_sum(s_val); s_dh = wave_sum(s_dh); s_dw = wave_sum(s_dw);
        }
        if (lane == 0) {
            gmask[((int64_t)b * K + k) * plane + rem] = s_val;
            goffset[((int64_t)b * 2 * K + 2 * k) * plane + rem] = s_dh * m;
            goffset[((int64_t)b * 2 * K + 2 * k + 1) * plane + rem] = s_dw * m;
        }
    }
}

// grad_x (see above): wave-private LDS windows, plain read-add-write. The geometry of a (pixel, tap) pair - window
// cell, the four bilinear weights times the mask, the row of grad_col - is the same for every channel: the workgroup
// computes it once into LDS, and the waves' loops over the pairs are a record read, one global load and the window
// update (with the geometry inside the loop a wave spent ~1700 cycles per pair, alone on its SIMD: 11.9 ms per call).
struct DcnPair {
    int row;            // (b * Ho * Wo + pixel) * K + tap: row of grad_col, -1: no contribution
    int cell;           // window cell of the low corner, -1: the corners leave the window (global atomics)
    float w1, w2, w3, w4;   // bilinear weight * mask, 0 for a corner outside the image
    int hl, wl;         // low corner in the input plane
};

__global__ __launch_bounds__(256, 1) void dcn_col2im_kernel(const float* __restrict__ offset, const float* __restrict__ mask,
                                                           const float* __restrict__ gcol, DcnGeom g, int tiles_x, int tiles_y,
                                                           int win_h, int win_w, float* __restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) float dcn_lds[];
    const int K = g.kh * g.kw, NP = DCN_TH * DCN_TW * K;       // (pixel, tap) pairs of the tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wpx = win_h * win_w;
    float* win = dcn_lds + (size_t)wave * wpx * 64;             // this wave's window [wpx][64]
    DcnPair* pairs = reinterpret_cast<DcnPair*>(dcn_lds + (size_t)4 * wpx * 64);
    const int b = blockIdx.x / (tiles_x * tiles_y);
    const int t = blockIdx.x - b * (tiles_x * tiles_y);
    const int ty0 = (t / tiles_x) * DCN_TH, tx0 = (t % tiles_x) * DCN_TW;
    const int wy0 = ty0 * g.sh - g.ph - DCN_R, wx0 = tx0 * g.sw - g.pw - DCN_R;    // window origin in the input plane
    const int64_t plane = (int64_t)g.Ho * g.Wo;
    float* gxb = gx + (int64_t)b * g.H * g.W * g.C;
    for (int pr = tid; pr < NP; pr += 256) {
        const int pl = pr / K, k = pr - pl * K;
        const int ho = ty0 + pl / DCN_TW, wo = tx0 + pl % DCN_TW;
        DcnPair q;
        q.row = -1; q.cell = -1; q.w1 = q.w2 = q.w3 = q.w4 = 0.f; q.hl = q.wl = 0;
        if (ho < g.Ho && wo < g.Wo) {
            const int rem = ho * g.Wo + wo;
            const float off_h = offset[((int64_t)b * 2 * K + 2 * k) * plane + rem];
            const float off_w = offset[((int64_t)b * 2 * K + 2 * k + 1) * plane + rem];
            const float m = mask[((int64_t)b * K + k) * plane + rem];
            const DcnTap tp = dcn_tap(g, ho, wo, k / g.kw, k % g.kw, off_h, off_w);
            if (tp.inside) {
                const float hh = 1.f - tp.lh, hw = 1.f - tp.lw;
                q.row = (int)(((int64_t)b * plane + rem) * K + k);
                q.w1 = tp.v1 ? hh * hw * m : 0.f; q.w2 = tp.v2 ? hh * tp.lw * m : 0.f;
                q.w3 = tp.v3 ? tp.lh * hw * m : 0.f; q.w4 = tp.v4 ? tp.lh * tp.lw * m : 0.f;
                q.hl = tp.hl; q.wl = tp.wl;
                const int ly = tp.hl - wy0, lx = tp.wl - wx0;                   // window coordinates of the low corner
                if (ly >= 0 && lx >= 0 && ly + 1 < win_h && lx + 1 < win_w) q.cell = ly * win_w + lx;
            }
        }
        pairs[pr] = q;
    }
    __syncthreads();
    for (int c0 = 0; c0 < g.C; c0 += 256) {
        const int c = c0 + wave * 64 + lane;
        for (int i = 0; i < wpx; ++i) win[i * 64 + lane] = 0.f;
        for (int pr0 = 0; pr0 < NP; pr0 += DCN_U) {
            // phase 1: request the gradient rows of DCN_U pairs; phase 2: the pairs in order (two samples that hit the same
            // cell must not race)
            float gc[DCN_U];
            DcnPair q[DCN_U];                                                    // records too: their LDS reads would otherwise wait
#pragma unroll                                                                   // behind the window stores of the pair before
            for (int u = 0; u < DCN_U; ++u) {
                q[u] = pairs[pr0 + u < NP ? pr0 + u : 0];
                if (pr0 + u >= NP) q[u].row = -1;
                gc[u] = q[u].row >= 0 ? gcol[(int64_t)q[u].row * g.C + c] : 0.f;
            }
            // all loads are requested here (otherwise the compiler sinks every load next to its use)
#pragma unroll
            for (int u = 0; u < DCN_U; ++u) asm volatile("" : "+v"(gc[u]));
#pragma unroll
            for (int u = 0; u < DCN_U; ++u) {
                if (q[u].row < 0) continue;                                      // wave-uniform
                if (q[u].cell >= 0) {                                            // mine alone: load, add, store
                    float* wp = win + (size_t)q[u].cell * 64 + lane;
                    const float q1 = wp[0], q2 = wp[64], q3 = wp[(size_t)win_w * 64], q4 = wp[(size_t)(win_w + 1) * 64];
                    wp[0] = fmaf(q[u].w1, gc[u], q1);
                    wp[64] = fmaf(q[u].w2, gc[u], q2);
                    wp[(size_t)win_w * 64] = fmaf(q[u].w3, gc[u], q3);
                    wp[(size_t)(win_w + 1) * 64] = fmaf(q[u].w4, gc[u], q4);
                } else {                                                         // far offset: straight to memory
                    float* o1 = gxb + ((int64_t)q[u].hl * g.W + q[u].wl) * g.C + c;
                    if (q[u].w1 != 0.f) atomicAdd(o1, q[u].w1 * gc[u]);
                    if (q[u].w2 != 0.f) atomicAdd(o1 + g.C, q[u].w2 * gc[u]);
                    if (q[u].w3 != 0.f) atomicAdd(o1 + (int64_t)g.W * g.C, q[u].w3 * gc[u]);
                    if (q[u].w4 != 0.f) atomicAdd(o1 + (int64_t)(g.W + 1) * g.C, q[u].w4 * gc[u]);
                }
            }
        }
        // flush my window: one 256-byte row of atomics per window pixel
        for (int px = 0; px < wpx; ++px) {
            const float v = win[px * 64 + lane];
            const int iy = wy0 + px / win_w, ix = wx0 + px % win_w;
            if (v != 0.f && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) atomicAdd(gxb + ((int64_t)iy * g.W + ix) * g.C + c, v);
        }
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
    return gga_dcn_im2col_amax(x, offset, mask, B, H, W, C, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, col, nullptr,
                               stream_);
}

extern "C" int gga_dcn_im2col_amax(const float* x, const float* offset, const float* mask, int B, int H, int W, int C, int kh,
                                   int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, float* col,
                                   uint32_t* amax_col, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && offset && mask && col, "gga_dcn_im2col: null pointer argument");
    DcnGeom g;
    if (int rc = dcn_geom("gga_dcn_im2col", B, H, W, C, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, &g)) return rc;
    const int64_t npix = (int64_t)B * g.Ho * g.Wo;
    const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL(dcn_im2col_kernel<1>, grid, block, 0, stream, x, offset, mask, g, col, amax_col); break;
        case 2: hipLaunchKernelGGL(dcn_im2col_kernel<2>, grid, block, 0, stream, x, offset, mask, g, col, amax_col); break;
        case 3: hipLaunchKernelGGL(dcn_im2col_kernel<3>, grid, block, 0, stream, x, offset, mask, g, col, amax_col); break;
        default: hipLaunchKernelGGL(dcn_im2col_kernel<4>, grid, block, 0, stream, x, offset, mask, g, col, amax_col); break;
    }
    GGA_CHECK_LAUNCH("dcn_im2col_kernel");
    return GGA_OK;
}

extern "C" int gga_dcn_col2im(const float* x, const float* offset, const float* mask, const float* grad_col, int B, int H,
                              int W, int C, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                              int dil_w, float* grad_x, float* grad_offset, float* grad_mask, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(x && offset && mask && grad_col && grad_offset && grad_mask, "gga_dcn_col2im: null pointer argument");
    GGA_REQUIRE((int64_t)B * H * W * kh * kw < 2147483647ll, "gga_dcn_col2im: too many (pixel, tap) pairs");
    DcnGeom g;
    if (int rc = dcn_geom("gga_dcn_col2im", B, H, W, C, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, &g)) return rc;
    const int64_t npix = (int64_t)B * g.Ho * g.Wo;
    const dim3 sgrid((unsigned)((npix + 3) / 4)), block(256);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL(dcn_col2im_sums_kernel<1>, sgrid, block, 0, stream, x, offset, mask, grad_col, g, grad_offset, grad_mask); break;
        case 2: hipLaunchKernelGGL(dcn_col2im_sums_kernel<2>, sgrid, block, 0, stream, x, offset, mask, grad_col, g, grad_offset, grad_mask); break;
        case 3: hipLaunchKernelGGL(dcn_col2im_sums_kernel<3>, sgrid, block, 0, stream, x, offset, mask, grad_col, g, grad_offset, grad_mask); break;
        default: hipLaunchKernelGGL(dcn_col2im_sums_kernel<4>, sgrid, block, 0, stream, x, offset, mask, grad_col, g, grad_offset, grad_mask); break;
    }
    GGA_CHECK_LAUNCH("dcn_col2im_sums_kernel");
    if (!grad_x) return GGA_OK;
    GGA_CHECK_HIP(hipMemsetAsync(grad_x, 0, (size_t)B * H * W * C * sizeof(float), stream), "dcn memset");
    const int tiles_x = (g.Wo + DCN_TW - 1) / DCN_TW, tiles_y = (g.Ho + DCN_TH - 1) / DCN_TH;
    // LDS window: receptive field of the tile + offset slack + the bilinear (+1) corner
    int win_h = (DCN_TH - 1) * stride_h + (kh - 1) * dil_h + 2 + 2 * DCN_R;
    int win_w = (DCN_TW - 1) * stride_w + (kw - 1) * dil_w + 2 + 2 * DCN_R;
    const size_t pair_bytes = (size_t)DCN_TH * DCN_TW * kh * kw * sizeof(DcnPair);          // the tile's (pixel, tap) records
    GGA_REQUIRE(pair_bytes + 16 * 1024 <= 160 * 1024, "gga_dcn_col2im: kernel %dx%d too large", kh, kw);
    int max_win = (int)((160 * 1024 - pair_bytes) / 1024);                                 // 1 KB per window pixel (256 channels)
    if (max_win > DCN_MAXWIN) max_win = DCN_MAXWIN;
    while (win_h * win_w > max_win) { if (win_h > 4) win_h -= 1; if (win_w > 4 && win_h * win_w > max_win) win_w -= 1; }   // smaller window: more direct atomics, same result
    const size_t lds = (size_t)win_h * win_w * 1024 + pair_bytes;
    GGA_CHECK_HIP(hipFuncSetAttribute((const void*)dcn_col2im_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                  "dcn col2im LDS size");
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3((unsigned)(B * tiles_x * tiles_y)), block, lds, stream, offset, mask, grad_col, g,
                       tiles_x, tiles_y, win_h, win_w, grad_x);
    GGA_CHECK_LAUNCH("dcn_col2im_kernel");
    return GGA_OK;
}
