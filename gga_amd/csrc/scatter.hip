// a3: PointPillars scatter forward / backward for gfx950 - the kernels the
// "HBM GB/s on voxel scatter" metric measures.
//
// Reference (mmdet3d/models/middle_encoders/pillar_scatter.py:62-102): per frame,
// zeros(C, ny*nx) -> boolean mask -> index_put -> stack. That is a zero-fill pass
// plus a 4-byte-granular column scatter (each pillar touches C different cache lines).
//
// NHWC canvas (channels-last memory, the default of the train step): a pillar is one contiguous
// 4*C-byte row, so the op is (1) stream zeros over the canvas, (2) copy the occupied rows to
// their cells with one 16 B piece per thread. Voxelizer output has one pillar per cell; for
// arbitrary input an atomicMax "winner" map (cell -> highest row) makes duplicates
// deterministic, and the rows pass resets the map words it used (self-cleaning, no per-call
// memset). Single-pass "gather" forms (every canvas byte written once, map -> row -> store
// per piece) measured 7.6 TB/s back to back but 4.2-4.8 TB/s inside a train step: re-running a
// kernel on the same buffers leaves map, rows and part of the canvas in the 256 MB MALL; with a
// cold memory system a plain fill keeps ~7 TB/s while the dependent load chain does not
// (DESIGN.md section 3).
//
// NCHW canvas (the reference layout): a pillar's channels are ny*nx*4 bytes apart, so the
// scatter is inverted into a gather over the output - every canvas byte is written once by
// coalesced 16 B-per-lane stores with the zero-fill fused in:
//   1. map pass     cell_map[b, y*nx+x] = pillar row   (4 B per pillar)
//   2. canvas pass  a workgroup owns 1024 consecutive cells; reads their map entries (and
//                   resets them to -1), stages the pillar rows in LDS, then stores one float4
//                   per channel and 4 cells: the pillar feature where a pillar exists, 0 elsewhere.
#include "gga_common.h"

__global__ __launch_bounds__(256) void scatter_map_kernel(const int32_t* __restrict__ coors, int64_t m,
                                                         const int32_t* __restrict__ num_valid, int batch,
                                                         int ny, int nx, int32_t* __restrict__ cell_map) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t lim = num_valid ? (int64_t)*num_valid : m;
    if (v >= m || v >= lim) return;
    const int4 c = reinterpret_cast<const int4*>(coors)[v];
    if ((unsigned)c.x >= (unsigned)batch || (unsigned)c.z >= (unsigned)ny || (unsigned)c.w >= (unsigned)nx) return;
    // highest row wins on duplicates (= sequential index_put of the reference's CPU path)
    atomicMax(&cell_map[((int64_t)c.x * ny + c.z) * nx + c.w], (int32_t)v);
}

// NCHW canvas, generic channel count. One thread = 4 consecutive cells of one frame, all
// channels (cells4 = ny*nx/4); the all-empty fast path is taken only when the WHOLE wave is
// empty, so a wave never issues two half-masked store streams.
template <int UNROLL>
__global__ __launch_bounds__(256) void scatter_canvas_nchw_v1_kernel(const float* __restrict__ feats,
                                                                    int32_t* __restrict__ cell_map, int channels,
                                                                    int64_t cells, int64_t cells4_total,
                                                                    float* __restrict__ canvas) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= cells4_total) return;
    const int64_t cells4 = cells >> 2;
    const int64_t b = q / cells4;
    const int64_t cq = q - b * cells4;
    int4* mp = reinterpret_cast<int4*>(cell_map) + q;
    const int4 idx = *mp;
    const bool any = (idx.x & idx.y & idx.z & idx.w) != -1;
    if (any) *mp = make_int4(-1, -1, -1, -1);
    float4* out = reinterpret_cast<float4*>(canvas + (b * channels) * cells) + cq;
    if (__ballot(any) == 0ull) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int c = 0; c < channels; ++c) out[(int64_t)c * cells4] = z;
        return;
    }
    const float* f0 = feats + (int64_t)(idx.x < 0 ? 0 : idx.x) * channels;
    const float* f1 = feats + (int64_t)(idx.y < 0 ? 0 : idx.y) * channels;
    const float* f2 = feats + (int64_t)(idx.z < 0 ? 0 : idx.z) * channels;
    const float* f3 = feats + (int64_t)(idx.w < 0 ? 0 : idx.w) * channels;
    for (int c = 0; c < channels; c += UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx.x >= 0) v[u].x = f0[c + u];
            if (idx.y >= 0) v[u].y = f1[c + u];
            if (idx.z >= 0) v[u].z = f2[c + u];
            if (idx.w >= 0) v[u].w = f3[c + u];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) out[(int64_t)(c + u) * cells4] = v[u];
    }
}

// NCHW canvas, C = 64 (the PointPillars canvas; at C = 128 - SparseEncoder.dense() - the generic kernel
// above measured 3x faster, 0.32 vs 0.94 ms): LDS-staged. A 256-thread workgroup owns 1024 consecutive cells of one frame.
//   (1) one 16 B map load per thread (+ reset), occupied cells get a compact tile slot;
//   (2) the pillar rows of the tile are fetched with fully coalesced 256 B wave loads
//       (lane = channel) into an LDS tile [slot][C+1];
//   (3) channel loop: every lane assembles its float4 from LDS (only occupied lanes read)
//       and issues one 1 KB-per-wave contiguous store per channel.
// Tiles with more than SC_CAP occupied cells (dense scenes) take the direct-gather path.
#define SC_CAP 128
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store4(float4* p, const float4& v) {
    if (NT) __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
    else *p = v;
}

template <int C, bool NT>
__global__ __launch_bounds__(256) void scatter_canvas_nchw_v2_kernel(const float* __restrict__ feats,
                                                                    int32_t* __restrict__ cell_map, int64_t cells,
                                                                    int64_t cells4_total,
                                                                    float* __restrict__ canvas) {
    __shared__ float tile[SC_CAP * (C + 1)];
    __shared__ int rows[SC_CAP];
    __shared__ int n_occ;
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const bool live = q < cells4_total;
    if (tid == 0) n_occ = 0;
    __syncthreads();
    int4 idx = make_int4(-1, -1, -1, -1);
    int4* mp = reinterpret_cast<int4*>(cell_map) + q;
    if (live) idx = *mp;
    const bool any = (idx.x & idx.y & idx.z & idx.w) != -1;
    if (any) *mp = make_int4(-1, -1, -1, -1);
    int4 slot = make_int4(-1, -1, -1, -1);
    if (any) {
        const int cnt = (idx.x >= 0) + (idx.y >= 0) + (idx.z >= 0) + (idx.w >= 0);
        int s0 = atomicAdd(&n_occ, cnt);
        if (idx.x >= 0) { slot.x = s0; if (s0 < SC_CAP) rows[s0] = idx.x; ++s0; }
        if (idx.y >= 0) { slot.y = s0; if (s0 < SC_CAP) rows[s0] = idx.y; ++s0; }
        if (idx.z >= 0) { slot.z = s0; if (s0 < SC_CAP) rows[s0] = idx.z; ++s0; }
        if (idx.w >= 0) { slot.w = s0; if (s0 < SC_CAP) rows[s0] = idx.w; ++s0; }
    }
    __syncthreads();
    const int nocc = n_occ;
    const int64_t cells4 = cells >> 2;
    const int64_t b = live ? q / cells4 : 0;
    const int64_t cq = q - b * cells4;
    float4* out = reinterpret_cast<float4*>(canvas + (b * C) * cells) + cq;
    if (nocc == 0) {
        if (!live) return;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int c = 0; c < C; ++c) store4<NT>(out + (int64_t)c * cells4, z);
        return;
    }
    if (nocc <= SC_CAP) {
        // (2) coalesced row fetch: 4 waves, each wave takes rows e = wave, wave+4, ...; lane = channel
        const int lane = tid & 63, wave = tid >> 6;
        for (int e = wave; e < nocc; e += 4) {
            const float* src = feats + (int64_t)rows[e] * C;
#pragma unroll
            for (int c0 = 0; c0 < C; c0 += 64)
                if (c0 + lane < C) tile[e * (C + 1) + c0 + lane] = src[c0 + lane];
        }
        __syncthreads();
        if (!live) return;
        const float* t0 = tile + (slot.x < 0 ? 0 : slot.x) * (C + 1);
        const float* t1 = tile + (slot.y < 0 ? 0 : slot.y) * (C + 1);
        const float* t2 = tile + (slot.z < 0 ? 0 : slot.z) * (C + 1);
        const float* t3 = tile + (slot.w < 0 ? 0 : slot.w) * (C + 1);
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot.x >= 0) v.x = t0[c];
            if (slot.y >= 0) v.y = t1[c];
            if (slot.z >= 0) v.z = t2[c];
            if (slot.w >= 0) v.w = t3[c];
            store4<NT>(out + (int64_t)c * cells4, v);
        }
        return;
    }
    if (!live) return;
    // dense tile: direct gather from global (rows come from L2 after first touch)
    const float* f0 = feats + (int64_t)(idx.x < 0 ? 0 : idx.x) * C;
    const float* f1 = feats + (int64_t)(idx.y < 0 ? 0 : idx.y) * C;
    const float* f2 = feats + (int64_t)(idx.z < 0 ? 0 : idx.z) * C;
    const float* f3 = feats + (int64_t)(idx.w < 0 ? 0 : idx.w) * C;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx.x >= 0) v.x = f0[c];
        if (idx.y >= 0) v.y = f1[c];
        if (idx.z >= 0) v.z = f2[c];
        if (idx.w >= 0) v.w = f3[c];
        store4<NT>(out + (int64_t)c * cells4, v);
    }
}

// NHWC canvas: fill, then one 16 B piece per thread copies the (winning) row of each occupied
// cell to its place; with a winner map, the row's first lane resets its map word in the same pass.
// (one plain 16-byte store per thread: 877.7 MB in 132 us = 6.65 TB/s stand-alone; non-temporal 137, four stores per thread 148-155 -
// what shipped until round 4 -, eight 161, hipMemsetAsync 136, 32 contiguous bytes per lane 370: tools_dev/micro/fill_rate.hip.
// One pass over the canvas instead of fill + row copies - every thread reads its cell's map word, then the pillar row's piece or
// nothing, then stores - was built and measured in the step: 0.253 ms against 0.160 for the pair; the dependent loads in front
// of every store cost more than writing the 7 % occupied cells twice.)
__global__ __launch_bounds__(256) void scatter_fill_kernel(float4* __restrict__ canvas4, int64_t total4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < total4) canvas4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool RESET>
__global__ __launch_bounds__(256) void scatter_rows_nhwc_kernel(const float4* __restrict__ feats4,
                                                               const int4* __restrict__ coors, int64_t m,
                                                               const int32_t* __restrict__ num_valid, int batch,
                                                               int ny, int nx, int c4, int shift,
                                                               int32_t* __restrict__ cell_map,
                                                               float4* __restrict__ canvas4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t lim = num_valid ? (int64_t)*num_valid : m;
    const int64_t r = shift >= 0 ? (t >> shift) : (t / c4);
    if (r >= m || r >= lim) return;
    const int piece = (int)(t - r * c4);
    const int4 c = coors[r];
    if ((unsigned)c.x >= (unsigned)batch || (unsigned)c.z >= (unsigned)ny || (unsigned)c.w >= (unsigned)nx) return;
    const int64_t cell = ((int64_t)c.x * ny + c.z) * nx + c.w;
    if (cell_map && cell_map[cell] != (int32_t)r) return;   // a duplicate of this cell with a higher row exists
    canvas4[cell * c4 + piece] = feats4[t];
    // RESET: a row's c4 lanes sit in one wavefront and read the map word with the same load, so
    // the one lane that resets it (after its own load returned) cannot be seen by the others
    if (RESET && piece == 0) cell_map[cell] = -1;
}

__global__ __launch_bounds__(256) void scatter_map_reset_kernel(const int32_t* __restrict__ coors, int64_t m,
                                                               const int32_t* __restrict__ num_valid, int batch,
                                                               int ny, int nx, int32_t* __restrict__ cell_map) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t lim = num_valid ? (int64_t)*num_valid : m;
    if (v >= m || v >= lim) return;
    const int4 c = reinterpret_cast<const int4*>(coors)[v];
    if ((unsigned)c.x >= (unsigned)batch || (unsigned)c.z >= (unsigned)ny || (unsigned)c.w >= (unsigned)nx) return;
    cell_map[((int64_t)c.x * ny + c.z) * nx + c.w] = -1;
}

// Backward = gather. One thread per (pillar, 4 channels).
//   NHWC: one 16 B load from the pillar's contiguous row.
//   NCHW: 4 strided 4 B loads; pillars are visited in row order, neighbouring cells share lines in L2.
__global__ __launch_bounds__(256) void scatter_bwd_kernel(const float* __restrict__ grad_canvas,
                                                         const int32_t* __restrict__ coors, int64_t m,
                                                         const int32_t* __restrict__ num_valid, int batch,
                                                         int channels, int ny, int nx, int layout,
                                                         float* __restrict__ grad_feats) {
    const int c4 = channels >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= m * c4) return;
    const int64_t v = t / c4;
    const int cc = (int)(t - v * c4);
    const int64_t lim = num_valid ? (int64_t)*num_valid : m;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < lim) {
        const int4 c = reinterpret_cast<const int4*>(coors)[v];
        if ((unsigned)c.x < (unsigned)batch && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx) {
            const int64_t cells = (int64_t)ny * nx;
            const int64_t cell = (int64_t)c.z * nx + c.w;
            if (layout == GGA_LAYOUT_NHWC) {
                g = reinterpret_cast<const float4*>(grad_canvas)[((int64_t)c.x * cells + cell) * c4 + cc];
            } else {
                const float* src = grad_canvas + ((int64_t)c.x * channels + cc * 4) * cells + cell;
                g.x = src[0]; g.y = src[cells]; g.z = src[2 * cells]; g.w = src[3 * cells];
            }
        }
    }
    reinterpret_cast<float4*>(grad_feats)[t] = g;
}


static void launch_canvas_nchw(hipStream_t stream, const float* feats, int32_t* cell_map, int channels,
                               int64_t cells, int batch, float* canvas) {
    const int64_t q = (int64_t)batch * cells / 4;
    const dim3 grid((unsigned)((q + 255) / 256)), block(256);
    if (channels == 64)
        hipLaunchKernelGGL((scatter_canvas_nchw_v2_kernel<64, true>), grid, block, 0, stream, feats, cell_map, cells, q, canvas);
    else if (channels % 8 == 0)
        hipLaunchKernelGGL(scatter_canvas_nchw_v1_kernel<8>, grid, block, 0, stream, feats, cell_map, channels, cells, q, canvas);
    else
        hipLaunchKernelGGL(scatter_canvas_nchw_v1_kernel<4>, grid, block, 0, stream, feats, cell_map, channels, cells, q, canvas);
}

extern "C" size_t gga_pillar_scatter_map_bytes(int batch, int ny, int nx) {
    return gga_align_up((size_t)batch * ny * nx * sizeof(int32_t), 256);
}

static int scatter_check(const char* fn, int64_t m, int batch, int channels, int ny, int nx, int layout) {
    GGA_REQUIRE(m >= 0 && batch >= 1 && ny >= 1 && nx >= 1, "%s: bad sizes (m=%lld batch=%d ny=%d nx=%d)", fn,
                (long long)m, batch, ny, nx);
    GGA_REQUIRE(channels >= 4 && channels % 4 == 0, "%s: channels (%d) must be a positive multiple of 4", fn, channels);
    GGA_REQUIRE(layout == GGA_LAYOUT_NCHW || layout == GGA_LAYOUT_NHWC, "%s: unknown layout %d", fn, layout);
    GGA_REQUIRE(layout == GGA_LAYOUT_NHWC || ((int64_t)ny * nx) % 4 == 0,
                "%s: NCHW layout needs ny*nx (%lld) to be a multiple of 4", fn, (long long)ny * nx);
    return GGA_OK;
}

extern "C" int gga_pillar_scatter_fwd(const float* feats, const int32_t* coors, int64_t m, const int32_t* num_valid,
                                      int batch, int channels, int ny, int nx, int layout, int unique_coors,
                                      int32_t* cell_map, float* canvas, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool need_map = !(unique_coors && layout == GGA_LAYOUT_NHWC);
    GGA_REQUIRE((cell_map || !need_map) && canvas && (m == 0 || (feats && coors)),
                "gga_pillar_scatter_fwd: null pointer argument");
    if (int rc = scatter_check("gga_pillar_scatter_fwd", m, batch, channels, ny, nx, layout)) return rc;
    const int64_t cells = (int64_t)ny * nx;
    hipEvent_t* tev = gga_timing_acquire(GGA_TIME_SCATTER_FWD, 0);
    const bool fill_rows = layout == GGA_LAYOUT_NHWC;
    // NHWC: the timing events bracket the whole op (fill, map, rows); NCHW: the canvas kernel
    if (fill_rows) GGA_TIME_START(tev, stream);
    if (fill_rows) {
        const int64_t total4 = (int64_t)batch * cells * (channels / 4);
        GGA_REQUIRE((total4 + 255) / 256 < 2147483647ll, "gga_pillar_scatter: canvas too large");
        const dim3 grid((unsigned)((total4 + 255) / 256)), block(256);
        hipLaunchKernelGGL(scatter_fill_kernel, grid, block, 0, stream, (float4*)canvas, total4);
        GGA_CHECK_LAUNCH("scatter_fill_kernel");
    }
    if (m > 0 && need_map) {
        hipLaunchKernelGGL(scatter_map_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, coors, m,
                           num_valid, batch, ny, nx, cell_map);
        GGA_CHECK_LAUNCH("scatter_map_kernel");
    }
    if (!fill_rows) GGA_TIME_START(tev, stream);
    if (layout == GGA_LAYOUT_NCHW) {
        launch_canvas_nchw(stream, feats, cell_map, channels, cells, batch, canvas);
        GGA_CHECK_LAUNCH("scatter_canvas_nchw_kernel");
        GGA_TIME_STOP(tev, stream);
    } else {
        const int c4 = channels / 4;
        const int shift = ((c4 & (c4 - 1)) == 0) ? (31 - __builtin_clz(c4)) : -1;
        const bool inplace_reset = (64 % c4) == 0;
        if (m > 0) {
            const dim3 grid((unsigned)((m * c4 + 255) / 256)), block(256);
            if (!need_map)
                hipLaunchKernelGGL(scatter_rows_nhwc_kernel<false>, grid, block, 0, stream, (const float4*)feats,
                                   (const int4*)coors, m, num_valid, batch, ny, nx, c4, shift, (int32_t*)nullptr,
                                   (float4*)canvas);
            else if (inplace_reset)
                hipLaunchKernelGGL(scatter_rows_nhwc_kernel<true>, grid, block, 0, stream, (const float4*)feats,
                                   (const int4*)coors, m, num_valid, batch, ny, nx, c4, shift, cell_map, (float4*)canvas);
            else
                hipLaunchKernelGGL(scatter_rows_nhwc_kernel<false>, grid, block, 0, stream, (const float4*)feats,
                                   (const int4*)coors, m, num_valid, batch, ny, nx, c4, shift, cell_map, (float4*)canvas);
            GGA_CHECK_LAUNCH("scatter_rows_nhwc_kernel");
        }
        GGA_TIME_STOP(tev, stream);
        if (m > 0 && need_map && !inplace_reset) {
            hipLaunchKernelGGL(scatter_map_reset_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream,
                               coors, m, num_valid, batch, ny, nx, cell_map);
            GGA_CHECK_LAUNCH("scatter_map_reset_kernel");
        }
    }
    return GGA_OK;
}

extern "C" int gga_pillar_scatter_bwd(const float* grad_canvas, const int32_t* coors, int64_t m,
                                      const int32_t* num_valid, int batch, int channels, int ny, int nx, int layout,
                                      float* grad_feats, void* stream_) {
    GGA_REQUIRE(m == 0 || (grad_canvas && coors && grad_feats), "gga_pillar_scatter_bwd: null pointer argument");
    if (int rc = scatter_check("gga_pillar_scatter_bwd", m, batch, channels, ny, nx, layout)) return rc;
    if (m == 0) return GGA_OK;
    const int64_t total = m * (channels / 4);
    hipLaunchKernelGGL(scatter_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       grad_canvas, coors, m, num_valid, batch, channels, ny, nx, layout, grad_feats);
    GGA_CHECK_LAUNCH("scatter_bwd_kernel");
    return GGA_OK;
}

// Bench-only: times the two forward kernels separately with HIP events on `stream`
// (SYNCHRONISES; never call it from a captured or latency-sensitive path).

// ----------------------------------------------------------------------------- pillar conv map
// Gather map of a dense 2D convolution restricted to the occupied cells of the canvas: for pillar
// r at (b, y, x) and kernel tap (ky, kx) the output cell that read it, as a row of the NHWC
// output [batch*oh*ow, C], or -1. One thread per (tap, pillar); rows past *num_valid get -1.
__global__ __launch_bounds__(256) void pillar_conv_map_kernel(const int4* __restrict__ coors, int64_t m,
                                                             const int32_t* __restrict__ num_valid, int kh, int kw,
                                                             int sh, int sw, int ph, int pw, int oh, int ow,
                                                             int32_t* __restrict__ map) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int k = blockIdx.y;
    const int ky = k / kw, kx = k - ky * kw;
    int32_t v = -1;
    if (!num_valid || r < *num_valid) {
        const int4 c = coors[r];                    // (b, z, y, x)
        const int ty = c.z + ph - ky, tx = c.w + pw - kx;
        if (ty >= 0 && tx >= 0 && ty % sh == 0 && tx % sw == 0) {
            const int oy = ty / sh, ox = tx / sw;
            if (oy < oh && ox < ow) v = (c.x * oh + oy) * ow + ox;
        }
    }
    map[(int64_t)k * m + r] = v;
}

// ------------------------------------------------------------------------------ sparse sites -> channels-last BEV map
// SparseEncoder's handover (middle_encoders/sparse_encoder.py:134-138: out.dense() -> view(N, C * D, H, W)) written straight
// into channels-last memory: site (b, d, y, x) with features f[c] owns out[b][y][x][c * D + d], c < C - 4-byte stores at a
// stride of D floats inside the pixel's C * D-float row. The canvas is zeroed by the caller's stream-ordered memset. Before,
// the map was scattered in NCHW (322 us at 8 x 256 x 200 x 176), copied to channels-last, and its gradient copied back and
// gathered with a stride of H * W floats per channel (635 us): 1.2 ms of the shipped config's 58 ms step.
__global__ __launch_bounds__(256) void sparse_bev_nhwc_fwd_kernel(const float4* __restrict__ f, const int4* __restrict__ coors,
                                                                int64_t n, int c4n, int D, int H, int W, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * c4n) return;
    const int64_t row = i / c4n;
    const int c4 = (int)(i - row * c4n);
    const int4 c = coors[row];                               // (b, z, y, x)
    const float4 v = f[i];
    float* dst = out + (((int64_t)c.x * H + c.z) * W + c.w) * (4 * c4n * D) + (int64_t)(4 * c4) * D + c.y;
    dst[0] = v.x; dst[D] = v.y; dst[2 * D] = v.z; dst[3 * D] = v.w;
}

__global__ __launch_bounds__(256) void sparse_bev_nhwc_bwd_kernel(const float* __restrict__ g, const int4* __restrict__ coors,
                                                                int64_t n, int c4n, int D, int H, int W, float4* __restrict__ gf) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * c4n) return;
    const int64_t row = i / c4n;
    const int c4 = (int)(i - row * c4n);
    const int4 c = coors[row];
    const float* src = g + (((int64_t)c.x * H + c.z) * W + c.w) * (4 * c4n * D) + (int64_t)(4 * c4) * D + c.y;
    gf[i] = make_float4(src[0], src[D], src[2 * D], src[3 * D]);
}

extern "C" int gga_sparse_bev_nhwc_fwd(const float* feats, const int32_t* coors, int64_t n, int batch, int channels, int depth,
                                       int height, int width, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(out && batch >= 1 && channels >= 4 && channels % 4 == 0 && depth >= 1 && height >= 1 && width >= 1 && n >= 0,
                "gga_sparse_bev_nhwc_fwd: bad sizes (channels %% 4 == 0)");
    GGA_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)batch * height * width * channels * depth * sizeof(float), stream),
                  "gga_sparse_bev_nhwc_fwd: memset");
    if (n == 0) return GGA_OK;
    GGA_REQUIRE(feats && coors && ((uintptr_t)feats & 15) == 0 && ((uintptr_t)coors & 15) == 0, "gga_sparse_bev_nhwc_fwd: null or unaligned pointer");
    const int64_t total = n * (channels / 4);
    hipLaunchKernelGGL(sparse_bev_nhwc_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const float4*)feats, (const int4*)coors, n, channels / 4, depth, height, width, out);
    GGA_CHECK_LAUNCH("sparse_bev_nhwc_fwd_kernel");
    return GGA_OK;
}

extern "C" int gga_sparse_bev_nhwc_bwd(const float* grad_out, const int32_t* coors, int64_t n, int batch, int channels, int depth,
                                       int height, int width, float* grad_feats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(batch >= 1 && channels >= 4 && channels % 4 == 0 && depth >= 1 && height >= 1 && width >= 1 && n >= 0,
                "gga_sparse_bev_nhwc_bwd: bad sizes (channels %% 4 == 0)");
    if (n == 0) return GGA_OK;
    GGA_REQUIRE(grad_out && coors && grad_feats && ((uintptr_t)grad_feats & 15) == 0 && ((uintptr_t)coors & 15) == 0,
                "gga_sparse_bev_nhwc_bwd: null or unaligned pointer");
    const int64_t total = n * (channels / 4);
    hipLaunchKernelGGL(sparse_bev_nhwc_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, grad_out,
                       (const int4*)coors, n, channels / 4, depth, height, width, (float4*)grad_feats);
    GGA_CHECK_LAUNCH("sparse_bev_nhwc_bwd_kernel");
    return GGA_OK;
}

extern "C" int gga_pillar_conv_map(const int32_t* coors, int64_t m, const int32_t* num_valid, int batch, int ny, int nx,
                                   int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int32_t* map,
                                   void* stream) {
    GGA_REQUIRE(coors && map, "gga_pillar_conv_map: null pointer argument");
    GGA_REQUIRE(m >= 0 && batch >= 1 && ny >= 1 && nx >= 1 && kh >= 1 && kw >= 1 && kh * kw <= 65535 && stride_h >= 1 &&
                    stride_w >= 1 && pad_h >= 0 && pad_w >= 0,
                "gga_pillar_conv_map: bad sizes");
    const int oh = (ny + 2 * pad_h - kh) / stride_h + 1, ow = (nx + 2 * pad_w - kw) / stride_w + 1;
    GGA_REQUIRE(oh >= 1 && ow >= 1 && (int64_t)batch * oh * ow < 2147483647ll, "gga_pillar_conv_map: output %dx%d", oh, ow);
    if (m == 0) return GGA_OK;
    hipLaunchKernelGGL(pillar_conv_map_kernel, dim3((unsigned)((m + 255) / 256), kh * kw), dim3(256), 0,
                       (hipStream_t)stream, (const int4*)coors, m, num_valid, kh, kw, stride_h, stride_w, pad_h, pad_w,
                       oh, ow, map);
    GGA_CHECK_LAUNCH("pillar_conv_map_kernel");
    return GGA_OK;
}
