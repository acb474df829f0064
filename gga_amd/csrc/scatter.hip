// a3: PointPillars scatter forward / backward for gfx950 — the kernel the
// "HBM GB/s on voxel scatter" metric measures.
//
// Reference (mmdet3d/models/middle_encoders/pillar_scatter.py:62-102): per frame,
// zeros(C, ny*nx) -> boolean mask -> index_put -> stack. That is a zero-fill pass
// plus a 4-byte-granular column scatter (each pillar touches C different cache lines).
//
// Here the scatter is turned into a GATHER over the dense output so every byte of the
// canvas is written exactly once, by coalesced 16 B-per-lane stores, with the zero-fill
// fused in:
//   1. map pass     cell_map[b, y*nx+x] = pillar row   (tiny: 4 B per pillar)
//   2. canvas pass  every thread owns 4 consecutive cells; reads their 4 map entries
//                   (one 16 B load), resets them to -1 (the map is self-cleaning, so no
//                   per-call memset of the map), then for every channel stores one float4
//                   = the pillar feature where a pillar exists, 0 elsewhere.
// HBM traffic = canvas bytes (write) + feature bytes (read, each row fetched once and then
// served from L1/L2 for the channel loop) + map bytes: the algorithmic minimum of
// SURVEY.md §8(d) plus the 4 B/cell map read+reset.
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

// NCHW canvas. One thread = 4 consecutive cells of one frame, all channels.
// cells4 = ny*nx/4 (ny*nx must be a multiple of 4).
template <int UNROLL>
__global__ __launch_bounds__(256) void scatter_canvas_nchw_kernel(const float* __restrict__ feats,
                                                                 int32_t* __restrict__ cell_map, int channels,
                                                                 int64_t cells, int64_t cells4_total,
                                                                 float* __restrict__ canvas) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;   // global quad index over batch*cells/4
    if (q >= cells4_total) return;
    const int64_t cells4 = cells >> 2;
    const int64_t b = q / cells4;
    const int64_t cq = q - b * cells4;
    int4* mp = reinterpret_cast<int4*>(cell_map) + q;
    const int4 idx = *mp;
    const bool any = (idx.x & idx.y & idx.z & idx.w) != -1;   // any entry != -1
    if (any) *mp = make_int4(-1, -1, -1, -1);                  // self-cleaning map
    float4* out = reinterpret_cast<float4*>(canvas + (b * channels) * cells) + cq;
    const int64_t cstride4 = cells4;                           // float4 stride between channels
    if (!any) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int c = 0; c < channels; ++c) out[(int64_t)c * cstride4] = z;
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
            v[u].x = idx.x >= 0 ? f0[c + u] : 0.f;
            v[u].y = idx.y >= 0 ? f1[c + u] : 0.f;
            v[u].z = idx.z >= 0 ? f2[c + u] : 0.f;
            v[u].w = idx.w >= 0 ? f3[c + u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) out[(int64_t)(c + u) * cstride4] = v[u];
    }
}

// NHWC canvas (channels-last memory): one thread = 4 channels of one cell; the
// channels/4 lanes of a cell read the same map word (broadcast) and write one
// contiguous row.
__global__ __launch_bounds__(256) void scatter_canvas_nhwc_kernel(const float* __restrict__ feats,
                                                                 int32_t* __restrict__ cell_map, int c4,
                                                                 int64_t total4, float* __restrict__ canvas) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int64_t cell = t / c4;
    const int cc = (int)(t - cell * c4);
    const int32_t idx = cell_map[cell];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx >= 0) v = reinterpret_cast<const float4*>(feats)[(int64_t)idx * c4 + cc];
    reinterpret_cast<float4*>(canvas)[t] = v;
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
                                      int batch, int channels, int ny, int nx, int layout, int32_t* cell_map,
                                      float* canvas, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(cell_map && canvas && (m == 0 || (feats && coors)), "gga_pillar_scatter_fwd: null pointer argument");
    if (int rc = scatter_check("gga_pillar_scatter_fwd", m, batch, channels, ny, nx, layout)) return rc;
    const int64_t cells = (int64_t)ny * nx;
    if (m > 0) {
        hipLaunchKernelGGL(scatter_map_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, coors, m,
                           num_valid, batch, ny, nx, cell_map);
        GGA_CHECK_LAUNCH("scatter_map_kernel");
    }
    if (layout == GGA_LAYOUT_NCHW) {
        const int64_t q = (int64_t)batch * cells / 4;
        if (channels % 8 == 0)
            hipLaunchKernelGGL(scatter_canvas_nchw_kernel<8>, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, stream,
                               feats, cell_map, channels, cells, q, canvas);
        else
            hipLaunchKernelGGL(scatter_canvas_nchw_kernel<4>, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, stream,
                               feats, cell_map, channels, cells, q, canvas);
        GGA_CHECK_LAUNCH("scatter_canvas_nchw_kernel");
    } else {
        const int c4 = channels / 4;
        const int64_t total4 = (int64_t)batch * cells * c4;
        hipLaunchKernelGGL(scatter_canvas_nhwc_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream,
                           feats, cell_map, c4, total4, canvas);
        GGA_CHECK_LAUNCH("scatter_canvas_nhwc_kernel");
        if (m > 0) {   // NHWC readers share map words, so the reset is its own (4 B/pillar) pass
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
extern "C" int gga_profile_pillar_scatter(const float* feats, const int32_t* coors, int64_t m, int batch,
                                          int channels, int ny, int nx, int layout, int32_t* cell_map,
                                          float* canvas, int iters, float* ms_map_host, float* ms_canvas_host,
                                          void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(feats && coors && cell_map && canvas && ms_map_host && ms_canvas_host && iters >= 1 && m > 0,
                "gga_profile_pillar_scatter: bad arguments");
    if (int rc = scatter_check("gga_profile_pillar_scatter", m, batch, channels, ny, nx, layout)) return rc;
    const int64_t cells = (int64_t)ny * nx;
    hipEvent_t e0, e1, e2;
    GGA_CHECK_HIP(hipEventCreate(&e0), "hipEventCreate");
    GGA_CHECK_HIP(hipEventCreate(&e1), "hipEventCreate");
    GGA_CHECK_HIP(hipEventCreate(&e2), "hipEventCreate");
    double tm = 0.0, tc = 0.0;
    for (int it = 0; it < iters; ++it) {
        (void)hipEventRecord(e0, stream);
        hipLaunchKernelGGL(scatter_map_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, coors, m,
                           (const int32_t*)nullptr, batch, ny, nx, cell_map);
        (void)hipEventRecord(e1, stream);
        if (layout == GGA_LAYOUT_NCHW) {
            const int64_t q = (int64_t)batch * cells / 4;
            if (channels % 8 == 0)
                hipLaunchKernelGGL(scatter_canvas_nchw_kernel<8>, dim3((unsigned)((q + 255) / 256)), dim3(256), 0,
                                   stream, feats, cell_map, channels, cells, q, canvas);
            else
                hipLaunchKernelGGL(scatter_canvas_nchw_kernel<4>, dim3((unsigned)((q + 255) / 256)), dim3(256), 0,
                                   stream, feats, cell_map, channels, cells, q, canvas);
            (void)hipEventRecord(e2, stream);
        } else {
            const int c4 = channels / 4;
            const int64_t total4 = (int64_t)batch * cells * c4;
            hipLaunchKernelGGL(scatter_canvas_nhwc_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                               stream, feats, cell_map, c4, total4, canvas);
            (void)hipEventRecord(e2, stream);
            hipLaunchKernelGGL(scatter_map_reset_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream,
                               coors, m, (const int32_t*)nullptr, batch, ny, nx, cell_map);
        }
        GGA_CHECK_HIP(hipEventSynchronize(e2), "hipEventSynchronize");
        float a = 0.f, b = 0.f;
        (void)hipEventElapsedTime(&a, e0, e1);
        (void)hipEventElapsedTime(&b, e1, e2);
        tm += a; tc += b;
    }
    GGA_CHECK_HIP(hipStreamSynchronize(stream), "hipStreamSynchronize");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    GGA_CHECK_LAUNCH("gga_profile_pillar_scatter");
    *ms_map_host = (float)(tm / iters);
    *ms_canvas_host = (float)(tc / iters);
    return GGA_OK;
}
