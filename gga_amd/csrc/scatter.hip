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
#include <stdlib.h>

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


// ---- variants (selected by launch_canvas_nchw) -----------------------------------------
// ceiling probe: the store pattern alone (no map, no gathers)
__global__ __launch_bounds__(256) void scatter_canvas_zero_probe_kernel(int channels, int64_t cells,
                                                                       int64_t cells4_total,
                                                                       float* __restrict__ canvas) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= cells4_total) return;
    const int64_t cells4 = cells >> 2;
    const int64_t b = q / cells4;
    const int64_t cq = q - b * cells4;
    float4* out = reinterpret_cast<float4*>(canvas + (b * channels) * cells) + cq;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int c = 0; c < channels; ++c) out[(int64_t)c * cells4] = z;
}

// V1: like V0 but the all-empty fast path is taken only when the WHOLE wave is empty, so a
// wave never issues two half-masked store streams.
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

// V2: LDS-staged. A 256-thread workgroup owns 1024 consecutive cells of one frame.
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

// NHWC canvas (channels-last memory): one thread = 4 channels of one cell; the
// channels/4 lanes of a cell read the same map word (broadcast) and write one
// contiguous row.
template <int U, bool NT>
__global__ __launch_bounds__(256) void scatter_canvas_nhwc_kernel(const float* __restrict__ feats,
                                                                 int32_t* __restrict__ cell_map, int c4,
                                                                 int64_t total4, float* __restrict__ canvas) {
    // U independent 16 B pieces per thread (block covers U*256 consecutive float4s), streaming stores
    const int shift = ((c4 & (c4 - 1)) == 0) ? (31 - __clz(c4)) : -1;
    const int64_t base = (int64_t)blockIdx.x * (U * 256) + threadIdx.x;
    int32_t idx[U];
    int cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t t = base + u * 256;
        idx[u] = -1; cc[u] = 0;
        if (t < total4) {
            const int64_t cell = shift >= 0 ? (t >> shift) : (t / c4);
            cc[u] = (int)(t - cell * c4);
            idx[u] = cell_map[cell];
        }
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx[u] >= 0) v[u] = reinterpret_cast<const float4*>(feats)[(int64_t)idx[u] * c4 + cc[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t t = base + u * 256;
        if (t < total4) store4<NT>(reinterpret_cast<float4*>(canvas) + t, v[u]);
    }
}

// NHWC canvas, wave-per-64-cells form (C4 = channels/4 divides 64): a wavefront owns 64
// consecutive cells = 64*C4 contiguous float4s. One coalesced 256 B map load per wave, then C4
// wave-wide 1 KB stores; each lane gets the map word of its cell by a cross-lane read. Empty cells
// (92 % of a KITTI canvas) are written with stores that depend on nothing but the map word, so
// the kernel runs as a fill with a minority of gather->store chains on the side: with a cold
// memory system the form above is bound by the latency of its map -> row -> store chain.
template <int C4, bool NT>
__global__ __launch_bounds__(256) void scatter_canvas_nhwc_wave_kernel(const float* __restrict__ feats,
                                                                      const int32_t* __restrict__ cell_map,
                                                                      int64_t total_cells,
                                                                      float* __restrict__ canvas) {
    constexpr int CPI = 64 / C4;                 // cells covered by one wave-wide store
    const int lane = threadIdx.x & 63;
    const int64_t cell0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (cell0 >= total_cells) return;
    const int mine = cell0 + lane < total_cells ? cell_map[cell0 + lane] : -2;
    const int sub = lane / C4, piece = lane - sub * C4;
    float4* out = reinterpret_cast<float4*>(canvas) + cell0 * C4 + lane;
    const float4* f4 = reinterpret_cast<const float4*>(feats);
    int idx[C4];
#pragma unroll
    for (int it = 0; it < C4; ++it) idx[it] = __shfl(mine, it * CPI + sub);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < C4; ++it)
        if (idx[it] == -1) store4<NT>(out + it * 64, zero);
    float4 v[C4];
#pragma unroll
    for (int it = 0; it < C4; ++it)
        if (idx[it] >= 0) v[it] = f4[(int64_t)idx[it] * C4 + piece];
#pragma unroll
    for (int it = 0; it < C4; ++it)
        if (idx[it] >= 0) store4<NT>(out + it * 64, v[it]);
}

// NHWC canvas, fill + rows form (default): measured with a cold memory system (the state inside a
// train step) a plain fill of the canvas runs at ~7 TB/s while every fused form above stays at
// 4.2-5.4 TB/s - its map -> row -> store chains and the reads mixed into the write stream cost
// more than writing the 7 % occupied cells twice. So: (1) stream zeros over the whole canvas,
// (2) winner map as before, (3) one 16 B piece per thread: the winning row of each occupied cell
// is copied to its place and its map word is reset in the same pass.
template <bool NT>
__global__ __launch_bounds__(256) void scatter_fill_kernel(float4* __restrict__ canvas4, int64_t total4) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t t = base + u * 256;
        if (t < total4) store4<NT>(canvas4 + t, zero);
    }
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

// GGA_SCATTER_NHWC_FORM: 2 = fill + rows (default), 1 = wave-per-64-cells gather, 0 = piece-per-thread gather
static int nhwc_form() {
    static int form = -1;
    if (form < 0) { const char* e = getenv("GGA_SCATTER_NHWC_FORM"); form = e ? atoi(e) : 2; }
    return form;
}
static int nhwc_nt() {
    static int nt = -1;
    if (nt < 0) { const char* e = getenv("GGA_SCATTER_NHWC_NT"); nt = e ? atoi(e) : 1; }
    return nt;
}

static void launch_canvas_nhwc(hipStream_t stream, const float* feats, int32_t* cell_map, int c4, int64_t total4,
                               float* canvas) {
    const int form = nhwc_form(), nt = nhwc_nt();
    if (form == 1 && (c4 == 8 || c4 == 16 || c4 == 32)) {
        const int64_t cells = total4 / c4;
        const dim3 grid((unsigned)((cells + 255) / 256)), block(256);
#define NW(C4) { if (nt) hipLaunchKernelGGL((scatter_canvas_nhwc_wave_kernel<C4, true>), grid, block, 0, stream, feats, cell_map, cells, canvas); \
                 else hipLaunchKernelGGL((scatter_canvas_nhwc_wave_kernel<C4, false>), grid, block, 0, stream, feats, cell_map, cells, canvas); }
        if (c4 == 8) NW(8) else if (c4 == 16) NW(16) else NW(32)
#undef NW
        return;
    }
    static int u = -1;
    if (u < 0) { const char* e = getenv("GGA_SCATTER_NHWC_UNROLL"); u = e ? atoi(e) : 1; }   // measured: 1 piece/thread + NT stores = 7.6 TB/s, 4 = 5.8, 16 = 5.6
#define NH(U) { if (nt) hipLaunchKernelGGL((scatter_canvas_nhwc_kernel<U, true>), dim3((unsigned)((total4 + U * 256 - 1) / (U * 256))), dim3(256), 0, stream, feats, cell_map, c4, total4, canvas); \
                else hipLaunchKernelGGL((scatter_canvas_nhwc_kernel<U, false>), dim3((unsigned)((total4 + U * 256 - 1) / (U * 256))), dim3(256), 0, stream, feats, cell_map, c4, total4, canvas); }
    if (u == 2) NH(2) else if (u == 4) NH(4) else if (u == 8) NH(8) else if (u == 16) NH(16) else NH(1)
#undef NH
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


// Development knob: GGA_SCATTER_VARIANT=0|1|2|9 picks the NCHW canvas kernel (default: best measured).
static int scatter_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GGA_SCATTER_VARIANT");
        v = e ? atoi(e) : 3;
    }
    return v;
}

static void launch_canvas_nchw(hipStream_t stream, const float* feats, int32_t* cell_map, int channels,
                               int64_t cells, int batch, float* canvas) {
    const int64_t q = (int64_t)batch * cells / 4;
    const dim3 grid((unsigned)((q + 255) / 256)), block(256);
    const int var = scatter_variant();
    if (var == 9) {
        hipLaunchKernelGGL(scatter_canvas_zero_probe_kernel, grid, block, 0, stream, channels, cells, q, canvas);
    } else if (var == 2 && channels == 64) {
        hipLaunchKernelGGL((scatter_canvas_nchw_v2_kernel<64, false>), grid, block, 0, stream, feats, cell_map, cells, q, canvas);
    } else if (var == 3 && channels == 64) {
        hipLaunchKernelGGL((scatter_canvas_nchw_v2_kernel<64, true>), grid, block, 0, stream, feats, cell_map, cells, q, canvas);
    } else if (var == 2 && channels == 128) {
        hipLaunchKernelGGL((scatter_canvas_nchw_v2_kernel<128, false>), grid, block, 0, stream, feats, cell_map, cells, q, canvas);
    } else if (var >= 1) {
        if (channels % 8 == 0)
            hipLaunchKernelGGL(scatter_canvas_nchw_v1_kernel<8>, grid, block, 0, stream, feats, cell_map, channels, cells, q, canvas);
        else
            hipLaunchKernelGGL(scatter_canvas_nchw_v1_kernel<4>, grid, block, 0, stream, feats, cell_map, channels, cells, q, canvas);
    } else {
        if (channels % 8 == 0)
            hipLaunchKernelGGL(scatter_canvas_nchw_kernel<8>, grid, block, 0, stream, feats, cell_map, channels, cells, q, canvas);
        else
            hipLaunchKernelGGL(scatter_canvas_nchw_kernel<4>, grid, block, 0, stream, feats, cell_map, channels, cells, q, canvas);
    }
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

// Bench-only in-place timing: while armed, every forward call brackets its canvas kernel with a
// pair of HIP events on the caller's stream (nothing is synchronised until the collect call).
#define SCATTER_TIMING_MAX 256
static hipEvent_t g_tev[SCATTER_TIMING_MAX][2];
static int g_tcap = 0, g_tcount = 0, g_tmade = 0;

extern "C" int gga_pillar_scatter_timing_begin(int max_samples) {
    GGA_REQUIRE(max_samples >= 0 && max_samples <= SCATTER_TIMING_MAX, "gga_pillar_scatter_timing_begin: 0 <= max_samples <= %d",
                SCATTER_TIMING_MAX);
    for (; g_tmade < max_samples; ++g_tmade) {
        GGA_CHECK_HIP(hipEventCreate(&g_tev[g_tmade][0]), "timing event");
        GGA_CHECK_HIP(hipEventCreate(&g_tev[g_tmade][1]), "timing event");
    }
    g_tcap = max_samples;
    g_tcount = 0;
    return GGA_OK;
}

extern "C" int gga_pillar_scatter_timing_collect(float* ms_host, int cap) {
    GGA_REQUIRE(ms_host || cap == 0, "gga_pillar_scatter_timing_collect: null pointer argument");
    const int n = g_tcount < cap ? g_tcount : cap;
    for (int i = 0; i < n; ++i) {
        GGA_CHECK_HIP(hipEventSynchronize(g_tev[i][1]), "timing sync");
        GGA_CHECK_HIP(hipEventElapsedTime(&ms_host[i], g_tev[i][0], g_tev[i][1]), "timing elapsed");
    }
    g_tcap = g_tcount = 0;
    return n;
}

extern "C" int gga_pillar_scatter_fwd(const float* feats, const int32_t* coors, int64_t m, const int32_t* num_valid,
                                      int batch, int channels, int ny, int nx, int layout, int unique_coors,
                                      int32_t* cell_map, float* canvas, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool need_map = !(unique_coors && layout == GGA_LAYOUT_NHWC && nhwc_form() == 2);
    GGA_REQUIRE((cell_map || !need_map) && canvas && (m == 0 || (feats && coors)),
                "gga_pillar_scatter_fwd: null pointer argument");
    if (int rc = scatter_check("gga_pillar_scatter_fwd", m, batch, channels, ny, nx, layout)) return rc;
    const int64_t cells = (int64_t)ny * nx;
    const bool timed = g_tcount < g_tcap;
    const bool fill_rows = layout == GGA_LAYOUT_NHWC && nhwc_form() == 2;
    // fill + rows form: the events bracket the whole op (fill, map, rows); otherwise the canvas kernel
    if (timed && fill_rows) GGA_CHECK_HIP(hipEventRecord(g_tev[g_tcount][0], stream), "timing record");
    if (fill_rows) {
        const int64_t total4 = (int64_t)batch * cells * (channels / 4);
        const dim3 grid((unsigned)((total4 + 1023) / 1024)), block(256);
        if (nhwc_nt()) hipLaunchKernelGGL(scatter_fill_kernel<true>, grid, block, 0, stream, (float4*)canvas, total4);
        else hipLaunchKernelGGL(scatter_fill_kernel<false>, grid, block, 0, stream, (float4*)canvas, total4);
        GGA_CHECK_LAUNCH("scatter_fill_kernel");
    }
    if (m > 0 && need_map) {
        hipLaunchKernelGGL(scatter_map_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, coors, m,
                           num_valid, batch, ny, nx, cell_map);
        GGA_CHECK_LAUNCH("scatter_map_kernel");
    }
    if (timed && !fill_rows) GGA_CHECK_HIP(hipEventRecord(g_tev[g_tcount][0], stream), "timing record");
    if (layout == GGA_LAYOUT_NCHW) {
        launch_canvas_nchw(stream, feats, cell_map, channels, cells, batch, canvas);
        GGA_CHECK_LAUNCH("scatter_canvas_nchw_kernel");
        if (timed) GGA_CHECK_HIP(hipEventRecord(g_tev[g_tcount++][1], stream), "timing record");
    } else if (nhwc_form() == 2) {
        const int c4 = channels / 4;
        const int64_t total4 = (int64_t)batch * cells * c4;
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
        if (timed) GGA_CHECK_HIP(hipEventRecord(g_tev[g_tcount++][1], stream), "timing record");
        if (m > 0 && need_map && !inplace_reset) {
            hipLaunchKernelGGL(scatter_map_reset_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream,
                               coors, m, num_valid, batch, ny, nx, cell_map);
            GGA_CHECK_LAUNCH("scatter_map_reset_kernel");
        }
        (void)total4;
    } else {
        const int c4 = channels / 4;
        const int64_t total4 = (int64_t)batch * cells * c4;
        launch_canvas_nhwc(stream, feats, cell_map, c4, total4, canvas);
        GGA_CHECK_LAUNCH("scatter_canvas_nhwc_kernel");
        if (timed) GGA_CHECK_HIP(hipEventRecord(g_tev[g_tcount++][1], stream), "timing record");
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
            launch_canvas_nchw(stream, feats, cell_map, channels, cells, batch, canvas);
            (void)hipEventRecord(e2, stream);
        } else {
            const int c4 = channels / 4;
            const int64_t total4 = (int64_t)batch * cells * c4;
            launch_canvas_nhwc(stream, feats, cell_map, c4, total4, canvas);
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
