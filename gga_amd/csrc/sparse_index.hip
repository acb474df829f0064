// a3': sparse 3D convolution for the SECOND-style SparseEncoder (gfx950).
// (this file: the index structures - hash index, output sites of strided convolutions, rule books, per-row offset masks;
//  the products are in sparse_conv.hip (split-plane matrix kernels) and sparse_conv_f32.hip (fp32 MFMA, any width))
//
// Reference: mmdet3d/models/middle_encoders/sparse_encoder.py:107-214 and
// mmdet3d/ops/sparse_block.py:82-199 build SubMConv3d / SparseConv3d layers from the
// un-vendored mmcv / spconv wheels (rule-book over a DENSE int grid of the whole volume, then
// per-offset gather -> GEMM -> scatter-add). Here, per resolution level:
//
//   index      open-addressing hash (64-bit CAS) cell id -> row. No dense grid (92 M cells).
//   out sites  (strided conv) every (input row, kernel offset) proposes an output cell; the hash
//              dedupes, the proposing pair with the smallest id owns the cell, an ordered
//              multi-block scan numbers the owners -> deterministic output order.
//   rulebook   GATHER form: nbr[k][out_row] = input row under kernel offset k, or -1; and the
//              transposed map for the backward-data pass (for SubM it is nbr[K-1-k]).
//   conv       output-stationary: a workgroup owns 64 output rows, walks the kernel offsets,
//              skips offsets no row of the tile uses, stages the gathered input rows and the
//              offset's weight slice in LDS and accumulates in registers. Each output row is
//              written once: no atomics, deterministic. Backward-data is the same kernel on the
//              transposed map / transposed weights.
//   bwd weight one workgroup per (kernel offset, row chunk): compacts the valid pairs of the
//              chunk, accumulates X^T G in registers, one float atomicAdd per weight per chunk.
#include <stdlib.h>

#include "gga_common.h"
#include <type_traits>
#include <hip/hip_fp16.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#define SP_EMPTY 0xFFFFFFFFFFFFFFFFull

struct SpDims { int B, D, H, W; };
struct SpConvGeom { int kz, ky, kx, sz, sy, sx, pz, py, px; };

struct SpIndex {              // view into a caller-provided buffer
    unsigned long long* keys; // [cap]
    int32_t* vals;            // [cap]
    uint32_t mask;
};

static inline uint64_t sp_cap(int64_t n) {
    uint64_t c = 1024;
    while (c < (uint64_t)(2 * n + 2)) c <<= 1;
    return c;
}

__device__ __forceinline__ uint32_t sp_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}
__device__ __forceinline__ unsigned long long sp_key(const SpDims d, int b, int z, int y, int x) {
    return (((unsigned long long)b * d.D + z) * d.H + y) * d.W + x;
}
__device__ __forceinline__ uint32_t sp_insert(const SpIndex ix, unsigned long long key) {
    uint32_t h = sp_hash(key) & ix.mask;
    while (true) {
        const unsigned long long prev = atomicCAS(&ix.keys[h], SP_EMPTY, key);
        if (prev == SP_EMPTY || prev == key) return h;
        h = (h + 1) & ix.mask;
    }
}
__device__ __forceinline__ int32_t sp_lookup(const SpIndex ix, unsigned long long key) {
    uint32_t h = sp_hash(key) & ix.mask;
    while (true) {
        const unsigned long long k = ix.keys[h];
        if (k == key) return ix.vals[h];
        if (k == SP_EMPTY) return -1;
        h = (h + 1) & ix.mask;
    }
}
__device__ __forceinline__ uint32_t sp_find_slot(const SpIndex ix, unsigned long long key) {
    uint32_t h = sp_hash(key) & ix.mask;
    while (ix.keys[h] != key) h = (h + 1) & ix.mask;
    return h;
}

static SpIndex sp_index_view(void* buf, int64_t n) {
    const uint64_t cap = sp_cap(n);
    SpIndex ix;
    ix.keys = (unsigned long long*)buf;
    ix.vals = (int32_t*)((char*)buf + cap * 8);
    ix.mask = (uint32_t)(cap - 1);
    return ix;
}

extern "C" size_t gga_sparse_index_bytes(int64_t n) { return sp_cap(n) * 12; }

// ------------------------------------------------------------------------------ index build
__global__ __launch_bounds__(256) void sp_index_insert_kernel(const int4* __restrict__ coors, int64_t n, SpDims d,
                                                             SpIndex ix) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = coors[i];
    if ((unsigned)c.x >= (unsigned)d.B || (unsigned)c.y >= (unsigned)d.D || (unsigned)c.z >= (unsigned)d.H ||
        (unsigned)c.w >= (unsigned)d.W)
        return;
    const uint32_t h = sp_insert(ix, sp_key(d, c.x, c.y, c.z, c.w));
    atomicMax(&ix.vals[h], (int32_t)i);      // duplicate coordinates: the highest row wins
}

extern "C" int gga_sparse_build_index(const int32_t* coors, int64_t n, int B, int D, int H, int W, void* index,
                                      size_t index_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(index && (n == 0 || coors), "gga_sparse_build_index: null pointer argument");
    GGA_REQUIRE(n >= 0 && B >= 1 && D >= 1 && H >= 1 && W >= 1, "gga_sparse_build_index: bad sizes");
    if (index_bytes < gga_sparse_index_bytes(n)) {
        gga_set_error("gga_sparse_build_index: index buffer %zu B < required %zu B", index_bytes,
                      gga_sparse_index_bytes(n));
        return GGA_ERR_WORKSPACE;
    }
    const uint64_t cap = sp_cap(n);
    SpIndex ix = sp_index_view(index, n);
    GGA_CHECK_HIP(hipMemsetAsync(ix.keys, 0xFF, cap * 12, stream), "sparse index memset");   // keys empty, the values behind them -1: one fill
    if (n > 0) {
        const SpDims d = { B, D, H, W };
        hipLaunchKernelGGL(sp_index_insert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const int4*)coors, n, d, ix);
        GGA_CHECK_LAUNCH("sp_index_insert_kernel");
    }
    return GGA_OK;
}

// ------------------------------------------------------------------------------ output sites
// candidate id = in_row * kvol + k. Output coordinate of (input coord c, offset k): (c + p - k) / s
// when divisible and inside the output grid.
__device__ __forceinline__ bool sp_out_coord(const SpConvGeom g, const SpDims od, int z, int y, int x, int k,
                                             int& oz, int& oy, int& ox) {
    const int kx = k % g.kx, ky = (k / g.kx) % g.ky, kz = k / (g.kx * g.ky);
    const int tz = z + g.pz - kz, ty = y + g.py - ky, tx = x + g.px - kx;
    if (tz < 0 || ty < 0 || tx < 0) return false;
    if (tz % g.sz || ty % g.sy || tx % g.sx) return false;
    oz = tz / g.sz; oy = ty / g.sy; ox = tx / g.sx;
    return oz < od.D && oy < od.H && ox < od.W;
}

__global__ __launch_bounds__(256) void sp_sites_propose_kernel(const int4* __restrict__ in_coors, int64_t n_in,
                                                              int kvol, SpConvGeom g, SpDims od, SpIndex ox_,
                                                              unsigned long long* __restrict__ first) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_in * kvol) return;
    const int64_t i = t / kvol;
    const int k = (int)(t - i * kvol);
    const int4 c = in_coors[i];
    int oz, oy, ox;
    if (!sp_out_coord(g, od, c.y, c.z, c.w, k, oz, oy, ox)) return;
    const uint32_t h = sp_insert(ox_, sp_key(od, c.x, oz, oy, ox));
    atomicMin(&first[h], (unsigned long long)t);
}

__global__ __launch_bounds__(256) void sp_sites_count_kernel(const int4* __restrict__ in_coors, int64_t n_in, int kvol,
                                                            SpConvGeom g, SpDims od, SpIndex ox_,
                                                            const unsigned long long* __restrict__ first,
                                                            int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const int4 c = in_coors[i];
    int n = 0;
    for (int k = 0; k < kvol; ++k) {
        int oz, oy, ox;
        if (!sp_out_coord(g, od, c.y, c.z, c.w, k, oz, oy, ox)) continue;
        const uint32_t h = sp_find_slot(ox_, sp_key(od, c.x, oz, oy, ox));
        n += (first[h] == (unsigned long long)(i * kvol + k));
    }
    cnt[i] = n;
}

// three-step exclusive scan of cnt[n] (block sums -> scan of sums -> apply), 1024 per block
__global__ __launch_bounds__(1024) void sp_scan_block_kernel(const int32_t* __restrict__ cnt, int64_t n,
                                                            int32_t* __restrict__ excl, int32_t* __restrict__ bsum) {
    __shared__ int wsum[16];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? cnt[i] : 0;
    int s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o, 64); if ((threadIdx.x & 63) >= o) s += t; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    if (i < n) excl[i] = base + s - v;
    if (threadIdx.x == 1023) bsum[blockIdx.x] = base + s;
}
__global__ __launch_bounds__(1024) void sp_scan_sums_kernel(int32_t* __restrict__ bsum, int nblk,
                                                           int32_t* __restrict__ total) {
    // single block, serial over chunks of 1024 (nblk is n/1024: a few hundred)
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nblk; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const int v = i < nblk ? bsum[i] : 0;
        int s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o, 64); if ((threadIdx.x & 63) >= o) s += t; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = s;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
        if (i < nblk) bsum[i] = base + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void sp_sites_assign_kernel(const int4* __restrict__ in_coors, int64_t n_in,
                                                             int kvol, SpConvGeom g, SpDims od, SpIndex ox_,
                                                             const unsigned long long* __restrict__ first,
                                                             const int32_t* __restrict__ excl,
                                                             const int32_t* __restrict__ bsum, int64_t cap_out,
                                                             int4* __restrict__ out_coors) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const int4 c = in_coors[i];
    int64_t row = (int64_t)excl[i] + bsum[i >> 10];
    for (int k = 0; k < kvol; ++k) {
        int oz, oy, ox;
        if (!sp_out_coord(g, od, c.y, c.z, c.w, k, oz, oy, ox)) continue;
        const uint32_t h = sp_find_slot(ox_, sp_key(od, c.x, oz, oy, ox));
        if (first[h] == (unsigned long long)(i * kvol + k)) {
            if (row < cap_out) out_coors[row] = make_int4(c.x, oz, oy, ox);
            ox_.vals[h] = (int32_t)row;
            ++row;
        }
    }
}

extern "C" size_t gga_sparse_out_sites_workspace_bytes(int64_t n_in, int kvol) {
    const int64_t cap = n_in * kvol;       // upper bound on distinct output cells
    return sp_cap(cap) * 8 + gga_align_up((size_t)n_in * 4, 256) * 2 + gga_align_up((size_t)((n_in + 1023) / 1024 + 1) * 4, 256);
}
extern "C" size_t gga_sparse_out_index_bytes(int64_t n_in, int kvol) { return gga_sparse_index_bytes(n_in * kvol); }

extern "C" int gga_sparse_conv_out_sites(const int32_t* in_coors, int64_t n_in, int B, const int32_t in_dhw[3],
                                         const int32_t kernel[3], const int32_t stride[3], const int32_t pad[3],
                                         int32_t out_dhw[3], int32_t* out_coors, int64_t cap_out, int32_t* n_out,
                                         void* out_index, size_t out_index_bytes, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(in_coors && in_dhw && kernel && stride && pad && out_dhw && out_coors && n_out && out_index && workspace,
                "gga_sparse_conv_out_sites: null pointer argument");
    GGA_REQUIRE(n_in >= 1 && B >= 1, "gga_sparse_conv_out_sites: bad sizes");
    const int kvol = kernel[0] * kernel[1] * kernel[2];
    GGA_REQUIRE(kvol >= 1 && kvol <= 64 && stride[0] >= 1 && stride[1] >= 1 && stride[2] >= 1,
                "gga_sparse_conv_out_sites: unsupported kernel/stride");
    for (int a = 0; a < 3; ++a) {
        out_dhw[a] = (in_dhw[a] + 2 * pad[a] - kernel[a]) / stride[a] + 1;
        GGA_REQUIRE(out_dhw[a] >= 1, "gga_sparse_conv_out_sites: empty output grid");
    }
    if (out_index_bytes < gga_sparse_out_index_bytes(n_in, kvol) ||
        workspace_bytes < gga_sparse_out_sites_workspace_bytes(n_in, kvol)) {
        gga_set_error("gga_sparse_conv_out_sites: index/workspace buffer too small");
        return GGA_ERR_WORKSPACE;
    }
    const int64_t ncand = n_in * kvol;
    const uint64_t cap = sp_cap(ncand);
    SpIndex ox_ = sp_index_view(out_index, ncand);
    char* w = (char*)workspace;
    unsigned long long* first = (unsigned long long*)w; w += cap * 8;
    int32_t* cnt = (int32_t*)w; w += gga_align_up((size_t)n_in * 4, 256);
    int32_t* excl = (int32_t*)w; w += gga_align_up((size_t)n_in * 4, 256);
    int32_t* bsum = (int32_t*)w;
    GGA_CHECK_HIP(hipMemsetAsync(ox_.keys, 0xFF, cap * 12, stream), "out sites memset");      // keys and the values behind them
    GGA_CHECK_HIP(hipMemsetAsync(first, 0xFF, cap * 8, stream), "out sites memset");
    const SpConvGeom g = { kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2] };
    const SpDims od = { B, out_dhw[0], out_dhw[1], out_dhw[2] };
    hipLaunchKernelGGL(sp_sites_propose_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, stream,
                       (const int4*)in_coors, n_in, kvol, g, od, ox_, first);
    GGA_CHECK_LAUNCH("sp_sites_propose_kernel");
    hipLaunchKernelGGL(sp_sites_count_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream,
                       (const int4*)in_coors, n_in, kvol, g, od, ox_, first, cnt);
    GGA_CHECK_LAUNCH("sp_sites_count_kernel");
    const int nblk = (int)((n_in + 1023) / 1024);
    hipLaunchKernelGGL(sp_scan_block_kernel, dim3(nblk), dim3(1024), 0, stream, cnt, n_in, excl, bsum);
    GGA_CHECK_LAUNCH("sp_scan_block_kernel");
    hipLaunchKernelGGL(sp_scan_sums_kernel, dim3(1), dim3(1024), 0, stream, bsum, nblk, n_out);
    GGA_CHECK_LAUNCH("sp_scan_sums_kernel");
    hipLaunchKernelGGL(sp_sites_assign_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream,
                       (const int4*)in_coors, n_in, kvol, g, od, ox_, first, excl, bsum, cap_out, (int4*)out_coors);
    GGA_CHECK_LAUNCH("sp_sites_assign_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ rulebooks
// forward (gather) map: nbr[k][r] = input row at out_coord*stride - pad + k
__global__ __launch_bounds__(256) void sp_rulebook_kernel(const int4* __restrict__ out_coors, int64_t n_out, int kvol,
                                                         SpConvGeom g, SpDims id, SpIndex in_ix,
                                                         int32_t* __restrict__ nbr) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_out * kvol) return;
    const int k = (int)(t / n_out);
    const int64_t r = t - (int64_t)k * n_out;
    const int4 c = out_coors[r];
    const int kx = k % g.kx, ky = (k / g.kx) % g.ky, kz = k / (g.kx * g.ky);
    const int z = c.y * g.sz - g.pz + kz, y = c.z * g.sy - g.py + ky, x = c.w * g.sx - g.px + kx;
    int32_t v = -1;
    if ((unsigned)z < (unsigned)id.D && (unsigned)y < (unsigned)id.H && (unsigned)x < (unsigned)id.W)
        v = sp_lookup(in_ix, sp_key(id, c.x, z, y, x));
    nbr[t] = v;
}
// transposed map: nbr_t[k][j] = output row r with nbr[k][r] == j
__global__ __launch_bounds__(256) void sp_rulebook_t_kernel(const int4* __restrict__ in_coors, int64_t n_in, int kvol,
                                                           SpConvGeom g, SpDims od, SpIndex out_ix,
                                                           int32_t* __restrict__ nbr_t) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_in * kvol) return;
    const int k = (int)(t / n_in);
    const int64_t j = t - (int64_t)k * n_in;
    const int4 c = in_coors[j];
    int oz, oy, ox;
    int32_t v = -1;
    if (sp_out_coord(g, od, c.y, c.z, c.w, k, oz, oy, ox)) v = sp_lookup(out_ix, sp_key(od, c.x, oz, oy, ox));
    nbr_t[t] = v;
}

extern "C" int gga_sparse_rulebook(const int32_t* out_coors, int64_t n_out, const int32_t* in_coors, int64_t n_in,
                                   int B, const int32_t in_dhw[3], const int32_t out_dhw[3], const int32_t kernel[3],
                                   const int32_t stride[3], const int32_t pad[3], const void* in_index,
                                   int64_t in_index_n, const void* out_index, int64_t out_index_n, int32_t* nbr,
                                   int32_t* nbr_t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(out_coors && in_coors && in_dhw && out_dhw && kernel && stride && pad && in_index && nbr,
                "gga_sparse_rulebook: null pointer argument");
    GGA_REQUIRE(n_out >= 1 && n_in >= 1, "gga_sparse_rulebook: empty level");
    GGA_REQUIRE(!nbr_t || out_index, "gga_sparse_rulebook: the transposed map needs the output index");
    const int kvol = kernel[0] * kernel[1] * kernel[2];
    const SpConvGeom g = { kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2] };
    const SpDims id = { B, in_dhw[0], in_dhw[1], in_dhw[2] };
    const SpDims od = { B, out_dhw[0], out_dhw[1], out_dhw[2] };
    SpIndex in_ix = sp_index_view((void*)in_index, in_index_n);   // *_index_n = the n the index was sized for
    hipLaunchKernelGGL(sp_rulebook_kernel, dim3((unsigned)((n_out * kvol + 255) / 256)), dim3(256), 0, stream,
                       (const int4*)out_coors, n_out, kvol, g, id, in_ix, nbr);
    GGA_CHECK_LAUNCH("sp_rulebook_kernel");
    if (nbr_t) {
        SpIndex out_ix = sp_index_view((void*)out_index, out_index_n);
        hipLaunchKernelGGL(sp_rulebook_t_kernel, dim3((unsigned)((n_in * kvol + 255) / 256)), dim3(256), 0, stream,
                           (const int4*)in_coors, n_in, kvol, g, od, out_ix, nbr_t);
        GGA_CHECK_LAUNCH("sp_rulebook_t_kernel");
    }
    return GGA_OK;
}

__global__ __launch_bounds__(256) void sp_rowmask_kernel(const int32_t* __restrict__ map, int64_t n, int kvol,
                                                        uint32_t* __restrict__ mask) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint32_t m = 0;
    for (int k = 0; k < kvol && k < 32; ++k) m |= (map[(int64_t)k * n + r] >= 0 ? 1u : 0u) << k;
    mask[r] = m;
}

extern "C" int gga_sparse_rowmask(const int32_t* map, int64_t n_rows, int kvol, uint32_t* mask, void* stream) {
    GGA_REQUIRE(map && mask && n_rows >= 1 && kvol >= 1, "gga_sparse_rowmask: bad arguments");
    hipLaunchKernelGGL(sp_rowmask_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, map,
                       n_rows, kvol, mask);
    GGA_CHECK_LAUNCH("sp_rowmask_kernel");
    return GGA_OK;
}

// The mask-sorted processing order of a rule book: rows in ascending order of their offset mask, ties in row order (the stable
// sort the kernels' tile skip and the tests rely on). Round 5: an LSD radix sort over the kvol mask bits (rocPRIM
// radix_sort_pairs, a counting iterator as the values: 8-10 launches) instead of the framework's stable sort of int32 keys,
// which is a merge sort on this stack - 22 launches of merge-path kernels per rule book, 1.3-1.8 ms of the sparse config's step.
// A stable sort has one answer, so the order is the same.
// (rocPRIM's default hands anything up to 2^20 items to its merge sort - the 42 launches per rule book this entry point exists
// to avoid; with the limit at 4096 a level goes through Onesweep: one histogram and one pass per 8 mask bits)
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096> sp_order_config;
static size_t sp_order_temp_bytes(int64_t n_rows, int bits) {
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs<sp_order_config>(nullptr, tb, (const int32_t*)nullptr, (int32_t*)nullptr, rocprim::counting_iterator<int32_t>(0),
                                    (int32_t*)nullptr, (size_t)n_rows, 0, (unsigned)bits, (hipStream_t)0);
    return (tb + 255) / 256 * 256;
}

extern "C" size_t gga_sparse_mask_order_workspace_bytes(int64_t n_rows) {
    if (n_rows < 1) return 0;
    return sp_order_temp_bytes(n_rows, 32) + ((size_t)n_rows * sizeof(int32_t) + 255) / 256 * 256;      // temporaries + the sorted keys
}

extern "C" int gga_sparse_mask_order(const uint32_t* mask, int64_t n_rows, int kvol, int32_t* order, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    GGA_REQUIRE(mask && order && workspace && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && kvol >= 1 && kvol <= 32,
                "gga_sparse_mask_order: bad arguments");
    if (workspace_bytes < gga_sparse_mask_order_workspace_bytes(n_rows)) {
        gga_set_error("gga_sparse_mask_order: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    // (kvol == 32: bit 31 is the sign of the int32 the mask is kept in; the keys are compared as signed, like the sort this replaces)
    size_t tb = sp_order_temp_bytes(n_rows, kvol);
    int32_t* keys_out = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(workspace) + sp_order_temp_bytes(n_rows, 32));
    const hipError_t e = rocprim::radix_sort_pairs<sp_order_config>(workspace, tb, reinterpret_cast<const int32_t*>(mask), keys_out,
                                                   rocprim::counting_iterator<int32_t>(0), order, (size_t)n_rows, 0, (unsigned)kvol,
                                                   (hipStream_t)stream);
    if (e != hipSuccess) {
        gga_set_error("gga_sparse_mask_order: rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
        return GGA_ERR_LAUNCH;
    }
    return GGA_OK;
}

// Rows of a level along a Z-order curve per sample (the halo form's tiles, sparse.py: morton_order): key = sample, then the
// bit-interleaved (z, y, x) - 17 elementwise launches and a merge sort of int64 keys in the framework; here one key kernel and an
// Onesweep radix sort over the bits the level's extent can set. Coordinates are distinct, so the order is THE ascending order of
// the keys whatever sorts them.
__device__ __forceinline__ uint64_t sp_spread3(uint32_t v) {          // bits of v (< 2^16) moved to every third position
    uint64_t x = v & 0xFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(256) void sp_morton_key_kernel(const int4* __restrict__ coors, int64_t n, int cbits,
                                                           uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = coors[i];                                            // (sample, z, y, x)
    keys[i] = ((uint64_t)(uint32_t)c.x << (3 * cbits)) | (sp_spread3((uint32_t)c.y) << 2) | (sp_spread3((uint32_t)c.z) << 1) |
              sp_spread3((uint32_t)c.w);
}
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096> sp_morton_config;
static size_t sp_morton_temp_bytes(int64_t n, int bits) {
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs<sp_morton_config>(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                                      rocprim::counting_iterator<int32_t>(0), (int32_t*)nullptr, (size_t)n, 0,
                                                      (unsigned)bits, (hipStream_t)0);
    return (tb + 255) / 256 * 256;
}
static inline size_t sp_morton_keys_bytes(int64_t n) { return ((size_t)n * 8 + 255) / 256 * 256; }

extern "C" size_t gga_sparse_morton_order_workspace_bytes(int64_t n_rows) {
    if (n_rows < 1) return 0;
    return sp_morton_temp_bytes(n_rows, 64) + 2 * sp_morton_keys_bytes(n_rows);
}

extern "C" int gga_sparse_morton_order(const int32_t* coors, int64_t n_rows, int batch_size, int max_extent, int32_t* order,
                                       void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(coors && order && workspace && n_rows >= 1 && n_rows < ((int64_t)1 << 31) && batch_size >= 1 &&
                max_extent >= 1 && max_extent <= 65536, "gga_sparse_morton_order: bad arguments");
    if (workspace_bytes < gga_sparse_morton_order_workspace_bytes(n_rows)) {
        gga_set_error("gga_sparse_morton_order: workspace too small");
        return GGA_ERR_WORKSPACE;
    }
    int cbits = 1, bbits = 1;
    while ((1 << cbits) < max_extent) ++cbits;                          // bits of a coordinate: 3 * cbits interleaved bits
    while ((1 << bbits) < batch_size) ++bbits;
    const int bits = 3 * cbits + bbits;
    unsigned char* w = static_cast<unsigned char*>(workspace) + sp_morton_temp_bytes(n_rows, 64);
    uint64_t* keys = reinterpret_cast<uint64_t*>(w);
    uint64_t* keys_out = reinterpret_cast<uint64_t*>(w + sp_morton_keys_bytes(n_rows));
    hipLaunchKernelGGL(sp_morton_key_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, (const int4*)coors,
                       n_rows, cbits, keys);
    GGA_CHECK_LAUNCH("sp_morton_key_kernel");
    size_t tb = sp_morton_temp_bytes(n_rows, bits);
    const hipError_t e = rocprim::radix_sort_pairs<sp_morton_config>(workspace, tb, (const uint64_t*)keys, keys_out,
                                                                     rocprim::counting_iterator<int32_t>(0), order,
                                                                     (size_t)n_rows, 0, (unsigned)bits, stream);
    if (e != hipSuccess) {
        gga_set_error("gga_sparse_morton_order: rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
        return GGA_ERR_LAUNCH;
    }
    return GGA_OK;
}
