// a1/a2: batched hard voxelization + HardSimpleVFE for gfx950.
//
// Reference semantics (mmdet3d/core/voxel/voxel_generator.py:137-208, the in-tree
// statement of mmcv.ops.Voxelization; call site mvx_two_stage_gga.py:211-236) are
// sequential: voxel id = order of the voxel's first point, slot = order of the point
// inside its voxel. The parallel formulation below reproduces that order exactly:
//
//   K1 insert   one thread per point: cell key -> open-addressing hash (64-bit CAS);
//               per slot: 64-bit atomicMin of (round, point index) and a counter.
//               After K1, the low word of `cur[slot]` is the voxel's FIRST point.
//   K2 flag     rank[i] = 0 if i is its voxel's first point, -1 otherwise.
//   K3 assign   (a) ordered exclusive scan of the "is first" flags = voxel id in
//                   first-come order (two-level: per-block counts, then block prefix +
//                   ballot scan); ids >= max_voxels are dropped;
//               (b) slot of a point inside its voxel = number of the voxel's points with a
//                   smaller index, counted over the voxel's bucket (K3c / K3d below).
//   K4 write    one thread per point: copy the point to voxels[vid, rank], the rank-0
//               point also writes coors / num_points.
//
// No dense (D,H,W) index grid (360 MB for the KITTI config) as the CPU reference uses.
#include "gga_common.h"

#define VOX_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define VOX_ROUND_HI 0x7FFFFFFFu

struct FrameOffsets {
    int32_t off[GGA_MAX_BATCH + 1];
    const int32_t* cnt;     // optional device-side point counts (frames stored at capacity offsets)
    __device__ __forceinline__ int n(int b) const {
        const int cap = off[b + 1] - off[b];
        if (!cnt) return cap;
        const int c = cnt[b];
        return c < 0 ? 0 : (c < cap ? c : cap);
    }
};

struct VoxGeom {
    float vs[3];
    float lo[3];
    int32_t grid[3];  // x, y, z
    int32_t max_points, max_voxels;
};

struct VoxWorkspace {
    unsigned long long* cur;   // [cap] (round, first/min index)
    unsigned long long* keys;  // [cap]
    int32_t* count;            // [cap]
    int32_t* vid;              // [cap]
    int32_t* slot;             // [total]
    int32_t* rank;             // [total]
    int32_t* act0;             // [total]
    int32_t* act1;             // [total]
    uint32_t cap_mask;
};

static inline uint64_t vox_cap(int64_t total_points) {
    uint64_t cap = 1024;
    while (cap < (uint64_t)(2 * total_points + 2)) cap <<= 1;
    return cap;
}

__device__ __forceinline__ uint32_t vox_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

__global__ __launch_bounds__(256) void vox_insert_kernel(const float* __restrict__ points, int ndim,
                                                        FrameOffsets fo, VoxGeom g, VoxWorkspace ws) {
    const int b = blockIdx.y;
    const int n = fo.n(b);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t gi = (int64_t)fo.off[b] + i;
    const float* p = points + gi * ndim;
    int32_t c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // floor((p - lo) / vs) in f32, true division (voxel_generator.py:189)
        float cf = floorf(__fdiv_rn(__fsub_rn(p[j], g.lo[j]), g.vs[j]));
        ok = ok && (cf >= 0.0f) && (cf < (float)g.grid[j]);
        c[j] = (int32_t)cf;
    }
    if (!ok) {
        ws.slot[gi] = -1;
        ws.rank[gi] = -2;
        return;
    }
    unsigned long long key =
        (((unsigned long long)b * g.grid[2] + c[2]) * g.grid[1] + c[1]) * g.grid[0] + c[0];
    uint32_t h = vox_hash(key) & ws.cap_mask;
    while (true) {
        unsigned long long prev = atomicCAS(&ws.keys[h], VOX_EMPTY_KEY, key);
        if (prev == VOX_EMPTY_KEY || prev == key) break;
        h = (h + 1) & ws.cap_mask;
    }
    atomicMin(&ws.cur[h], ((unsigned long long)VOX_ROUND_HI << 32) | (uint32_t)i);
    atomicAdd(&ws.count[h], 1);
    ws.slot[gi] = (int32_t)h;
}

__global__ __launch_bounds__(256) void vox_flag_kernel(FrameOffsets fo, VoxWorkspace ws) {
    const int b = blockIdx.y;
    const int n = fo.n(b);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t gi = (int64_t)fo.off[b] + i;
    const int32_t h = ws.slot[gi];
    if (h < 0) return;
    ws.rank[gi] = ((uint32_t)ws.cur[h] == (uint32_t)i) ? 0 : -1;
}

// K3, fully parallel (was: one 1024-thread workgroup per frame doing the ordered scan chunk by chunk and
// up to max_points - 1 rounds of global atomicMin: 240 us on 16 of 256 CUs at 16 frames x 20 k points).
//   K3a count    per (frame, 1024-point block): number of first-point flags.
//   K3b assign   same grid: block prefix (sum of the earlier blocks' counts: <= a few dozen values) + ballot scan
//                = voxel id in first-come order; kept voxels also get a region of count[h] entries of the
//                frame's bucket array (atomic cursor: any layout will do).
//   K3c fill     every point of a kept voxel drops its index into the voxel's region (unordered).
//   K3d rank     rank of a point = number of smaller indices in its voxel's region, stopping at max_points -
//                O(points per voxel) per point, no rounds, no order dependence.
// `cur[h]` is free after K2 and is reused as (region start, fill counter).
struct BlkOffsets { int32_t off[GGA_MAX_BATCH + 1]; };      // first 1024-point block of every frame (kernel argument, by value)

__global__ __launch_bounds__(1024) void vox_count_kernel(FrameOffsets fo, VoxWorkspace ws, BlkOffsets bo,
                                                        int32_t* __restrict__ blk_cnt) {
    const int32_t* blk_off = bo.off;
    const int b = blockIdx.y;
    if ((int)blockIdx.x >= blk_off[b + 1] - blk_off[b]) return;
    const int n = fo.n(b), i = blockIdx.x * 1024 + threadIdx.x;
    const bool flag = (i < n) && (ws.rank[fo.off[b] + i] == 0);
    __shared__ int wave_tot[16];
    const unsigned long long bal = __ballot(flag);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += wave_tot[w];
        blk_cnt[blk_off[b] + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void vox_assign_kernel(FrameOffsets fo, VoxGeom g, VoxWorkspace ws,
                                                         BlkOffsets bo,
                                                         const int32_t* __restrict__ blk_cnt, int32_t* __restrict__ cursor,
                                                         int32_t* __restrict__ voxel_num) {
    const int32_t* blk_off = bo.off;
    const int b = blockIdx.y;
    const int nblk = blk_off[b + 1] - blk_off[b];
    if ((int)blockIdx.x >= nblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = fo.off[b], n = fo.n(b);
    __shared__ int wave_tot[16];
    __shared__ int running_s;
    if (wave == 0) {                                   // sum of the earlier blocks' counts
        int s = 0;
        for (int x = lane; x < (int)blockIdx.x; x += 64) s += blk_cnt[blk_off[b] + x];
        s = wave_sum(s);
        if (lane == 0) running_s = s;
    }
    const int i = blockIdx.x * 1024 + tid;
    const bool flag = (i < n) && (ws.rank[base + i] == 0);
    const unsigned long long bal = __ballot(flag);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int wpre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wave_tot[w];
        wpre += (w < wave) ? t : 0;
        tot += t;
    }
    const int v = running_s + wpre + pre;
    if (flag) {
        const int32_t h = ws.slot[base + i];
        if (v < g.max_voxels) {
            ws.vid[h] = v;
            const int c = ws.count[h];
            const int region = (g.max_points > 1 && c > 1) ? atomicAdd(&cursor[b], c) : 0;
            ws.cur[h] = (unsigned long long)(uint32_t)region;        // (region start, fill = 0)
        } else {
            ws.vid[h] = -1;               // voxel beyond max_voxels: dropped with all its points
            ws.rank[base + i] = -1;
        }
    }
    if ((int)blockIdx.x == nblk - 1 && tid == 0) {
        const int total = running_s + tot;
        voxel_num[b] = total < g.max_voxels ? total : g.max_voxels;
    }
}

__global__ __launch_bounds__(256) void vox_fill_kernel(FrameOffsets fo, VoxWorkspace ws) {
    const int b = blockIdx.y;
    const int n = fo.n(b), i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t gi = (int64_t)fo.off[b] + i;
    const int32_t h = ws.slot[gi];
    if (h < 0 || ws.vid[h] < 0 || ws.count[h] <= 1) return;
    int32_t* rf = reinterpret_cast<int32_t*>(&ws.cur[h]);
    const int pos = rf[0] + atomicAdd(&rf[1], 1);
    ws.act0[fo.off[b] + pos] = i;
}

__global__ __launch_bounds__(256) void vox_rank_kernel(FrameOffsets fo, VoxGeom g, VoxWorkspace ws) {
    const int b = blockIdx.y;
    const int n = fo.n(b), i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t gi = (int64_t)fo.off[b] + i;
    if (ws.rank[gi] != -1) return;                      // first points (0) and out-of-range points (-2) are settled
    const int32_t h = ws.slot[gi];
    if (ws.vid[h] < 0) return;                          // dropped voxel
    const int c = ws.count[h];
    const int32_t* bucket = ws.act0 + fo.off[b] + reinterpret_cast<const int32_t*>(&ws.cur[h])[0];
    int r = 0;
    for (int j = 0; j < c && r < g.max_points; ++j) r += bucket[j] < i;
    if (r < g.max_points) ws.rank[gi] = r;
}

__global__ __launch_bounds__(256) void vox_write_kernel(const float* __restrict__ points, int ndim,
                                                       FrameOffsets fo, int batch, VoxGeom g, VoxWorkspace ws,
                                                       const int32_t* __restrict__ voxel_num_in,
                                                       int32_t* __restrict__ voxel_total,
                                                       float* __restrict__ voxels, int32_t* __restrict__ coors,
                                                       int32_t* __restrict__ num_points) {
    const int b = blockIdx.y;
    __shared__ int vbase_s;
    if (threadIdx.x == 0) {
        int s = 0;
        for (int bb = 0; bb < b; ++bb) s += voxel_num_in[bb];
        vbase_s = s;
        if (blockIdx.x == 0 && b == batch - 1) *voxel_total = s + voxel_num_in[b];
    }
    __syncthreads();
    const int n = fo.n(b);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t gi = (int64_t)fo.off[b] + i;
    const int r = ws.rank[gi];
    if (r < 0) return;
    const int32_t h = ws.slot[gi];
    const int64_t row = (int64_t)vbase_s + ws.vid[h];
    const float* p = points + gi * ndim;
    float* dst = voxels + (row * g.max_points + r) * ndim;
    if (ndim == 4) {
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(p);
    } else {
        for (int j = 0; j < ndim; ++j) dst[j] = p[j];
    }
    if (r == 0) {
        unsigned long long key = ws.keys[h];
        const int cx = (int)(key % (unsigned)g.grid[0]); key /= (unsigned)g.grid[0];
        const int cy = (int)(key % (unsigned)g.grid[1]); key /= (unsigned)g.grid[1];
        const int cz = (int)(key % (unsigned)g.grid[2]);
        reinterpret_cast<int4*>(coors)[row] = make_int4(b, cz, cy, cx);
        const int cnt = ws.count[h];
        num_points[row] = cnt < g.max_points ? cnt : g.max_points;
    }
}

// ---------------------------------------------------------------------------------
extern "C" void gga_voxel_grid_size(const gga_voxel_params* prm, int32_t grid_xyz[3]) {
    for (int j = 0; j < 3; ++j) {
        float gsz = (prm->pc_range[3 + j] - prm->pc_range[j]) / prm->voxel_size[j];
        grid_xyz[j] = (int32_t)__builtin_rintf(gsz);
    }
}

extern "C" size_t gga_hard_voxelize_workspace_bytes(int batch, int64_t total_points) {
    const uint64_t cap = vox_cap(total_points);
    size_t bytes = cap * (8 + 8 + 4 + 4);
    bytes += gga_align_up((size_t)total_points * 4, 256) * 4;
    bytes += gga_align_up(((size_t)total_points / 1024 + 3 * (size_t)batch + 16) * 4, 256);     // block offsets / counts, cursors
    return bytes + 1024;
}

static int hard_voxelize_impl(const float* points, int ndim, const int64_t* offsets_host, const int32_t* counts_dev,
                              int batch, const gga_voxel_params* prm, float* voxels, int32_t* coors,
                              int32_t* num_points, int32_t* voxel_num, void* workspace, size_t workspace_bytes,
                              void* stream_);

extern "C" int gga_hard_voxelize_batch(const float* points, int ndim, const int64_t* offsets_host, int batch,
                                       const gga_voxel_params* prm, float* voxels, int32_t* coors,
                                       int32_t* num_points, int32_t* voxel_num, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
    return hard_voxelize_impl(points, ndim, offsets_host, nullptr, batch, prm, voxels, coors, num_points, voxel_num,
                              workspace, workspace_bytes, stream_);
}

extern "C" int gga_hard_voxelize_prepared(const float* points, int ndim, const int64_t* capacity_offsets_host,
                                          const int32_t* counts_dev, int batch, const gga_voxel_params* prm,
                                          float* voxels, int32_t* coors, int32_t* num_points, int32_t* voxel_num,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    GGA_REQUIRE(counts_dev, "gga_hard_voxelize_prepared: null pointer argument");
    return hard_voxelize_impl(points, ndim, capacity_offsets_host, counts_dev, batch, prm, voxels, coors, num_points,
                              voxel_num, workspace, workspace_bytes, stream_);
}

static int hard_voxelize_impl(const float* points, int ndim, const int64_t* offsets_host, const int32_t* counts_dev,
                              int batch, const gga_voxel_params* prm, float* voxels, int32_t* coors,
                              int32_t* num_points, int32_t* voxel_num, void* workspace, size_t workspace_bytes,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(points && offsets_host && prm && voxels && coors && num_points && voxel_num && workspace,
                "gga_hard_voxelize_batch: null pointer argument");
    GGA_REQUIRE(batch >= 1 && batch <= GGA_MAX_BATCH, "gga_hard_voxelize_batch: batch %d not in [1, %d]", batch,
                GGA_MAX_BATCH);
    GGA_REQUIRE(ndim >= 3 && ndim <= 16, "gga_hard_voxelize_batch: ndim %d not in [3, 16]", ndim);
    GGA_REQUIRE(prm->max_points >= 1 && prm->max_voxels >= 1, "gga_hard_voxelize_batch: max_points/max_voxels < 1");
    const int64_t total = offsets_host[batch];
    GGA_REQUIRE(offsets_host[0] == 0 && total >= 0 && total < (1ll << 30),
                "gga_hard_voxelize_batch: offsets must start at 0 and total points < 2^30");
    FrameOffsets fo;
    fo.cnt = counts_dev;
    int max_n = 0;
    for (int b = 0; b <= batch; ++b) {
        fo.off[b] = (int32_t)offsets_host[b];
        if (b > 0) {
            GGA_REQUIRE(offsets_host[b] >= offsets_host[b - 1], "gga_hard_voxelize_batch: offsets not monotone");
            const int nb = (int)(offsets_host[b] - offsets_host[b - 1]);
            max_n = nb > max_n ? nb : max_n;
        }
    }
    if (gga_hard_voxelize_workspace_bytes(batch, total) > workspace_bytes) {
        gga_set_error("gga_hard_voxelize_batch: workspace %zu B < required %zu B", workspace_bytes,
                      gga_hard_voxelize_workspace_bytes(batch, total));
        return GGA_ERR_WORKSPACE;
    }
    VoxGeom g;
    for (int j = 0; j < 3; ++j) { g.vs[j] = prm->voxel_size[j]; g.lo[j] = prm->pc_range[j]; }
    gga_voxel_grid_size(prm, g.grid);
    GGA_REQUIRE(g.grid[0] > 0 && g.grid[1] > 0 && g.grid[2] > 0, "gga_hard_voxelize_batch: empty grid");
    g.max_points = prm->max_points;
    g.max_voxels = prm->max_voxels;

    const uint64_t cap = vox_cap(total);
    VoxWorkspace ws;
    char* w = (char*)workspace;
    ws.cur = (unsigned long long*)w;  w += cap * 8;
    ws.keys = (unsigned long long*)w; w += cap * 8;
    ws.count = (int32_t*)w;           w += cap * 4;
    ws.vid = (int32_t*)w;             w += cap * 4;
    const size_t per = gga_align_up((size_t)total * 4, 256);
    ws.slot = (int32_t*)w; w += per;
    ws.rank = (int32_t*)w; w += per;
    ws.act0 = (int32_t*)w; w += per;
    ws.act1 = (int32_t*)w; w += per;
    ws.cap_mask = (uint32_t)(cap - 1);
    // small tables: per-frame block offsets [batch + 1] (host -> kernel argument copy), block counts, cursors [batch]
    int32_t* small = (int32_t*)w;
    BlkOffsets bo;
    bo.off[0] = 0;
    int max_blk = 0;
    for (int b = 0; b < batch; ++b) {
        const int nb = (fo.off[b + 1] - fo.off[b] + 1023) / 1024;
        bo.off[b + 1] = bo.off[b] + nb;
        max_blk = nb > max_blk ? nb : max_blk;
    }
    int32_t* cursor = small;                      // [batch]
    int32_t* blk_cnt = cursor + batch;            // [sum of blocks]

    const size_t cap_rows = (size_t)batch * prm->max_voxels;
    GGA_CHECK_HIP(hipMemsetAsync(ws.cur, 0xFF, cap * 16, stream), "voxelize memset(hash)");
    GGA_CHECK_HIP(hipMemsetAsync(ws.count, 0, cap * 4, stream), "voxelize memset(count)");
    GGA_CHECK_HIP(hipMemsetAsync(voxels, 0, cap_rows * prm->max_points * ndim * sizeof(float), stream),
                  "voxelize memset(voxels)");
    GGA_CHECK_HIP(hipMemsetAsync(coors, 0, cap_rows * 4 * sizeof(int32_t), stream), "voxelize memset(coors)");
    GGA_CHECK_HIP(hipMemsetAsync(num_points, 0, cap_rows * sizeof(int32_t), stream), "voxelize memset(num_points)");
    if (total == 0 || max_n == 0) {
        GGA_CHECK_HIP(hipMemsetAsync(voxel_num, 0, (batch + 1) * sizeof(int32_t), stream), "voxelize memset(voxel_num)");
        return GGA_OK;
    }
    dim3 grid((max_n + 255) / 256, batch);
    hipLaunchKernelGGL(vox_insert_kernel, grid, dim3(256), 0, stream, points, ndim, fo, g, ws);
    GGA_CHECK_LAUNCH("vox_insert_kernel");
    hipLaunchKernelGGL(vox_flag_kernel, grid, dim3(256), 0, stream, fo, ws);
    GGA_CHECK_LAUNCH("vox_flag_kernel");
    GGA_CHECK_HIP(hipMemsetAsync(cursor, 0, batch * sizeof(int32_t), stream), "voxelize memset(cursor)");
    GGA_CHECK_HIP(hipMemsetAsync(voxel_num, 0, (batch + 1) * sizeof(int32_t), stream), "voxelize memset(voxel_num)");     // frames without a block
    const dim3 bgrid(max_blk, batch);
    hipLaunchKernelGGL(vox_count_kernel, bgrid, dim3(1024), 0, stream, fo, ws, bo, blk_cnt);
    GGA_CHECK_LAUNCH("vox_count_kernel");
    hipLaunchKernelGGL(vox_assign_kernel, bgrid, dim3(1024), 0, stream, fo, g, ws, bo, blk_cnt, cursor, voxel_num);
    GGA_CHECK_LAUNCH("vox_assign_kernel");
    if (g.max_points > 1) {
        hipLaunchKernelGGL(vox_fill_kernel, grid, dim3(256), 0, stream, fo, ws);
        GGA_CHECK_LAUNCH("vox_fill_kernel");
        hipLaunchKernelGGL(vox_rank_kernel, grid, dim3(256), 0, stream, fo, g, ws);
        GGA_CHECK_LAUNCH("vox_rank_kernel");
    }
    hipLaunchKernelGGL(vox_write_kernel, grid, dim3(256), 0, stream, points, ndim, fo, batch, g, ws, voxel_num,
                       voxel_num + batch, voxels, coors, num_points);
    GGA_CHECK_LAUNCH("vox_write_kernel");
    return GGA_OK;
}

// a2. HardSimpleVFE (voxel_encoder.py:43-45): one thread per (voxel, feature).
__global__ __launch_bounds__(256) void voxel_mean_kernel(const float* __restrict__ voxels,
                                                        const int32_t* __restrict__ num_points, int64_t m,
                                                        int P, int ndim, int nf, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= m * nf) return;
    const int64_t v = t / nf;
    const int f = (int)(t - v * nf);
    const float* src = voxels + v * P * ndim + f;
    float s = 0.0f;
    for (int p = 0; p < P; ++p) s += src[p * ndim];   // same left-to-right order as sum(dim=1)
    out[t] = __fdiv_rn(s, (float)num_points[v]);
}

extern "C" int gga_voxel_mean(const float* voxels, const int32_t* num_points, int64_t m, int max_points, int ndim,
                              int num_features, float* out, void* stream) {
    GGA_REQUIRE(voxels && num_points && out, "gga_voxel_mean: null pointer argument");
    GGA_REQUIRE(m >= 0 && max_points >= 1 && num_features >= 1 && num_features <= ndim,
                "gga_voxel_mean: bad sizes (m=%lld P=%d ndim=%d nf=%d)", (long long)m, max_points, ndim, num_features);
    if (m == 0) return GGA_OK;
    const int64_t total = m * num_features;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       voxels, num_points, m, max_points, ndim, num_features, out);
    GGA_CHECK_LAUNCH("voxel_mean_kernel");
    return GGA_OK;
}
