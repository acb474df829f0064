// Offline GGA label generation primitives, tools/data_converter/utils_gga.py:
//   region_grow                utils_gga.py:6-38    in-box point clusters by region growing
//   points_in_frustm_indices   utils_gga.py:87-100  -> box_np_ops.points_in_convex_polygon_3d_jit
//                              (mmdet3d/core/bbox/box_np_ops.py:641-705): points inside convex polyhedra
//   calculate_ground           utils_gga.py:103-133 RANSAC ground plane: scoring of candidate planes
// All arithmetic is float64 in the reference's operation order (no fma contraction), so the masks
// are bit-identical to numpy's.
#include "gga_common.h"

#define NOFMA(x) asm volatile("" : "+v"(x))

// ------------------------------------------------------------------------------ region growing
// One 1024-thread workgroup runs one region_grow call (blockIdx.x = threshold index: the seven
// thresholds the reference tries per object share their masks and are independent). The breadth
// first growth is sequential by definition - the early exit on the in-box ratio is checked after
// every dequeued point - so pops are processed one at a time, each evaluated by the whole
// workgroup: distance of the popped point to every search point (sqrt of the left-to-right sum of
// squares, as np.linalg.norm), first-minimum argmin, then the not-yet-reached points closer than
// the threshold are appended to the queue IN INDEX ORDER (thread t owns a contiguous slice of the
// search list; a workgroup scan of the per-thread counts gives the append positions).
#define RG_THREADS 1024

struct RgWork {          // per threshold, all in global memory
    int32_t* sidx;       // [N]  search list: point index of search entry s
    int32_t* queue;      // [N]  FIFO of search entries
    uint8_t* smask;      // [N]  seed_mask over search entries
    uint8_t* mask;       // [N]  remaining origin points
    uint8_t* best;       // [N]  seed_mask_all of the best region so far
    uint8_t* cur;        // [N]  seed_mask_all of the region being grown
};

__device__ __forceinline__ int rg_block_scan_excl(int v, int* total, int* lds) {
    // exclusive scan of one int per thread over 1024 threads; lds: >= 17 ints
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < RG_THREADS / 64; ++w) { const int t = lds[w]; lds[w] = s; s += t; } lds[16] = s; }
    __syncthreads();
    const int base = lds[wave];
    *total = lds[16];
    __syncthreads();
    return base + x - v;
}

#define RG_CACHE 32
// sum of squared differences, left to right as np.linalg.norm's reduction (its sqrt is taken by the caller)
__device__ __forceinline__ double rg_sq(const double* __restrict__ a, const double* __restrict__ b, int dim) {
    double s = 0.0;
    for (int j = 0; j < dim; ++j) {
        const double d = a[j] - b[j];
        double q = d * d;
        NOFMA(q);
        s = j == 0 ? q : s + q;
    }
    return s;
}

__global__ __launch_bounds__(RG_THREADS) void region_grow_kernel(const double* __restrict__ pc, int n, int dim,
                                                                const uint8_t* __restrict__ mask_search,
                                                                const uint8_t* __restrict__ mask_origin,
                                                                const double* __restrict__ thresholds, double ratio,
                                                                int use_ratio, char* __restrict__ workspace,
                                                                size_t work_stride, uint8_t* __restrict__ out) {
    const int tid = threadIdx.x;
    const double thresh = thresholds[blockIdx.x];
    char* w = workspace + (size_t)blockIdx.x * work_stride;
    const size_t a4 = gga_align_up((size_t)n * 4, 256), a1 = gga_align_up((size_t)n, 256);
    RgWork wk;
    wk.sidx = (int32_t*)w; w += a4;
    wk.queue = (int32_t*)w; w += a4;
    wk.smask = (uint8_t*)w; w += a1;
    wk.mask = (uint8_t*)w; w += a1;
    wk.best = (uint8_t*)w; w += a1;
    wk.cur = (uint8_t*)w;
    uint8_t* res = out + (size_t)blockIdx.x * n;
    __shared__ int lds[32];
    __shared__ double red_d[RG_THREADS / 64];
    __shared__ int red_i[RG_THREADS / 64];
    __shared__ int sh_i[4];

    // search list in index order (stable compaction of mask_search), mask = mask_origin, best = 0
    const int chunk_n = (n + RG_THREADS - 1) / RG_THREADS;
    const int n0 = min(n, tid * chunk_n), n1 = min(n, n0 + chunk_n);
    int cnt = 0;
    for (int i = n0; i < n1; ++i) cnt += mask_search[i] ? 1 : 0;
    int S;
    int pos = rg_block_scan_excl(cnt, &S, lds);
    for (int i = n0; i < n1; ++i) {
        if (mask_search[i]) wk.sidx[pos++] = i;
        wk.mask[i] = mask_origin[i] ? 1 : 0;
        wk.best[i] = 0;
    }
    __syncthreads();
    if (S == 0) {            // nothing to grow into (numpy's argmin of an empty array raises here)
        for (int i = n0; i < n1; ++i) res[i] = 0;
        return;
    }
    const int chunk_s = (S + RG_THREADS - 1) / RG_THREADS;
    const int s0 = min(S, tid * chunk_s), s1 = min(S, s0 + chunk_s);
    long long best_len = 0;

    while (true) {
        // seed = first point with mask == 1
        int first = n;
        for (int i = n0; i < n1; ++i) if (wk.mask[i]) { first = i; break; }
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        if ((tid & 63) == 0) red_i[tid >> 6] = first;
        __syncthreads();
        if (tid == 0) { int f = n; for (int k = 0; k < RG_THREADS / 64; ++k) f = min(f, red_i[k]); sh_i[0] = f; }
        __syncthreads();
        const int seed = sh_i[0];
        __syncthreads();
        if (seed >= n) break;
        for (int s = s0; s < s1; ++s) wk.smask[s] = 0;
        for (int i = n0; i < n1; ++i) wk.cur[i] = 0;
        __syncthreads();

        int head = 0, tail = 0;            // queue[head..tail): search entries waiting; pop -1 = the seed itself
        long long reached = 0, reached_origin = 0;
        bool first_pop = true, flag = true;
        while (first_pop || head < tail) {
            double temp[4];
            const int src = first_pop ? seed : wk.sidx[wk.queue[head]];
            for (int j = 0; j < dim; ++j) temp[j] = pc[(size_t)src * dim + j];
            if (!first_pop) ++head;
            first_pop = false;
            // squared distances once per pop (kept in registers when a thread owns <= RG_CACHE entries);
            // sqrt is monotone and correctly rounded, so it is only evaluated where it can decide:
            // next to the minimum (first-minimum ties) and next to the threshold
            double sq[RG_CACHE];
            const bool cached = chunk_s <= RG_CACHE;
            double bs = 1.0 / 0.0; int bi = 0x7fffffff;
            if (cached) {
#pragma unroll
                for (int c = 0; c < RG_CACHE; ++c) {
                    const int s = s0 + c;
                    sq[c] = 1.0 / 0.0;
                    if (s < s1) { sq[c] = rg_sq(pc + (size_t)wk.sidx[s] * dim, temp, dim); if (sq[c] < bs) { bs = sq[c]; bi = s; } }
                }
            } else {
                for (int s = s0; s < s1; ++s) {
                    const double v = rg_sq(pc + (size_t)wk.sidx[s] * dim, temp, dim);
                    if (v < bs) { bs = v; bi = s; }
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double od = __shfl_xor(bs, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (od < bs || (od == bs && oi < bi)) { bs = od; bi = oi; }
            }
            if ((tid & 63) == 0) { red_d[tid >> 6] = bs; red_i[tid >> 6] = bi; }
            __syncthreads();
            double smin = red_d[0];
            for (int k = 1; k < RG_THREADS / 64; ++k) smin = fmin(smin, red_d[k]);
            __syncthreads();
            // (a) argmin of sqrt(s): the first entry whose rounded distance equals the minimum
            const double dmin = sqrt(smin), s_tie = smin * (1.0 + 1e-15);
            int ai = 0x7fffffff;
            if (cached) {
#pragma unroll
                for (int c = RG_CACHE - 1; c >= 0; --c)
                    if (s0 + c < s1 && sq[c] <= s_tie && sqrt(sq[c]) == dmin) ai = s0 + c;
            } else {
                for (int s = s0; s < s1; ++s) {
                    const double v = rg_sq(pc + (size_t)wk.sidx[s] * dim, temp, dim);
                    if (v <= s_tie && sqrt(v) == dmin) { ai = s; break; }
                }
            }
            for (int o = 32; o > 0; o >>= 1) ai = min(ai, __shfl_xor(ai, o, 64));
            if ((tid & 63) == 0) red_i[tid >> 6] = ai;
            __syncthreads();
            if (tid == 0) {
                int i = red_i[0];
                for (int k = 1; k < RG_THREADS / 64; ++k) i = min(i, red_i[k]);
                int add = 0, addo = 0;
                if (i < S && !wk.smask[i]) { wk.smask[i] = 1; const int g = wk.sidx[i]; wk.cur[g] = 1; add = 1; addo = mask_origin[g] ? 1 : 0; }
                sh_i[1] = add; sh_i[2] = addo;
            }
            __syncthreads();
            reached += sh_i[1]; reached_origin += sh_i[2];
            // (b) newly reached points: closer than the threshold and not reached before
            const double t_in = thresh * thresh * (1.0 - 1e-15), t_out = thresh * thresh * (1.0 + 1e-15);
            unsigned int near_bits = 0;         // per-thread result bits in the cached case
            int c_new = 0, co = 0;
            if (cached) {
#pragma unroll
                for (int c = 0; c < RG_CACHE; ++c) {
                    const double v = sq[c];                 // +inf past the end of the slice
                    if (v <= t_out && !wk.smask[s0 + c] && (v < t_in || sqrt(v) < thresh)) { ++c_new; near_bits |= 1u << c; }
                }
            } else {
                for (int s = s0; s < s1; ++s) {
                    if (wk.smask[s]) continue;
                    const double v = rg_sq(pc + (size_t)wk.sidx[s] * dim, temp, dim);
                    if (v < t_in || (v <= t_out && sqrt(v) < thresh)) ++c_new;
                }
            }
            int tot;
            int at = tail + rg_block_scan_excl(c_new, &tot, lds);
            if (cached) {
                while (near_bits) {
                    const int c = __ffs(near_bits) - 1;
                    near_bits &= near_bits - 1;
                    const int s = s0 + c;
                    wk.queue[at++] = s; wk.smask[s] = 1;
                    const int g = wk.sidx[s]; wk.cur[g] = 1; co += mask_origin[g] ? 1 : 0;
                }
            } else {
                for (int s = s0; s < s1; ++s) {
                    if (wk.smask[s]) continue;
                    const double v = rg_sq(pc + (size_t)wk.sidx[s] * dim, temp, dim);
                    if (v < t_in || (v <= t_out && sqrt(v) < thresh)) {
                        wk.queue[at++] = s; wk.smask[s] = 1;
                        const int g = wk.sidx[s]; wk.cur[g] = 1; co += mask_origin[g] ? 1 : 0;
                    }
                }
            }
            int toto;
            rg_block_scan_excl(co, &toto, lds);
            tail += tot; reached += tot; reached_origin += toto;
            __syncthreads();
            if (use_ratio && (double)reached_origin / (double)(float)reached < ratio) { flag = false; break; }
        }
        if (flag && reached > best_len) {
            best_len = reached;
            for (int i = n0; i < n1; ++i) wk.best[i] = wk.cur[i];
        }
        for (int i = n0; i < n1; ++i) if (wk.cur[i]) wk.mask[i] = 0;
        // a seed outside the search set is never reached; the reference would loop forever on it
        if (seed >= n0 && seed < n1) wk.mask[seed] = 0;
        __syncthreads();
    }
    for (int i = n0; i < n1; ++i) res[i] = use_ratio ? (wk.best[i] && mask_origin[i] ? 1 : 0) : wk.best[i];
}

extern "C" size_t gga_region_grow_workspace_bytes(int64_t n_points, int n_thresholds) {
    if (n_points < 0 || n_thresholds < 1) return 0;
    return (size_t)n_thresholds * (2 * gga_align_up((size_t)n_points * 4, 256) + 4 * gga_align_up((size_t)n_points, 256));
}

extern "C" int gga_region_grow(const double* pc, int64_t n_points, int dim, const uint8_t* mask_search,
                               const uint8_t* mask_origin, const double* thresholds, int n_thresholds, double ratio,
                               int use_ratio, uint8_t* out_masks, void* workspace, size_t workspace_bytes, void* stream) {
    GGA_REQUIRE(n_points >= 0 && n_points < (1ll << 30) && dim >= 1 && dim <= 4 && n_thresholds >= 1,
                "gga_region_grow: bad sizes (n=%lld dim=%d thresholds=%d; dim <= 4)", (long long)n_points, dim, n_thresholds);
    if (n_points == 0) return GGA_OK;
    GGA_REQUIRE(pc && mask_search && mask_origin && thresholds && out_masks && workspace, "gga_region_grow: null pointer argument");
    const size_t need = gga_region_grow_workspace_bytes(n_points, n_thresholds);
    if (workspace_bytes < need) {
        gga_set_error("gga_region_grow: workspace %zu B < required %zu B", workspace_bytes, need);
        return GGA_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(region_grow_kernel, dim3(n_thresholds), dim3(RG_THREADS), 0, (hipStream_t)stream, pc, (int)n_points, dim,
                       mask_search, mask_origin, thresholds, ratio, use_ratio, (char*)workspace, need / n_thresholds, out_masks);
    GGA_CHECK_LAUNCH("region_grow_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ points in convex polyhedra
// ret[i, j] = all_k ( p_i . normal[j,k] + d[j,k] < 0 ), products summed left to right.
__global__ __launch_bounds__(256) void points_in_polyhedra_kernel(const double* __restrict__ pts, int64_t n, int pstride,
                                                                 const double* __restrict__ normal, const double* __restrict__ d,
                                                                 int n_poly, int n_surf, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = pts[i * pstride], y = pts[i * pstride + 1], z = pts[i * pstride + 2];
    for (int j = 0; j < n_poly; ++j) {
        bool in = true;
        for (int k = 0; k < n_surf; ++k) {
            const double* nv = normal + ((size_t)j * n_surf + k) * 3;
            double a = x * nv[0], b = y * nv[1], c = z * nv[2];
            NOFMA(a); NOFMA(b); NOFMA(c);
            double s = a + b; s = s + c; s = s + d[(size_t)j * n_surf + k];
            if (s >= 0) { in = false; break; }
        }
        out[i * n_poly + j] = in ? 1 : 0;
    }
}

extern "C" int gga_points_in_convex_polyhedra(const double* points, int64_t n_points, int point_stride, const double* normal_vec,
                                              const double* d, int n_polyhedra, int n_surfaces, uint8_t* out, void* stream) {
    GGA_REQUIRE(n_points >= 0 && point_stride >= 3 && n_polyhedra >= 1 && n_surfaces >= 1, "gga_points_in_convex_polyhedra: bad sizes");
    if (n_points == 0) return GGA_OK;
    GGA_REQUIRE(points && normal_vec && d && out, "gga_points_in_convex_polyhedra: null pointer argument");
    hipLaunchKernelGGL(points_in_polyhedra_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       points, n_points, point_stride, normal_vec, d, n_polyhedra, n_surfaces, out);
    GGA_CHECK_LAUNCH("points_in_polyhedra_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ RANSAC plane scoring
// For candidate planes a.p = 1 (utils_gga.py:121-125): diff_i = |p_i . plane - 1| / ||plane||,
// inlier = diff < threshold; counts[c] = number of inliers (the masks are produced on request for
// the winning candidates). np.matmul of an [N,3] by [3] vector sums the products left to right.
__global__ __launch_bounds__(256) void plane_inliers_kernel(const double* __restrict__ pts, int64_t n, int pstride,
                                                           const double* __restrict__ planes, int n_planes, double thresh,
                                                           int32_t* __restrict__ counts, uint8_t* __restrict__ masks) {
    const int c = blockIdx.y;
    const double a = planes[3 * c], b = planes[3 * c + 1], cc = planes[3 * c + 2];
    double a2 = a * a, b2 = b * b, c2 = cc * cc;
    NOFMA(a2); NOFMA(b2); NOFMA(c2);
    const double norm = sqrt((a2 + b2) + c2);
    int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double u = pts[i * pstride] * a, v = pts[i * pstride + 1] * b, w = pts[i * pstride + 2] * cc;
        NOFMA(u); NOFMA(v); NOFMA(w);
        const double diff = fabs(((u + v) + w) - 1.0) / norm;
        const bool in = diff < thresh;
        if (masks) masks[(size_t)c * n + i] = in ? 1 : 0;
        local += in ? 1 : 0;
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counts[c], local);
}

extern "C" int gga_plane_inliers(const double* points, int64_t n_points, int point_stride, const double* planes, int n_planes,
                                 double threshold, int32_t* counts, uint8_t* masks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(n_points >= 0 && point_stride >= 3 && n_planes >= 1, "gga_plane_inliers: bad sizes");
    GGA_REQUIRE(planes && counts && (n_points == 0 || points), "gga_plane_inliers: null pointer argument");
    GGA_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)n_planes * 4, stream), "plane_inliers memset");
    if (n_points == 0) return GGA_OK;
    int bx = (int)((n_points + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(plane_inliers_kernel, dim3(bx, n_planes), dim3(256), 0, stream, points, n_points, point_stride, planes,
                       n_planes, threshold, counts, masks);
    GGA_CHECK_LAUNCH("plane_inliers_kernel");
    return GGA_OK;
}
