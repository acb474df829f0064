// Point-level tail of the GGA train pipeline as one batched device op:
//   ObjectSample_GGA.remove_points_in_boxes_v2   mmdet3d/datasets/pipelines/gga_processing.py:58-68
//       scene points whose BEV distance (float64, sqrt of the sum of squares as scipy's cdist) to
//       any pasted object's centre is < min_distance are dropped
//   points.cat([sampled_points, points])         gga_processing.py:176
//   PointsRangeFilter -> BasePoints.in_range_3d  transforms_3d.py:942-977, base_points.py:203-225
//       strict float32 inequalities on both sides
//   PointShuffle                                 transforms_3d.py:858-883
//       a seeded bijection of the kept rows (the reference draws torch.randperm on the host)
// Frames are independent: blockIdx.y = frame. A frame's virtual row list is [its pasted objects'
// points, its scene points]; rows are compacted in order (stable), so with seed 0 the output
// equals the reference's row for row. Output rows of frame f start at its capacity offset
// (pasted + scene rows before it); the kept count stays on the device for the voxelizer
// (gga_hard_voxelize_prepared) - no host round trip.
#include "gga_common.h"

#define PP_CHUNK 256

struct PrepFrames {
    int32_t cap_off[GGA_MAX_BATCH + 1];   // virtual rows before frame f (= output capacity offset)
    int32_t n_samp[GGA_MAX_BATCH];        // pasted rows of frame f (the first rows of its virtual list)
    int32_t samp_off[GGA_MAX_BATCH];      // row offset into `sampled`
    int32_t scene_off[GGA_MAX_BATCH];     // row offset into `scene`
    int32_t ctr_off[GGA_MAX_BATCH + 1];   // centre offsets
    uint64_t seed[GGA_MAX_BATCH];
};

__device__ __forceinline__ bool pp_keep(const float* __restrict__ p, bool is_scene, const double* __restrict__ ctr,
                                        int n_ctr, double min_distance, const float* __restrict__ rng) {
    const float x = p[0], y = p[1], z = p[2];
    bool keep = (x > rng[0]) & (y > rng[1]) & (z > rng[2]) & (x < rng[3]) & (y < rng[4]) & (z < rng[5]);
    if (keep && is_scene) {
        for (int c = 0; c < n_ctr; ++c) {
            const double dx = (double)x - ctr[2 * c], dy = (double)y - ctr[2 * c + 1];
            double sx = dx * dx, sy = dy * dy;
            asm volatile("" : "+v"(sx), "+v"(sy));      // no fma: s = dx*dx; s += dy*dy as the C reference
            const double d = sqrt(sx + sy);
            if (d < min_distance) { keep = false; break; }
        }
    }
    return keep;
}

__device__ __forceinline__ const float* pp_row(const PrepFrames& fr, int f, int i, const float* scene, const float* sampled,
                                               int ndim, bool* is_scene) {
    if (i < fr.n_samp[f]) { *is_scene = false; return sampled + ((int64_t)fr.samp_off[f] + i) * ndim; }
    *is_scene = true;
    return scene + ((int64_t)fr.scene_off[f] + (i - fr.n_samp[f])) * ndim;
}

// pass 1: keep flag per virtual row + kept rows per 256-row chunk
__global__ __launch_bounds__(PP_CHUNK) void pp_flag_kernel(const float* __restrict__ scene, const float* __restrict__ sampled,
                                                          const double* __restrict__ centers, PrepFrames fr, int ndim,
                                                          double min_distance, const float* __restrict__ rng,
                                                          uint8_t* __restrict__ flags, int32_t* __restrict__ chunk_cnt,
                                                          int chunks_per_frame) {
    const int f = blockIdx.y, i = blockIdx.x * PP_CHUNK + threadIdx.x;
    const int n = fr.cap_off[f + 1] - fr.cap_off[f];
    bool keep = false;
    if (i < n) {
        bool is_scene;
        const float* p = pp_row(fr, f, i, scene, sampled, ndim, &is_scene);
        keep = pp_keep(p, is_scene, centers + 2 * (int64_t)fr.ctr_off[f], fr.ctr_off[f + 1] - fr.ctr_off[f], min_distance, rng);
        flags[fr.cap_off[f] + i] = keep ? 1 : 0;
    }
    __shared__ int wsum[PP_CHUNK / 64];
    const int c = __popcll(__ballot(keep));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < PP_CHUNK / 64; ++w) t += wsum[w];
        chunk_cnt[f * chunks_per_frame + blockIdx.x] = t;
    }
}

// pass 2: exclusive scan of a frame's chunk counts (one workgroup per frame), kept total -> counts[f]
__global__ __launch_bounds__(1024) void pp_scan_kernel(int32_t* __restrict__ chunk_cnt, int chunks_per_frame,
                                                      int32_t* __restrict__ counts) {
    const int f = blockIdx.x, tid = threadIdx.x;
    int32_t* c = chunk_cnt + (int64_t)f * chunks_per_frame;
    __shared__ int part[1024];
    __shared__ int carry;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < chunks_per_frame; base += 1024) {
        const int i = base + tid;
        const int v = i < chunks_per_frame ? c[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan
            const int t = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        if (i < chunks_per_frame) c[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 0) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) counts[f] = carry;
}

// seeded bijection of [0, n): balanced Feistel network over the next even power of two, cycle-walked
__device__ __forceinline__ uint32_t pp_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t pp_permute(uint32_t r, uint32_t n, uint64_t seed) {
    if (n <= 1) return r;
    int half = (32 - __clz(n - 1) + 1) / 2;           // bits per half; domain 2^(2*half) >= n
    if (half < 1) half = 1;
    const uint32_t mask = (1u << half) - 1u;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t x = r;
    do {
        uint32_t L = x >> half, R = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t F = pp_mix(R ^ (round & 1 ? k1 : k0) ^ (0x9e3779b9u * (uint32_t)(round + 1))) & mask;
            const uint32_t t = L ^ F;
            L = R; R = t;
        }
        x = (L << half) | R;
    } while (x >= n);
    return x;
}

// pass 3: stable compaction (+ permutation) of the kept rows to the frame's capacity offset
__global__ __launch_bounds__(PP_CHUNK) void pp_write_kernel(const float* __restrict__ scene, const float* __restrict__ sampled,
                                                           PrepFrames fr, int ndim, const uint8_t* __restrict__ flags,
                                                           const int32_t* __restrict__ chunk_base, int chunks_per_frame,
                                                           const int32_t* __restrict__ counts, float* __restrict__ out) {
    const int f = blockIdx.y, i = blockIdx.x * PP_CHUNK + threadIdx.x;
    const int n = fr.cap_off[f + 1] - fr.cap_off[f];
    const bool keep = i < n && flags[fr.cap_off[f] + i];
    __shared__ int wsum[PP_CHUNK / 64];
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    if (!keep) return;
    int rank = chunk_base[f * chunks_per_frame + blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += wsum[w];
    const uint64_t seed = fr.seed[f];
    const uint32_t dst = seed ? pp_permute((uint32_t)rank, (uint32_t)counts[f], seed) : (uint32_t)rank;
    bool is_scene;
    const float* p = pp_row(fr, f, i, scene, sampled, ndim, &is_scene);
    float* o = out + ((int64_t)fr.cap_off[f] + dst) * ndim;
    for (int j = 0; j < ndim; ++j) o[j] = p[j];
}

static int pp_chunks(const int64_t* scene_off, const int64_t* samp_off, int n_frames) {
    int64_t mx = 0;
    for (int f = 0; f < n_frames; ++f) {
        const int64_t n = (scene_off[f + 1] - scene_off[f]) + (samp_off ? samp_off[f + 1] - samp_off[f] : 0);
        mx = n > mx ? n : mx;
    }
    return (int)((mx + PP_CHUNK - 1) / PP_CHUNK);
}

extern "C" size_t gga_points_prepare_workspace_bytes(int n_frames, int64_t n_total, int64_t max_frame_rows) {
    if (n_frames < 1 || n_total < 0 || max_frame_rows < 0) return 0;
    const size_t chunks = (size_t)((max_frame_rows + PP_CHUNK - 1) / PP_CHUNK);
    return gga_align_up((size_t)n_total, 256) + gga_align_up((size_t)n_frames * (chunks ? chunks : 1) * 4, 256) + 256;
}

extern "C" int gga_points_prepare_batch(const float* scene, const int64_t* scene_offsets_host, const float* sampled,
                                        const int64_t* sampled_offsets_host, const double* centers_xy,
                                        const int64_t* center_offsets_host, int n_frames, int ndim, double min_distance,
                                        const float* pc_range_dev, const uint64_t* shuffle_seeds_host,
                                        float* out_points, int32_t* out_counts, void* workspace, size_t workspace_bytes,
                                        void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(scene_offsets_host && pc_range_dev && out_counts && workspace, "gga_points_prepare_batch: null pointer argument");
    GGA_REQUIRE(n_frames >= 1 && n_frames <= GGA_MAX_BATCH, "gga_points_prepare_batch: n_frames %d not in [1, %d]", n_frames,
                GGA_MAX_BATCH);
    GGA_REQUIRE(ndim >= 3 && ndim <= 16, "gga_points_prepare_batch: ndim %d not in [3, 16]", ndim);
    GGA_REQUIRE(min_distance >= 0.0, "gga_points_prepare_batch: negative min_distance");
    PrepFrames fr;
    int64_t total = 0;
    fr.cap_off[0] = 0;
    fr.ctr_off[0] = 0;
    for (int f = 0; f < n_frames; ++f) {
        const int64_t ns = scene_offsets_host[f + 1] - scene_offsets_host[f];
        const int64_t np = sampled_offsets_host ? sampled_offsets_host[f + 1] - sampled_offsets_host[f] : 0;
        const int64_t nc = center_offsets_host ? center_offsets_host[f + 1] - center_offsets_host[f] : 0;
        GGA_REQUIRE(ns >= 0 && np >= 0 && nc >= 0, "gga_points_prepare_batch: offsets not monotone");
        fr.n_samp[f] = (int32_t)np;
        fr.samp_off[f] = sampled_offsets_host ? (int32_t)sampled_offsets_host[f] : 0;
        fr.scene_off[f] = (int32_t)scene_offsets_host[f];
        total += ns + np;
        GGA_REQUIRE(total < (1ll << 30), "gga_points_prepare_batch: more than 2^30 rows");
        fr.cap_off[f + 1] = (int32_t)total;
        fr.ctr_off[f + 1] = fr.ctr_off[f] + (int32_t)nc;
        fr.seed[f] = shuffle_seeds_host ? shuffle_seeds_host[f] : 0;
    }
    GGA_REQUIRE(total == 0 || (out_points && (scene || scene_offsets_host[n_frames] == 0)), "gga_points_prepare_batch: null pointer argument");
    GGA_REQUIRE(!sampled_offsets_host || sampled || sampled_offsets_host[n_frames] == 0, "gga_points_prepare_batch: sampled points missing");
    GGA_REQUIRE(fr.ctr_off[n_frames] == 0 || centers_xy, "gga_points_prepare_batch: centres missing");
    const int chunks = pp_chunks(scene_offsets_host, sampled_offsets_host, n_frames);
    const size_t need = gga_points_prepare_workspace_bytes(n_frames, total, (int64_t)chunks * PP_CHUNK);
    if (workspace_bytes < need) {
        gga_set_error("gga_points_prepare_batch: workspace %zu B < required %zu B", workspace_bytes, need);
        return GGA_ERR_WORKSPACE;
    }
    if (total == 0 || chunks == 0) {
        GGA_CHECK_HIP(hipMemsetAsync(out_counts, 0, (size_t)n_frames * 4, stream), "points_prepare memset");
        return GGA_OK;
    }
    uint8_t* flags = (uint8_t*)workspace;
    int32_t* chunk_cnt = (int32_t*)((char*)workspace + gga_align_up((size_t)total, 256));
    const dim3 grid(chunks, n_frames);
    hipLaunchKernelGGL(pp_flag_kernel, grid, dim3(PP_CHUNK), 0, stream, scene, sampled, centers_xy, fr, ndim, min_distance,
                       pc_range_dev, flags, chunk_cnt, chunks);
    GGA_CHECK_LAUNCH("pp_flag_kernel");
    hipLaunchKernelGGL(pp_scan_kernel, dim3(n_frames), dim3(1024), 0, stream, chunk_cnt, chunks, out_counts);
    GGA_CHECK_LAUNCH("pp_scan_kernel");
    hipLaunchKernelGGL(pp_write_kernel, grid, dim3(PP_CHUNK), 0, stream, scene, sampled, fr, ndim, flags, chunk_cnt, chunks,
                       out_counts, out_points);
    GGA_CHECK_LAUNCH("pp_write_kernel");
    return GGA_OK;
}
