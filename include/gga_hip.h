/*
 * gga_hip.h — C ABI of libgga_hip.so: the MI355X (gfx950) kernels of the GGA
 * training hot path.
 *
 * Boundary rules
 *   - plain C: device pointers, sizes, POD structs. No torch / ATen types.
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void*. Every entry point only
 *     ENQUEUES work on it: no allocation, no synchronisation, no host read-back,
 *     so a caller may capture any sequence of calls into a hipGraph.
 *   - memory is owned by the caller (the PyTorch caching allocator in gga_amd/);
 *     scratch is passed in as `workspace` and sized by the *_workspace_bytes()
 *     helper next to each entry point.
 *   - return 0 on success, a negative gga_status otherwise; gga_last_error()
 *     returns a thread-local message for the last failure.
 *   - no global mutable state, re-entrant across processes (one rank per GPU).
 *
 * What each entry point replaces in the reference (paths under /root/reference,
 * a pure-Python mmdet3d fork whose native ops come from the un-vendored
 * mmcv-full / mmdet wheels — see SURVEY.md §2.2): cited per function below.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 */
#ifndef GGA_HIP_H_
#define GGA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GGA_OK = 0,
    GGA_ERR_INVALID_ARG = -1,   /* null pointer, bad size, unsupported shape */
    GGA_ERR_WORKSPACE = -2,     /* workspace too small */
    GGA_ERR_LAUNCH = -3         /* hipGetLastError() != hipSuccess after a launch */
} gga_status;

const char* gga_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int gga_abi_version(void);

/* Measurement only (bench.py's `roofline` objects): kernel timing sessions. After
 * gga_timing_begin(site, n, key) the next n calls of that site's entry point bracket their
 * launches with a pair of HIP events on the caller's stream (no synchronisation, so the step
 * being timed is not disturbed); gga_timing_collect waits for those events and writes the
 * durations in milliseconds, returning how many were taken (negative status on error) and
 * disarming the site. `key` = 0 times every call; otherwise only calls whose shape key matches
 * (see the site list). Thread-safe; one session per site at a time. */
enum {
    GGA_TIME_SCATTER_FWD = 0,   /* gga_pillar_scatter_fwd: NHWC fill [+ map] + rows, NCHW canvas kernel */
    GGA_TIME_DENSE_CONV = 1,    /* gga_dense_conv3x3*: key = GGA_TIMING_CONV_KEY(cin, cout, H*W) */
    GGA_TIME_SPARSE_CONV = 2,   /* gga_sparse_conv_apply[_split]: key = GGA_TIMING_CONV_KEY(cin, cout, 0); gga_sparse_conv_apply_halo: (cin, cout, kvol) */
    GGA_TIME_SPARSE_WGRAD = 3,  /* gga_sparse_conv_wgrad: same key */
    GGA_TIME_DENSE_WGRAD = 4,   /* gga_dense_wgrad3x3: key as GGA_TIME_DENSE_CONV */
    GGA_TIME_SITES = 5
};
#define GGA_TIMING_CONV_KEY(cin, cout, hw) \
    (((int64_t)(cin) << 48) | ((int64_t)(cout) << 32) | ((int64_t)(hw) & 0xffffffffll))
int gga_timing_begin(int site, int max_samples, int64_t key);
int gga_timing_collect(int site, float* ms_host, int cap);

/* ------------------------------------------------------------------------- */
/* a1. Hard voxelization, whole batch in one call.                            */
/* Replaces: mmcv.ops.Voxelization(deterministic=True) called once per frame  */
/* in a Python loop + torch.cat + F.pad —                                     */
/*   mmdet3d/models/detectors/mvx_two_stage_gga.py:211-236 (voxelize),        */
/*   semantics restated in mmdet3d/core/voxel/voxel_generator.py:137-208.     */
/* First-come semantics, bit-exact: voxel id = order of first point, slot =   */
/* order of the point inside its voxel, voxels beyond max_voxels and points   */
/* beyond max_points dropped.                                                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    float voxel_size[3];     /* x, y, z */
    float pc_range[6];       /* xmin, ymin, zmin, xmax, ymax, zmax */
    int32_t max_points;      /* per voxel */
    int32_t max_voxels;      /* per frame */
} gga_voxel_params;

/* grid = round((max - min) / voxel_size) in f32, as mmcv / the numpy voxelizer do. */
void gga_voxel_grid_size(const gga_voxel_params* prm, int32_t grid_xyz[3]);

size_t gga_hard_voxelize_workspace_bytes(int batch, int64_t total_points);

/*
 * points         [total_points, ndim] f32, frames concatenated
 * offsets_host   [batch + 1] i64 row offsets of each frame in `points`
 * voxels         [batch * max_voxels, max_points, ndim] f32  (zero-filled here)
 * coors          [batch * max_voxels, 4] i32  (b, z, y, x)
 * num_points     [batch * max_voxels] i32
 * voxel_num      [batch + 1] i32: voxels per frame, [batch] = total M.
 * Frames are compacted: frame b occupies rows [sum_{b'<b} M_b', +M_b); rows >= M
 * are left zero.
 */
int gga_hard_voxelize_batch(const float* points, int ndim, const int64_t* offsets_host, int batch,
                            const gga_voxel_params* prm, float* voxels, int32_t* coors,
                            int32_t* num_points, int32_t* voxel_num, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Same, for frames that sit at capacity offsets with their real point counts on the device
 * (the output of gga_points_prepare_batch): frame b's points are rows
 * [capacity_offsets_host[b], +min(counts_dev[b], capacity)). No host synchronisation. */
int gga_hard_voxelize_prepared(const float* points, int ndim, const int64_t* capacity_offsets_host,
                               const int32_t* counts_dev, int batch, const gga_voxel_params* prm,
                               float* voxels, int32_t* coors, int32_t* num_points, int32_t* voxel_num,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* a0. Point-level tail of the train data pipeline,    */
/*     batched over frames, between loading and voxelization.                  */
/* ------------------------------------------------------------------------- */
/* Per frame, over the row list [pasted objects' points, scene points]:
 *  - ObjectSample_GGA.remove_points_in_boxes_v2 (mmdet3d/datasets/pipelines/gga_processing.py:
 *    58-68): scene rows whose BEV distance (f64, sqrt(dx*dx + dy*dy) like scipy cdist) to any
 *    pasted object's centre is < min_distance are dropped;
 *  - points.cat([sampled_points, points]) (gga_processing.py:176);
 *  - PointsRangeFilter / BasePoints.in_range_3d (transforms_3d.py:942-977,
 *    core/points/base_points.py:203-225): strict f32 inequalities against pc_range_dev[6];
 *  - PointShuffle (transforms_3d.py:858-883): rows permuted by a bijection seeded with
 *    shuffle_seeds_host[f] (0 = keep the order; the reference draws torch.randperm).
 * Kept rows are compacted in order (stable). scene / sampled: [rows, ndim] f32 with host row
 * offsets [n_frames+1] (sampled*, center* may be NULL = nothing pasted); centers_xy [n,2] f64.
 * Output: frame f's rows start at capacity offset (pasted + scene rows of the frames before it)
 * in out_points; out_counts[f] (device i32) = kept rows. */
size_t gga_points_prepare_workspace_bytes(int n_frames, int64_t total_rows, int64_t max_frame_rows);
int gga_points_prepare_batch(const float* scene, const int64_t* scene_offsets_host, const float* sampled,
                             const int64_t* sampled_offsets_host, const double* centers_xy,
                             const int64_t* center_offsets_host, int n_frames, int ndim, double min_distance,
                             const float* pc_range_dev, const uint64_t* shuffle_seeds_host, float* out_points,
                             int32_t* out_counts, void* workspace, size_t workspace_bytes, void* stream);

/* a2. HardSimpleVFE: mean of the valid points of each voxel.
 * Replaces mmdet3d/models/voxel_encoders/voxel_encoder.py:43-45.
 * out [m, num_features] = sum_p voxels[m, p, :num_features] / num_points[m]. */
int gga_voxel_mean(const float* voxels, const int32_t* num_points, int64_t m, int max_points,
                   int ndim, int num_features, float* out, void* stream);

/* ------------------------------------------------------------------------- */
/* a2'. Fused PillarFeatureNet (one PFNLayer, legacy=True, mode='max'):       */
/* decorate (cluster-mean / pillar-centre offsets) + Linear(10->64, no bias)  */
/* + BatchNorm1d (batch statistics over all m*P rows, padding included) +     */
/* ReLU + max over the P points, and its backward w.r.t. the parameters.      */
/* Replaces mmdet3d/models/voxel_encoders/pillar_encoder.py:93-159 and        */
/* voxel_encoders/utils.py:145-182 (PFNLayer.forward).                        */
/* ------------------------------------------------------------------------- */
typedef struct {
    float voxel_size[3];   /* vx, vy, vz */
    float offsets[3];      /* x_offset = vx/2 + pc_range[0], ... (pillar_encoder.py:87-90) */
    float eps, momentum;   /* BatchNorm1d(eps=1e-3, momentum=0.01) */
    int32_t training;      /* 1: batch statistics + running-stat update, 0: running statistics */
    int32_t in_features;   /* point features (x, y, z, r) = 4 */
    int32_t channels;      /* output channels = 64 */
} gga_pfn_params;

size_t gga_pfn_workspace_bytes(int64_t m);
/*
 * voxels [m,P,4] f32, num_points [m] i32, coors [m,4] i32 (b,z,y,x)
 * weight [64,10], gamma/beta [64], running_mean/var [64] (updated in place when training)
 * out [m,64] f32; argmax [m,64] u8 (index of the winning point, 255 = a padding row)
 * saved [239] f64: first/second moments of the decorated features, per-channel mean / invstd,
 *   and the row count the statistics were taken over
 * num_valid: optional device i32 scalar; only pillars < *num_valid exist (capacity-sized buffers of
 *   gga_hard_voxelize_batch read without a host sync): the BatchNorm statistics run over
 *   min(m, *num_valid) * max_points rows and the out rows past it are zero.
 */
int gga_pfn_fwd(const float* voxels, const int32_t* num_points, const int32_t* coors, int64_t m,
                const int32_t* num_valid, int max_points, const gga_pfn_params* prm, const float* weight, const float* gamma,
                const float* beta, float* running_mean, float* running_var, float* out,
                uint8_t* argmax, double* saved, void* workspace, size_t workspace_bytes,
                void* stream);
/* grad_weight [64,10], grad_gamma [64], grad_beta [64] from grad_out [m,64] (points carry no grad). */
int gga_pfn_bwd(const float* voxels, const int32_t* num_points, const int32_t* coors, int64_t m,
                const int32_t* num_valid, int max_points, const gga_pfn_params* prm, const float* weight, const float* gamma,
                const float* out, const uint8_t* argmax, const double* saved, const float* grad_out,
                float* grad_weight, float* grad_gamma, float* grad_beta, void* workspace,
                size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* a3. PointPillars scatter (the kernel the "HBM GB/s on voxel scatter"       */
/* metric measures) and its backward.                                         */
/* Replaces mmdet3d/models/middle_encoders/pillar_scatter.py:62-102           */
/* (per-frame zeros + boolean mask + index_put + stack).                      */
/* ------------------------------------------------------------------------- */
enum { GGA_LAYOUT_NCHW = 0, GGA_LAYOUT_NHWC = 1 };

/* cell_map: [batch * ny * nx] i32 scratch that MUST hold -1 everywhere on entry
 * and holds -1 everywhere again on return (the canvas pass resets the entries it
 * consumes), so one hipMemset at allocation time serves every later call. */
size_t gga_pillar_scatter_map_bytes(int batch, int ny, int nx);

/*
 * feats      [m, channels] f32;  coors [m, 4] i32 (b, z, y, x)
 * num_valid  optional device i32 scalar: only rows < *num_valid are scattered
 *            (lets the caller keep a capacity-sized buffer without a host sync)
 * canvas     dense output, written in full (zero where no pillar):
 *            NCHW: [batch, channels, ny, nx]   NHWC: [batch, ny, nx, channels]
 * channels must be a multiple of 4. Duplicate (b, y, x): the highest row wins
 * (= the sequential index_put of the reference's CPU path).
 * unique_coors != 0: the caller guarantees that no (b, y, x) repeats (true for the output of
 * gga_hard_voxelize_batch and for the sites of a sparse tensor); the NHWC path then needs no
 * winner map (cell_map may be NULL) - with duplicates any one of the rows wins, as in the
 * reference's CUDA index_put. NHWC = zero fill + one 16 B copy per thread of the occupied rows.
 */
int gga_pillar_scatter_fwd(const float* feats, const int32_t* coors, int64_t m,
                           const int32_t* num_valid, int batch, int channels, int ny, int nx,
                           int layout, int unique_coors, int32_t* cell_map, float* canvas, void* stream);

/* grad_feats [m, channels] = grad_canvas gathered at each pillar's cell
 * (rows >= *num_valid get zeros). */
int gga_pillar_scatter_bwd(const float* grad_canvas, const int32_t* coors, int64_t m,
                           const int32_t* num_valid, int batch, int channels, int ny, int nx,
                           int layout, float* grad_feats, void* stream);

/* SparseEncoder's handover to the 2D backbone (middle_encoders/sparse_encoder.py:134-138: spatial_features =
 * out.dense().view(N, C * D, H, W)) in channels-last memory: out [batch, height, width, channels * depth] f32 with
 * out[b][y][x][c * depth + z] = feats[row][c] for the site coors[row] = (b, z, y, x), zeros elsewhere (the call zeroes out);
 * distinct sites. _bwd: grad_feats [n, channels] gathered from the map's gradient in the same layout. channels % 4 == 0. */
int gga_sparse_bev_nhwc_fwd(const float* feats, const int32_t* coors, int64_t n, int batch, int channels, int depth, int height,
                            int width, float* out, void* stream);
int gga_sparse_bev_nhwc_bwd(const float* grad_out, const int32_t* coors, int64_t n, int batch, int channels, int depth,
                            int height, int width, float* grad_feats, void* stream);

/* Gather map of a dense 2D convolution (kernel kh x kw, given stride / zero padding, dilation 1)
 * over the canvas, restricted to its occupied cells: map[k = ky*kw + kx][r] = row of the NHWC
 * convolution output [batch*oh*ow, C] that read pillar r through tap (ky, kx), or -1
 * (oh = (ny + 2*pad_h - kh)/stride_h + 1, likewise ow; rows >= *num_valid get -1). With it the
 * backward of the first SECOND convolution (second.py:49-57 applied to the canvas of
 * pillar_scatter.py:58-102) runs on the pillars only: its input gradient is needed just at the
 * occupied cells (all the scatter's backward reads) and its weight gradient only sees non-zero
 * input there - gga_sparse_conv_apply[_split] / gga_sparse_conv_wgrad with this map, the
 * output gradient as `x` (rows = output cells) and the pillar features as the other operand. */
int gga_pillar_conv_map(const int32_t* coors, int64_t m, const int32_t* num_valid, int batch, int ny, int nx,
                        int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int32_t* map,
                        void* stream);


/* ------------------------------------------------------------------------- */
/* a3'. Sparse 3D convolution (SubMConv3d / SparseConv3d) for SparseEncoder.  */
/* Replaces the mmcv / spconv natives behind                                  */
/*   mmdet3d/models/middle_encoders/sparse_encoder.py:107-214 and             */
/*   mmdet3d/ops/sparse_block.py:82-199 (SparseConvTensor, rule-book build,   */
/*   per-offset gather-GEMM-scatter). Coordinates are [n,4] i32 (b,z,y,x);    */
/*   weights are [kz,ky,kx,Cin,Cout] (the mmcv layout, write_spconv2.py:47).  */
/* ------------------------------------------------------------------------- */
/* Hash index of one resolution level (cell -> row). `index` is caller memory of
 * gga_sparse_index_bytes(n) bytes; remember n: later calls need it as *_index_n. */
size_t gga_sparse_index_bytes(int64_t n);
int gga_sparse_build_index(const int32_t* coors, int64_t n, int B, int D, int H, int W, void* index,
                           size_t index_bytes, void* stream);

/* Output sites of a strided SparseConv3d: every cell reached by some (input site, offset).
 * out_dhw is computed here ((in + 2p - k)/s + 1). out_coors [cap_out,4]; *n_out (device i32)
 * = number of sites, ordered by the first (input row, offset) that reaches them. out_index
 * becomes the hash index of the OUTPUT level, sized for n_in*kvol
 * (gga_sparse_out_index_bytes); pass that product as its *_index_n later. */
size_t gga_sparse_out_index_bytes(int64_t n_in, int kvol);
size_t gga_sparse_out_sites_workspace_bytes(int64_t n_in, int kvol);
int gga_sparse_conv_out_sites(const int32_t* in_coors, int64_t n_in, int B, const int32_t in_dhw[3],
                              const int32_t kernel[3], const int32_t stride[3], const int32_t pad[3],
                              int32_t out_dhw[3], int32_t* out_coors, int64_t cap_out, int32_t* n_out,
                              void* out_index, size_t out_index_bytes, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Rule books in gather form. nbr [kvol, n_out]: input row under offset k of each output row
 * (-1: none). nbr_t [kvol, n_in] (optional): output row fed by each input row through offset k
 * (backward-data). Submanifold conv: out == in coords, stride 1, pad k/2, nbr_t = NULL
 * (it is nbr[kvol-1-k]). */
int gga_sparse_rulebook(const int32_t* out_coors, int64_t n_out, const int32_t* in_coors, int64_t n_in,
                        int B, const int32_t in_dhw[3], const int32_t out_dhw[3], const int32_t kernel[3],
                        const int32_t stride[3], const int32_t pad[3], const void* in_index,
                        int64_t in_index_n, const void* out_index, int64_t out_index_n, int32_t* nbr,
                        int32_t* nbr_t, void* stream);

/* Bit k of mask[r] is set when map[k][r] >= 0 (kvol <= 32). Sorting rows by this mask gives
 * tiles whose rows use the same kernel offsets; the conv kernel skips the others. */
int gga_sparse_rowmask(const int32_t* map, int64_t n_rows, int kvol, uint32_t* mask, void* stream);
/* order[i] = the row with the i-th smallest mask, rows with equal masks in row order (a stable sort of the int32 masks: the
 * processing order gga_sparse_conv_apply_split takes as `perm`; replaces the framework's stable sort, sparse.py). kvol <= 32
 * mask bits take part; masks are compared as signed 32-bit values when kvol == 32. workspace: device memory of
 * gga_sparse_mask_order_workspace_bytes(n_rows) bytes. */
/* order[i] = the row with the i-th smallest Z-order key (sample, then the bit-interleaved (z, y, x)) of coors [n_rows][4] =
 * (sample, z, y, x): consecutive rows are close in space - the tiles of the halo form (gga_sparse_halo_build). Coordinates must be
 * distinct, 0 <= coordinate < max_extent <= 65536, 0 <= sample < batch_size. */
size_t gga_sparse_morton_order_workspace_bytes(int64_t n_rows);
int gga_sparse_morton_order(const int32_t* coors, int64_t n_rows, int batch_size, int max_extent, int32_t* order,
                            void* workspace, size_t workspace_bytes, void* stream);
size_t gga_sparse_mask_order_workspace_bytes(int64_t n_rows);
int gga_sparse_mask_order(const uint32_t* mask, int64_t n_rows, int kvol, int32_t* order, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Weights in the MFMA fragment order the conv kernel stages through LDS (one 16-byte copy per
 * thread instead of a transposing scatter): packed[k][chunk][lane][g][t][j] =
 * W[k][chunk*32 + 2*(4g+j) + lane/32][t*32 + lane%32], zero beyond cin/cout, with
 * t < (cout<=32 ? 1 : cout<=64 ? 2 : 4). `weight` is [kvol,cin,cout] (transpose 0) or
 * [kvol,cout,cin] (transpose 1: the forward weight of a cout->cin conv, for its backward-data
 * pass). Replaces the weight handling inside mmcv's indice_conv (sparse_block.py:9-20 call sites). */
size_t gga_sparse_packed_weight_bytes(int kvol, int cin, int cout);
int gga_sparse_pack_weight(const float* weight, int kvol, int cin, int cout, int transpose, float* packed,
                           void* stream);

/* y[r,:] = sum_k x[map[kk][r],:] @ W[k], kk = flip ? kvol-1-k : k, W given as
 * gga_sparse_pack_weight(.., cin, cout, ..).
 *   forward        : map = nbr,   flip 0
 *   backward-data  : x = grad_out, map = nbr_t (or nbr with flip 1 for SubM), W packed with
 *                    transpose 1, cin/cout swapped
 * perm (optional, [n_rows]): processing order of the rows (e.g. argsort of the row masks);
 * rowmask (optional, [n_rows]): gga_sparse_rowmask of `map`. Output rows are not permuted. */
int gga_sparse_conv_apply(const float* x, const int32_t* map, const float* packed_weight, const int32_t* perm,
                          const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                          float* y, void* stream);
/* The same convolution with fp32 carried by three bfloat16 planes per operand: a fp32 number is
 * the exact sum of three bf16 numbers, so every product is the sum of nine bf16 products, each
 * exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (1.7x the fp32 MFMA rate on
 * gfx950; error vs float64 not larger than the fp32 MFMA's). The kernels issue six of the nine: the
 * three products of the low planes are together below 2^-23 of the product (one fp32 ulp) and the
 * measured error of a convolution does not change; build with -DX9_NINE for all nine.
 * gga_sparse_pack_weight_split splits
 * and lays out the weights: packed[k][chunk][plane][col][32 ch] bf16; arguments as for
 * gga_sparse_pack_weight / gga_sparse_conv_apply. */
size_t gga_sparse_split_weight_bytes(int kvol, int cin, int cout);
int gga_sparse_pack_weight_split(const float* weight, int kvol, int cin, int cout, int transpose, void* packed,
                                 void* stream);
int gga_sparse_conv_apply_split(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip,
                                float* y, void* stream);
/* The same with y a column block of a wider [n_rows, y_row_stride] matrix. With an arithmetic rule book over the
 * pixels of a channels-last image this gather-GEMM is also the stride-2 3x3 convolution of a SECOND stage
 * (mmdet3d/models/backbones/second.py:49-57) and the kernel = stride transposed convolution of SECONDFPN
 * (necks/second_fpn.py:52-69), forward and backward-data (gga_amd/strided_conv.py). */
int gga_sparse_conv_apply_split_strided(const float* x, const int32_t* map, const void* split_weight,
                                        const int32_t* perm, const uint32_t* rowmask, int64_t n_rows, int kvol, int cin,
                                        int cout, int flip, float* y, int64_t y_row_stride, void* stream);

/* Dense 3x3 / stride 1 / zero-pad 1 convolution of a channels-last image on the same bf16x9
 * path: x [B,H,W,cin] (cin a multiple of 32), y [B,H,W,cout] (cout 64 or 128), split_weight =
 * gga_sparse_pack_weight_split(weight [9][cin][cout] tap-major (ky*3+kx), kvol 9). The SECOND
 * block convolutions and the first convolution of every head branch (backbones/second.py:58-63,
 * dense_heads/centerpoint_head.py:58-68); backward-data is the same call on grad_y with the
 * taps reversed and the channel roles swapped. Each workgroup fetches its input halo once per
 * 32-channel chunk, splits it into the three bf16 planes on the way into LDS and reuses it for
 * all nine taps. */
/* The same convolution that also leaves the batch statistics of its output for the BatchNorm
 * that follows: stats [gga_dense_conv3x3_tiles(B,H,W,cout)][2][cout] f64 = per-tile sum and sum of
 * squares per channel - the `partials` of gga_bn_relu_fwd_partials / gga_bn_stats_partials, so
 * the BatchNorm does not read y a second time for its reduction. stats may be NULL. */
int64_t gga_dense_conv3x3_tiles(int B, int H, int W, int cout);   /* rows of `stats` of a three-plane launch (= ..._tiles_planes(.., 3)) */
/* rows of `stats` for a launch of this arithmetic: the two-plane launches run the producer / consumer form
 * (dense_conv_ws.hip: tiles of 8 x 32 pixels at 128 output channels, 16 x 32 at 64; GGA_DC_WS=0 puts them back on the
 * lock-step kernel), the three-plane launches tiles of 8 or 16 rows by shape. */
int64_t gga_dense_conv3x3_tiles_planes(int B, int H, int W, int cout, int planes);
int64_t gga_dense_conv3x3_stat_rows(int B, int H, int W, int cout, int planes, int n_slices);   /* rows of `stats` (per slice): one per tile (lock-step kernel) or one per workgroup (producer / consumer form, <= 256) */
int gga_dense_conv3x3_tile_rows(int B, int H, int W, int cout, int planes);      /* 8 or 16: the tile_rows to hand gga_dense_conv3x3_levels for slices of this shape */
int gga_dense_conv3x3_stats(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                            float* y, double* stats, void* stream);
/* Weight gradient of the same convolution, bf16x9 with the pixel index as the GEMM's K (transposed
 * LDS reads): x [B,H,W,cin], grad_y [B,H,W,cout] channels-last, cin and cout multiples of 64;
 * grad_weight is written as a [cout, cin, 3, 3] tensor with the given element strides
 * (overwritten, not accumulated). Replaces the convolution-backward-weights call behind autograd
 * for backbones/second.py:58-63 and dense_heads/centerpoint_head.py:58-68,120-121. */
size_t gga_dense_wgrad3x3_workspace_bytes(int B, int H, int W, int cin, int cout);
int gga_dense_wgrad3x3(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                       float* grad_weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                       int64_t stride_kx, int transposed, void* workspace, size_t workspace_bytes, void* stream);

/* Two-fp16-plane arithmetic for the dense kernels (the default of gga_amd/dense_conv.py): each operand is scaled by
 * the power of two that puts its largest finite magnitude into [2^14, 2^15) and split into two round-to-nearest fp16
 * planes (22 significand bits), so a product needs THREE matrix products instead of the six of the three-bf16-plane
 * form; the result is scaled back exactly. Absolute accuracy per element: 2^-39 of the tensor's largest magnitude
 * (fp16 has no fp32 exponent range) - see DESIGN.md 5. `planes` = 3 selects the bf16 form (no absmax needed).
 * gga_absmax_bits: bits of the largest finite |x| of a [rows, width] f32 matrix into *out_bits (device). */
int gga_absmax_bits(const float* x, int64_t rows, int width, int64_t row_stride, uint32_t* out_bits, void* stream);
/* Operands for MANY convolution weights in two launches (gga_amd/weight_bank.py keeps the tables; weights change once per step, all
 * together): gga_absmax_table zeroes `slots[0 .. n_slots)` and leaves in every entry's slot the bits of the largest finite |x| of
 * its block of `n` floats (several entries may share a slot: the absmax of their union); gga_pack_weights_table writes, per entry,
 * kvol x n_c x n_col elements W[tap][c0 + c][col0 + col] = src[k0 * s_k0 + k1 * s_k1 + c * s_c + col * s_col] (tap = k0 * kw + k1,
 * taken in reverse order when `reverse`) into a packed operand of n_in input channels and `co` (padded, 32 * ceil(n_out / 32) for
 * n_out <= 128) columns in the stage layout of gga_sparse_pack_weight_planes (layout 0) or gga_dense_conv3x3_pack_planes (layout
 * 1); padding is never written (the caller zeroes operands once). `first`: running sum of the entries' element counts; `total`:
 * the sum; `first_block`: running sum of gga_absmax_table_blocks(n). Tables live in device memory. */
typedef struct {
    const float* src;
    uint16_t* dst;
    const uint32_t* amax;     /* absmax bits of the operand's weights (planes == 2), or NULL */
    int64_t s_k0, s_k1, s_c, s_col;
    int64_t first;
    int32_t kw, kvol, n_c, n_col, c0, col0, n_in, co, layout, reverse;
} GgaPackEntry;
typedef struct {
    const float* src;
    uint32_t* slot;
    int64_t n;
    int64_t first_block;
} GgaAmaxEntry;
int64_t gga_absmax_table_blocks(int64_t n);
int gga_absmax_table(const GgaAmaxEntry* table_device, int n_entries, int64_t n_blocks, uint32_t* slots, int n_slots, void* stream);
int gga_pack_weights_table(const GgaPackEntry* table_device, int n_entries, int64_t total, int planes, void* stream);
/* the gather-GEMM kernels (sparse, strided and transposed convolutions) in the same two forms */
int gga_sparse_pack_weight_planes(const float* weight, int kvol, int cin, int cout, int transpose, int planes,
                                  const uint32_t* amax_weight, void* packed, void* stream);
int gga_sparse_conv_apply_planes(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                 const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip, float* y,
                                 int64_t y_row_stride, int planes, const uint32_t* amax_x, const uint32_t* amax_weight,
                                 void* stream);
/* gga_sparse_conv_apply_planes that also leaves the per-channel sums of its output for the BatchNorm that follows
 * (ops/sparse_block.py:117-134, backbones/second.py:49-57, necks/second_fpn.py:52-69: conv -> BN -> ReLU): stats
 * [gga_sparse_conv_apply_tiles(n_rows)][2][cout] f64 = per workgroup of 128 rows the sum and the sum of squares - the
 * partials gga_bn_relu_fwd_ex takes. stats == NULL: exactly gga_sparse_conv_apply_planes. */
int64_t gga_sparse_conv_apply_tiles(int64_t n_rows);
int gga_sparse_conv_apply_stats(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip, float* y,
                                int64_t y_row_stride, int planes, const uint32_t* amax_x, const uint32_t* amax_weight,
                                double* stats, void* stream);
/* gga_sparse_conv_apply_stats run as the backward-data pass that produces the gradient of z = relu(bn(bn_x)) (the
 * conv -> BatchNorm1d -> ReLU -> conv chains of middle_encoders/sparse_encoder.py:107-214): rows are masked by the ReLU
 * (recomputed from bn_x [n_rows, row stride bn_x_row_stride] and gamma, beta, mean, invstd exactly as the forward decided)
 * before they are stored, and stats (required) holds the sums of g and g * xhat for gga_bn_relu_bwd_partials. */
int gga_sparse_conv_apply_bn_bwd(const float* x, const int32_t* map, const void* split_weight, const int32_t* perm,
                                 const uint32_t* rowmask, int64_t n_rows, int kvol, int cin, int cout, int flip, float* y,
                                 int64_t y_row_stride, int planes, const uint32_t* amax_x, const uint32_t* amax_weight,
                                 double* stats, const float* bn_x, int64_t bn_x_row_stride, const float* bn_gamma,
                                 const float* bn_beta, const float* bn_mean, const float* bn_invstd, void* stream);
/* Halo form of gga_sparse_conv_apply_bn_bwd for SUBMANIFOLD convolutions (same reference path: SubMConv3d of
 * middle_encoders/sparse_encoder.py:107-214 / ops/sparse_block.py:117-199; same arithmetic and epilogues). The rows are
 * processed in tiles of gga_sparse_halo_tile_rows() (256) rows that lie close together in space; the distinct input rows a
 * tile's kvol x 256 rule-book entries name (its halo) are staged once per 32-channel chunk in LDS and every offset reads
 * them there. gga_sparse_halo_build makes the tiling from the rule book and a spatial order of its rows (gga_amd/sparse.py::
 * _Halo: Z-order of the level's coordinates):
 *   tile_rows    int32 [n_tiles * 256]         output row (= row of the rule book) of each tile slot, -1 = empty slot
 *   halo_rows    int32 [n_tiles][capacity]     tile t's halo: its first halo_counts[t] entries, distinct input rows
 *   halo_counts  int32 [n_tiles]
 *   local_map    uint16 [n_tiles][kvol][256]   position of the neighbour in the tile's halo, 0xFFFF = none; offset k of a
 *                                              flipped (backward-data) launch reads entry kvol-1-k
 * n_tiles = ceil(n_rows / 256); capacity >= kvol * 256 (no tile can overflow); apply: 8 <= kvol <= 27; cin % 32 == 0; cout 64
 * or 128; two fp16 planes only (planes == 2). stats (may be NULL) is sized as for gga_sparse_conv_apply_stats
 * ([gga_sparse_conv_apply_tiles(n_rows)][2][cout]); the rows no tile writes are zeroed. */
int64_t gga_sparse_halo_tile_rows(void);
int gga_sparse_halo_build(const int32_t* nbr, const int32_t* tile_rows, int64_t n_rows, int64_t n_tiles, int kvol,
                          int halo_capacity, int32_t* halo_rows, int32_t* halo_counts, uint16_t* local_map, void* stream);
int gga_sparse_conv_apply_halo(const float* x, const void* split_weight, const int32_t* tile_rows,
                               const int32_t* halo_counts, int halo_capacity, const int32_t* halo_rows,
                               const uint16_t* local_map, int64_t n_rows, int64_t n_tiles, int kvol, int cin, int cout, int flip,
                               float* y, int64_t y_row_stride, int planes, const uint32_t* amax_x, const uint32_t* amax_weight,
                               double* stats, const float* bn_x, int64_t bn_x_row_stride, const float* bn_gamma,
                               const float* bn_beta, const float* bn_mean, const float* bn_invstd, void* stream);
int gga_sparse_conv_wgrad_planes(const float* x, int64_t x_row_stride, const float* grad_out, int64_t grad_out_row_stride,
                                 const int32_t* nbr, int64_t n_rows, int kvol, int cin, int cout, float* grad_weight,
                                 int planes, const uint32_t* amax_x, const uint32_t* amax_grad_out, void* workspace,
                                 size_t workspace_bytes, void* stream);
int gga_dense_conv3x3_pack_planes(const float* weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                  int64_t stride_kx, int cin, int cout, int backward, int planes,
                                  const uint32_t* amax_weight, void* packed, void* stream);
int gga_dense_conv3x3_planes(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout, float* y,
                             int64_t y_pixel_stride, int transposed, double* stats, int planes, const uint32_t* amax_x,
                             const uint32_t* amax_weight, void* stream);
/* gga_dense_conv3x3_planes run as the backward-data convolution that produces the gradient of z = relu(bn(bn_x)) (the
 * conv -> BatchNorm -> ReLU -> conv chains of backbones/second.py:46-63): the epilogue masks the tile with the ReLU
 * (recomputed from bn_x and gamma, beta, mean, invstd of the `cout` channels of this launch, exactly as the forward
 * pass decided it) before it is stored, and `stats` ([tiles][2][cout] f64, required) receives the per-tile sums of
 * g and g * xhat instead - the reduce pass of that BatchNorm's backward, which gga_bn_relu_bwd_partials then skips.
 * bn_x == NULL: exactly gga_dense_conv3x3_planes. */
/* gga_dense_conv3x3_planes over several images of different sizes in ONE launch (the tower convolutions of an FPN head,
 * anchor_free_mono3d_head.py:160-250, share their weights over the levels, and the small levels alone leave most of the
 * chip idle): entry e convolves x[e] [B, heights[e], widths[e], cin] with split_weight[e] (entries that are output slices of
 * one convolution bring their own operand) into y[e] (pixel stride y_pixel_stride floats). bias (NULL: none): per entry the
 * `cout` bias values of its output channels (NULL entries: none), added in the epilogue - this is also how a single map with
 * a bias is run. tile_rows: 8, or 16 (cout 128 only: the form gga_dense_conv3x3_planes picks for maps with at least 384
 * 16-row tiles - large and small maps go into separate launches); transposed: every entry walks its map transposed (as
 * gga_dense_conv3x3_slice; split_weight packed accordingly); stats (NULL: none): per entry the f64 [tiles][2][cout]
 * BatchNorm sums of its output (gga_dense_conv3x3_stats; tiles of the entry = B * ceil(w / 32) * ceil(h / tile_rows) in its
 * tile space). At most 16 entries. */
int gga_dense_conv3x3_levels(int n_entries, const float* const* x, const int32_t* heights, const int32_t* widths,
                             const void* const* split_weight, int B, int cin, int cout, float* const* y, int64_t y_pixel_stride,
                             int planes, const uint32_t* const* amax_x, const uint32_t* amax_weight, const float* const* bias,
                             int tile_rows, int transposed, double* const* stats, void* stream);
int gga_dense_conv3x3_bn_bwd_pays(int B, int H, int W, int cout);   /* 1: the epilogue costs less than the reduce pass (H, W of the tile space); three planes */
int gga_dense_conv3x3_bn_bwd_pays_planes(int B, int H, int W, int cout, int planes);   /* the same for a launch of this arithmetic (two planes: always) */
int gga_dense_conv3x3_bn_bwd(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout, float* y,
                             int64_t y_pixel_stride, int transposed, double* stats, int planes, const uint32_t* amax_x,
                             const uint32_t* amax_weight, const float* bn_x, int64_t bn_x_pixel_stride, const float* bn_gamma,
                             const float* bn_beta, const float* bn_mean, const float* bn_invstd, void* stream);
int gga_dense_wgrad3x3_planes(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                              float* grad_weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                              int64_t stride_kx, int transposed, int planes, const uint32_t* amax_x,
                              const uint32_t* amax_grad_y, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the absmax of an operand given per 64-channel block (amax_x[ci / 64], amax_grad_y[co / 64]) when its
 * *_per_block flag is set: dW[ci][co] only sees channel ci of x and channel co of grad_y, so on two fp16 planes a block whose
 * values lie far below the tensor's largest magnitude keeps its own 22 significant bits (the 960-channel gradient of the 15 head
 * branches: regression branches with a few object cells beside the heat-map branches).
 * One image of an operand (H * W * max(cin, cout) floats) must stay below 2 GiB: the kernel addresses a staged piece by a 32-bit
 * offset inside its image row beside a scalar row base (GGA_ERR_* otherwise). */
int gga_dense_wgrad3x3_block_amax(const float* x, const float* grad_y, int B, int H, int W, int cin, int cout,
                                  float* grad_weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                                  int64_t stride_kx, int transposed, int planes, const uint32_t* amax_x, int amax_x_per_block,
                                  const uint32_t* amax_grad_y, int amax_grad_y_per_block, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* The same with y as a 64- or 128-channel slice of a wider channels-last tensor: pixel p of the
 * result starts at y + p * y_pixel_stride (floats). A convolution with more output channels runs as
 * one call per slice (each with the weights of its slice): the backward-data of the 384 -> 64
 * shared convolution of the head (centerpoint_head.py:255-262) is three 64 -> 128 calls. */
/* transposed != 0: the 8 x 32-pixel tiles run 32 pixels along H instead of W (maps whose width is a
 * poor multiple of 32, e.g. 124 x 108 or 62 x 54); split_weight must then be packed with the ky / kx
 * strides swapped, and stats holds gga_dense_conv3x3_tiles(B, W, H, cout) rows. Results are identical. */
int gga_dense_conv3x3_slice(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout,
                            float* y, int64_t y_pixel_stride, int transposed, double* stats, void* stream);

/* split_weight for gga_dense_conv3x3 straight from the framework's [cout, cin, 3, 3] parameter with
 * arbitrary element strides (channels-last parameters included): size
 * gga_sparse_split_weight_bytes(9, cin, cout); backward != 0 packs the operand of the
 * backward-data convolution (a cout -> cin convolution: size ..._bytes(9, cout, cin)). */
int gga_dense_conv3x3_pack(const float* weight, int64_t stride_co, int64_t stride_ci, int64_t stride_ky,
                           int64_t stride_kx, int cin, int cout, int backward, void* packed, void* stream);
int gga_dense_conv3x3(const float* x, const void* split_weight, int B, int H, int W, int cin, int cout, float* y,
                      void* stream);

/* grad_weight [kvol,cin,cout] = sum_r x[nbr[k][r]]^T grad_out[r]  (zero-filled here).
 * Native fp32 MFMA with one float atomicAdd per weight and 2048-row chunk: the summation order of
 * the chunks (not the values summed) varies from run to run. */
int gga_sparse_conv_wgrad(const float* x, const float* grad_out, const int32_t* nbr, int64_t n_rows,
                          int kvol, int cin, int cout, float* grad_weight, void* stream);
/* The same gradient on the bf16 planes (six partial products, as gga_sparse_conv_apply_split) and
 * DETERMINISTIC: pairs are compacted in row order, every workgroup writes its partial sum to
 * workspace [chunks][kvol][CI][CO] and a second kernel adds the chunks in a fixed order in f64.
 * No atomics; about 1.7x the fp32-MFMA form at 128 channels. */
size_t gga_sparse_conv_wgrad_workspace_bytes(int64_t n_rows, int kvol, int cin, int cout);
int gga_sparse_conv_wgrad_split(const float* x, const float* grad_out, const int32_t* nbr, int64_t n_rows,
                                int kvol, int cin, int cout, float* grad_weight, void* workspace,
                                size_t workspace_bytes, void* stream);
/* The same with x / grad_out column blocks (<= 128 wide) of wider matrices: row strides in floats.
 * Every x row a rule-book entry names must start below byte 4 GiB of x (32-bit pair offsets; the kernel TRAPS - the stream
 * reports a fault - on a row beyond it: 8 M rows of 128 floats).
 * Non-finite inputs: absent neighbours, tail pairs and (dense weight gradient) pixels outside the image are not selected to zero
 * but read a real element - row 0 / the clamped pixel - and multiply it by a scale of 0 (one instruction less per fragment beside
 * the matrix instructions). With finite tensors that is 0; an Inf / NaN in the substituted element turns that padded contribution
 * into NaN, i.e. a non-finite value in x or grad_out can reach weight-gradient entries it would not reach through a select. The
 * train Runner's range guard reports non-finite operands on its guarded iterations. */
int gga_sparse_conv_wgrad_split_strided(const float* x, int64_t x_row_stride, const float* grad_out,
                                        int64_t grad_out_row_stride, const int32_t* nbr, int64_t n_rows, int kvol, int cin,
                                        int cout, float* grad_weight, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* a4/a5 (elementwise part). Fused training-mode BatchNorm (+ residual add)   */
/* (+ ReLU) over a [rows, channels] row-major tensor = the sparse [N,C]        */
/* features and the memory of a channels-last [B,C,H,W] activation.           */
/* Replaces the separate BN / ReLU / add kernels of                            */
/*   mmdet3d/models/backbones/second.py:46-63, necks/second_fpn.py:66-69,      */
/*   dense_heads/centerpoint_head.py (ConvModule), ops/sparse_block.py:117-134 */
/* y = relu(bn(x) + residual); channels % 4 == 0 and channels/4 must divide 256. */
/* ------------------------------------------------------------------------- */
size_t gga_bn_relu_workspace_bytes(int64_t rows, int channels);
size_t gga_bn_relu_mask_bytes(int64_t rows, int channels);   /* 1 bit / element ReLU sign */
/* saved [2*channels] f32 receives mean / invstd; running stats are updated when training. */
int gga_bn_relu_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                    float momentum, int training, int relu, float* y, void* mask_bits, float* saved,
                    void* workspace, size_t workspace_bytes, void* stream);
/* batch-statistics backward: grad_x, optional grad_residual (= masked grad_y), grad_gamma/beta.
 * relu: 0 none, 1 mask_bits = the bits written by the forward, 2 mask_bits = scale_shift of
 * gga_bn_stats (mask recomputed from x). */
int gga_bn_relu_bwd(const float* grad_y, const float* x, const void* mask_bits, const float* gamma,
                    const float* saved, int64_t rows, int channels, int relu, float* grad_x,
                    float* grad_residual, float* grad_gamma, float* grad_beta, void* workspace,
                    size_t workspace_bytes, void* stream);
/* The same with y (forward) / grad_y (backward) as a column block of a wider row-major matrix:
 * row r of the operand starts at y + r * y_row_stride (floats; a multiple of 4, >= channels; the
 * pointer 16 B aligned). The three SECONDFPN branches write their BN+ReLU output straight into
 * their channel slice of the concatenated map and read their slice of its gradient in place
 * (necks/second_fpn.py:85-91: torch.cat of the deblock outputs) - no concat copy, no split copy. */
int gga_bn_relu_fwd_strided(const float* x, const float* residual, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                            float momentum, int training, int relu, float* y, int64_t y_row_stride,
                            void* mask_bits, float* saved, void* workspace, size_t workspace_bytes, void* stream);
int gga_bn_relu_bwd_strided(const float* grad_y, int64_t grad_y_row_stride, const float* x, const void* mask_bits,
                            const float* gamma, const float* saved, int64_t rows, int channels, int relu,
                            float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Batch statistics only (no normalised output): saved [2*channels] = mean / invstd, scale_shift
 * [2*channels] = gamma*invstd, beta - mean*gamma*invstd; running stats updated when training. For
 * consumers that apply the affine + ReLU themselves while loading (gga_head_conv3x3_* with
 * in_scale_shift), so the normalised activation is never written. Its backward is
 * gga_bn_relu_bwd[_strided] with relu = 2 and mask_bits = scale_shift: the ReLU mask is
 * recomputed as fma(x, scale, shift) > 0 instead of read from stored bits. */
/* Supersets of the calls above with one more output: the bits of the largest finite magnitude written (y /
 * grad_x), max-combined into *amax (the caller zeroes it; several calls may share it) - what the two-fp16-plane
 * convolution kernels that consume the tensor derive their scale from, without a pass of their own.
 * gga_bn_relu_fwd_ex: partials != NULL = the producer's per-channel sums (as gga_bn_relu_fwd_partials).
 * gga_bn_relu_bwd_ex: training = 0 is the backward of an evaluation-mode forward (running statistics, e.g. a frozen
 * backbone with norm_eval): grad_x = gamma * invstd * g, grad_gamma / grad_beta as in training mode. */
int gga_bn_relu_fwd_ex(const float* x, const float* residual, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t rows, int channels, float eps, float momentum,
                       int training, int relu, float* y, int64_t y_row_stride, void* mask_bits, float* saved,
                       const double* partials, int n_partials, uint32_t* amax_y, void* workspace, size_t workspace_bytes,
                       void* stream);
int gga_bn_relu_bwd_ex(const float* grad_y, int64_t grad_y_row_stride, const float* x, const void* mask_bits,
                       const float* gamma, const float* saved, int64_t rows, int channels, int relu, int training, float* grad_x,
                       float* grad_residual, float* grad_gamma, float* grad_beta, uint32_t* amax_grad_x, void* workspace,
                       size_t workspace_bytes, void* stream);
/* sums[c] = sum over the rows of x [rows, channels] (row major; channels / 4 must divide 256) - the bias gradient of a
 * convolution from its output gradient in channels-last memory. Workspace: gga_bn_relu_workspace_bytes(rows, channels). */
int gga_column_sums(const float* x, int64_t rows, int channels, float* sums, void* workspace, size_t workspace_bytes,
                    void* stream);
/* The backward pass whose reduce pass was done by the producer of the gradient (gga_dense_conv3x3_bn_bwd): grad_masked
 * is already multiplied by the ReLU mask, partials [n_partials][2][channels] f64 hold the sums of g and g * xhat. */
int gga_bn_relu_bwd_partials(const float* grad_masked, int64_t grad_row_stride, const float* x, const float* gamma,
                             const float* saved, int64_t rows, int channels, int training, const double* partials,
                             int n_partials, float* grad_x, float* grad_gamma, float* grad_beta, uint32_t* amax_grad_x,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Fused GroupNorm (+ ReLU) over a channels-last [B, rows_per_sample, channels] activation (statistics per sample and
 * group of channels / groups channels): replaces the GN + ReLU of the ConvModules in the PGD / FCOS3D head
 * (mmdet3d/models/dense_heads/anchor_free_mono3d_head.py:160-250, norm_cfg type 'GN'), whose framework kernels need NCHW
 * memory. stat [B][groups][2] = mean, rstd; scale_shift [B][2][channels]; both are kept for the backward, which
 * recomputes the ReLU mask from x. channels/4 must divide 256, channels per group % 4 == 0. amax_* as in gga_bn_relu_*_ex. */
size_t gga_gn_relu_workspace_bytes(int B, int channels);
int gga_gn_relu_fwd(const float* x, const float* gamma, const float* beta, int B, int64_t rows_per_sample, int channels,
                    int groups, float eps, int relu, float* y, float* stat, float* scale_shift, uint32_t* amax_y,
                    void* workspace, size_t workspace_bytes, void* stream);
int gga_gn_relu_bwd(const float* grad_y, const float* x, const float* gamma, const float* stat, const float* scale_shift,
                    int B, int64_t rows_per_sample, int channels, int groups, int relu, float* grad_x, float* grad_gamma,
                    float* grad_beta, uint32_t* amax_grad_x, void* workspace, size_t workspace_bytes, void* stream);

int gga_bn_stats(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                 int64_t rows, int channels, float eps, float momentum, int training, float* saved,
                 float* scale_shift, void* workspace, size_t workspace_bytes, void* stream);

/* Training-mode forward / statistics with the per-channel sums already reduced by the producer of
 * x: partials [n_partials][2][channels] f64 (sum, sum of squares over disjoint row sets that
 * together cover all `rows`), e.g. from gga_dense_conv3x3_stats. Otherwise as
 * gga_bn_relu_fwd_strided / gga_bn_stats. */
int gga_bn_relu_fwd_partials(const float* x, const float* residual, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int64_t rows, int channels, float eps,
                             float momentum, int relu, float* y, int64_t y_row_stride, void* mask_bits,
                             float* saved, const double* partials, int n_partials, void* workspace,
                             size_t workspace_bytes, void* stream);
int gga_bn_stats_partials(const float* gamma, const float* beta, float* running_mean, float* running_var,
                          int64_t rows, int channels, float eps, float momentum, float* saved, float* scale_shift,
                          const double* partials, int n_partials, void* stream);
/* The same for a BatchNorm whose channels are columns column_offset .. + channels of partial rows [2][partials_width] (a
 * convolution launch that computed the inputs of several BatchNorms side by side: the head's paired first convolutions). */
int gga_bn_stats_partials_cols(const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t rows,
                               int channels, float eps, float momentum, float* saved, float* scale_shift, const double* partials,
                               int n_partials, int partials_width, int column_offset, void* stream);

/* a5 (output convs of the head branches): 3x3 conv, 64 input channels -> 1..4 output channels,
 * stride 1, pad 1, + bias. Replaces the last layer of each SeparateHead branch,
 * mmdet3d/models/dense_heads/centerpoint_head.py:70-79 (HBM-bound; N = 9 taps x cout on the
 * matrix cores). x: channels-last memory [B,H,W,64] of a [B,64,H,W] tensor, or a 64-channel column block of
 * a wider one (x_pixel_stride floats between pixels, 64 when dense); weight [cout,64,3,3];
 * y [B,cout,H,W] NCHW-contiguous. in_scale_shift (optional, [2*64]): the convolution input is
 * relu(x * scale + shift) per channel, applied while loading - the BatchNorm + ReLU of the
 * branch's ConvModule (centerpoint_head.py:58-68) fused into its consumer. */
int gga_head_conv3x3_fwd(const float* x, int64_t x_pixel_stride, const float* in_scale_shift, const float* weight,
                         const float* bias, int B, int H, int W, int cin, int cout, float* y, void* stream);
/* grad_weight [cout,64,3,3] and grad_bias [cout] (optional) from grad_y [B,cout,H,W] */
size_t gga_head_conv3x3_workspace_bytes(int cout);
int gga_head_conv3x3_wgrad(const float* x, int64_t x_pixel_stride, const float* in_scale_shift, const float* grad_y,
                           int B, int H, int W, int cin, int cout, float* grad_weight, float* grad_bias,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the whole branch tail BatchNorm(training) -> ReLU -> this conv w.r.t. the BatchNorm input x
 * (centerpoint_head.py:58-79 under autograd): grad_x [B,H,W,64], grad_gamma / grad_beta [64] (optional) from
 * grad_y [B,cout,H,W], the conv weight and the statistics gga_bn_stats left (saved = mean / invstd,
 * scale_shift). The conv's input gradient is rebuilt from grad_y inside the two BatchNorm-backward passes and
 * never stored. x / grad_x may be 64-channel column blocks of wider tensors (pixel strides in floats).
 * amax_grad_x (optional): atomicMax target for the bits of the largest finite |grad_x| written (not reset here).
 * workspace: gga_bn_relu_workspace_bytes(B*H*W, 64). */
int gga_head_tail_bwd(const float* grad_y, const float* x, int64_t x_pixel_stride, const float* scale_shift,
                      const float* gamma, const float* saved, const float* weight, int B, int H, int W, int cin, int cout,
                      float* grad_x, int64_t grad_x_pixel_stride, float* grad_gamma, float* grad_beta, uint32_t* amax_grad_x,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* a6/a7. Heat-map target splat on the device.                                */
/* Replaces the per-object numpy gaussian + H2D copy + torch.max(out=) of     */
/* mmdet3d/core/utils/gaussian.py:25-54 called from                           */
/* mmdet3d/models/dense_heads/centerpoint_head_gga.py:576.                    */
/* ------------------------------------------------------------------------- */
/*
 * heatmap       [n_maps, H, W] f32, zero-filled here, then max-splatted
 * objs          [n_obj, 4] i32: (map index, cx, cy, radius)
 * patch_table   f32 gaussian patches for radius 0..max_radius, concatenated;
 *               patch r is (2r+1)^2 values at patch_offsets[r] (built on the host
 *               with the reference's f64 formula so the values are bit-identical)
 */
int gga_heatmap_splat(float* heatmap, int n_maps, int H, int W, const int32_t* objs, int n_obj,
                      const float* patch_table, const int32_t* patch_offsets, int max_radius,
                      void* stream);

/* ------------------------------------------------------------------------- */
/* a8. clip_sigmoid + GaussianFocalLoss (mean over max(num_pos,1)), fused.    */
/* Replaces mmdet3d/models/utils/clip_sigmoid.py:16 + mmdet GaussianFocalLoss */
/* + the host-syncing num_pos.item() of centerpoint_head_gga.py:650-655.      */
/* ------------------------------------------------------------------------- */
size_t gga_focal_loss_workspace_bytes(int64_t n);
/* out[0] = scale * sum(loss) / (max(num_pos,1) + eps_f32); out[1] = num_pos. */
int gga_focal_loss_fwd(const float* logits, const float* target, int64_t n, float alpha,
                       float gamma, float scale, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);
/* grad_logits[i] = (*grad_out) * scale * dloss_i/dlogit_i / (max(num_pos,1)+eps);
 * fwd_out is the `out` of the forward call (num_pos is read from fwd_out[1]). */
int gga_focal_loss_bwd(const float* logits, const float* target, int64_t n, float alpha,
                       float gamma, float scale, const float* fwd_out, const float* grad_out,
                       float* grad_logits, void* stream);

/* ------------------------------------------------------------------------- */
/* a9. Gather the 8 regression channels at the object cells, and its backward */
/* (scatter-add into the dense head-map gradients).                           */
/* Replaces cat + permute + contiguous + gather of                            */
/* centerpoint_head_gga.py:141-164,657-676.                                   */
/* ------------------------------------------------------------------------- */
/* reg [B,2,H,W] height [B,1,H,W] dim [B,3,H,W] rot [B,2,H,W] -> pred [B,K,8] */
int gga_gather_pred_fwd(const float* reg, const float* height, const float* dim, const float* rot,
                        const int64_t* ind, int B, int K, int H, int W, float* pred, void* stream);
/* The four grad maps are zero-filled here, then grad_pred is scatter-added;
 * slots with mask == 0 are skipped (their weight is zero in every loss term). */
int gga_gather_pred_bwd(const float* grad_pred, const int64_t* ind, const uint8_t* mask, int B,
                        int K, int H, int W, float* g_reg, float* g_height, float* g_dim,
                        float* g_rot, void* stream);

/* ------------------------------------------------------------------------- */
/* a10-a13. The GGA geometry-aware losses for one task, forward + analytic    */
/* gradient in one pass: rotation, decode, 8 corners, lidar2img projection,   */
/* 2D box (BPL), semantic ratio (SRL), point-to-box alignment (PAL).          */
/* Replaces centerpoint_head_gga.py:167-341 (GGA_calculate_rotation,          */
/* get_distance_single/bev, get_prediction_single) and the loss assembly      */
/* :678-721 with mmdet L1Loss(reduction='mean', loss_weight).                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t B, K;                 /* frames, object slots per frame (max_objs) */
    int32_t fm_w;                 /* feature-map width (ind = y * fm_w + x) */
    float voxel_size[2];          /* train_cfg.voxel_size[:2] */
    float out_size_factor;
    float pc_range[2];            /* train_cfg.point_cloud_range[:2] */
    float code_weights[5];        /* train_cfg.code_weights */
    float l1_loss_weight;         /* L1Loss.loss_weight (0.25) */
    float w_bpl, w_srl, w_pal;    /* head:697-721 multipliers (0.3, 0.1, 0.1) */
} gga_loss_params;

/* indices into losses[] */
enum { GGA_L_BPL = 0, GGA_L_SRL = 1, GGA_L_PAL_MIN = 2, GGA_L_PAL_X = 3, GGA_L_PAL_Y = 4,
       GGA_L_NUM = 5 };

size_t gga_box_losses_workspace_bytes(int B, int K);

/*
 * pred        [B,K,8] f32 (dx, dy, z, log l, log w, log h, sin, cos)
 * ind         [B,K] i64; mask [B,K] u8; anno_box [B,K,5] f32 (x1,y1,x2,y2,srl)
 * lidar2img   [B,K,4,4] f32; bound_mask [B,K,4] u8
 * ibp_xy      [n_pts,2] f32 in-box points of every object (xy, packed)
 * ibp_offsets [n_ibp_obj + 1] i32 point ranges; ibp_slot [n_ibp_obj] i32 = b*K + k
 * outputs
 *   losses      [5] f32, final dict values (weights and 1/(num+1e-4+eps) applied)
 *   box_out     [B,K,12] f32: rot, l, w, u_min, v_min, u_max, v_max, X, Y, p2c_min, p2c_x, p2c_y
 *               (intermediates, for parity tests / logging)
 *   grad_pred   [5,B,K,8] f32: d losses[t] / d pred, per term
 */
int gga_box_losses_fwd(const float* pred, const int64_t* ind, const uint8_t* mask,
                       const float* anno_box, const float* lidar2img, const uint8_t* bound_mask,
                       const float* ibp_xy, const int32_t* ibp_offsets, const int32_t* ibp_slot,
                       int n_ibp_obj, const gga_loss_params* prm, float* losses, float* box_out,
                       float* grad_pred, void* workspace, size_t workspace_bytes, void* stream);

/* grad_pred_out [B,K,8] = sum_t grad_losses[t] * grad_pred[t] (grad_losses [5] on device). */
int gga_box_losses_bwd(const float* grad_pred, const float* grad_losses, int B, int K,
                       float* grad_pred_out, void* stream);

/* ------------------------------------------------------------------------- */
/* SURVEY.md §8(f) rank 1 — inference / pseudo-label post-processing.        */
/* ------------------------------------------------------------------------- */
/* Rotated BEV IoU. Replaces mmcv.ops.box_iou_rotated as called by
 * mmdet3d/core/bbox/structures/base_box3d.py:469 (BaseInstance3DBoxes.overlaps).
 * boxes [n,5] / [m,5] = (x, y, w, h, angle[rad]); out [n,m] (or [n] when aligned);
 * mode_iof: intersection over the area of box1 instead of the union. */
int gga_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int m, int mode_iof,
                        int aligned, float* out, void* stream);

/* Rotated NMS on score-sorted boxes. Replaces mmcv.ops.nms_rotated's native half as called by
 * mmdet3d/core/post_processing/box3d_nms.py:264 (nms_bev) <- centerpoint_head_gga.py:885-890;
 * like mmcv, the caller sorts by descending score. keep [n] i64 receives the positions (in the
 * sorted order) of the surviving boxes, *num_keep (device) their count (<= max_keep if > 0). */
size_t gga_nms_rotated_workspace_bytes(int n);
int gga_nms_rotated_sorted(const float* boxes_sorted, int n, float iou_threshold, int max_keep,
                           int64_t* keep, int32_t* num_keep, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Circular NMS of CenterPoint (mmdet3d/core/post_processing/box3d_nms.py:181-225, test_cfg
 * nms_type='circle'): xy_sorted [n,2] f32 centres by descending score; a kept centre suppresses the
 * later ones with (dx*dx + dy*dy) <= thresh, the distance in the reference's float32 arithmetic.
 * keep / num_keep / workspace as for gga_nms_rotated_sorted (same workspace size). */
int gga_circle_nms_sorted(const float* xy_sorted, int n, double thresh, int max_keep, int64_t* keep,
                          int32_t* num_keep, void* workspace, size_t workspace_bytes, void* stream);


/* Points in rotated 3D boxes. Replaces mmcv.ops.points_in_boxes_part / points_in_boxes_all as
 * called by base_box3d.py:534,566. points [B,M,3]; boxes [B,T,7] = (x, y, z_bottom, dx, dy, dz,
 * yaw). all = 0: out [B,M] i32 index of the first box containing the point or -1;
 * all = 1: out [B,M,T] i32 flags. */
int gga_points_in_boxes(const float* points, const float* boxes, int B, int M, int T, int all,
                        int32_t* out, void* stream);

/* Detections of a whole batch in one launch: the per-(frame, task) post-processing of CenterHead_GGA.get_bboxes /
 * get_task_detections (mmdet3d/models/dense_heads/centerpoint_head_gga.py:725-934) behind CenterPointBBoxCoder.decode's
 * top-k (mmdet3d/core/bbox/coders/centerpoint_bbox_coders.py:117-229), with nms_bev -> mmcv.ops.nms_rotated
 * (mmdet3d/core/post_processing/box3d_nms.py:231-268) inside. boxes [n_tasks, n_frames, k, box_dim] = the decoded boxes
 * (x, y, z gravity centre, dx, dy, dz, yaw [, vx, vy]) of every task's k top-scored cells in descending score order, scores /
 * labels [n_tasks, n_frames, k] (labels as the coder's float class index within the task). Per frame and task: coder mask
 * (centre inside coder_range [6] inclusive, score > coder_score_threshold when has_coder_score_threshold), head threshold
 * (score >= score_threshold when > 0), rotated BEV NMS over the first pre_max_size survivors on the boxes as nms_bev sees them
 * (centre -/+ extent / 2 and back), IoU > nms_threshold suppresses, at most post_max_size kept (<= 0: no cap), range filter
 * (limit_range [6] inclusive; NULL: none); tasks concatenated in order, z moved to the bottom centre, label + class_offset[task]
 * (0 + offset when single_class[task]). out_boxes [n_frames, n_tasks * k, box_dim], out_scores / out_labels [n_frames,
 * n_tasks * k], out_count [n_frames]: frame b's detections are rows 0 .. out_count[b]. k <= 128. workspace:
 * gga_centerpoint_detect_workspace_bytes (the per-task counts between the two launches). */
size_t gga_centerpoint_detect_workspace_bytes(int n_tasks, int n_frames);
int gga_centerpoint_detect(const float* boxes, const float* scores, const float* labels, int n_tasks, int n_frames, int k,
                           int box_dim, const float* coder_range, float coder_score_threshold, int has_coder_score_threshold,
                           float score_threshold, const float* limit_range, float nms_threshold, int pre_max_size,
                           int post_max_size, const int32_t* class_offset, const int32_t* single_class, float* out_boxes,
                           float* out_scores, int32_t* out_labels, int32_t* out_count, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Pseudo-label matching: image-plane IoU of every detection with the ground truths of its own
 * frame and the argmax, i.e. `calculate_iou_partly(dt_annos, gt_annos, metric=0)` followed by
 * `np.argmax(c_overlap, axis=-1)` in tools/utils_pseudo_labels_gga.py:44-59 (IoU arithmetic:
 * image_box_overlap, mmdet3d/core/evaluation/kitti_utils/eval.py:86-114, criterion -1).
 * dt_boxes [n_dt,4], gt_boxes [n_gt,4] = (x1,y1,x2,y2) f64, both concatenated over frames with
 * device offsets dt_offsets / gt_offsets [n_frames+1]. round_f32: the detections were float32
 * (the usual case: float32 predictions, float64 KITTI labels) - their own area is then float32
 * arithmetic and each overlap is rounded to float32, as the reference's typing gives. match [n_dt]: index of the first
 * maximum within the frame's ground truths (-1 if it has none); best_iou [n_dt] optional;
 * overlaps optional: frame f's [n_dt_f, n_gt_f] matrix at overlap_offsets[f]. */
int gga_image_box_match(const double* dt_boxes, const int64_t* dt_offsets, const double* gt_boxes,
                        const int64_t* gt_offsets, int n_frames, int64_t n_dt, int round_f32, int64_t* match,
                        double* best_iou, double* overlaps, const int64_t* overlap_offsets, void* stream);

/* ------------------------------------------------------------------------- */
/* Offline GGA label generation primitives,             */
/* tools/data_converter/utils_gga.py. float64, reference operation order.       */
/* ------------------------------------------------------------------------- */
/* region_grow (utils_gga.py:6-38) for n_thresholds distance thresholds at once (the reference
 * calls it for thresh = 0.1 .. 0.7 with the same masks, kitti_converter_gga.py:373-381).
 * pc [n_points, dim] f64 (dim <= 4; the reference passes homogeneous camera coordinates),
 * mask_search / mask_origin [n_points] u8 (origin must be a subset of search),
 * ratio: the early-exit in-box ratio (use_ratio 0 = the reference's ratio=None).
 * out_masks [n_thresholds, n_points] u8 = mask_best * mask_origin (or mask_best for ratio None). */
size_t gga_region_grow_workspace_bytes(int64_t n_points, int n_thresholds);
int gga_region_grow(const double* pc, int64_t n_points, int dim, const uint8_t* mask_search,
                    const uint8_t* mask_origin, const double* thresholds, int n_thresholds, double ratio,
                    int use_ratio, uint8_t* out_masks, void* workspace, size_t workspace_bytes, void* stream);

/* points_in_convex_polygon_3d_jit (mmdet3d/core/bbox/box_np_ops.py:641-705) as used by
 * points_in_frustm_indices (utils_gga.py:87-100): out[i, j] = 1 iff for every surface k of
 * polyhedron j: p_i . normal_vec[j,k] + d[j,k] < 0. points [n, point_stride >= 3] f64,
 * normal_vec [n_polyhedra, n_surfaces, 3], d [n_polyhedra, n_surfaces] (surface_equ_3d). */
int gga_points_in_convex_polyhedra(const double* points, int64_t n_points, int point_stride, const double* normal_vec,
                                   const double* d, int n_polyhedra, int n_surfaces, uint8_t* out, void* stream);

/* RANSAC scoring of calculate_ground (utils_gga.py:121-125): for every candidate plane a.p = 1,
 * counts[c] = #{ i : |p_i . plane_c - 1| / ||plane_c|| < threshold }; masks (optional)
 * [n_planes, n_points] u8 the inlier masks. */
int gga_plane_inliers(const double* points, int64_t n_points, int point_stride, const double* planes, int n_planes,
                      double threshold, int32_t* counts, uint8_t* masks, void* stream);

/* ------------------------------------------------------------------------- */
/* §8(f)4. Modulated deformable convolution (DCNv2), sampling half.           */
/* Replaces the un-vendored mmcv ModulatedDeformConv2dPack natives behind     */
/* `dcn_on_last_conv=True` of the PGD / FCOS3D heads                          */
/* (mmdet3d/models/dense_heads/anchor_free_mono3d_head.py:187-211,            */
/* configs/_base_/models/pgd.py:47). deform_groups = groups = 1.              */
/* ------------------------------------------------------------------------- */
/* x [B,H,W,C] f32 channels-last (C a multiple of 256, <= 1024); offset [B, 2*kh*kw, Ho, Wo] f32 NCHW with
 * channel 2k = vertical, 2k+1 = horizontal offset of tap k = i*kw + j (mmcv's layout); mask [B, kh*kw, Ho, Wo]
 * (already through the sigmoid). col [B*Ho*Wo, kh*kw*C] f32:
 * col[p, k, c] = mask[p,k] * bilinear(x[..,c], p*stride - pad + tap*dil + offset[p,k]); samples outside
 * (-1, H) x (-1, W) are 0, corners outside the image contribute 0. The convolution is then
 * y = col @ W[Cout, kh, kw, C]^T + bias (a library GEMM on the caller's side). */
int gga_dcn_im2col(const float* x, const float* offset, const float* mask, int B, int H, int W, int C, int kh, int kw,
                   int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, float* col, void* stream);
/* the same with one more output: the bits of the largest finite |col| written, max-combined into *amax_col (zeroed by the
 * caller; NULL: not wanted) - the scale of the two-fp16-plane matrix kernels that multiply col by the weight */
int gga_dcn_im2col_amax(const float* x, const float* offset, const float* mask, int B, int H, int W, int C, int kh, int kw,
                        int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, float* col, uint32_t* amax_col,
                        void* stream);
/* Backward of the sampling: grad_col [B*Ho*Wo, kh*kw*C] -> grad_x [B,H,W,C] (zero-filled here, float atomic
 * adds: the scatter targets are data dependent; NULL to skip), grad_offset / grad_mask in the layouts above. */
int gga_dcn_col2im(const float* x, const float* offset, const float* mask, const float* grad_col, int B, int H, int W,
                   int C, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                   float* grad_x, float* grad_offset, float* grad_mask, void* stream);

/* FCOS3D / PGD target assignment (FCOSMono3DHead._get_target_single, fcos_mono3d_head.py:773-956, for the
 * whole batch in one launch). points [P,2] f32: all levels concatenated, level l = rows
 * level_begin_host[l] .. level_begin_host[l+1] (host int32 [n_levels]; the last level ends at P) with stride
 * strides_host[l] and regress range regress_ranges_host[2l], [2l+1]. Ground truths of all images concatenated,
 * image b = rows gt_offsets[b] .. gt_offsets[b+1] (device int64 [batch+1]): gt_bboxes [G,4], centers2d [G,2],
 * depths [G], gt_bboxes_3d [G,code] (yaw already made local), labels int64. Outputs [batch, P(, .)]:
 * labels, bbox_targets (l, t, r, b), labels_3d, bbox_targets_3d (dx, dy, depth, gt[3:]), centerness
 * exp(-alpha * |d| / (1.414 * stride * radius)), attr. Points with no admissible ground truth get the
 * background labels and (like the reference) the regression targets of the image's first box. */
int gga_fcos3d_targets(const float* points, int n_points, int n_levels, const int32_t* level_begin_host,
                       const float* strides_host, const float* regress_ranges_host, float center_sample_radius,
                       const int64_t* gt_offsets, int batch, const float* gt_bboxes, const float* centers2d,
                       const float* depths, const float* gt_bboxes_3d, int code_size, const int64_t* gt_labels,
                       const int64_t* gt_labels_3d, const int64_t* attr_labels, int64_t background_label,
                       int64_t attr_background_label, float centerness_alpha, int64_t* labels, float* bbox_targets,
                       int64_t* labels_3d, float* bbox_targets_3d, float* centerness_targets, int64_t* attr_targets,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GGA_HIP_H_ */
