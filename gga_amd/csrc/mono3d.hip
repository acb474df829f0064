// FCOS3D / PGD target assignment for gfx950: one thread per (image, feature-map point) walks that
// image's ground truths and picks the one whose projected 3D centre is nearest among those that
// (a) contain the point in their centre-sampling box and (b) regress within the level's range.
//
// Reference: FCOSMono3DHead._get_target_single (mmdet3d/models/dense_heads/fcos_mono3d_head.py:
// 773-956), ~60 broadcast tensor ops over [points, gts] per image in a Python loop over images.
// Arithmetic follows the reference's float32 operations one by one (no fused multiply-add: the
// nearest-centre argmin and the two inclusion tests decide integer labels, which must be exact).
#include "gga_common.h"

#define M3D_INF 1e8f

struct M3dLevels {
    int n_levels;
    int begin[8];           // first point of each level in the concatenated point list (+ total at [n_levels])
    float stride_radius[8]; // strides[l] * center_sample_radius
    float lo[8], hi[8];     // regress range of the level
};

__global__ __launch_bounds__(256) void fcos3d_targets_kernel(
    const float* __restrict__ points, int P, M3dLevels lv, const int64_t* __restrict__ gt_offsets, int B,
    const float* __restrict__ gt_bboxes, const float* __restrict__ centers2d, const float* __restrict__ depths,
    const float* __restrict__ gt_bboxes_3d, int code, const int64_t* __restrict__ gt_labels,
    const int64_t* __restrict__ gt_labels_3d, const int64_t* __restrict__ attr_labels, int64_t background, int64_t attr_background,
    float centerness_alpha, int64_t* __restrict__ labels, float* __restrict__ bbox_targets, int64_t* __restrict__ labels_3d,
    float* __restrict__ bbox_targets_3d, float* __restrict__ centerness, int64_t* __restrict__ attr_out) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * P) return;
    const int b = (int)(i / P), p = (int)(i - (int64_t)b * P);
    int l = 0;
    while (l + 1 < lv.n_levels && p >= lv.begin[l + 1]) ++l;
    const float sr = lv.stride_radius[l], lo = lv.lo[l], hi = lv.hi[l];
    const float xs = points[2 * p], ys = points[2 * p + 1];
    const int64_t g0 = gt_offsets[b], g1 = gt_offsets[b + 1];
    float* bt = bbox_targets + i * 4;
    float* t3 = bbox_targets_3d + i * code;
    if (g1 == g0) {            // no ground truth in this image
        labels[i] = background; labels_3d[i] = background; attr_out[i] = attr_background; centerness[i] = 0.f;
        bt[0] = bt[1] = bt[2] = bt[3] = 0.f;
        for (int c = 0; c < code; ++c) t3[c] = 0.f;
        return;
    }
    float best = 0.f;
    int64_t bi = -1;
    for (int64_t g = g0; g < g1; ++g) {
        const float cx = centers2d[2 * g], cy = centers2d[2 * g + 1];
        const float dx = xs - cx, dy = ys - cy;
        const float left = xs - gt_bboxes[4 * g], top = ys - gt_bboxes[4 * g + 1];
        const float right = gt_bboxes[4 * g + 2] - xs, bottom = gt_bboxes[4 * g + 3] - ys;
        // centre-sampling box [c - sr, c + sr]: the reference forms the box corners first
        const float bx0 = cx - sr, by0 = cy - sr, bx1 = cx + sr, by1 = cy + sr;
        const float cmin = fminf(fminf(xs - bx0, ys - by0), fminf(bx1 - xs, by1 - ys));
        const float mx = fmaxf(fmaxf(left, top), fmaxf(right, bottom));
        const float dxx = dx * dx, dyy = dy * dy;
        float d = sqrtf(dxx + dyy);
        if (!(cmin > 0.f)) d = M3D_INF;
        if (!(mx >= lo && mx <= hi)) d = M3D_INF;
        if (bi < 0 || d < best) { best = d; bi = g; }      // first minimum, as torch.min
    }
    const bool none = best == M3D_INF;
    labels[i] = none ? background : gt_labels[bi];
    labels_3d[i] = none ? background : gt_labels_3d[bi];
    attr_out[i] = none ? attr_background : attr_labels[bi];
    bt[0] = xs - gt_bboxes[4 * bi]; bt[1] = ys - gt_bboxes[4 * bi + 1];
    bt[2] = gt_bboxes[4 * bi + 2] - xs; bt[3] = gt_bboxes[4 * bi + 3] - ys;
    const float dx = xs - centers2d[2 * bi], dy = ys - centers2d[2 * bi + 1];
    t3[0] = dx; t3[1] = dy; t3[2] = depths[bi];
    for (int c = 3; c < code; ++c) t3[c] = gt_bboxes_3d[bi * code + c];
    const float dxx = dx * dx, dyy = dy * dy;
    const float denom = 1.414f * sr;
    const float rel = sqrtf(dxx + dyy) / denom;
    centerness[i] = expf(-centerness_alpha * rel);
}

extern "C" int gga_fcos3d_targets(const float* points, int n_points, int n_levels, const int32_t* level_begin_host,
                                  const float* strides_host, const float* regress_ranges_host, float center_sample_radius,
                                  const int64_t* gt_offsets, int batch, const float* gt_bboxes, const float* centers2d,
                                  const float* depths, const float* gt_bboxes_3d, int code_size, const int64_t* gt_labels,
                                  const int64_t* gt_labels_3d, const int64_t* attr_labels, int64_t background_label,
                                  int64_t attr_background_label, float centerness_alpha, int64_t* labels, float* bbox_targets,
                                  int64_t* labels_3d, float* bbox_targets_3d, float* centerness_targets, int64_t* attr_targets,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(points && level_begin_host && strides_host && regress_ranges_host && gt_offsets && labels && bbox_targets &&
                    labels_3d && bbox_targets_3d && centerness_targets && attr_targets,
                "gga_fcos3d_targets: null pointer argument");
    GGA_REQUIRE(n_points >= 1 && batch >= 1 && n_levels >= 1 && n_levels <= 7 && code_size >= 7 && code_size <= 16,
                "gga_fcos3d_targets: bad sizes (points=%d batch=%d levels=%d code=%d)", n_points, batch, n_levels, code_size);
    M3dLevels lv;
    lv.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        lv.begin[l] = level_begin_host[l];
        lv.stride_radius[l] = strides_host[l] * center_sample_radius;       // float32 product, as stride * radius on a float32 tensor
        lv.lo[l] = regress_ranges_host[2 * l];
        lv.hi[l] = regress_ranges_host[2 * l + 1];
    }
    lv.begin[n_levels] = n_points;
    const int64_t total = (int64_t)batch * n_points;
    hipLaunchKernelGGL(fcos3d_targets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, points, n_points, lv,
                       gt_offsets, batch, gt_bboxes, centers2d, depths, gt_bboxes_3d, code_size, gt_labels, gt_labels_3d,
                       attr_labels, background_label, attr_background_label, centerness_alpha, labels, bbox_targets, labels_3d,
                       bbox_targets_3d, centerness_targets, attr_targets);
    GGA_CHECK_LAUNCH("fcos3d_targets_kernel");
    return GGA_OK;
}
