// §8(f) rank 1 — inference / pseudo-label post-processing ops for gfx950:
// rotated BEV IoU, rotated NMS, points-in-boxes.
//
// The reference calls the un-vendored mmcv natives:
//   mmcv.ops.nms_rotated      <- mmdet3d/core/post_processing/box3d_nms.py:231-268 (nms_bev),
//                                used by CenterHead_GGA.get_task_detections (head:885-890)
//   mmcv.ops.box_iou_rotated  <- mmdet3d/core/bbox/structures/base_box3d.py:469 (overlaps)
//   mmcv.ops.points_in_boxes_part / _all <- base_box3d.py:534,566
// Their published semantics are restated (rotated IoU = exact polygon overlap of the two
// rectangles; greedy suppression of lower-scored boxes with IoU > thr; a point is inside a box
// when |z - zc| <= dz/2 and the yaw-aligned offsets are strictly inside +-dx/2, +-dy/2) and pinned
// by the reference's own known-answer tests (tests/test_utils/test_nms.py:82-120,
// tests/test_utils/test_box3d.py:1122-1187,1683-1790).
//
// NMS runs fully on the device (mmcv copies the N x N bit mask to the host for the greedy scan):
//   nms_mask_kernel   64x64 IoU blocks -> bit mask rows (only the upper triangle)
//   nms_scan_kernel   one wavefront walks the score-sorted boxes, the "removed" bit set lives in
//                     registers (one 64-bit word per lane per 4096 boxes), kept rows OR their mask
//                     row into it.
#include "gga_common.h"

struct P2 { float x, y; };

__device__ __forceinline__ float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }

__device__ __forceinline__ void rect_corners(const float* b, float sx, float sy, P2 out[4]) {
    const float c = cosf(b[4]), s = sinf(b[4]);
    const float hw = b[2] * 0.5f, hh = b[3] * 0.5f;
    const float cx = b[0] - sx, cy = b[1] - sy;
    const float dx[4] = { -hw, hw, hw, -hw }, dy[4] = { -hh, -hh, hh, hh };
#pragma unroll
    for (int i = 0; i < 4; ++i) { out[i].x = cx + dx[i] * c - dy[i] * s; out[i].y = cy + dx[i] * s + dy[i] * c; }
}

// exact overlap area of two rotated rectangles (x, y, w, h, angle): Sutherland-Hodgman clip of
// rectangle 1 by the four half planes of rectangle 2 (both convex, counter-clockwise)
__device__ float rotated_inter_area(const float* b1, const float* b2) {
    // shift both to their mid point for precision, as mmcv's box_iou_rotated_utils does
    const float sx = (b1[0] + b2[0]) * 0.5f, sy = (b1[1] + b2[1]) * 0.5f;
    P2 poly[10], tmp[10], q[4];
    rect_corners(b1, sx, sy, poly);
    rect_corners(b2, sx, sy, q);
    int n = 4;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const P2 a = q[e], bq = q[(e + 1) & 3];
        const P2 ed = { bq.x - a.x, bq.y - a.y };
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const P2 p = poly[i], r = poly[(i + 1) % n];
            const float dp = cross2(ed, P2{ p.x - a.x, p.y - a.y });
            const float dr = cross2(ed, P2{ r.x - a.x, r.y - a.y });
            if (dp >= 0.0f) tmp[m++] = p;
            if ((dp >= 0.0f) != (dr >= 0.0f)) {
                const float t = dp / (dp - dr);
                tmp[m++] = P2{ p.x + t * (r.x - p.x), p.y + t * (r.y - p.y) };
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
    }
    if (n < 3) return 0.0f;
    float area = 0.0f;
    for (int i = 0; i < n; ++i) area += cross2(poly[i], poly[(i + 1) % n]);
    return fabsf(area) * 0.5f;
}

__device__ __forceinline__ float rotated_iou(const float* b1, const float* b2, int mode_iof) {
    const float a1 = b1[2] * b1[3], a2 = b2[2] * b2[3];
    if (a1 < 1e-14f || a2 < 1e-14f) return 0.0f;
    const float inter = rotated_inter_area(b1, b2);
    const float base = mode_iof ? a1 : (a1 + a2 - inter);
    return inter / base;
}

// ------------------------------------------------------------------------------ pairwise IoU
__global__ __launch_bounds__(256) void box_iou_rotated_kernel(const float* __restrict__ b1, int n,
                                                             const float* __restrict__ b2, int m, int mode_iof,
                                                             int aligned, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (aligned) {
        if (t >= n) return;
        out[t] = rotated_iou(b1 + t * 5, b2 + t * 5, mode_iof);
    } else {
        if (t >= (int64_t)n * m) return;
        const int i = (int)(t / m), j = (int)(t - (int64_t)i * m);
        out[t] = rotated_iou(b1 + (int64_t)i * 5, b2 + (int64_t)j * 5, mode_iof);
    }
}

extern "C" int gga_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int m, int mode_iof, int aligned,
                                   float* out, void* stream) {
    GGA_REQUIRE(n >= 0 && m >= 0 && (!aligned || n == m), "gga_box_iou_rotated: bad sizes (n=%d m=%d aligned=%d)", n, m, aligned);
    const int64_t total = aligned ? n : (int64_t)n * m;
    if (total == 0) return GGA_OK;
    GGA_REQUIRE(boxes1 && boxes2 && out, "gga_box_iou_rotated: null pointer argument");
    hipLaunchKernelGGL(box_iou_rotated_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       boxes1, n, boxes2, m, mode_iof, aligned, out);
    GGA_CHECK_LAUNCH("box_iou_rotated_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ rotated NMS
// boxes are already sorted by descending score (the caller sorts, as mmcv's python wrapper does)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thr,
                                                     unsigned long long* __restrict__ mask, int colblocks) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;                               // only j > i matters
    __shared__ float cols[64 * 5];
    const int t = threadIdx.x;
    const int ncol = min(64, n - cb * 64);
    if (t < ncol)
#pragma unroll
        for (int k = 0; k < 5; ++k) cols[t * 5 + k] = boxes[(int64_t)(cb * 64 + t) * 5 + k];
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    float me[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) me[k] = boxes[(int64_t)i * 5 + k];
    unsigned long long bits = 0;
    const int j0 = (rb == cb) ? t + 1 : 0;
    for (int j = j0; j < ncol; ++j)
        if (rotated_iou(me, cols + j * 5, 0) > thr) bits |= 1ull << j;
    mask[(int64_t)i * colblocks + cb] = bits;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int colblocks,
                                                     int max_keep, int64_t* __restrict__ keep,
                                                     int32_t* __restrict__ num_keep) {
    // lane l owns words l, l+64, ... of the "removed" set (<= 8 words per lane: n <= 32768)
    const int lane = threadIdx.x;
    unsigned long long removed[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) removed[w] = 0;
    int nk = 0;
    for (int i = 0; i < n && nk < max_keep; ++i) {
        const int word = i >> 6, owner = word & 63, slot = word >> 6;
        unsigned long long wv = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) if (w == slot) wv = removed[w];
        const unsigned long long ow = __shfl(wv, owner, 64);
        if ((ow >> (i & 63)) & 1ull) continue;
        if (lane == 0) keep[nk] = i;
        ++nk;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int cw = lane + 64 * w;
            if (cw < colblocks && cw >= word) removed[w] |= mask[(int64_t)i * colblocks + cw];
        }
    }
    if (lane == 0) *num_keep = nk;
}

// Circular NMS (mmdet3d/core/post_processing/box3d_nms.py:181-225): score-sorted centres; a kept
// centre suppresses every later one within squared distance `thresh`. Same two-kernel scheme: 64x64
// blocks of the "within thresh" relation as bit-mask rows, then the one-wavefront greedy scan.
// The distance follows the reference's float32 arithmetic (numba on a float32 array: subtract,
// square, add as separately rounded float32 operations; the comparison with the python-float
// threshold happens in float64).
__global__ __launch_bounds__(64) void circle_mask_kernel(const float* __restrict__ xy, int n, double thr,
                                                        unsigned long long* __restrict__ mask, int colblocks) {
#pragma clang fp contract(off)
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    __shared__ float cols[64 * 2];
    const int t = threadIdx.x;
    const int ncol = min(64, n - cb * 64);
    if (t < ncol) { cols[2 * t] = xy[(int64_t)(cb * 64 + t) * 2]; cols[2 * t + 1] = xy[(int64_t)(cb * 64 + t) * 2 + 1]; }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    const float x = xy[(int64_t)i * 2], y = xy[(int64_t)i * 2 + 1];
    unsigned long long bits = 0;
    const int j0 = (rb == cb) ? t + 1 : 0;
    for (int j = j0; j < ncol; ++j) {
        const float dx = x - cols[2 * j], dy = y - cols[2 * j + 1];
        const float dxx = dx * dx, dyy = dy * dy;
        const float d = dxx + dyy;
        if ((double)d <= thr) bits |= 1ull << j;
    }
    mask[(int64_t)i * colblocks + cb] = bits;
}

extern "C" size_t gga_nms_rotated_workspace_bytes(int n) {
    const size_t cb = (size_t)(n + 63) / 64;
    return (size_t)n * cb * 8 + 256;
}

extern "C" int gga_nms_rotated_sorted(const float* boxes_sorted, int n, float iou_threshold, int max_keep,
                                      int64_t* keep, int32_t* num_keep, void* workspace, size_t workspace_bytes,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(num_keep && n >= 0 && n <= 32768, "gga_nms_rotated_sorted: n=%d not in [0, 32768]", n);
    if (n == 0) {
        GGA_CHECK_HIP(hipMemsetAsync(num_keep, 0, 4, stream), "nms memset");
        return GGA_OK;
    }
    GGA_REQUIRE(boxes_sorted && keep && workspace, "gga_nms_rotated_sorted: null pointer argument");
    if (workspace_bytes < gga_nms_rotated_workspace_bytes(n)) {
        gga_set_error("gga_nms_rotated_sorted: workspace %zu B < required %zu B", workspace_bytes,
                      gga_nms_rotated_workspace_bytes(n));
        return GGA_ERR_WORKSPACE;
    }
    const int cb = (n + 63) / 64;
    unsigned long long* mask = (unsigned long long*)workspace;
    GGA_CHECK_HIP(hipMemsetAsync(mask, 0, (size_t)n * cb * 8, stream), "nms memset");
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, stream, boxes_sorted, n, iou_threshold, mask, cb);
    GGA_CHECK_LAUNCH("nms_mask_kernel");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, stream, mask, n, cb, max_keep > 0 ? max_keep : n, keep,
                       num_keep);
    GGA_CHECK_LAUNCH("nms_scan_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ points in boxes
// box = (x, y, z_bottom, dx, dy, dz, yaw). mmcv's check_pt_in_box3d.
__device__ __forceinline__ bool pt_in_box3d(float px, float py, float pz, const float* b) {
    const float cz = b[2] + b[5] * 0.5f;
    if (fabsf(pz - cz) > b[5] * 0.5f) return false;
    const float sx = px - b[0], sy = py - b[1];
    const float c = cosf(-b[6]), s = sinf(-b[6]);
    const float lx = sx * c - sy * s, ly = sx * s + sy * c;
    return (lx > -b[3] * 0.5f) & (lx < b[3] * 0.5f) & (ly > -b[4] * 0.5f) & (ly < b[4] * 0.5f);
}

__global__ __launch_bounds__(256) void points_in_boxes_kernel(const float* __restrict__ pts, const float* __restrict__ boxes,
                                                             int B, int M, int T, int all, int32_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float* p = pts + ((int64_t)b * M + i) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float* bb = boxes + (int64_t)b * T * 7;
    if (all) {
        int32_t* o = out + ((int64_t)b * M + i) * T;
        for (int t = 0; t < T; ++t) o[t] = pt_in_box3d(px, py, pz, bb + t * 7) ? 1 : 0;
    } else {
        int32_t idx = -1;
        for (int t = 0; t < T; ++t)
            if (pt_in_box3d(px, py, pz, bb + t * 7)) { idx = t; break; }
        out[(int64_t)b * M + i] = idx;
    }
}

// all = 1 writes B*M*T flags: one thread per (point, box) with the box index fastest, so a
// wavefront writes 256 contiguous bytes; the boxes of the tile (centre z, half sizes, cos/sin of
// -yaw: the same expressions as pt_in_box3d, evaluated once per box and workgroup) sit in LDS.
#define PIB_PTS 64
__global__ __launch_bounds__(256) void points_in_boxes_all_kernel(const float* __restrict__ pts,
                                                                 const float* __restrict__ boxes, int M, int T,
                                                                 int32_t* __restrict__ out) {
    __shared__ float sb[256][8];
    const int b = blockIdx.y, p0 = blockIdx.x * PIB_PTS;
    const int np = min(PIB_PTS, M - p0);
    const float* bb = boxes + (int64_t)b * T * 7;
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int tn = min(256, T - t0);
        __syncthreads();
        if ((int)threadIdx.x < tn) {
            const float* q = bb + (int64_t)(t0 + threadIdx.x) * 7;
            float* d = sb[threadIdx.x];
            d[0] = q[0]; d[1] = q[1]; d[2] = q[2] + q[5] * 0.5f; d[3] = q[3] * 0.5f; d[4] = q[4] * 0.5f; d[5] = q[5] * 0.5f;
            d[6] = cosf(-q[6]); d[7] = sinf(-q[6]);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < np * tn; e += 256) {
            const int p = e / tn, t = e - p * tn;
            const float* pt = pts + ((int64_t)b * M + p0 + p) * 3;
            const float* d = sb[t];
            bool in = !(fabsf(pt[2] - d[2]) > d[5]);
            const float sx = pt[0] - d[0], sy = pt[1] - d[1];
            const float lx = sx * d[6] - sy * d[7], ly = sx * d[7] + sy * d[6];
            in = in & (lx > -d[3]) & (lx < d[3]) & (ly > -d[4]) & (ly < d[4]);
            out[((int64_t)b * M + p0 + p) * T + t0 + t] = in ? 1 : 0;
        }
    }
}

extern "C" int gga_points_in_boxes(const float* points, const float* boxes, int B, int M, int T, int all, int32_t* out,
                                   void* stream) {
    GGA_REQUIRE(B >= 1 && M >= 0 && T >= 0, "gga_points_in_boxes: bad sizes");
    if (M == 0) return GGA_OK;
    GGA_REQUIRE(points && out && (T == 0 || boxes), "gga_points_in_boxes: null pointer argument");
    if (all) {
        if (T == 0) return GGA_OK;
        hipLaunchKernelGGL(points_in_boxes_all_kernel, dim3((M + PIB_PTS - 1) / PIB_PTS, B), dim3(256), 0,
                           (hipStream_t)stream, points, boxes, M, T, out);
        GGA_CHECK_LAUNCH("points_in_boxes_all_kernel");
        return GGA_OK;
    }
    hipLaunchKernelGGL(points_in_boxes_kernel, dim3((M + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, points, boxes,
                       B, M, T, all, out);
    GGA_CHECK_LAUNCH("points_in_boxes_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ pseudo-label matching
// KITTI image-plane IoU (image_box_overlap, criterion -1) of every detection with the ground
// truths of its own frame, and the index of the best one (numpy argmax: first maximum). Doubles
// throughout, one multiply/add/divide per reference operation (the empty asm statements stop
// hipcc from contracting a*b+c into an fma, which would change the last bit). f32_boxes = the
// detections are float32: their own area is float32 arithmetic and each overlap is rounded to
// float32 (the reference's typing with float32 detections and float64 ground truth).
__device__ __forceinline__ double img_iou(const double* b, const double* q, int f32_boxes) {
    const double qa0 = q[2] - q[0], qa1 = q[3] - q[1];
    double qarea = qa0 * qa1;
    asm volatile("" : "+v"(qarea));
    const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
    if (!(iw > 0)) return 0.0;
    const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
    if (!(ih > 0)) return 0.0;
    double barea;
    if (f32_boxes) {                           // float32 detections: their own area is float32 arithmetic
        const float w = (float)b[2] - (float)b[0], h = (float)b[3] - (float)b[1];
        float a = w * h;
        asm volatile("" : "+v"(a));
        barea = (double)a;
    } else {
        barea = (b[2] - b[0]) * (b[3] - b[1]);
        asm volatile("" : "+v"(barea));
    }
    double inter = iw * ih;
    asm volatile("" : "+v"(inter));
    double ua = barea + qarea;
    asm volatile("" : "+v"(ua));
    ua = ua - inter;
    return inter / ua;
}

__global__ __launch_bounds__(256) void image_box_match_kernel(const double* __restrict__ dt, const int64_t* __restrict__ dt_off,
                                                             const double* __restrict__ gt, const int64_t* __restrict__ gt_off,
                                                             int n_frames, int64_t n_dt, int round_f32,
                                                             int64_t* __restrict__ match, double* __restrict__ best_iou,
                                                             double* __restrict__ overlaps,
                                                             const int64_t* __restrict__ ov_off) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_dt) return;
    int lo = 0, hi = n_frames;                 // frame f with dt_off[f] <= i < dt_off[f+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (dt_off[mid] <= i) lo = mid; else hi = mid; }
    const int f = lo;
    const int64_t g0 = gt_off[f], ng = gt_off[f + 1] - g0;
    const double* b = dt + i * 4;
    int64_t arg = -1;
    double best = 0.0;
    for (int64_t g = 0; g < ng; ++g) {
        double v = img_iou(b, gt + (g0 + g) * 4, round_f32);
        if (round_f32) v = (double)(float)v;
        if (overlaps) overlaps[ov_off[f] + (i - dt_off[f]) * ng + g] = v;
        if (arg < 0 || v > best) { best = v; arg = g; }
    }
    match[i] = arg;                            // -1 when the frame has no ground truth
    if (best_iou) best_iou[i] = best;
}

extern "C" int gga_image_box_match(const double* dt_boxes, const int64_t* dt_offsets, const double* gt_boxes,
                                   const int64_t* gt_offsets, int n_frames, int64_t n_dt, int round_f32, int64_t* match,
                                   double* best_iou, double* overlaps, const int64_t* overlap_offsets, void* stream) {
    GGA_REQUIRE(n_frames >= 1 && n_dt >= 0, "gga_image_box_match: bad sizes");
    if (n_dt == 0) return GGA_OK;
    GGA_REQUIRE(dt_boxes && dt_offsets && gt_offsets && match, "gga_image_box_match: null pointer argument");
    GGA_REQUIRE(!overlaps || overlap_offsets, "gga_image_box_match: overlaps needs overlap_offsets");
    hipLaunchKernelGGL(image_box_match_kernel, dim3((unsigned)((n_dt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dt_boxes, dt_offsets, gt_boxes, gt_offsets, n_frames, n_dt, round_f32, match, best_iou, overlaps,
                       overlap_offsets);
    GGA_CHECK_LAUNCH("image_box_match_kernel");
    return GGA_OK;
}

extern "C" int gga_circle_nms_sorted(const float* xy_sorted, int n, double thresh, int max_keep, int64_t* keep,
                                     int32_t* num_keep, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(num_keep && n >= 0 && n <= 32768, "gga_circle_nms_sorted: n=%d not in [0, 32768]", n);
    if (n == 0) {
        GGA_CHECK_HIP(hipMemsetAsync(num_keep, 0, 4, stream), "nms memset");
        return GGA_OK;
    }
    GGA_REQUIRE(xy_sorted && keep && workspace, "gga_circle_nms_sorted: null pointer argument");
    if (workspace_bytes < gga_nms_rotated_workspace_bytes(n)) {
        gga_set_error("gga_circle_nms_sorted: workspace %zu B < required %zu B", workspace_bytes,
                      gga_nms_rotated_workspace_bytes(n));
        return GGA_ERR_WORKSPACE;
    }
    const int cb = (n + 63) / 64;
    unsigned long long* mask = (unsigned long long*)workspace;
    GGA_CHECK_HIP(hipMemsetAsync(mask, 0, (size_t)n * cb * 8, stream), "nms memset");
    hipLaunchKernelGGL(circle_mask_kernel, dim3(cb, cb), dim3(64), 0, stream, xy_sorted, n, thresh, mask, cb);
    GGA_CHECK_LAUNCH("circle_mask_kernel");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, stream, mask, n, cb, max_keep > 0 ? max_keep : n, keep,
                       num_keep);
    GGA_CHECK_LAUNCH("nms_scan_kernel");
    return GGA_OK;
}

// ------------------------------------------------------------------------------ CenterPoint detections of a whole batch
// CenterHead_GGA.get_bboxes + get_task_detections (centerpoint_head_gga.py:725-934) after the coder's top-k decode
// (centerpoint_bbox_coders.py:117-229), for ALL frames and tasks in ONE launch (round 6: the per-(frame, task) python loop
// of the reference - boolean masks, sorts, two NMS launches and a count read-back each, ~25 launches and 4 host
// synchronisations per pair - was 4-5 ms per frame, 78 % of the pseudo-label run's device-side time per frame).
// One workgroup per (frame, task); thread i owns candidate i of the K <= 128 decoded boxes (already in descending score
// order: torch.topk) and the steps are the reference's, in its order (a second small launch does step 6 per frame):
//   1 coder mask      centre inside post_center_range (inclusive) and score > the coder's threshold      (bbox_coders:221-229)
//   2 head threshold  score >= test_cfg.score_threshold when that is > 0                                     (head:836-846)
//   3 NMS boxes       bev (x, y, dx, dy, yaw) -> xywhr2xyxyr -> nms_bev's conversion back to xywhr: centre -/+ extent / 2,
//                     then (x1 + x2) / 2 and x2 - x1 - the same separately rounded float operations           (head:885-890, box3d_nms.py:258-262)
//   4 rotated NMS     the first pre_max_size candidates; exact polygon IoU (rotated_iou above) > nms_thr suppresses the
//                     lower-scored box; greedy in score order; at most post_max_size survivors
//   5 range filter    centre inside post_center_limit_range (inclusive)                                       (head:905-911)
//   6 merge           tasks concatenated in task order, z moved from the gravity centre to the bottom (z - dz * 0.5, two
//                     rounded operations), label = class within the task + the classes of the tasks before     (head:797-817)
// Output: frame b's detections compacted at out_*[b, 0 .. count[b]).
#define CPD_THREADS 256
#define CPD_K 128
// rotated_inter_area / rotated_iou with the boxes' cos / sin handed in (computed once per candidate instead of once per pair:
// the same float operations on the same values, so the overlap is bit-identical to rotated_iou's)
__device__ __forceinline__ void rect_corners_cs(const float* b, float c, float s, float sx, float sy, P2 out[4]) {
    const float hw = b[2] * 0.5f, hh = b[3] * 0.5f;
    const float cx = b[0] - sx, cy = b[1] - sy;
    const float dx[4] = { -hw, hw, hw, -hw }, dy[4] = { -hh, -hh, hh, hh };
#pragma unroll
    for (int i = 0; i < 4; ++i) { out[i].x = cx + dx[i] * c - dy[i] * s; out[i].y = cy + dx[i] * s + dy[i] * c; }
}

__device__ float rotated_iou_cs(const float* b1, float c1, float s1, const float* b2, float c2, float s2) {
    const float a1 = b1[2] * b1[3], a2 = b2[2] * b2[3];
    if (a1 < 1e-14f || a2 < 1e-14f) return 0.0f;
    const float sx = (b1[0] + b2[0]) * 0.5f, sy = (b1[1] + b2[1]) * 0.5f;
    P2 poly[10], tmp[10], q[4];
    rect_corners_cs(b1, c1, s1, sx, sy, poly);
    rect_corners_cs(b2, c2, s2, sx, sy, q);
    int n = 4;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const P2 a = q[e], bq = q[(e + 1) & 3];
        const P2 ed = { bq.x - a.x, bq.y - a.y };
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const P2 p = poly[i], r = poly[(i + 1) % n];
            const float dp = cross2(ed, P2{ p.x - a.x, p.y - a.y });
            const float dr = cross2(ed, P2{ r.x - a.x, r.y - a.y });
            if (dp >= 0.0f) tmp[m++] = p;
            if ((dp >= 0.0f) != (dr >= 0.0f)) {
                const float t = dp / (dp - dr);
                tmp[m++] = P2{ p.x + t * (r.x - p.x), p.y + t * (r.y - p.y) };
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
    }
    float inter = 0.0f;
    if (n >= 3) {
        float area = 0.0f;
        for (int i = 0; i < n; ++i) area += cross2(poly[i], poly[(i + 1) % n]);
        inter = fabsf(area) * 0.5f;
    }
    return inter / (a1 + a2 - inter);
}

// One 256-thread workgroup per (frame, task): steps 1-5 of the list above; the survivors go to the task's own segment
// out_*[b, t * K ...] with their number in seg_count[b, t]. Thread pair (i, i + 128) shares candidate i's row of the overlap
// mask: each takes half of the later candidates (the row's work falls with i, the split keeps the long rows short).
__global__ __launch_bounds__(CPD_THREADS) void centerpoint_detect_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                                        const float* __restrict__ labels, int T, int B, int K, int D,
                                                                        const float* __restrict__ coder_range, float coder_thr,
                                                                        int has_coder_thr, float score_thr,
                                                                        const float* __restrict__ limit_range, float nms_thr,
                                                                        int pre_max, int post_max, const int32_t* __restrict__ class_offset,
                                                                        const int32_t* __restrict__ single_class,
                                                                        float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                                        int32_t* __restrict__ out_labels, int32_t* __restrict__ seg_count) {
#pragma clang fp contract(off)
    __shared__ float nb[CPD_K * 5], ncs[CPD_K * 2];
    __shared__ unsigned long long mask[CPD_K][2][2];       // [row][half of the pair][word]
    __shared__ int cand[CPD_K], kept[CPD_K];
    __shared__ int wave_n[2], n_kept_s;
    const int b = blockIdx.x % B, t = blockIdx.x / B, tid = threadIdx.x, i = tid & (CPD_K - 1), half = tid >> 7;
    const int lane = tid & 63, wave = (tid >> 6) & 1;      // (of the first 128 threads: the candidates' owners)
    const float* bx = boxes + ((int64_t)(t * B + b) * K + (i < K ? i : 0)) * D;
    float v[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) v[d] = d < D ? bx[d] : 0.0f;
    const float s = scores[(int64_t)(t * B + b) * K + (i < K ? i : 0)];
    bool ok = i < K && half == 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) ok = ok && v[d] >= coder_range[d] && v[d] <= coder_range[3 + d];
    if (has_coder_thr) ok = ok && s > coder_thr;
    if (score_thr > 0.0f) ok = ok && s >= score_thr;
    // rank among the survivors, in candidate (= score) order
    const unsigned long long bal = __ballot(ok);
    if (half == 0 && lane == 0) wave_n[wave] = __popcll(bal);
    __syncthreads();
    const int rank = __popcll(bal & ((1ull << lane) - 1ull)) + (wave ? wave_n[0] : 0);
    int n = wave_n[0] + wave_n[1];
    if (n > pre_max) n = pre_max;
    if (ok && rank < n) {
        cand[rank] = i;
        const float hw = v[3] / 2.0f, hh = v[4] / 2.0f;
        const float x1 = v[0] - hw, y1 = v[1] - hh, x2 = v[0] + hw, y2 = v[1] + hh;
        nb[rank * 5 + 0] = (x1 + x2) / 2.0f; nb[rank * 5 + 1] = (y1 + y2) / 2.0f;
        nb[rank * 5 + 2] = x2 - x1; nb[rank * 5 + 3] = y2 - y1; nb[rank * 5 + 4] = v[6];
        ncs[rank * 2] = cosf(v[6]); ncs[rank * 2 + 1] = sinf(v[6]);
    }
    __syncthreads();
    if (i < n) {
        unsigned long long m0 = 0, m1 = 0;
        float me[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) me[k] = nb[i * 5 + k];
        const float mc = ncs[i * 2], ms = ncs[i * 2 + 1];
        const int mid = i + 1 + (n - i - 1) / 2;
        const int j0 = half ? mid : i + 1, j1 = half ? n : mid;
        for (int j = j0; j < j1; ++j)
            if (rotated_iou_cs(me, mc, ms, nb + j * 5, ncs[j * 2], ncs[j * 2 + 1]) > nms_thr) { if (j < 64) m0 |= 1ull << j; else m1 |= 1ull << (j - 64); }
        mask[i][half][0] = m0; mask[i][half][1] = m1;
    }
    __syncthreads();
    if (tid == 0) {                                        // greedy scan in score order (n <= 128: two words of "removed")
        unsigned long long r0 = 0, r1 = 0;
        int nk = 0;
        for (int c = 0; c < n && nk < post_max; ++c) {
            if (((c < 64 ? r0 >> c : r1 >> (c - 64)) & 1ull)) continue;
            kept[nk++] = c;
            r0 |= mask[c][0][0] | mask[c][1][0]; r1 |= mask[c][0][1] | mask[c][1][1];
        }
        n_kept_s = nk;
    }
    __syncthreads();
    const int nk = n_kept_s;
    // range filter of the survivors; thread q < 128 owns survivor q (its box is candidate cand[kept[q]])
    bool in = false;
    int src = 0;
    if (half == 0 && i < nk) {
        src = cand[kept[i]];
        const float* sb = boxes + ((int64_t)(t * B + b) * K + src) * D;
        in = true;
        if (limit_range)
#pragma unroll
            for (int d = 0; d < 3; ++d) in = in && sb[d] >= limit_range[d] && sb[d] <= limit_range[3 + d];
    }
    const unsigned long long bal2 = __ballot(in);
    __syncthreads();                                       // (wave_n is reused)
    if (half == 0 && lane == 0) wave_n[wave] = __popcll(bal2);
    __syncthreads();
    const int pos = t * K + __popcll(bal2 & ((1ull << lane) - 1ull)) + (wave ? wave_n[0] : 0);
    if (in) {
        const float* sb = boxes + ((int64_t)(t * B + b) * K + src) * D;
        float* ob = out_boxes + ((int64_t)b * T * K + pos) * D;
        for (int d = 0; d < D; ++d) ob[d] = sb[d];
        const float halfh = sb[5] * 0.5f;
        ob[2] = sb[2] - halfh;
        out_scores[(int64_t)b * T * K + pos] = scores[(int64_t)(t * B + b) * K + src];
        const int cls = single_class[t] ? 0 : (int)labels[(int64_t)(t * B + b) * K + src];
        out_labels[(int64_t)b * T * K + pos] = cls + class_offset[t];
    }
    if (tid == 0) seg_count[b * T + t] = wave_n[0] + wave_n[1];
}

// step 6: frame b's task segments [t * K, t * K + seg_count[b, t]) moved together in task order (in place: a segment only ever
// moves towards the front, and the workgroup finishes one before it starts the next)
__global__ __launch_bounds__(CPD_K) void centerpoint_merge_kernel(int T, int K, int D, const int32_t* __restrict__ seg_count,
                                                                  float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                                  int32_t* __restrict__ out_labels, int32_t* __restrict__ out_count) {
    const int b = blockIdx.x, i = threadIdx.x;
    int base = seg_count[b * T];
    for (int t = 1; t < T; ++t) {
        const int n = seg_count[b * T + t];
        float v[9];
        float sc = 0.0f;
        int lb = 0;
        const int64_t src = (int64_t)b * T * K + t * K + i, dst = (int64_t)b * T * K + base + i;
        if (i < n) {
            for (int d = 0; d < D; ++d) v[d] = out_boxes[src * D + d];
            sc = out_scores[src]; lb = out_labels[src];
        }
        __syncthreads();
        if (i < n && base != t * K) {
            for (int d = 0; d < D; ++d) out_boxes[dst * D + d] = v[d];
            out_scores[dst] = sc; out_labels[dst] = lb;
        }
        __syncthreads();
        base += n;
    }
    if (i == 0) out_count[b] = base;
}

extern "C" size_t gga_centerpoint_detect_workspace_bytes(int n_tasks, int n_frames) { return (size_t)n_tasks * n_frames * 4 + 16; }

extern "C" int gga_centerpoint_detect(const float* boxes, const float* scores, const float* labels, int n_tasks, int n_frames,
                                      int k, int box_dim, const float* coder_range, float coder_score_threshold,
                                      int has_coder_score_threshold, float score_threshold, const float* limit_range,
                                      float nms_threshold, int pre_max_size, int post_max_size, const int32_t* class_offset,
                                      const int32_t* single_class, float* out_boxes, float* out_scores, int32_t* out_labels,
                                      int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    GGA_REQUIRE(n_tasks >= 1 && n_frames >= 0 && k >= 1 && k <= CPD_K && box_dim >= 7 && box_dim <= 9,
                "gga_centerpoint_detect: bad sizes (tasks=%d frames=%d k=%d box_dim=%d; k <= 128, 7 <= box_dim <= 9)", n_tasks,
                n_frames, k, box_dim);
    if (n_frames == 0) return GGA_OK;
    GGA_REQUIRE(boxes && scores && labels && coder_range && class_offset && single_class && out_boxes && out_scores && out_labels &&
                    out_count && workspace, "gga_centerpoint_detect: null pointer argument");
    if (workspace_bytes < gga_centerpoint_detect_workspace_bytes(n_tasks, n_frames)) {
        gga_set_error("gga_centerpoint_detect: workspace %zu B < required %zu B", workspace_bytes,
                      gga_centerpoint_detect_workspace_bytes(n_tasks, n_frames));
        return GGA_ERR_WORKSPACE;
    }
    int32_t* seg_count = (int32_t*)workspace;
    hipLaunchKernelGGL(centerpoint_detect_kernel, dim3((unsigned)(n_frames * n_tasks)), dim3(CPD_THREADS), 0, (hipStream_t)stream, boxes,
                       scores, labels, n_tasks, n_frames, k, box_dim, coder_range, coder_score_threshold, has_coder_score_threshold,
                       score_threshold, limit_range, nms_threshold, pre_max_size > 0 ? pre_max_size : k,
                       post_max_size > 0 ? post_max_size : k, class_offset, single_class, out_boxes, out_scores, out_labels, seg_count);
    GGA_CHECK_LAUNCH("centerpoint_detect_kernel");
    hipLaunchKernelGGL(centerpoint_merge_kernel, dim3((unsigned)n_frames), dim3(CPD_K), 0, (hipStream_t)stream, n_tasks, k, box_dim,
                       seg_count, out_boxes, out_scores, out_labels, out_count);
    GGA_CHECK_LAUNCH("centerpoint_merge_kernel");
    return GGA_OK;
}
