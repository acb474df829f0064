// a9-a13: gather of the regression channels and the GGA geometry-aware losses
// (Boundary Projection, Semantic Ratio, Point-to-Box Alignment) for gfx950.
//
// Reference: mmdet3d/models/dense_heads/centerpoint_head_gga.py
//   :141-164,657-676  cat + permute + contiguous + gather        -> gga_gather_pred_*
//   :167-182          GGA_calculate_rotation (atan2)
//   :250-341          get_prediction_single (decode, 8 corners, lidar2img, min/max)
//   :184-248          get_distance_single/bev (per-object Python loop, ~20 launches each)
//   :678-721          loss assembly with mmdet L1Loss(reduction='mean', loss_weight)
// plus mmdet3d/core/bbox/structures/utils.py:66-106 (rotation_3d_in_axis).
//
// The reference runs ~10^3 tiny launches per step here; this file runs three:
//   box_slot_kernel  one thread per object slot: forward values + analytic d/d pred of BPL, SRL
//   pal_kernel       one wavefront per object: sums over its in-box points with wave shuffles,
//                    values + analytic gradient w.r.t. the predicted BEV box
//   box_reduce_kernel fixed-order (deterministic) reduction to the five dict values
// All arithmetic is fp32 in the reference's evaluation order; the loss is a fixed-weight sum,
// so the gradient w.r.t. pred is produced in the same pass (no autograd graph, no recompute).
#include <float.h>

#include "gga_common.h"

// ----------------------------------------------------------------------------- gather
__global__ __launch_bounds__(256) void gather_pred_kernel(const float* __restrict__ reg,
                                                         const float* __restrict__ height,
                                                         const float* __restrict__ dim,
                                                         const float* __restrict__ rot,
                                                         const int64_t* __restrict__ ind, int n, int K, int64_t hw,
                                                         float* __restrict__ pred) {
    const int t = blockIdx.x * 256 + threadIdx.x;   // one thread per (slot, channel)
    if (t >= n * 8) return;
    const int s = t >> 3, c = t & 7;
    const int64_t b = s / K;
    const int64_t i = ind[s];
    const float* src;
    switch (c) {
        case 0: case 1: src = reg + (b * 2 + c) * hw; break;
        case 2: src = height + b * hw; break;
        case 3: case 4: case 5: src = dim + (b * 3 + (c - 3)) * hw; break;
        default: src = rot + (b * 2 + (c - 6)) * hw; break;
    }
    pred[t] = src[i];
}

__global__ __launch_bounds__(256) void gather_pred_bwd_kernel(const float* __restrict__ grad_pred,
                                                             const int64_t* __restrict__ ind,
                                                             const uint8_t* __restrict__ mask, int n, int K,
                                                             int64_t hw, float* __restrict__ g_reg,
                                                             float* __restrict__ g_height, float* __restrict__ g_dim,
                                                             float* __restrict__ g_rot) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * 8) return;
    const int s = t >> 3, c = t & 7;
    if (!mask[s]) return;                 // zero weight in every loss term
    const int64_t b = s / K;
    const int64_t i = ind[s];
    // Two objects of a frame may share a cell. Deterministic without atomics: the FIRST live slot of a cell (in slot order)
    // writes the sum over all of that cell's live slots, added in slot order; the others write nothing. K <= 500 slots
    // per frame and a few dozen live ones: the scans are a few thousand cached loads per step.
    const int s0 = (int)b * K, k = s - s0;
    for (int j = 0; j < k; ++j)
        if (mask[s0 + j] && ind[s0 + j] == i) return;
    float g = grad_pred[t];
    for (int j = k + 1; j < K; ++j)
        if (mask[s0 + j] && ind[s0 + j] == i) g += grad_pred[(int64_t)(s0 + j) * 8 + c];
    float* dst;
    switch (c) {
        case 0: case 1: dst = g_reg + (b * 2 + c) * hw; break;
        case 2: dst = g_height + b * hw; break;
        case 3: case 4: case 5: dst = g_dim + (b * 3 + (c - 3)) * hw; break;
        default: dst = g_rot + (b * 2 + (c - 6)) * hw; break;
    }
    dst[i] = g;
}

extern "C" int gga_gather_pred_fwd(const float* reg, const float* height, const float* dim, const float* rot,
                                   const int64_t* ind, int B, int K, int H, int W, float* pred, void* stream) {
    GGA_REQUIRE(reg && height && dim && rot && ind && pred, "gga_gather_pred_fwd: null pointer argument");
    GGA_REQUIRE(B >= 1 && K >= 1 && H >= 1 && W >= 1, "gga_gather_pred_fwd: bad sizes");
    const int n = B * K;
    hipLaunchKernelGGL(gather_pred_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, reg, height,
                       dim, rot, ind, n, K, (int64_t)H * W, pred);
    GGA_CHECK_LAUNCH("gather_pred_kernel");
    return GGA_OK;
}

extern "C" int gga_gather_pred_bwd(const float* grad_pred, const int64_t* ind, const uint8_t* mask, int B, int K,
                                   int H, int W, float* g_reg, float* g_height, float* g_dim, float* g_rot,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(grad_pred && ind && mask && g_reg && g_height && g_dim && g_rot,
                "gga_gather_pred_bwd: null pointer argument");
    GGA_REQUIRE(B >= 1 && K >= 1 && H >= 1 && W >= 1, "gga_gather_pred_bwd: bad sizes");
    const size_t hw = (size_t)H * W * sizeof(float);
    const size_t fl = (size_t)H * W * B;
    if (g_height == g_reg + 2 * fl && g_dim == g_height + fl && g_rot == g_dim + 3 * fl) {      // four views of one allocation: one memset
        GGA_CHECK_HIP(hipMemsetAsync(g_reg, 0, hw * B * 8, stream), "gather bwd memset");
    } else {
        GGA_CHECK_HIP(hipMemsetAsync(g_reg, 0, hw * B * 2, stream), "gather bwd memset");
        GGA_CHECK_HIP(hipMemsetAsync(g_height, 0, hw * B, stream), "gather bwd memset");
        GGA_CHECK_HIP(hipMemsetAsync(g_dim, 0, hw * B * 3, stream), "gather bwd memset");
        GGA_CHECK_HIP(hipMemsetAsync(g_rot, 0, hw * B * 2, stream), "gather bwd memset");
    }
    const int n = B * K;
    hipLaunchKernelGGL(gather_pred_bwd_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, stream, grad_pred, ind, mask,
                       n, K, (int64_t)H * W, g_reg, g_height, g_dim, g_rot);
    GGA_CHECK_LAUNCH("gather_pred_bwd_kernel");
    return GGA_OK;
}

// ----------------------------------------------------------------------------- losses
// avg_factor = (num + 1e-4) as an f32 tensor, then + eps_f32 inside mmdet's mean
// (centerpoint_head_gga.py:673,693 + weight_reduce_loss). Every block recomputes
// num = sum(mask) (a few KB) so no extra launch / host sync is needed.
__device__ float block_avg_factor(const uint8_t* __restrict__ mask, int n) {
    __shared__ int s_cnt[16];
    int c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c += mask[i] ? 1 : 0;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_cnt[w];
    __syncthreads();
    const float num = (float)tot;
    return (num + 1e-4f) + FLT_EPSILON;
}

struct Decoded {
    float r, c, s, X, Y, l, w, h, Zb, inv_n2;   // inv_n2 = 1 / (sin^2 + cos^2) of the raw head outputs
};

__device__ __forceinline__ Decoded decode_slot(const float* __restrict__ p, int64_t ind, const gga_loss_params& q) {
    Decoded d;
    d.r = atan2f(p[6], p[7]);                                       // head:169-171
    const int64_t ix = ind % q.fm_w, iy = ind / q.fm_w;
    d.X = (((float)ix + p[0]) * q.voxel_size[0]) * q.out_size_factor + q.pc_range[0];   // head:293-294
    d.Y = (((float)iy + p[1]) * q.voxel_size[1]) * q.out_size_factor + q.pc_range[1];
    d.l = expf(p[3]); d.w = expf(p[4]); d.h = expf(p[5]);
    d.Zb = p[2] + (-d.h * 0.5f);                                    // head:310-316
    d.c = cosf(d.r); d.s = sinf(d.r);
    d.inv_n2 = 1.0f / (p[6] * p[6] + p[7] * p[7]);
    return d;
}

__device__ __forceinline__ void corner_local(int qi, const Decoded& d, float& lx, float& ly, float& lz) {
    // unravel_index order [0,1,3,2,4,5,7,6] minus (.5,.5,0)  (head:259-266):
    //   x offset: -,-,-,-,+,+,+,+   y offset: -,-,+,+,-,-,+,+   z offset: 0,1,1,0,0,1,1,0
    const float ox = (qi & 4) ? 0.5f : -0.5f;
    const float oy = (qi & 2) ? 0.5f : -0.5f;
    const float oz = (((qi >> 1) ^ qi) & 1) ? 1.0f : 0.0f;
    lx = d.l * ox; ly = d.w * oy; lz = d.h * oz;
}

// project corner qi; optionally the gradient of (u, v) w.r.t. the 8 pred channels
template <bool GRAD>
__device__ __forceinline__ void corner_uv(int qi, const Decoded& d, const float* __restrict__ M,
                                          const gga_loss_params& prm, float& u, float& v, float* gu, float* gv) {
    float lx, ly, lz;
    corner_local(qi, d, lx, ly, lz);
    const float x = (lx * d.c + ly * (-d.s)) + d.X;                 // utils.py:79-106 (counter-clockwise)
    const float y = (lx * d.s + ly * d.c) + d.Y;
    const float z = lz + d.Zb;
    const float q0 = M[0] * x + M[1] * y + M[2] * z + M[3];         // head:326
    const float q1 = M[4] * x + M[5] * y + M[6] * z + M[7];
    const float q2 = M[8] * x + M[9] * y + M[10] * z + M[11];
    const float dep = fmaxf(q2, 0.1f);                              // head:329
    u = q0 / dep; v = q1 / dep;
    if (GRAD) {
        const float k = q2 > 0.1f ? 1.0f : 0.0f;
        const float inv = 1.0f / dep;
        const float sx = prm.voxel_size[0] * prm.out_size_factor, sy = prm.voxel_size[1] * prm.out_size_factor;
        const float dxr = -lx * d.s - ly * d.c, dyr = lx * d.c - ly * d.s;   // d(x,y)/d rot
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float val = a == 0 ? u : v;
            const float* R = M + 4 * a;
            const float gx = (R[0] - val * k * M[8]) * inv;
            const float gy = (R[1] - val * k * M[9]) * inv;
            const float gz = (R[2] - val * k * M[10]) * inv;
            float* g = a == 0 ? gu : gv;
            g[0] = gx * sx;
            g[1] = gy * sy;
            g[2] = gz;
            g[3] = gx * (lx * d.c) + gy * (lx * d.s);
            g[4] = gx * (-ly * d.s) + gy * (ly * d.c);
            g[5] = gz * (lz - 0.5f * d.h);
            g[6] = gx * dxr + gy * dyr;   // d/d rot; the caller chains it through atan2(sin, cos)
            g[7] = 0.0f;
        }
    }
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) - (x < 0.0f); }

// box_out row layout: rot, l, w, umin, vmin, umax, vmax, X, Y, p2c_min, p2c_x, p2c_y
#define BOX_OUT_W 12

__global__ __launch_bounds__(256) void box_slot_kernel(const float* __restrict__ pred,
                                                      const int64_t* __restrict__ ind,
                                                      const uint8_t* __restrict__ mask,
                                                      const float* __restrict__ anno,
                                                      const float* __restrict__ lidar2img,
                                                      const uint8_t* __restrict__ bound_mask, gga_loss_params prm,
                                                      float* __restrict__ box_out, float* __restrict__ grad_pred,
                                                      float* __restrict__ part) {
    const int n = prm.B * prm.K;
    const float avg = block_avg_factor(mask, n);
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const float* p = pred + (int64_t)s * 8;
    const float* M = lidar2img + (int64_t)s * 16;
    const Decoded d = decode_slot(p, ind[s], prm);
    // forward: 2D box = min/max of the 8 projected corners (head:332-335)
    float bx[4] = { INFINITY, INFINITY, -INFINITY, -INFINITY };
    int sel[4] = { 0, 0, 0, 0 };
    for (int qi = 0; qi < 8; ++qi) {
        float u, v;
        corner_uv<false>(qi, d, M, prm, u, v, nullptr, nullptr);
        if (u < bx[0]) { bx[0] = u; sel[0] = qi; }
        if (v < bx[1]) { bx[1] = v; sel[1] = qi; }
        if (u > bx[2]) { bx[2] = u; sel[2] = qi; }
        if (v > bx[3]) { bx[3] = v; sel[3] = qi; }
    }
    float* bo = box_out + (int64_t)s * BOX_OUT_W;
    bo[0] = d.r; bo[1] = d.l; bo[2] = d.w;
    bo[3] = bx[0]; bo[4] = bx[1]; bo[5] = bx[2]; bo[6] = bx[3];
    bo[7] = d.X; bo[8] = d.Y;

    const float m = mask[s] ? 1.0f : 0.0f;
    if (m == 0.0f) return;     // weight 0 in every term (part / grad buffers are pre-zeroed)
    const float* a = anno + (int64_t)s * 5;
    // bbox_weights = mask * isnotnan(target) * code_weights   (head:678-684)
    float bw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) bw[j] = m * (isnan(a[j]) ? 0.0f : 1.0f) * prm.code_weights[j];

    // ---- Boundary Projection Loss (head:714-720)
    const float cb = prm.l1_loss_weight * prm.w_bpl / avg;
    float bpl = 0.0f;
    float g[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int j = 0; j < 4; ++j) {
        const float wj = bw[j] * (bound_mask[(int64_t)s * 4 + j] ? 1.0f : 0.0f);
        const float diff = bx[j] - a[j];
        bpl += fabsf(diff) * wj;       // NaN target * 0 stays NaN, as in the reference
        if (wj != 0.0f) {
            float u, v, gu[8], gv[8];
            corner_uv<true>(sel[j], d, M, prm, u, v, gu, gv);
            const float* gsel = (j & 1) ? gv : gu;
            const float k = cb * wj * sgn(diff);
#pragma unroll
            for (int c = 0; c < 6; ++c) g[c] += k * gsel[c];
            // d rot / d (sin, cos) of the raw head outputs (atan2)
            g[6] += k * gsel[6] * (p[7] * d.inv_n2);
            g[7] += k * gsel[6] * (-p[6] * d.inv_n2);
        }
    }
    part[0 * n + s] = bpl;
    float* gp = grad_pred + ((int64_t)GGA_L_BPL * n + s) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) gp[c] = g[c];

    // ---- Semantic Ratio Loss (head:703-712): max(l,w) - min(l,w) * srl
    const float coef = a[4];
    const bool l_is_max = d.l >= d.w;
    const float rw = l_is_max ? d.w : d.l, rl = l_is_max ? d.l : d.w;
    const float srl = rl - rw * coef;
    part[1 * n + s] = fabsf(srl) * bw[4];
    const float ks = prm.l1_loss_weight * prm.w_srl / avg * bw[4] * sgn(srl);
    float* gs = grad_pred + ((int64_t)GGA_L_SRL * n + s) * 8;
    gs[3] = ks * (l_is_max ? d.l : -coef * d.l);
    gs[4] = ks * (l_is_max ? -coef * d.w : d.w);
}

// One wavefront per object with in-box points (head:184-239).
__global__ __launch_bounds__(256) void pal_kernel(const float* __restrict__ pred, const int64_t* __restrict__ ind,
                                                 const uint8_t* __restrict__ mask, const float* __restrict__ anno,
                                                 const float2* __restrict__ ibp_xy,
                                                 const int32_t* __restrict__ ibp_offsets,
                                                 const int32_t* __restrict__ ibp_slot, int n_obj, gga_loss_params prm,
                                                 float* __restrict__ box_out, float* __restrict__ grad_pred,
                                                 float* __restrict__ part) {
    const int n = prm.B * prm.K;
    const float avg = block_avg_factor(mask, n);
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_obj) return;
    const int s = ibp_slot[o];
    if (s < 0 || s >= n) return;
    const float* p = pred + (int64_t)s * 8;
    const Decoded d = decode_slot(p, ind[s], prm);
    const float c = d.c, sn = d.s;
    // clockwise rotation of the centre and the points (head:201-202)
    const float Cx = d.X * c + d.Y * sn, Cy = d.X * (-sn) + d.Y * c;
    const float hl = d.l / 2.0f, hw = d.w / 2.0f;
    const float xmin = Cx - hl, xmax = Cx + hl, ymin = Cy - hw, ymax = Cy + hw;
    // accumulators: value + d/d(X, Y, l, w, rot) for the three sums
    float vmin = 0, vx = 0, vy = 0;
    float gmin[5] = { 0, 0, 0, 0, 0 }, gx[5] = { 0, 0, 0, 0, 0 }, gy[5] = { 0, 0, 0, 0, 0 };
    const int beg = ibp_offsets[o], end = ibp_offsets[o + 1];
    for (int i = beg + lane; i < end; i += 64) {
        const float2 pt = ibp_xy[i];
        const float rx = pt.x * c + pt.y * sn, ry = pt.x * (-sn) + pt.y * c;
        const float e1 = rx - xmin, e2 = rx - xmax, e3 = ry - ymin, e4 = ry - ymax;
        const float a1 = fabsf(e1), a2 = fabsf(e2), a3 = fabsf(e3), a4 = fabsf(e4);
        // torch.min(dim) over [dx1, dx2, dy1, dy2]: first minimum
        float best = a1; int bj = 0;
        if (a2 < best) { best = a2; bj = 1; }
        if (a3 < best) { best = a3; bj = 2; }
        if (a4 < best) { best = a4; bj = 3; }
        vmin += best;
        const float ax = rx - Cx, ay = ry - Cy;
        // d ax / d(X, Y, rot) = (-c, -s, ay);  d ay / d(X, Y, rot) = (s, -c, -ax)
        {
            const float e = bj == 0 ? e1 : bj == 1 ? e2 : bj == 2 ? e3 : e4;
            const float sg = sgn(e);
            if (bj < 2) {
                gmin[0] += sg * (-c); gmin[1] += sg * (-sn); gmin[4] += sg * ay;
                gmin[2] += sg * (bj == 0 ? 0.5f : -0.5f);
            } else {
                gmin[0] += sg * sn; gmin[1] += sg * (-c); gmin[4] += sg * (-ax);
                gmin[3] += sg * (bj == 2 ? 0.5f : -0.5f);
            }
        }
        const float ex = fabsf(ax) - 2 * hl, ey = fabsf(ay) - 2 * hw;   // head:216-219
        if (ex > 0.0f) {
            vx += ex;
            const float sg = sgn(ax);
            gx[0] += sg * (-c); gx[1] += sg * (-sn); gx[4] += sg * ay; gx[2] += -1.0f;
        }
        if (ey > 0.0f) {
            vy += ey;
            const float sg = sgn(ay);
            gy[0] += sg * sn; gy[1] += sg * (-c); gy[4] += sg * (-ax); gy[3] += -1.0f;
        }
    }
    vmin = wave_sum(vmin); vx = wave_sum(vx); vy = wave_sum(vy);
#pragma unroll
    for (int j = 0; j < 5; ++j) { gmin[j] = wave_sum(gmin[j]); gx[j] = wave_sum(gx[j]); gy[j] = wave_sum(gy[j]); }
    if (lane != 0) return;
    float* bo = box_out + (int64_t)s * BOX_OUT_W;
    bo[9] = vmin; bo[10] = vx; bo[11] = vy;
    if (!mask[s]) return;
    const float w0 = (isnan(anno[(int64_t)s * 5]) ? 0.0f : 1.0f) * prm.code_weights[0];   // bbox_weights[..., 0]
    part[2 * n + s] = fabsf(vmin) * w0;
    part[3 * n + s] = fabsf(vx) * w0;
    part[4 * n + s] = fabsf(vy) * w0;
    const float k = prm.l1_loss_weight * prm.w_pal / avg * w0;
    const float sx = prm.voxel_size[0] * prm.out_size_factor, sy = prm.voxel_size[1] * prm.out_size_factor;
    const float* G[3] = { gmin, gx, gy };
    const float V[3] = { vmin, vx, vy };
    for (int t = 0; t < 3; ++t) {
        float* gp = grad_pred + ((int64_t)(GGA_L_PAL_MIN + t) * n + s) * 8;
        const float kk = k * sgn(V[t]);
        gp[0] = kk * G[t][0] * sx;
        gp[1] = kk * G[t][1] * sy;
        gp[3] = kk * G[t][2] * d.l;
        gp[4] = kk * G[t][3] * d.w;
        gp[6] = kk * G[t][4] * (p[7] * d.inv_n2);
        gp[7] = kk * G[t][4] * (-p[6] * d.inv_n2);
    }
}

__global__ __launch_bounds__(1024) void box_reduce_kernel(const float* __restrict__ part,
                                                         const uint8_t* __restrict__ mask, gga_loss_params prm,
                                                         float* __restrict__ losses) {
    // one pass: every thread carries the five partial sums, wave shuffle, then a fixed-order fold
    const int n = prm.B * prm.K;
    const float avg = block_avg_factor(mask, n);
    __shared__ double sh[GGA_L_NUM][16];
    double acc[GGA_L_NUM];
#pragma unroll
    for (int t = 0; t < GGA_L_NUM; ++t) acc[t] = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
#pragma unroll
        for (int t = 0; t < GGA_L_NUM; ++t) acc[t] += (double)part[t * n + i];
    }
#pragma unroll
    for (int t = 0; t < GGA_L_NUM; ++t) {
        const double a = wave_sum(acc[t]);
        if ((threadIdx.x & 63) == 0) sh[t][threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x < GGA_L_NUM) {
        const int t = threadIdx.x;
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += sh[t][w];
        const float wt = t == 0 ? prm.w_bpl : (t == 1 ? prm.w_srl : prm.w_pal);
        losses[t] = (prm.l1_loss_weight * ((float)tot / avg)) * wt;   // L1Loss then the head multiplier
    }
}

__global__ __launch_bounds__(256) void box_bwd_kernel(const float* __restrict__ grad_pred,
                                                     const float* __restrict__ grad_losses, int n8,
                                                     float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n8) return;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < GGA_L_NUM; ++k) {
        const float gl = grad_losses[k];
        if (gl != 0.0f) acc += gl * grad_pred[(int64_t)k * n8 + t];
    }
    out[t] = acc;
}

extern "C" size_t gga_box_losses_workspace_bytes(int B, int K) {
    return (size_t)GGA_L_NUM * B * K * sizeof(float);
}

extern "C" int gga_box_losses_fwd(const float* pred, const int64_t* ind, const uint8_t* mask, const float* anno_box,
                                  const float* lidar2img, const uint8_t* bound_mask, const float* ibp_xy,
                                  const int32_t* ibp_offsets, const int32_t* ibp_slot, int n_ibp_obj,
                                  const gga_loss_params* prm, float* losses, float* box_out, float* grad_pred,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(pred && ind && mask && anno_box && lidar2img && bound_mask && prm && losses && box_out && grad_pred &&
                    workspace,
                "gga_box_losses_fwd: null pointer argument");
    GGA_REQUIRE(prm->B >= 1 && prm->K >= 1 && prm->fm_w >= 1, "gga_box_losses_fwd: bad B/K/fm_w");
    GGA_REQUIRE(n_ibp_obj == 0 || (ibp_xy && ibp_offsets && ibp_slot), "gga_box_losses_fwd: null in-box-point arrays");
    const int n = prm->B * prm->K;
    if (workspace_bytes < gga_box_losses_workspace_bytes(prm->B, prm->K)) {
        gga_set_error("gga_box_losses_fwd: workspace %zu B < required %zu B", workspace_bytes,
                      gga_box_losses_workspace_bytes(prm->B, prm->K));
        return GGA_ERR_WORKSPACE;
    }
    float* part = (float*)workspace;
    GGA_CHECK_HIP(hipMemsetAsync(part, 0, (size_t)GGA_L_NUM * n * sizeof(float), stream), "box losses memset");
    if (box_out == grad_pred + (size_t)GGA_L_NUM * n * 8) {         // two views of one allocation: one memset
        GGA_CHECK_HIP(hipMemsetAsync(grad_pred, 0, ((size_t)GGA_L_NUM * n * 8 + (size_t)n * BOX_OUT_W) * sizeof(float), stream), "box losses memset");
    } else {
        GGA_CHECK_HIP(hipMemsetAsync(grad_pred, 0, (size_t)GGA_L_NUM * n * 8 * sizeof(float), stream), "box losses memset");
        GGA_CHECK_HIP(hipMemsetAsync(box_out, 0, (size_t)n * BOX_OUT_W * sizeof(float), stream), "box losses memset");
    }
    hipLaunchKernelGGL(box_slot_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pred, ind, mask, anno_box,
                       lidar2img, bound_mask, *prm, box_out, grad_pred, part);
    GGA_CHECK_LAUNCH("box_slot_kernel");
    if (n_ibp_obj > 0) {
        hipLaunchKernelGGL(pal_kernel, dim3((n_ibp_obj + 3) / 4), dim3(256), 0, stream, pred, ind, mask, anno_box,
                           (const float2*)ibp_xy, ibp_offsets, ibp_slot, n_ibp_obj, *prm, box_out, grad_pred, part);
        GGA_CHECK_LAUNCH("pal_kernel");
    }
    hipLaunchKernelGGL(box_reduce_kernel, dim3(1), dim3(1024), 0, stream, part, mask, *prm, losses);
    GGA_CHECK_LAUNCH("box_reduce_kernel");
    return GGA_OK;
}

extern "C" int gga_box_losses_bwd(const float* grad_pred, const float* grad_losses, int B, int K,
                                  float* grad_pred_out, void* stream) {
    GGA_REQUIRE(grad_pred && grad_losses && grad_pred_out && B >= 1 && K >= 1,
                "gga_box_losses_bwd: null pointer or bad sizes");
    const int n8 = B * K * 8;
    hipLaunchKernelGGL(box_bwd_kernel, dim3((n8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_pred,
                       grad_losses, n8, grad_pred_out);
    GGA_CHECK_LAUNCH("box_bwd_kernel");
    return GGA_OK;
}
