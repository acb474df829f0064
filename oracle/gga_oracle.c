/*
 * gga_oracle.c — CPU restatement of the GGA training hot path. TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it. The product path (gga_amd/) never
 * links or imports anything under oracle/ and fails loudly without the HIP library.
 *
 * Each function follows the reference's algorithm (file:line under /root/reference)
 * in the reference's arithmetic type and evaluation order; no reference source is
 * copied. Parity is PINNED: tests/test_oracle.py checks every function against
 * golden vectors produced by importing the reference itself
 * (tools_dev/make_golden.py -> tests/golden/ npz files), including the reference's own
 * known-answer tests (test_voxel_generator.py:7-22, test_utils.py:12-17,
 * test_box3d.py:1598-1607).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fno-fast-math)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* a1. hard voxelization — mmdet3d/core/voxel/voxel_generator.py:137-208     */
/*     (the in-tree statement of mmcv.ops.Voxelization's semantics,          */
/*      call site mmdet3d/models/detectors/mvx_two_stage_gga.py:225)         */
/* ------------------------------------------------------------------------ */
typedef struct { int64_t key; int32_t val; } slot_t;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

/* grid = round((hi - lo) / vs) in f32, round-half-even like np.round / torch.round
 * (voxel_generator.py:182-185) */
void gga_oracle_grid_size(const float vs[3], const float rng[6], int32_t grid[3]) {
    for (int j = 0; j < 3; ++j) {
        float g = (rng[3 + j] - rng[j]) / vs[j];
        grid[j] = (int32_t)rintf(g);
    }
}

/* points [n, ndim] f32 -> voxels [max_voxels, max_points, ndim] (caller zero-fills),
 * coors [max_voxels, 3] (z, y, x), num_points [max_voxels]. Returns voxel count.
 * The reference indexes a dense (D,H,W) int grid (voxel_generator.py:115); a hash
 * map keyed by the linear cell id gives the same first-come numbering. */
int64_t gga_oracle_hard_voxelize(const float* points, int64_t n, int ndim,
                                 const float vs[3], const float rng[6],
                                 int max_points, int max_voxels,
                                 float* voxels, int32_t* coors, int32_t* num_points) {
    int32_t grid[3];
    gga_oracle_grid_size(vs, rng, grid);
    uint64_t cap = 64;
    while (cap < (uint64_t)(2 * n + 2)) cap <<= 1;
    slot_t* tab = (slot_t*)malloc(cap * sizeof(slot_t));
    if (!tab) return -1;
    for (uint64_t i = 0; i < cap; ++i) tab[i].key = -1;
    int64_t voxel_num = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t c[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {                              /* :187-193 */
            float cf = floorf((points[i * ndim + j] - rng[j]) / vs[j]);
            if (!(cf >= 0.0f) || cf >= (float)grid[j]) { failed = 1; break; }
            c[j] = (int32_t)cf;
        }
        if (failed) continue;
        int64_t key = ((int64_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
        uint64_t h = mix64((uint64_t)key) & (cap - 1);
        while (tab[h].key != -1 && tab[h].key != key) h = (h + 1) & (cap - 1);
        int32_t vid;
        if (tab[h].key == -1) {                                     /* :195-201 */
            if (voxel_num >= max_voxels) continue;
            vid = (int32_t)voxel_num++;
            tab[h].key = key; tab[h].val = vid;
            coors[vid * 3 + 0] = c[2]; coors[vid * 3 + 1] = c[1]; coors[vid * 3 + 2] = c[0];
        } else {
            vid = tab[h].val;
        }
        int32_t num = num_points[vid];                              /* :202-205 */
        if (num < max_points) {
            memcpy(voxels + ((int64_t)vid * max_points + num) * ndim, points + i * ndim,
                   sizeof(float) * ndim);
            num_points[vid] = num + 1;
        }
    }
    free(tab);
    return voxel_num;
}

/* a2. HardSimpleVFE — mmdet3d/models/voxel_encoders/voxel_encoder.py:43-45 */
void gga_oracle_voxel_mean(const float* voxels, const int32_t* num_points, int64_t m,
                           int max_points, int ndim, int num_features, float* out) {
    for (int64_t v = 0; v < m; ++v)
        for (int f = 0; f < num_features; ++f) {
            float s = 0.0f;
            for (int p = 0; p < max_points; ++p) s += voxels[(v * max_points + p) * ndim + f];
            out[v * num_features + f] = s / (float)num_points[v];
        }
}

/* a2'. PillarFeatureNet decoration (legacy=True) —
 * mmdet3d/models/voxel_encoders/pillar_encoder.py:106-154. out [m, P, 10]:
 * (x-vx, y-vy, z-vz, r, x-mean, y-mean, z-mean, x-vx, y-vy, z-vz), padding rows zeroed. */
void gga_oracle_pfn_decorate(const float* voxels, const int32_t* num_points, const int32_t* coors4,
                             int64_t m, int P, float vx, float vy, float vz,
                             float x_off, float y_off, float z_off, float* out) {
    for (int64_t v = 0; v < m; ++v) {
        float mean[3];
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int p = 0; p < P; ++p) s += voxels[(v * P + p) * 4 + j];
            mean[j] = s / (float)num_points[v];
        }
        float cen[3] = { (float)coors4[v * 4 + 3] * vx + x_off,
                         (float)coors4[v * 4 + 2] * vy + y_off,
                         (float)coors4[v * 4 + 1] * vz + z_off };
        for (int p = 0; p < P; ++p) {
            const float* q = voxels + (v * P + p) * 4;
            float* o = out + (v * P + p) * 10;
            if (p < num_points[v]) {
                for (int j = 0; j < 3; ++j) {
                    o[j] = q[j] - cen[j]; o[4 + j] = q[j] - mean[j]; o[7 + j] = q[j] - cen[j];
                }
                o[3] = q[3];
            } else {
                for (int j = 0; j < 10; ++j) o[j] = 0.0f;
            }
        }
    }
}

/* PFNLayer forward, training-mode BatchNorm1d —
 * mmdet3d/models/voxel_encoders/utils.py:161-171 (linear, BN over all m*P rows,
 * ReLU, max over the P points). feats [m,P,10], W [C,10]; out [m,C];
 * mean/var (biased) of the pre-BN activations are returned for the running stats. */
void gga_oracle_pfn_layer(const float* feats, int64_t m, int P, int cin, const float* W, int C,
                          const float* gamma, const float* beta, float eps,
                          float* out, float* bmean, float* bvar) {
    int64_t rows = m * P;
    float* z = (float*)malloc(sizeof(float) * rows * C);
    for (int64_t r = 0; r < rows; ++r)
        for (int c = 0; c < C; ++c) {
            float s = 0.0f;
            for (int k = 0; k < cin; ++k) s += feats[r * cin + k] * W[c * cin + k];
            z[r * C + c] = s;
        }
    for (int c = 0; c < C; ++c) {
        double s = 0.0, ss = 0.0;
        for (int64_t r = 0; r < rows; ++r) s += z[r * C + c];
        double mu = s / (double)rows;
        for (int64_t r = 0; r < rows; ++r) { double d = z[r * C + c] - mu; ss += d * d; }
        bmean[c] = (float)mu; bvar[c] = (float)(ss / (double)rows);
        float inv = 1.0f / sqrtf(bvar[c] + eps);
        for (int64_t v = 0; v < m; ++v) {
            float best = -INFINITY;
            for (int p = 0; p < P; ++p) {
                float y = (z[(v * P + p) * C + c] - bmean[c]) * inv * gamma[c] + beta[c];
                y = y > 0.0f ? y : 0.0f;
                if (y > best) best = y;
            }
            out[v * C + c] = best;
        }
    }
    free(z);
}

/* a3. PointPillarsScatter.forward_batch — mmdet3d/models/middle_encoders/pillar_scatter.py:62-102
 * canvas [B, C, ny, nx] (caller zero-fills); index = coors[:,2]*nx + coors[:,3] (:84);
 * later rows overwrite earlier ones (sequential index_put). */
void gga_oracle_pillar_scatter(const float* feats, const int32_t* coors4, int64_t m, int C,
                               int B, int ny, int nx, float* canvas) {
    for (int64_t v = 0; v < m; ++v) {
        int b = coors4[v * 4];
        if (b < 0 || b >= B) continue;
        int64_t idx = (int64_t)coors4[v * 4 + 2] * nx + coors4[v * 4 + 3];
        for (int c = 0; c < C; ++c)
            canvas[((int64_t)b * C + c) * ny * nx + idx] = feats[v * C + c];
    }
}

/* ------------------------------------------------------------------------ */
/* a7. gaussian targets — mmdet3d/core/utils/gaussian.py:6-86                 */
/* ------------------------------------------------------------------------ */
double gga_oracle_gaussian_radius(double height, double width, double min_overlap) {
    double a1 = 1, b1 = height + width;
    double c1 = width * height * (1 - min_overlap) / (1 + min_overlap);
    double r1 = (b1 + sqrt(b1 * b1 - 4 * a1 * c1)) / 2;
    double a2 = 4, b2 = 2 * (height + width);
    double c2 = (1 - min_overlap) * width * height;
    double r2 = (b2 + sqrt(b2 * b2 - 4 * a2 * c2)) / 2;
    double a3 = 4 * min_overlap, b3 = -2 * min_overlap * (height + width);
    double c3 = (min_overlap - 1) * width * height;
    double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    double r = r1 < r2 ? r1 : r2;
    return r < r3 ? r : r3;
}

/* draw_heatmap_gaussian (gaussian.py:25-54) on an [H, W] f32 map; sigma = (2r+1)/6,
 * patch in f64, entries < eps_f64 * max zeroed (:20-21), cast to f32, elementwise max. */
void gga_oracle_draw_gaussian(float* heatmap, int H, int W, int cx, int cy, int radius) {
    int d = 2 * radius + 1;
    double sigma = (double)d / 6.0;
    int left = cx < radius ? cx : radius, right = (W - cx) < (radius + 1) ? (W - cx) : (radius + 1);
    int top = cy < radius ? cy : radius, bottom = (H - cy) < (radius + 1) ? (H - cy) : (radius + 1);
    for (int yy = -top; yy < bottom; ++yy)
        for (int xx = -left; xx < right; ++xx) {
            double g = exp(-((double)(xx * xx + yy * yy)) / (2 * sigma * sigma));
            if (g < 2.220446049250313e-16 * 1.0) g = 0.0;   /* h.max() == 1 at the centre */
            float gf = (float)g;
            float* p = heatmap + (int64_t)(cy + yy) * W + (cx + xx);
            if (gf > *p) *p = gf;
        }
}

/* ------------------------------------------------------------------------ */
/* a8. clip_sigmoid + GaussianFocalLoss(alpha, gamma), reduction 'mean' with  */
/*     avg_factor = max(num_pos, 1) — mmdet3d/models/utils/clip_sigmoid.py:16, */
/*     centerpoint_head_gga.py:650-655; mmdet gaussian_focal_loss (restated).  */
/*     Returns the loss before the x5.0 of head:719. grad (may be NULL) gets   */
/*     d loss / d logit.                                                       */
/* ------------------------------------------------------------------------ */
float gga_oracle_focal_loss(const float* logits, const float* target, int64_t n,
                            float alpha, float gamma, float* grad, double* num_pos_out) {
    const float eps = 1e-12f, lo = 1e-4f, hi = 1.0f - 1e-4f;
    double acc = 0.0; int64_t npos = 0;
    for (int64_t i = 0; i < n; ++i) if (target[i] == 1.0f) ++npos;
    float avg = (float)((double)(npos > 1 ? npos : 1) + 1.1920928955078125e-07);
    for (int64_t i = 0; i < n; ++i) {
        float s = 1.0f / (1.0f + expf(-logits[i]));
        float p = s < lo ? lo : (s > hi ? hi : s);
        float t = target[i];
        float posw = (t == 1.0f) ? 1.0f : 0.0f;
        float negw = powf(1.0f - t, gamma);
        float pl = -logf(p + eps) * powf(1.0f - p, alpha) * posw;
        float nl = -logf(1.0f - p + eps) * powf(p, alpha) * negw;
        acc += (double)(pl + nl);
        if (grad) {
            /* d/dp, then through clamp (pass-through inside [lo, hi]) and sigmoid */
            double dp = 0.0;
            double omp = 1.0 - p;
            if (posw != 0.0f) {
                dp += -(1.0 / (p + eps)) * pow(omp, alpha);
                if (alpha != 0.0f) dp += -log(p + eps) * (-(double)alpha) * pow(omp, alpha - 1.0);
            }
            if (negw != 0.0f) {
                dp += (1.0 / (omp + eps)) * pow(p, alpha) * negw;
                if (alpha != 0.0f) dp += -log(omp + eps) * alpha * pow(p, alpha - 1.0) * negw;
            }
            double pass = (s >= lo && s <= hi) ? 1.0 : 0.0;
            grad[i] = (float)(dp * pass * (double)s * (1.0 - (double)s) / (double)avg);
        }
    }
    if (num_pos_out) *num_pos_out = (double)npos;
    return (float)acc / avg;
}

/* ------------------------------------------------------------------------ */
/* a9-a11. gather + decode + corners + projection                              */
/*   centerpoint_head_gga.py:141-164 (gather), :167-171 (rot), :250-341        */
/*   (get_prediction_single), core/bbox/structures/utils.py:66-106 (rotation)  */
/* ------------------------------------------------------------------------ */
/* maps are NCHW: reg [B,2,H,W], height [B,1,H,W], dim [B,3,H,W], rot [B,2,H,W];
 * ind [B,K] i64 = y*W + x. pred [B,K,8] = (dx, dy, z, ll, lw, lh, sin, cos). */
void gga_oracle_gather_pred(const float* reg, const float* height, const float* dim,
                            const float* rot, const int64_t* ind, int B, int K, int H, int W,
                            float* pred) {
    int64_t hw = (int64_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) {
            int64_t i = ind[b * K + k];
            float* p = pred + ((int64_t)b * K + k) * 8;
            p[0] = reg[((int64_t)b * 2 + 0) * hw + i];
            p[1] = reg[((int64_t)b * 2 + 1) * hw + i];
            p[2] = height[(int64_t)b * hw + i];
            p[3] = dim[((int64_t)b * 3 + 0) * hw + i];
            p[4] = dim[((int64_t)b * 3 + 1) * hw + i];
            p[5] = dim[((int64_t)b * 3 + 2) * hw + i];
            p[6] = rot[((int64_t)b * 2 + 0) * hw + i];
            p[7] = rot[((int64_t)b * 2 + 1) * hw + i];
        }
}

/* n slots. lidar2img [n,16] row-major. Outputs: rot [n], pred_ratio [n,2],
 * pred_iou [n,4] (xmin, ymin, xmax, ymax), pred_box_bev [n,5] (x, y, l, w, rot). */
void gga_oracle_box_project(const float* pred, const int64_t* ind, const float* lidar2img,
                            int64_t n, int fm_w, float vs0, float vs1, float osf,
                            float pc0, float pc1, float* rot_out, float* pred_ratio,
                            float* pred_iou, float* pred_bev) {
    static const float OX[8] = { -.5f, -.5f, -.5f, -.5f, .5f, .5f, .5f, .5f };
    static const float OY[8] = { -.5f, -.5f, .5f, .5f, -.5f, -.5f, .5f, .5f };
    static const float OZ[8] = { 0.f, 1.f, 1.f, 0.f, 0.f, 1.f, 1.f, 0.f };
    for (int64_t i = 0; i < n; ++i) {
        const float* p = pred + i * 8;
        const float* M = lidar2img + i * 16;
        float r = atan2f(p[6], p[7]);                               /* head:169-171 */
        int64_t ix = ind[i] % fm_w, iy = ind[i] / fm_w;
        float X = (((float)ix + p[0]) * vs0) * osf + pc0;           /* head:293-294 */
        float Y = (((float)iy + p[1]) * vs1) * osf + pc1;
        float l = expf(p[3]), w = expf(p[4]), h = expf(p[5]);
        float Zb = p[2] + (-h * 0.5f);                              /* head:310-316 */
        float c = cosf(r), s = sinf(r);
        float umin = INFINITY, vmin = INFINITY, umax = -INFINITY, vmax = -INFINITY;
        for (int q = 0; q < 8; ++q) {
            float lx = l * OX[q], ly = w * OY[q], lz = h * OZ[q];   /* head:262-266 */
            float x = (lx * c + ly * (-s)) + X;                     /* utils.py:79-106 */
            float y = (lx * s + ly * c) + Y;
            float z = lz + Zb;
            float q0 = M[0] * x + M[1] * y + M[2] * z + M[3];       /* head:326 */
            float q1 = M[4] * x + M[5] * y + M[6] * z + M[7];
            float q2 = M[8] * x + M[9] * y + M[10] * z + M[11];
            float d = q2 > 0.1f ? q2 : 0.1f;                        /* head:329 */
            float u = q0 / d, v = q1 / d;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
        }
        rot_out[i] = r;
        pred_ratio[i * 2] = l; pred_ratio[i * 2 + 1] = w;
        pred_iou[i * 4] = umin; pred_iou[i * 4 + 1] = vmin;
        pred_iou[i * 4 + 2] = umax; pred_iou[i * 4 + 3] = vmax;
        pred_bev[i * 5] = X; pred_bev[i * 5 + 1] = Y; pred_bev[i * 5 + 2] = l;
        pred_bev[i * 5 + 3] = w; pred_bev[i * 5 + 4] = r;
    }
}

/* a12. Point-to-Box Alignment for one object — centerpoint_head_gga.py:184-239.
 * pts [ni, stride] f64 (x, y first; cast to f32 like `.float()`), bev = (x,y,l,w,rot).
 * out3 = (dmin, dx, dy) sums. */
void gga_oracle_pal_object(const double* pts, int64_t ni, int stride, const float bev[5],
                           float out3[3]) {
    float c = cosf(bev[4]), s = sinf(bev[4]);
    float Cx = bev[0] * c + bev[1] * s, Cy = bev[0] * (-s) + bev[1] * c;   /* :202 */
    float hl = bev[2] / 2.0f, hw = bev[3] / 2.0f;
    float xmin = Cx - hl, xmax = Cx + hl, ymin = Cy - hw, ymax = Cy + hw;
    double smin = 0.0, sx = 0.0, sy = 0.0;
    for (int64_t i = 0; i < ni; ++i) {
        float px = (float)pts[i * stride], py = (float)pts[i * stride + 1];
        float rx = px * c + py * s, ry = px * (-s) + py * c;               /* :201 */
        float d = fminf(fminf(fabsf(rx - xmin), fabsf(rx - xmax)),
                        fminf(fabsf(ry - ymin), fabsf(ry - ymax)));        /* :210-226 */
        float ex = fabsf(rx - Cx) - 2 * hl, ey = fabsf(ry - Cy) - 2 * hw;  /* :216-219 */
        smin += d; sx += ex > 0 ? ex : 0; sy += ey > 0 ? ey : 0;
    }
    out3[0] = (float)smin; out3[1] = (float)sx; out3[2] = (float)sy;
}

/* a13. mmdet L1Loss(reduction='mean') with weight and avg_factor (restated):
 * loss_weight * sum(|pred-target| * weight) / (avg_factor + eps_f32). */
float gga_oracle_l1_loss(const float* pred, const float* target, const float* weight,
                         int64_t n, float avg_factor, float loss_weight) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc += (double)(fabsf(pred[i] - target[i]) * weight[i]);
    return loss_weight * ((float)acc / (avg_factor + 1.1920928955078125e-07f));
}

/* ------------------------------------------------------------------------ */
/* post-processing ops of the inference path. The    */
/* reference calls mmcv natives (not in the tree); their published          */
/* semantics are restated and pinned by the reference's known-answer tests  */
/* tests/test_utils/test_nms.py:82-120, test_box3d.py:1122-1187,1683-1790.  */
/* ------------------------------------------------------------------------ */
typedef struct { double x, y; } dpt;
static double dcross(dpt a, dpt b) { return a.x * b.y - a.y * b.x; }

static void rect_corners_d(const float* b, dpt out[4]) {
    double c = cos((double)b[4]), s = sin((double)b[4]);
    double hw = b[2] * 0.5, hh = b[3] * 0.5;
    double dx[4] = { -hw, hw, hw, -hw }, dy[4] = { -hh, -hh, hh, hh };
    for (int i = 0; i < 4; ++i) { out[i].x = b[0] + dx[i] * c - dy[i] * s; out[i].y = b[1] + dx[i] * s + dy[i] * c; }
}

/* exact overlap area of two rotated rectangles (x, y, w, h, angle), convex clipping in f64 */
double gga_oracle_rotated_inter(const float* b1, const float* b2) {
    dpt poly[16], tmp[16], q[4];
    rect_corners_d(b1, poly);
    rect_corners_d(b2, q);
    int n = 4;
    for (int e = 0; e < 4 && n > 0; ++e) {
        dpt a = q[e], bq = q[(e + 1) & 3], ed = { bq.x - a.x, bq.y - a.y };
        int m = 0;
        for (int i = 0; i < n; ++i) {
            dpt p = poly[i], r = poly[(i + 1) % n];
            dpt pa = { p.x - a.x, p.y - a.y }, ra = { r.x - a.x, r.y - a.y };
            double dp = dcross(ed, pa), dr = dcross(ed, ra);
            if (dp >= 0) tmp[m++] = p;
            if ((dp >= 0) != (dr >= 0)) { double t = dp / (dp - dr); dpt ip = { p.x + t * (r.x - p.x), p.y + t * (r.y - p.y) }; tmp[m++] = ip; }
        }
        n = m;
        memcpy(poly, tmp, sizeof(dpt) * n);
    }
    if (n < 3) return 0.0;
    double area = 0.0;
    for (int i = 0; i < n; ++i) area += dcross(poly[i], poly[(i + 1) % n]);
    return fabs(area) * 0.5;
}

float gga_oracle_rotated_iou(const float* b1, const float* b2, int mode_iof) {
    double a1 = (double)b1[2] * b1[3], a2 = (double)b2[2] * b2[3];
    if (a1 < 1e-14 || a2 < 1e-14) return 0.0f;
    double inter = gga_oracle_rotated_inter(b1, b2);
    return (float)(inter / (mode_iof ? a1 : (a1 + a2 - inter)));
}

/* greedy rotated NMS on score-sorted boxes (mmcv nms_rotated): box j is suppressed by a kept,
 * higher-scored box i when IoU(i, j) > thr. Returns the number of kept positions. */
int gga_oracle_nms_rotated_sorted(const float* boxes, int n, float thr, int64_t* keep) {
    char* dead = (char*)calloc((size_t)n + 1, 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!dead[j] && gga_oracle_rotated_iou(boxes + i * 5, boxes + j * 5, 0) > thr) dead[j] = 1;
    }
    free(dead);
    return nk;
}

/* mmcv check_pt_in_box3d: box = (x, y, z_bottom, dx, dy, dz, yaw) */
static int pt_in_box(const float* p, const float* b) {
    float cz = b[2] + b[5] * 0.5f;
    if (fabsf(p[2] - cz) > b[5] * 0.5f) return 0;
    float sx = p[0] - b[0], sy = p[1] - b[1];
    float c = cosf(-b[6]), s = sinf(-b[6]);
    float lx = sx * c - sy * s, ly = sx * s + sy * c;
    return (lx > -b[3] * 0.5f) & (lx < b[3] * 0.5f) & (ly > -b[4] * 0.5f) & (ly < b[4] * 0.5f);
}

/* points [M,3], boxes [T,7]; all=0: out [M] first box index or -1; all=1: out [M,T] flags */
void gga_oracle_points_in_boxes(const float* pts, int M, const float* boxes, int T, int all, int32_t* out) {
    for (int i = 0; i < M; ++i) {
        if (all) { for (int t = 0; t < T; ++t) out[i * T + t] = pt_in_box(pts + i * 3, boxes + t * 7); }
        else {
            out[i] = -1;
            for (int t = 0; t < T; ++t) if (pt_in_box(pts + i * 3, boxes + t * 7)) { out[i] = t; break; }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * KITTI image-plane box overlap (mmdet3d/core/evaluation/kitti_utils/eval.py:86-114, criterion
 * -1), as called with (detections, ground truths) by pseudo_label_matching_kitti
 * (tools/utils_pseudo_labels_gga.py:44) -> overlaps [N,K]; round_f32 mirrors the reference's
 * `np.zeros((N, K), dtype=boxes.dtype)` when the detections are float32.
 * ------------------------------------------------------------------------------------------- */
void gga_oracle_image_box_overlap(const double* boxes, int N, const double* query, int K, int round_f32, double* out) {
    for (int k = 0; k < K; ++k) {
        const double* q = query + 4 * k;
        double qarea = (q[2] - q[0]) * (q[3] - q[1]);
        for (int n = 0; n < N; ++n) {
            const double* b = boxes + 4 * n;
            double v = 0.0;
            double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
            if (iw > 0) {
                double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
                if (ih > 0) {
                    /* float32 detections: their own area is float32 arithmetic (array scalar
                     * typing), everything mixed with the float64 ground truth is float64 */
                    double barea = round_f32 ? (double)((float)((float)b[2] - (float)b[0]) * (float)((float)b[3] - (float)b[1]))
                                             : (b[2] - b[0]) * (b[3] - b[1]);
                    double ua = barea + qarea - iw * ih;
                    v = iw * ih / ua;
                }
            }
            out[(size_t)n * K + k] = round_f32 ? (double)(float)v : v;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Point-level tail of the train pipeline for ONE frame: remove_points_in_boxes_v2
 * (mmdet3d/datasets/pipelines/gga_processing.py:58-68, scipy cdist = sqrt of the sum of squared
 * differences in double), points.cat([sampled, scene]) (:176), in_range_3d
 * (mmdet3d/core/points/base_points.py:203-225, strict float32 inequalities). Returns kept rows.
 * ------------------------------------------------------------------------------------------- */
int64_t gga_oracle_points_prepare(const float* scene, int64_t n_scene, const float* sampled, int64_t n_sampled,
                                  const double* centers_xy, int64_t n_centers, int ndim, double min_distance,
                                  const float* rng, float* out) {
    int64_t m = 0;
    for (int64_t v = 0; v < n_sampled + n_scene; ++v) {
        const int is_scene = v >= n_sampled;
        const float* p = is_scene ? scene + (v - n_sampled) * ndim : sampled + v * ndim;
        if (!(p[0] > rng[0] && p[1] > rng[1] && p[2] > rng[2] && p[0] < rng[3] && p[1] < rng[4] && p[2] < rng[5])) continue;
        int near = 0;
        if (is_scene)
            for (int64_t c = 0; c < n_centers && !near; ++c) {
                double dx = (double)p[0] - centers_xy[2 * c], dy = (double)p[1] - centers_xy[2 * c + 1];
                double s = dx * dx;
                s += dy * dy;
                near = sqrt(s) < min_distance;
            }
        if (near) continue;
        for (int j = 0; j < ndim; ++j) out[m * ndim + j] = p[j];
        ++m;
    }
    return m;
}

/* ---------------------------------------------------------------------------------------------
 * Offline label generation primitives, tools/data_converter/utils_gga.py.
 * region_grow (:6-38): breadth-first region growing from every not-yet-covered origin point over
 * the search set, early exit when the share of in-origin points among the reached ones drops below
 * `ratio`; the largest completed region wins. pc [n, dim] f64; masks u8; out u8 [n].
 * ------------------------------------------------------------------------------------------- */
static double rg_norm(const double* a, const double* b, int dim) {
    double s = 0.0;
    for (int j = 0; j < dim; ++j) { double d = a[j] - b[j]; double q = d * d; s = j == 0 ? q : s + q; }
    return sqrt(s);
}

void gga_oracle_region_grow(const double* pc, int64_t n, int dim, const uint8_t* mask_search, const uint8_t* mask_origin,
                            double thresh, double ratio, int use_ratio, uint8_t* out) {
    int64_t S = 0;
    int64_t* sidx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < n; ++i) if (mask_search[i]) sidx[S++] = i;
    uint8_t* mask = (uint8_t*)malloc((size_t)n + 1);
    uint8_t* best = (uint8_t*)calloc((size_t)n + 1, 1);
    uint8_t* cur = (uint8_t*)malloc((size_t)n + 1);
    uint8_t* smask = (uint8_t*)malloc((size_t)S + 1);
    int64_t* queue = (int64_t*)malloc(sizeof(int64_t) * (size_t)(S + 1));
    memcpy(mask, mask_origin, (size_t)n);
    long long best_len = 0;
    for (int64_t seed = 0; S > 0 && seed < n; ++seed) {
        if (!mask[seed]) continue;                       /* pc[mask==1][0]: masks only shrink, so scan forward */
        memset(smask, 0, (size_t)S);
        memset(cur, 0, (size_t)n);
        int64_t head = 0, tail = 0;
        long long reached = 0, reached_origin = 0;
        int flag = 1, first = 1;
        while (first || head < tail) {
            const double* temp = first ? pc + seed * dim : pc + sidx[queue[head]] * dim;
            if (!first) ++head;
            first = 0;
            double bd = 1.0 / 0.0; int64_t bi = -1;
            for (int64_t s = 0; s < S; ++s) { double d = rg_norm(pc + sidx[s] * dim, temp, dim); if (d < bd) { bd = d; bi = s; } }
            if (bi >= 0 && !smask[bi]) { smask[bi] = 1; cur[sidx[bi]] = 1; ++reached; reached_origin += mask_origin[sidx[bi]] ? 1 : 0; }
            for (int64_t s = 0; s < S; ++s)
                if (!smask[s] && rg_norm(pc + sidx[s] * dim, temp, dim) < thresh) {
                    queue[tail++] = s; smask[s] = 1; cur[sidx[s]] = 1; ++reached; reached_origin += mask_origin[sidx[s]] ? 1 : 0;
                }
            if (use_ratio && (double)reached_origin / (double)(float)reached < ratio) { flag = 0; break; }
        }
        if (flag && reached > best_len) { best_len = reached; memcpy(best, cur, (size_t)n); }
        for (int64_t i = 0; i < n; ++i) if (cur[i]) mask[i] = 0;
        mask[seed] = 0;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = use_ratio ? (best[i] && mask_origin[i]) : best[i];
    free(sidx); free(mask); free(best); free(cur); free(smask); free(queue);
}

/* points_in_convex_polygon_3d_jit (mmdet3d/core/bbox/box_np_ops.py:641-676) */
void gga_oracle_points_in_polyhedra(const double* pts, int64_t n, int stride, const double* normal, const double* d, int n_poly,
                                    int n_surf, uint8_t* out) {
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < n_poly; ++j) {
            int in = 1;
            for (int k = 0; k < n_surf && in; ++k) {
                const double* nv = normal + ((size_t)j * n_surf + k) * 3;
                double s = pts[i * stride] * nv[0] + pts[i * stride + 1] * nv[1];
                s = s + pts[i * stride + 2] * nv[2];
                s = s + d[(size_t)j * n_surf + k];
                if (s >= 0) in = 0;
            }
            out[i * n_poly + j] = (uint8_t)in;
        }
}

/* inlier test of calculate_ground (utils_gga.py:122-124) for one plane a.p = 1 */
int64_t gga_oracle_plane_inliers(const double* pts, int64_t n, int stride, const double* plane, double thresh, uint8_t* mask) {
    double norm = sqrt((plane[0] * plane[0] + plane[1] * plane[1]) + plane[2] * plane[2]);
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        double v = (pts[i * stride] * plane[0] + pts[i * stride + 1] * plane[1]) + pts[i * stride + 2] * plane[2];
        int in = fabs(v - 1.0) / norm < thresh;
        if (mask) mask[i] = (uint8_t)in;
        c += in;
    }
    return c;
}
