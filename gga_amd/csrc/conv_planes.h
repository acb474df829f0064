// Shared by the split-plane matrix kernels (sparse_conv.hip: gather-GEMM forward / backward-data / weight gradient of the
// sparse 3D, strided, transposed and pillar convolutions; dense_conv.hip: 3x3 stride-1 convolutions of the BEV trunk and the
// head): operand planes of an fp32 value - three truncated bf16 planes or two round-to-nearest fp16 planes of the value
// scaled to its tensor's largest magnitude -, the power-of-two scale, fragment vector types and the stage constants of the
// packed weight layout (gga_sparse_pack_weight_planes / gga_dense_conv3x3_pack_planes).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#define MF_TK 32                         // input channels per packed weight stage
typedef float mf_v16 __attribute__((ext_vector_type(16)));
typedef short dw_v4s __attribute__((ext_vector_type(4)));      // operand of ds_read_b64_tr_b16 (the weight-gradient kernels)

static inline int mf_nt(int cout) { return cout <= 32 ? 1 : (cout <= 64 ? 2 : 4); }      // 32-column tiles of an output width

// ------------------------------------------------------------------------------ fp32 through bf16 planes
// The same convolution on v_mfma_f32_32x32x16_bf16. An fp32 number is the exact sum of three
// bfloat16 numbers (8 + 8 + 8 significand bits, by truncation), so a*b is the sum of nine bf16
// products, each exact in fp32; the matrix core accumulates them in fp32. Nine bf16 MFMAs cover
// K = 16 in 9*32 cycles where the fp32 MFMA needs 8*64: measured 256 vs 149 fp32-equivalent
// TFLOP/s on this part, with an error against float64 no larger than the native fp32 MFMA's
// (2.0e-7 vs 4.5e-7 of sum|a*b| at K = 256; tools_dev/micro/bf16x9_probe.hip). The weights are split
// when they are packed; the gathered inputs are split on their way into LDS.
// LDS image per plane: A [128 rows][32 ch], B [CO cols][32 ch] bf16, 80-byte rows (64 + 16 pad:
// conflict-free ds_read_b128). Lane (r = lane%32, h = lane/32) of k-step s reads the 8 channels
// 16s + 8h .. +7 of its row / column: one 16-byte read per plane.
typedef __bf16 mf_v8bf __attribute__((ext_vector_type(8)));
#define X9_ROWB 80                       // bytes per LDS row

__device__ __forceinline__ void x9_split(float x, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const uint32_t u1 = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(u1);            // exact
    const uint32_t u2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(u2);           // exact, <= 8 significant bits
    p1 = u1 >> 16; p2 = u2 >> 16; p3 = __float_as_uint(r2) >> 16;
}

// two values at once, packed for the LDS images: word p = {plane p of b, plane p of a} (a in the low
// half). v_perm_b32 picks the two high halves directly, so no shift / or is spent on packing.
__device__ __forceinline__ void x9_split2(float a, float b, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
    const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    w1 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);     // exact
    const uint32_t va = __float_as_uint(ra), vb = __float_as_uint(rb);
    w2 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xFFFF0000u), sb = rb - __uint_as_float(vb & 0xFFFF0000u);   // exact, <= 8 bits
    w3 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

// ---- fp32 through TWO fp16 planes (the dense kernels' default arithmetic, NP = 2) ---------------------------------
// With a power-of-two scale s that puts the tensor's largest finite magnitude into [2^14, 2^15), v = x * s (exact) is
// h0 + h1 + r with h0 = fp16(v), h1 = fp16(v - h0), both round-to-nearest: |r| <= max(2^-22 |v|, 2^-25) - 22 significand
// bits where bf16 needs three planes for 24, and (a0 + a1)(b0 + b1) needs THREE matrix products (a1 * b1 < 2^-21 |ab| is
// dropped) instead of six. What fp16 does not have is fp32's exponent range: an element is kept to an ABSOLUTE accuracy of
// 2^-39 of its tensor's largest magnitude, so its relative accuracy falls below 2^-22 once it is smaller than 2^-17 of
// that maximum. For a sum of products that is an error of at most ~1e-12 * max|a| * sum|b| - far below the fp32
// accumulation error - but it is not fp32's element-wise semantics for tensors spanning more than ~2^38 in magnitude
// (test_dense_conv3x3_arithmetic_contract). Non-finite inputs: Inf splits into Inf + NaN, as on the bf16 path.
typedef _Float16 mf_v8h __attribute__((ext_vector_type(8)));

// scale 2^(14 - floor(log2(amax))) from the bits of the largest finite magnitude (0: empty / all-zero tensor -> 1)
__device__ __forceinline__ int h2_scale_exp(uint32_t amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xFF);
    if (e == 0) return 127;
    const int sb = 268 - e;
    return sb < 2 ? 2 : (sb > 252 ? 252 : sb);
}
__device__ __forceinline__ float h2_scale(int sb) { return __uint_as_float((uint32_t)sb << 23); }
__device__ __forceinline__ float h2_descale(int sb) { return __uint_as_float((uint32_t)(254 - sb) << 23); }
// two scaled values at once: word p = {plane p of b, plane p of a} (a in the low half)
// Four vector instructions per pair: v_cvt_pk_f16_f32, the two residuals by v_fma_mix_f32 (a - h0 with h0 read as the fp16 half it
// is: exact, as the cvt + sub pair the compiler makes of the plain expression - 8 instructions per pair, and every one of them is
// taken from the matrix pipe's issue slots), v_cvt_pk_f16_f32.
__device__ __forceinline__ void h2_split2(float a, float b, uint32_t& w0, uint32_t& w1) {
    typedef _Float16 h2_v2h __attribute__((ext_vector_type(2)));
    const h2_v2h h0 = {(_Float16)a, (_Float16)b};                   // round to nearest even
    const uint32_t hb = __builtin_bit_cast(uint32_t, h0);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hb), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hb), "v"(b));
    const h2_v2h h1 = {(_Float16)ra, (_Float16)rb};
    w0 = hb;
    w1 = __builtin_bit_cast(uint32_t, h1);
}

// The same planes of (a * s, b * s) for a power-of-two scale s, in FOUR vector instructions (round 5): each 16-bit half comes
// straight out of v_fma_mixlo / mixhi_f16 - hi = f16(a * s), lo = f16(a * s - hi), the product exact in the instruction's fp32
// arithmetic (s is a power of two), so the words are bit-identical to h2_split2(a * s, b * s, ..) (tools_dev/micro/split_probe.hip:
// 0 of 134 M words differ, overflow to Inf / NaN included) without the scale multiply and the second v_cvt_pk_f16_f32. Every
// vector instruction a staging wave issues costs the matrix-instruction wave on the same SIMD ~9 cycles
// (tools_dev/micro/ws_interference_probe.hip), so the count matters more than the instructions' own 4 cycles.
__device__ __forceinline__ void h2_split2s(float a, float b, float s, uint32_t& w0, uint32_t& w1) {
    uint32_t hb, lb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hb) : "v"(a), "v"(s));                  // (the other half: whatever was there, replaced next)
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hb) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(a), "v"(s), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(b), "v"(s), "v"(hb));
    w0 = hb;
    w1 = lb;
}

// h2_split2s for a scale that is the same in every lane (a kernel-wide operand scale): the scale stays in a scalar register - as a
// "v" operand the compiler copies it into a vector register in front of every use it cannot hoist (one v_mov per staged piece in
// the producer waves of dense_conv_ws.hip). Must NOT be given a per-lane value.
__device__ __forceinline__ void h2_split2u(float a, float b, float s_uniform, uint32_t& w0, uint32_t& w1) {
    uint32_t hb, lb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hb) : "v"(a), "s"(s_uniform));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hb) : "v"(b), "s"(s_uniform));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(a), "s"(s_uniform), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(b), "s"(s_uniform), "v"(hb));
    w0 = hb;
    w1 = lb;
}
