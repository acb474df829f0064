// Shared helpers for the gfx950 kernels behind include/gga_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gga_hip.h"

#define GGA_WAVE 64

void gga_set_error(const char* fmt, ...);
// event pair [2] of the next sample of an armed timing session (api.cc), or nullptr
hipEvent_t* gga_timing_acquire(int site, int64_t key);
#define GGA_TIME_START(tev_, stream_) \
    do { if (tev_) GGA_CHECK_HIP(hipEventRecord((tev_)[0], stream_), "timing record"); } while (0)
#define GGA_TIME_STOP(tev_, stream_) \
    do { if (tev_) GGA_CHECK_HIP(hipEventRecord((tev_)[1], stream_), "timing record"); } while (0)

// bn_relu.hip, for the kernels that produce BatchNorm-backward partial sums themselves
int gga_bn_bwd_finalize(const double* partials, int nblocks, int channels, int64_t rows, const float* gamma,
                        const float* saved, float* grad_gamma, float* grad_beta, void* workspace, float** coef,
                        hipStream_t stream);

#define GGA_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            gga_set_error(__VA_ARGS__);             \
            return GGA_ERR_INVALID_ARG;             \
        }                                           \
    } while (0)

#define GGA_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            gga_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
            return GGA_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

#define GGA_CHECK_HIP(expr, name)                                                 \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            gga_set_error("%s: %s", name, hipGetErrorString(e_));                 \
            return GGA_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

__host__ __device__ static inline size_t gga_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// frames per call of the batched point ops (per-frame offsets travel to the kernels by value)
#define GGA_MAX_BATCH 128

// ---- wave / block reductions (wave = 64 lanes on CDNA) ----------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Largest finite magnitude (as float bits) seen by a workgroup -> *out by ONE guarded atomicMax (same-address atomics
// serialise in the L2: thousands of them cost more than the pass they decorate). Call from every thread of a
// workgroup of at most 1024 threads; `m` = the thread's running maximum of (bits & 0x7fffffff) over finite values.
// BatchNorm as one multiply-add per element, y = x * scale + shift. Every kernel that needs the pair (the forward
// finalisation, and the backward kernels that recompute the ReLU mask from x) derives it through this one function, so the
// recomputed mask is bit for bit the forward pass's decision.
__device__ __forceinline__ void gga_bn_scale_shift(float gamma, float beta, float mean, float invstd, float& scale,
                                                   float& shift) {
    scale = gamma * invstd;
    shift = fmaf(-mean, scale, beta);
}

__device__ __forceinline__ uint32_t gga_amax_of(float v, uint32_t m) {
    const uint32_t u = __float_as_uint(v) & 0x7FFFFFFFu;
    return (u < 0x7F800000u && u > m) ? u : m;
}
__device__ __forceinline__ void gga_amax_commit(uint32_t m, uint32_t* out) {
    __shared__ uint32_t gga_amax_wm[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int nw = (blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
    if ((tid & 63) == 0) gga_amax_wm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nw; ++w) m = gga_amax_wm[w] > m ? gga_amax_wm[w] : m;
        if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
    }
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
