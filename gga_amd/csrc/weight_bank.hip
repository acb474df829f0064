// Operands of the split-plane matrix kernels for MANY convolution weights in two launches: one absmax pass and one packing pass
// over device-side tables of entries, instead of one absmax_kernel + one *_pack_weight_kernel per convolution and direction
// every step (79 launches of ~5 us in the PointPillars step, ~150 in the sparse step; weights change once per step - in the
// optimizer - and all of them together). An operand may be assembled from several weight tensors (entries with their own
// channel / column offsets): the two first convolutions a head-branch launch covers, the 15 branch weights of the 960 -> 64
// backward-data convolution - no concatenated copy is made. Layouts: those of sp_pack_weight_split_kernel (gather-GEMM,
// 32-channel stages, swizzled quarters) and dense_pack_weight_kernel (3x3 dense, 16-channel stages); gga_amd/weight_bank.py
// keeps the tables.
#include "gga_common.h"
#include "conv_planes.h"

struct PackEntry {              // mirrors gga_hip.h::GgaPackEntry
    const float* src;
    uint16_t* dst;
    const uint32_t* amax;
    int64_t s_k0, s_k1, s_c, s_col;
    int64_t first;              // index of the entry's first element in the launch's flat element space
    int32_t kw, kvol, n_c, n_col, c0, col0, n_in, co, layout, reverse;
};
static_assert(sizeof(PackEntry) == sizeof(GgaPackEntry), "PackEntry must match the ABI struct");

struct AmaxEntry {              // mirrors gga_hip.h::GgaAmaxEntry
    const float* src;
    uint32_t* slot;
    int64_t n;                  // floats, a dense block of memory (any permutation of a tensor's elements)
    int64_t first_block;        // index of the entry's first workgroup
};
static_assert(sizeof(AmaxEntry) == sizeof(GgaAmaxEntry), "AmaxEntry must match the ABI struct");

template <typename E, typename F>
__device__ __forceinline__ int find_entry(const E* __restrict__ t, int n, int64_t i, F first_of) {
    int lo = 0, hi = n - 1;                     // last entry whose first <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_of(t[mid]) <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

#define WB_AMAX_PER_BLOCK (256 * 8)             // floats a workgroup reduces

__global__ __launch_bounds__(256) void wb_absmax_kernel(const AmaxEntry* __restrict__ table, int n_entries) {
    const int e = find_entry(table, n_entries, (int64_t)blockIdx.x, [](const AmaxEntry& a) { return a.first_block; });
    const AmaxEntry en = table[e];
    const int64_t base = ((int64_t)blockIdx.x - en.first_block) * WB_AMAX_PER_BLOCK;
    uint32_t m = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < en.n) {
            const uint32_t q = __float_as_uint(en.src[i]) & 0x7FFFFFFFu;
            if (q < 0x7F800000u && q > m) m = q;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > __hip_atomic_load(en.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(en.slot, m);
    }
}

__global__ __launch_bounds__(256) void wb_pack_kernel(const PackEntry* __restrict__ table, int n_entries, int64_t total, int np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = find_entry(table, n_entries, i, [](const PackEntry& a) { return a.first; });
    const PackEntry en = table[e];
    int64_t j = i - en.first;                               // (tap, column, channel) of the entry, channel fastest
    const int cl = (int)(j % en.n_c); j /= en.n_c;
    const int colloc = (int)(j % en.n_col);
    const int tap = (int)(j / en.n_col);
    const int t = en.reverse ? en.kvol - 1 - tap : tap;
    const int k0 = t / en.kw, k1 = t - k0 * en.kw;
    const float v = en.src[k0 * en.s_k0 + k1 * en.s_k1 + cl * en.s_c + colloc * en.s_col];
    const int c = en.c0 + cl, col = en.col0 + colloc;
    const int nchunks = (en.n_in + MF_TK - 1) / MF_TK;
    uint16_t* dst;
    int64_t plane;
    if (en.layout == 0) {                                   // gather-GEMM stage layout (sp_pack_weight_split_kernel)
        const int ch = c & 31;
        const int64_t stage = (int64_t)tap * nchunks + (c >> 5);
        plane = (int64_t)en.co * 32;
        dst = en.dst + stage * (np * plane) + (int64_t)col * 32 + ((((ch >> 3) ^ ((col >> 2) & 3)) << 3) | (ch & 7));
    } else {                                                // dense 3x3 stage layout (dense_pack_weight_kernel)
        const int64_t stage16 = (int64_t)tap * (2 * nchunks) + (c >> 4);
        plane = (int64_t)en.co * 16;
        dst = en.dst + stage16 * (np * plane) + (int64_t)col * 16 + (c & 15);
    }
    if (np == 3) {
        uint32_t p1, p2, p3;
        x9_split(v, p1, p2, p3);
        dst[0] = (uint16_t)p1; dst[plane] = (uint16_t)p2; dst[2 * plane] = (uint16_t)p3;
    } else {
        uint32_t w0, w1;
        h2_split2(v * h2_scale(h2_scale_exp(*en.amax)), 0.0f, w0, w1);
        dst[0] = (uint16_t)(w0 & 0xFFFFu); dst[plane] = (uint16_t)(w1 & 0xFFFFu);
    }
}

extern "C" int64_t gga_absmax_table_blocks(int64_t n) { return n <= 0 ? 0 : (n + WB_AMAX_PER_BLOCK - 1) / WB_AMAX_PER_BLOCK; }

extern "C" int gga_absmax_table(const GgaAmaxEntry* table_device, int n_entries, int64_t n_blocks, uint32_t* slots,
                                int n_slots, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(table_device && slots && n_entries >= 0 && n_slots >= 0 && n_blocks >= 0 && n_blocks < 2147483647ll,
                "gga_absmax_table: bad arguments");
    if (n_slots) GGA_CHECK_HIP(hipMemsetAsync(slots, 0, (size_t)n_slots * sizeof(uint32_t), stream), "gga_absmax_table: memset");
    if (n_entries == 0 || n_blocks == 0) return GGA_OK;
    hipLaunchKernelGGL(wb_absmax_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, (const AmaxEntry*)table_device, n_entries);
    GGA_CHECK_LAUNCH("wb_absmax_kernel");
    return GGA_OK;
}

extern "C" int gga_pack_weights_table(const GgaPackEntry* table_device, int n_entries, int64_t total, int planes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GGA_REQUIRE(table_device && n_entries >= 0 && total >= 0 && (planes == 2 || planes == 3), "gga_pack_weights_table: bad arguments");
    if (n_entries == 0 || total == 0) return GGA_OK;
    const int64_t nb = (total + 255) / 256;
    GGA_REQUIRE(nb < 2147483647ll, "gga_pack_weights_table: too many elements for one launch");
    hipLaunchKernelGGL(wb_pack_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const PackEntry*)table_device, n_entries, total, planes);
    GGA_CHECK_LAUNCH("wb_pack_kernel");
    return GGA_OK;
}
