// How fast can 877.7 MB (the PointPillars canvas of 16 frames) be zero-filled? Variants of the store loop, 1 GiB of other
// traffic between launches (cold caches), median of 9.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, int U>
__global__ __launch_bounds__(256) void fill(float4* p, long long total4) {
    const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long t = base + u * 256;
        if (t < total4) {
            if (MODE == 0) p[t] = make_float4(0, 0, 0, 0);
            if (MODE == 1) __builtin_nontemporal_store(v4f{0, 0, 0, 0}, reinterpret_cast<v4f*>(p + t));
        }
    }
}
// persistent grid-stride: `blocks` workgroups walk the buffer
template <int MODE>
__global__ __launch_bounds__(256) void fill_gs(float4* p, long long total4) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long long)gridDim.x * 256) {
        if (MODE == 0) p[t] = make_float4(0, 0, 0, 0);
        else __builtin_nontemporal_store(v4f{0, 0, 0, 0}, reinterpret_cast<v4f*>(p + t));
    }
}
// each workgroup owns a contiguous span (one span per XCD-interleaved workgroup): span bytes = total / blocks
template <int MODE>
__global__ __launch_bounds__(256) void fill_span(float4* p, long long total4, long long span4) {
    const long long s0 = (long long)blockIdx.x * span4, s1 = s0 + span4 < total4 ? s0 + span4 : total4;
    for (long long t = s0 + threadIdx.x; t < s1; t += 256) {
        if (MODE == 0) p[t] = make_float4(0, 0, 0, 0);
        else __builtin_nontemporal_store(v4f{0, 0, 0, 0}, reinterpret_cast<v4f*>(p + t));
    }
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fill1(float4* p, long long total4) {
    const long long t = (long long)blockIdx.x * THREADS + threadIdx.x;
    if (t < total4) __builtin_nontemporal_store(v4f{0, 0, 0, 0}, reinterpret_cast<v4f*>(p + t));
}
// two adjacent float4 per thread (32 contiguous bytes per lane)
__global__ __launch_bounds__(256) void fill_pair(float4* p, long long total4) {
    const long long t = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (t + 1 < total4) { __builtin_nontemporal_store(v4f{0, 0, 0, 0}, reinterpret_cast<v4f*>(p + t)); __builtin_nontemporal_store(v4f{0, 0, 0, 0}, reinterpret_cast<v4f*>(p + t + 1)); }
}
int main() {
    const long long bytes = 16LL * 496 * 432 * 64 * 4, total4 = bytes / 16;
    float4* d; char* trash; hipMalloc(&d, bytes); hipMalloc(&trash, 1LL << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        std::vector<float> ts;
        for (int r = 0; r < 11; ++r) {
            hipMemsetAsync(trash, r, 1LL << 30, 0);
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-44s %.1f us  %.2f TB/s\n", name, ts[ts.size() / 2] * 1e3, bytes / (ts[ts.size() / 2] * 1e-3) / 1e12);
    };
    timeit("plain, 4 per thread", [&] { fill<0, 4><<<(unsigned)((total4 + 1023) / 1024), 256>>>(d, total4); });
    timeit("nontemporal, 4 per thread (shipped)", [&] { fill<1, 4><<<(unsigned)((total4 + 1023) / 1024), 256>>>(d, total4); });
    timeit("nontemporal, 1 per thread", [&] { fill<1, 1><<<(unsigned)((total4 + 255) / 256), 256>>>(d, total4); });
    timeit("nontemporal, 8 per thread", [&] { fill<1, 8><<<(unsigned)((total4 + 2047) / 2048), 256>>>(d, total4); });
    timeit("nontemporal, 16 per thread", [&] { fill<1, 16><<<(unsigned)((total4 + 4095) / 4096), 256>>>(d, total4); });
    timeit("plain, 16 per thread", [&] { fill<0, 16><<<(unsigned)((total4 + 4095) / 4096), 256>>>(d, total4); });
    timeit("nontemporal, 2 per thread", [&] { fill<1, 2><<<(unsigned)((total4 + 511) / 512), 256>>>(d, total4); });
    timeit("1 per thread, 64-thread workgroups", [&] { fill1<64><<<(unsigned)((total4 + 63) / 64), 64>>>(d, total4); });
    timeit("1 per thread, 128-thread workgroups", [&] { fill1<128><<<(unsigned)((total4 + 127) / 128), 128>>>(d, total4); });
    timeit("1 per thread, 512-thread workgroups", [&] { fill1<512><<<(unsigned)((total4 + 511) / 512), 512>>>(d, total4); });
    timeit("1 per thread, 1024-thread workgroups", [&] { fill1<1024><<<(unsigned)((total4 + 1023) / 1024), 1024>>>(d, total4); });
    timeit("32 contiguous bytes per lane", [&] { fill_pair<<<(unsigned)((total4 / 2 + 255) / 256), 256>>>(d, total4); });
    timeit("plain, 1 per thread", [&] { fill<0, 1><<<(unsigned)((total4 + 255) / 256), 256>>>(d, total4); });
    for (int blocks : {256, 512}) {
        char nm[64]; snprintf(nm, 64, "grid-stride nontemporal, %d workgroups", blocks);
        timeit(nm, [&] { fill_gs<1><<<blocks, 256>>>(d, total4); });
    }
    for (int blocks : {1024, 2048, 8192}) {
        char nm[64]; snprintf(nm, 64, "contiguous spans nontemporal, %d workgroups", blocks);
        const long long span4 = (total4 + blocks - 1) / blocks;
        timeit(nm, [&] { fill_span<1><<<blocks, 256>>>(d, total4, span4); });
    }
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    return 0;
}
