#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <numeric>
int main() {
    const int n = 510000;
    std::vector<uint32_t> h(n); for (int i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u) & ((1u << 27) - 1);
    uint32_t *k, *ko; int *vo; hipMalloc(&k, n * 4); hipMalloc(&ko, n * 4); hipMalloc(&vo, n * 4);
    hipMemcpy(k, h.data(), n * 4, hipMemcpyHostToDevice);
    size_t bytes = 0;
    rocprim::counting_iterator<int> iota(0);
    rocprim::radix_sort_pairs(nullptr, bytes, k, ko, iota, vo, n, 0, 27, 0);
    void* tmp; hipMalloc(&tmp, bytes);
    printf("temp bytes %zu\n", bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) rocprim::radix_sort_pairs(tmp, bytes, k, ko, iota, vo, n, 0, 27, 0);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) rocprim::radix_sort_pairs(tmp, bytes, k, ko, iota, vo, n, 0, 27, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("radix_sort_pairs u32 27 bits, n=%d: %.1f us\n", n, ms / 20 * 1e3);
    std::vector<int> p(n); hipMemcpy(p.data(), vo, n * 4, hipMemcpyDeviceToHost);
    std::vector<int> ref(n); std::iota(ref.begin(), ref.end(), 0);
    std::stable_sort(ref.begin(), ref.end(), [&](int a, int b) { return h[a] < h[b]; });
    printf("matches stable sort: %d\n", (int)(p == ref));
    // 64-bit keys, 52 bits
    std::vector<uint64_t> h6(n); for (int i = 0; i < n; ++i) h6[i] = ((uint64_t)(i * 2654435761u) * 0x9E3779B97F4A7C15ull) >> 12;
    uint64_t *k6, *k6o; hipMalloc(&k6, n * 8); hipMalloc(&k6o, n * 8); hipMemcpy(k6, h6.data(), n * 8, hipMemcpyHostToDevice);
    size_t b6 = 0; rocprim::radix_sort_pairs(nullptr, b6, k6, k6o, iota, vo, n, 0, 52, 0);
    void* t6; hipMalloc(&t6, b6);
    for (int r = 0; r < 3; ++r) rocprim::radix_sort_pairs(t6, b6, k6, k6o, iota, vo, n, 0, 52, 0);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) rocprim::radix_sort_pairs(t6, b6, k6, k6o, iota, vo, n, 0, 52, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("radix_sort_pairs u64 52 bits: %.1f us (temp %zu)\n", ms / 20 * 1e3, b6);
    return 0;
}
