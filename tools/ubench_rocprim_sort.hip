// How fast is rocPRIM's radix sort (onesweep) on the binning workload?  6e6 (u32 key, i32 value) pairs, 23 significant bits.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench_rocprim_sort.bin tools/ubench_rocprim_sort.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

int main()
{
    const size_t n = 6000000;
    const int bits = 23;
    std::vector<uint32_t> hk(n); std::vector<int> hv(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; ++i) { hk[i] = rng() & ((1u << bits) - 1); hv[i] = (int)i; }
    uint32_t *k0, *k1; int *v0, *v1;
    hipMalloc(&k0, 4 * n); hipMalloc(&k1, 4 * n); hipMalloc(&v0, 4 * n); hipMalloc(&v1, 4 * n);
    hipMemcpy(k0, hk.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), 4 * n, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    void* tmp; hipMalloc(&tmp, tmp_bytes);
    printf("temp storage %zu bytes\n", tmp_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int end_bit : {23, 16, 24, 32}) {
        for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, end_bit, 0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, end_bit, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("rocprim radix_sort_pairs, %zu pairs, bits [0,%d): %.1f us\n", n, end_bit, 1e3 * ms / reps);
    }
    // check
    std::vector<uint32_t> ok(n);
    rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    hipMemcpy(ok.data(), k1, 4 * n, hipMemcpyDeviceToHost);
    bool sorted = true; for (size_t i = 1; i < n; ++i) if (ok[i - 1] > ok[i]) { sorted = false; break; }
    printf("sorted: %d\n", (int)sorted);
    return 0;
}
