// Write bandwidth of a scatter on MI355X as a function of RUN LENGTH: n (key, value) pairs are written to a permuted position such
// that L consecutive lanes write L consecutive elements (a "run") and consecutive runs land far apart -- what one radix pass does
// when a 4096-pair tile is spread over B bins (L ~ 4096 / B pairs per run: 16 for 8-bit digits, 1-2 for 12-bit digits).
// Prices the "two passes of 12 + 11 bits instead of three of 8" idea.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_scatter_runs.bin tools/ubench_scatter_runs.hip && tools/ubench_scatter_runs.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_scatter(const uint32_t* __restrict__ kin, const int32_t* __restrict__ vin, uint32_t* __restrict__ kout, int32_t* __restrict__ vout,
                          uint32_t n, uint32_t L, uint32_t nruns, uint32_t mult)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t run = i / L, within = i - run * L;
    const uint32_t dst_run = (uint32_t)(((uint64_t)run * mult) % nruns);      // a bijection on the runs (mult coprime to nruns)
    const uint32_t pos = dst_run * L + within;
    kout[pos] = kin[i];
    vout[pos] = vin[i];
}

static uint32_t gcd(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }

int main()
{
    const uint32_t n = 6u << 20;          // ~6.3 M pairs, the headline frame's K
    uint32_t *kin, *kout; int32_t *vin, *vout;
    hipMalloc(&kin, 4ull * n); hipMalloc(&kout, 4ull * n); hipMalloc(&vin, 4ull * n); hipMalloc(&vout, 4ull * n);
    hipMemset(kin, 1, 4ull * n); hipMemset(vin, 2, 4ull * n);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("{\"pairs\": %u, \"bytes_moved_per_pass\": %llu, \"runs\": [", n, 16ull * n);
    const uint32_t Ls[] = {1, 2, 4, 8, 16, 32, 64, 256};
    for (int li = 0; li < 8; ++li) {
        const uint32_t L = Ls[li], nruns = n / L;
        uint32_t mult = 1000003u; while (gcd(mult, nruns) != 1) mult += 2;
        for (int w = 0; w < 3; ++w) k_scatter<<<(n + 255) / 256, 256>>>(kin, vin, kout, vout, n, L, nruns, mult);
        hipEventRecord(a);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) k_scatter<<<(n + 255) / 256, 256>>>(kin, vin, kout, vout, n, L, nruns, mult);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= reps;
        printf("%s{\"run_length_pairs\": %u, \"us\": %.1f, \"GBps_of_16B_per_pair\": %.0f}", li ? ", " : "", L, ms * 1e3, 16.0 * n / (ms * 1e-3) / 1e9);
    }
    printf("]}\n");
    return 0;
}
