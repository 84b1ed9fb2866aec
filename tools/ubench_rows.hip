// Micro-benchmark: writing / reading N rows of 56 floats (224 B), one row per thread with 14 float4 accesses
// (the access pattern of k_bwd_points / k_project) against a wave-cooperative coalesced form through LDS.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_rows.bin tools/ubench_rows.hip && tools/ubench_rows.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ROW 56
__global__ __launch_bounds__(256) void w_strided(float* __restrict__ out, int n, float v)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4* r = reinterpret_cast<float4*>(out + (size_t)ROW * i);
#pragma unroll
    for (int k = 0; k < ROW / 4; ++k) r[k] = make_float4(v + k, v, v + i, v);
}

// each wave stages its 64 rows in LDS (row stride 57 floats: conflict-free column writes), then writes them out contiguously
__global__ __launch_bounds__(256) void w_lds(float* __restrict__ out, int n, float v)
{
    __shared__ float st[4][64 * 57];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float* my = st[wave] + lane * 57;
#pragma unroll
    for (int k = 0; k < ROW / 4; ++k) { my[4 * k] = v + k; my[4 * k + 1] = v; my[4 * k + 2] = v + i; my[4 * k + 3] = v; }
    __builtin_amdgcn_wave_barrier();
    const int row0 = blockIdx.x * 256 + wave * 64;
    const int rows = min(64, n - row0);
    if (rows <= 0) return;
    float* dst = out + (size_t)ROW * row0;
    for (int e = lane; e < rows * ROW; e += 64) dst[e] = st[wave][(e / ROW) * 57 + (e % ROW)];
}

// same, 16-byte stores: element e4 = 4 floats
__global__ __launch_bounds__(256) void w_lds4(float* __restrict__ out, int n, float v)
{
    __shared__ float st[4][64 * 60];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4* my = reinterpret_cast<float4*>(st[wave] + lane * 60);
#pragma unroll
    for (int k = 0; k < ROW / 4; ++k) my[k] = make_float4(v + k, v, v + i, v);
    __builtin_amdgcn_wave_barrier();
    const int row0 = blockIdx.x * 256 + wave * 64;
    const int rows = min(64, n - row0);
    if (rows <= 0) return;
    float4* dst = reinterpret_cast<float4*>(out + (size_t)ROW * row0);
    for (int e = lane; e < rows * 14; e += 64) dst[e] = *reinterpret_cast<const float4*>(st[wave] + (e / 14) * 60 + (e % 14) * 4);
}

__global__ __launch_bounds__(256) void r_strided(const float* __restrict__ in, int n, float* __restrict__ sink)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = reinterpret_cast<const float4*>(in + (size_t)ROW * i);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < ROW / 4; ++k) { float4 x = r[k]; acc += x.x + x.y + x.z + x.w; }
    if (acc == 1234.5f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void r_lds4(const float* __restrict__ in, int n, float* __restrict__ sink)
{
    __shared__ float st[4][64 * 60];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 256 + wave * 64;
    const int rows = min(64, n - row0);
    if (rows <= 0) return;
    const float4* src = reinterpret_cast<const float4*>(in + (size_t)ROW * row0);
    for (int e = lane; e < rows * 14; e += 64) *reinterpret_cast<float4*>(st[wave] + (e / 14) * 60 + (e % 14) * 4) = src[e];
    __builtin_amdgcn_wave_barrier();
    const float4* my = reinterpret_cast<const float4*>(st[wave] + lane * 60);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < ROW / 4; ++k) { float4 x = my[k]; acc += x.x + x.y + x.z + x.w; }
    if (acc == 1234.5f) sink[0] = acc;
}

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const int n = 500000;
    float *buf, *sink;
    hipMalloc(&buf, sizeof(float) * ROW * (size_t)n + 4096); hipMalloc(&sink, 64);
    hipMemset(buf, 0, sizeof(float) * ROW * (size_t)n);
    const int nb = (n + 255) / 256;
    const double mb = sizeof(float) * ROW * (double)n / 1e6;
    float t;
    t = timeit([&] { w_strided<<<nb, 256>>>(buf, n, 1.f); }, 50); printf("write strided  %.1f us  %.0f GB/s\n", t * 1e3, mb / t);
    t = timeit([&] { w_lds<<<nb, 256>>>(buf, n, 1.f); }, 50);     printf("write lds b32  %.1f us  %.0f GB/s\n", t * 1e3, mb / t);
    t = timeit([&] { w_lds4<<<nb, 256>>>(buf, n, 1.f); }, 50);    printf("write lds b128 %.1f us  %.0f GB/s\n", t * 1e3, mb / t);
    t = timeit([&] { r_strided<<<nb, 256>>>(buf, n, sink); }, 50); printf("read  strided  %.1f us  %.0f GB/s\n", t * 1e3, mb / t);
    t = timeit([&] { r_lds4<<<nb, 256>>>(buf, n, sink); }, 50);    printf("read  lds b128 %.1f us  %.0f GB/s\n", t * 1e3, mb / t);
    return 0;
}
