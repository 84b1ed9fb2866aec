// gs_scan.h -- exclusive prefix sums used by compaction, tile-count offsets and the radix sort.
#pragma once
#include "gs_common.h"

// Single-block exclusive scan of n u32 values (n up to a few hundred thousand).  total -> *total_out.
static __global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* in, uint32_t* out, int n,
                                                      uint32_t* total_out)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < n ? in[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_tot[w];
        uint32_t carry = carry_s;
        if (i < n) out[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}


// ---- three-phase scan for large n: reduce 2048-element chunks, scan the chunk sums, apply ----
#define GS_SCAN_CHUNK 2048

static __global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ in, int n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t ws[4];
    int base = blockIdx.x * GS_SCAN_CHUNK;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { int i = base + k * 256 + threadIdx.x; if (i < n) acc += in[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

static __global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* in, uint32_t* out, int n,
                                                           const uint32_t* __restrict__ chunk_offsets)
{
    __shared__ uint32_t ws[4];
    __shared__ uint32_t carry_s;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = chunk_offsets[blockIdx.x];
    __syncthreads();
    int base = blockIdx.x * GS_SCAN_CHUNK;
    for (int k = 0; k < 8; ++k) {
        int i = base + k * 256 + threadIdx.x;
        uint32_t v = i < n ? in[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += ws[w];
        uint32_t carry = carry_s;
        if (i < n) out[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + incl;
        __syncthreads();
    }
}

// in -> out (may alias), n elements; tmp must hold 2*ceil(n/2048) u32.
static inline void gs_scan_u32(const uint32_t* in, uint32_t* out, int n, uint32_t* tmp, uint32_t* total, hipStream_t s,
                               GsProf* prof = nullptr)
{
    if (n <= 0) return;
    if (n <= 8192) { GS_TIMED(prof, KID_SCAN_BLOCKS, s, k_scan_blocks<<<1, 1024, 0, s>>>(in, out, n, total)); return; }
    int chunks = (n + GS_SCAN_CHUNK - 1) / GS_SCAN_CHUNK;
    GS_TIMED(prof, KID_SCAN_REDUCE, s, k_scan_reduce<<<chunks, 256, 0, s>>>(in, n, tmp));
    GS_TIMED(prof, KID_SCAN_BLOCKS, s, k_scan_blocks<<<1, 1024, 0, s>>>(tmp, tmp + chunks, chunks, total));
    GS_TIMED(prof, KID_SCAN_APPLY, s, k_scan_apply<<<chunks, 256, 0, s>>>(in, out, n, tmp + chunks));
}
