// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (diagnostic tool, not part of the library).
// Measures cycles per wave64 VALU instruction per SIMD for: independent / dependent fma chains, DPP adds,
// v_exp, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 independent fma chains
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (MODE == 1) {   // one dependent chain
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(m), "v"(c));
            }
        } else if (MODE == 2) {   // 8 independent dpp adds (quad_perm)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 3) {   // 8 independent row_bcast dpp adds
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 4) {   // 8 independent v_exp
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 5) {   // 8 independent v_cndmask + v_cmp pairs
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            }
        } else if (MODE == 7) {   // 4 independent v_pk_fma_f32 (2 floats per lane each)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            }
        } else if (MODE == 8) {   // 4 independent v_pk_mul_f32
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            }
        } else if (MODE == 6) {   // 8 independent v_mul_f32 (2-operand)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
}

template <int MODE>
double run(int waves_per_simd, int iters, float* d)
{
    // 256 CUs x 4 SIMDs; one block of 64*4*w threads per CU would need dispatcher luck; use blocks of 256 threads (1 wave per SIMD) x w per CU
    int blocks = 256 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)iters * 64.0 * waves_per_simd;     // 64 VALU instr per iteration per wave
    return ms * 1e-3 / instr_per_simd;                                // seconds per instruction per SIMD
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char* names[] = { "fma x8 independent", "fma dependent chain", "add_dpp quad_perm x8", "add_dpp row_bcast x8", "v_exp x8", "cmp+cndmask x4", "v_mul x8", "v_pk_fma x4 (64 instr/iter)", "v_pk_mul x4 (64 instr/iter)" };
    const int iters = 20000;
    for (int mode = 0; mode < 9; ++mode) {
        printf("%-24s", names[mode]);
        for (int w : { 1, 2, 4, 8 }) {
            double s = 0;
            switch (mode) {
            case 0: s = run<0>(w, iters, d); break; case 1: s = run<1>(w, iters, d); break; case 2: s = run<2>(w, iters, d); break;
            case 3: s = run<3>(w, iters, d); break; case 4: s = run<4>(w, iters, d); break; case 5: s = run<5>(w, iters, d); break;
            case 6: s = run<6>(w, iters, d); break; case 7: s = run<7>(w, iters, d); break; case 8: s = run<8>(w, iters, d); break;
            }
            printf("  w=%d: %.2f ns/instr/SIMD (%.2f cyc @2.4GHz)", w, s * 1e9, s * 2.4e9);
        }
        printf("\n");
    }
    return 0;
}
