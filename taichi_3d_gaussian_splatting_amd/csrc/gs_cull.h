// gs_cull.h -- conservative per-quadrant culling shared by the forward and backward blend kernels.
//
// A splat is dropped for an 8x8 pixel quadrant only when alpha = exp(e) * rescale * opacity is
// certainly below 1/255 at every pixel centre of the quadrant, i.e. when the reference's own test
// (RAST:451-452 forward, RAST:634 backward) would skip it for all 64 pixels.  The bound is the exact
// maximum of the Gaussian exponent over the quadrant's rectangle (minimum of the conic quadratic
// over the four edges), compared in the log domain with a slack that covers every rounding error
// of the per-pixel f32 evaluation.  Culling therefore never changes a result.
#pragma once
#include "gs_common.h"

// min over the rectangle [X0,X1]x[Y0,Y1] (pixel centre minus mean) of q = a x^2 + 2 b x y + c y^2, for a,c > 0.
// nb_c = -b/c and nb_a = -b/a are precomputed per splat (approximate reciprocals are fine: the
// caller adds a slack that dwarfs their error).
__device__ __forceinline__ float gs_rect_min_quadratic(float a, float b, float c, float nb_c, float nb_a,
                                                      float X0, float X1, float Y0, float Y1)
{
    if (X0 <= 0.0f && X1 >= 0.0f && Y0 <= 0.0f && Y1 >= 0.0f) return 0.0f;
    float best;
    {
        float y = __builtin_amdgcn_fmed3f(nb_c * X0, Y0, Y1);
        best = X0 * (a * X0 + 2.0f * b * y) + c * y * y;
        y = __builtin_amdgcn_fmed3f(nb_c * X1, Y0, Y1);
        best = fminf(best, X1 * (a * X1 + 2.0f * b * y) + c * y * y);
        float x = __builtin_amdgcn_fmed3f(nb_a * Y0, X0, X1);
        best = fminf(best, Y0 * (c * Y0 + 2.0f * b * x) + a * x * x);
        x = __builtin_amdgcn_fmed3f(nb_a * Y1, X0, X1);
        best = fminf(best, Y1 * (c * Y1 + 2.0f * b * x) + a * x * x);
    }
    return best;
}

struct CullSplat { float u, v, a, b, c, nb_c, nb_a, cut, sa, sb, sc; bool pd; };

__device__ __forceinline__ CullSplat gs_cull_prepare(float4 A, float4 B, float4 C)
{
    CullSplat s;
    s.u = A.x; s.v = A.y; s.a = A.z; s.b = A.w; s.c = B.x; s.cut = C.w;
    s.pd = s.a > 0.0f && s.c > 0.0f && s.a * s.c > s.b * s.b;
    s.nb_c = -s.b * __builtin_amdgcn_rcpf(s.c);
    s.nb_a = -s.b * __builtin_amdgcn_rcpf(s.a);
    s.sa = 4.0e-6f * fabsf(s.a); s.sb = 8.0e-6f * fabsf(s.b); s.sc = 4.0e-6f * fabsf(s.c);
    return s;
}

// true when the splat cannot reach alpha >= 1/255 anywhere in the 8x8 rectangle whose first pixel
// centre is (rx0, ry0).  Any NaN makes the test false (= keep the splat).
__device__ __forceinline__ bool gs_cull(const CullSplat& s, float rx0, float ry0)
{
    const float X0 = rx0 - s.u, X1 = X0 + 7.0f, Y0 = ry0 - s.v, Y1 = Y0 + 7.0f;
    const float ax = fmaxf(fabsf(X0), fabsf(X1)), ay = fmaxf(fabsf(Y0), fabsf(Y1));
    // rounding slack of the f32 exponent evaluated per pixel (terms can cancel for skewed conics)
    const float slack = 0.02f + (s.sa * ax * ax + s.sc * ay * ay + s.sb * ax * ay);
    const float qmin = gs_rect_min_quadratic(s.a, s.b, s.c, s.nb_c, s.nb_a, X0, X1, Y0, Y1);
    return s.pd && (-0.5f * qmin + slack < s.cut);
}

