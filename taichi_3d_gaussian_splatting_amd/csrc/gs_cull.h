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

// The rectangle of an 8x8 quadrant's pixels that are still of interest (wave-uniform, from a 64-bit lane mask with
// lane = 8 * row + column): first column / row and extent in pixels.  A splat only has to be kept if it can reach
// 1/255 at one of THOSE pixels -- the others have saturated (forward) or lie beyond their last contributor (backward)
// and take nothing from it -- so the cull rectangle shrinks with the mask and the late part of a walk, where a few
// pixels keep a wave going, evaluates a fraction of the splats.  Exactly as invisible in the results as the quadrant cull.
struct CullRect { float x0, y0, wx, wy; };
__device__ __forceinline__ CullRect gs_live_rect(unsigned long long mask, float qx0, float qy0)
{
    // rows: which bytes of the mask are non-zero; columns: the OR of the eight bytes
    const int ymin = __builtin_ctzll(mask) >> 3, ymax = (63 - __builtin_clzll(mask)) >> 3;
    uint32_t c = (uint32_t)mask | (uint32_t)(mask >> 32);
    c |= c >> 16; c |= c >> 8; c &= 0xffu;
    const int xmin = __builtin_ctz(c), xmax = 31 - __builtin_clz(c);
    CullRect r;
    r.x0 = qx0 + (float)xmin; r.y0 = qy0 + (float)ymin; r.wx = (float)(xmax - xmin); r.wy = (float)(ymax - ymin);
    return r;
}

// true when the splat cannot reach alpha >= 1/255 anywhere in the rectangle of pixel centres [rx0, rx0 + wx] x [ry0, ry0 + wy]
// (wx = wy = 7: a whole 8x8 quadrant whose first pixel centre is (rx0, ry0)).  Any NaN makes the test false (= keep the splat).
__device__ __forceinline__ bool gs_cull(const CullSplat& s, float rx0, float ry0, float wx = 7.0f, float wy = 7.0f)
{
    const float X0 = rx0 - s.u, X1 = X0 + wx, Y0 = ry0 - s.v, Y1 = Y0 + wy;
    const float ax = fmaxf(fabsf(X0), fabsf(X1)), ay = fmaxf(fabsf(Y0), fabsf(Y1));
    // rounding slack of the f32 exponent evaluated per pixel (terms can cancel for skewed conics)
    const float slack = 0.02f + (s.sa * ax * ax + s.sc * ay * ay + s.sb * ax * ay);
    const float qmin = gs_rect_min_quadratic(s.a, s.b, s.c, s.nb_c, s.nb_a, X0, X1, Y0, Y1);
    return s.pd && (-0.5f * qmin + slack < s.cut);
}

