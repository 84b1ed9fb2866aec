// k_loss.hip -- the step right after the rasteriser in the reference's training loop:
//   L = (1 - lambda) * L1 + lambda * (1 - SSIM)          LossFunction.py:20-38
// with SSIM as pytorch_msssim.ssim(data_range=1, size_average=True) computes it (11-tap Gaussian window,
// sigma 1.5, "valid" separable filtering, K1 = 0.01, K2 = 0.03), forward AND the gradient w.r.t. the predicted
// image in three launches instead of the ~40 small torch kernels of conv2d-based SSIM + autograd:
//   k_loss_ssim_maps  forward: per 32x54 tile the five filtered maps, the SSIM value and the three derivative maps
//                     dS/d(mu1), dS/d(E[x^2]), dS/d(E[xy]); per-block SSIM and L1 partial sums
//   k_loss_finish     fixed-order sum of the partials -> {L, L1, LD_SSIM}
//   k_loss_grad       backward: per 32x54 tile the transposed (full) filtering of the derivative maps, chain to dS/dx,
//                     the L1 sign term, the upstream scalar and (optionally) torch.clamp's mask
// Both filters run the same way: the HORIZONTAL pass straight from global memory, a thread taking four adjacent outputs of a
// row from sixteen consecutive inputs (four 16-byte loads; every input is used by up to four of its outputs in registers),
// the results through LDS, the VERTICAL pass seven outputs of a column per thread from seventeen LDS reads per map.
// The derivative maps live in a PADDED array P (3, Hp, Wp), P[y][x] = map[y - 10][x - 12], zero outside the map:
// the transposed filter of the backward then is a plain forward stencil without bounds tests, and the column shift of 12
// keeps the sixteen-float row loads of both kernels 16-byte aligned.
// The predicted image may be any strided (3,H,W) view -- in particular the permuted (H,W,3) output of the rasteriser, read in
// place -- and torch.clamp(., 0, 1) (GaussianPointTrainer.py:173) can be applied on the fly (value and gradient as torch's).
// Also here: gs_adam_step's kernel (torch.optim.Adam semantics, GaussianPointTrainer.py:131-134,183-184).
// No float atomics, results reproducible.
#include "gs_common.h"

#define LT 32                 // tile width
#define LTY 54                // tile height
#define LW 11                 // window
#define LH (LW - 1)           // halo
#define LPR (LTY + LH)        // patch rows = 64
#define LHS 36                // floats per row of the LDS intermediate (16-byte aligned rows)
#define LHR (LPR + 2)         // its rows: the last row group's two unused outputs read two rows past the patch
#define LXS 12                // column shift of the padded maps
#define LVR 7                 // vertical pass: outputs per thread (8 row groups x 7 >= 54)
// the forward kernel's own tile height (its LDS intermediate holds five maps): 54 or 22 (patch rows a multiple of 32)
#ifndef GS_LOSS_MTY
#define GS_LOSS_MTY 22       // measured: 62.0 us against 67.3 at 1920x1088 (24.5 KB of LDS and six blocks per CU instead of 47.5 KB and three)
#endif
#define MTY GS_LOSS_MTY
#define MPR (MTY + LH)
#define MHR (MPR + 2)
#define MVR ((MTY + 7) / 8)
#define MITEMS (MPR * 8 / 256)

struct GsGaussWin { float g[LW]; };

__host__ __device__ inline int gs_loss_hp(int H) { return LTY * ((H + LTY - 1) / LTY) + LH; }
__host__ __device__ inline int gs_loss_wp(int W) { return LT * ((W + LT - 1) / LT) + LXS; }

// sixteen consecutive pixels of channel ch, row iy, starting at column cs (cs % 4 == 0), zero outside the image.
//   MODE 0: any strides, one load per pixel
//   MODE 1: unit pixel stride (a (3,H,W) array): four 16-byte loads.  W % 4 == 0 and 16-byte aligned rows: a float4 is all in or all out
//   MODE 2: interleaved channels (the rasteriser's (H,W,3) array: channel stride 1, pixel stride 3): twelve 16-byte loads bring the
//           48 floats of the sixteen pixels, the channel's sixteen are picked out of the registers
#define GS_LOSS_SCALAR 0
#define GS_LOSS_VEC 1
#define GS_LOSS_VEC3 2
template <int CH>
__device__ __forceinline__ void gs_loss_pick4(const float* __restrict__ src, float* v)
{
    const float4* q = reinterpret_cast<const float4*>(src);
    const float4 a = q[0], b = q[1], d = q[2];
    const float f[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w };
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[3 * i + CH];
}
template <int MODE>
__device__ __forceinline__ void gs_loss_load16(const GsLossImage& im, int ch, int iy, int cs, int H, int W, float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.0f;
    if (iy < 0 || iy >= H) return;
    const float* row = im.p + (long long)ch * im.sc + (long long)iy * im.sy;
    if (MODE == GS_LOSS_VEC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cs + 4 * j;
            if (c >= 0 && c + 3 < W) {
                const float4 q = *reinterpret_cast<const float4*>(row + c);
                v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
            }
        }
    } else if (MODE == GS_LOSS_VEC3) {
        const float* row0 = im.p + (long long)iy * im.sy;               // channel 0 of the row's first pixel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cs + 4 * j;
            if (c >= 0 && c + 3 < W) {
                if (ch == 0) gs_loss_pick4<0>(row0 + 3 * c, v + 4 * j);
                else if (ch == 1) gs_loss_pick4<1>(row0 + 3 * c, v + 4 * j);
                else gs_loss_pick4<2>(row0 + 3 * c, v + 4 * j);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int c = cs + i; if (c >= 0 && c < W) v[i] = row[(long long)c * im.sx]; }
    }
    if (im.clamp) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] < 0.0f ? 0.0f : (v[i] > 1.0f ? 1.0f : v[i]);        // (keeps a NaN, like torch.clamp)
    }
}

__device__ __forceinline__ float gs_loss_pixel(const GsLossImage& im, int ch, int iy, int ix, bool clamp)
{
    float v = im.p[(long long)ch * im.sc + (long long)iy * im.sy + (long long)ix * im.sx];
    if (clamp) v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return v;
}

template <int XMODE, int YMODE>
__global__ __launch_bounds__(256) void k_loss_ssim_maps(const GsLossImage X, const GsLossImage Y, int H, int W, int Hp, int Wp,
                                                        const GsGaussWin win, float* __restrict__ mapA, float* __restrict__ mapB,
                                                        float* __restrict__ mapC, float* __restrict__ partial_ssim,
                                                        float* __restrict__ partial_l1)
{
    __shared__ __attribute__((aligned(16))) float hz[5][MHR][LHS];
    __shared__ float wsum[2][4];
    const int ch = blockIdx.z;
    const int X0 = blockIdx.x * LT, Y0 = blockIdx.y * MTY;       // this block's corner of P
    const int t = threadIdx.x;
    float local_l1 = 0.0f;
    // horizontal pass: patch row r is image row Y0 - 10 + r; output column X0 + 4 g + j is map column X0 + 4 g + j - 12 and
    // takes image columns (X0 + 4 g - 12) + j .. + j + 10
#pragma unroll 1
    for (int it = 0; it < MITEMS; ++it) {
        const int item = t + 256 * it;
        const int r = item >> 3, g = item & 7;
        float xin[16], yin[16];
        gs_loss_load16<XMODE>(X, ch, Y0 - LH + r, X0 + 4 * g - LXS, H, W, xin);
        gs_loss_load16<YMODE>(Y, ch, Y0 - LH + r, X0 + 4 * g - LXS, H, W, yin);
        // the L1 term of the four pixels the last 16-byte load brought (row Y0 - 10 + r, columns X0 + 4 g .. + 3): every pixel of the
        // block's own rows and columns once, from registers (pixels outside the image were loaded as 0 for both images)
        if (r >= LH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) local_l1 += fabsf(xin[12 + j] - yin[12 + j]);
        }
        float acc[5][4];
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = 0.0f;
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const float a = xin[i], b = yin[i];
            const float aa = a * a, bb = b * b, ab = a * b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = i - j;
                if (k < 0 || k > LH) continue;
                const float w = win.g[k];
                acc[0][j] = __builtin_fmaf(w, a, acc[0][j]); acc[1][j] = __builtin_fmaf(w, b, acc[1][j]);
                acc[2][j] = __builtin_fmaf(w, aa, acc[2][j]); acc[3][j] = __builtin_fmaf(w, bb, acc[3][j]);
                acc[4][j] = __builtin_fmaf(w, ab, acc[4][j]);
            }
        }
#pragma unroll
        for (int m = 0; m < 5; ++m)
            *reinterpret_cast<float4*>(&hz[m][r][4 * g]) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
    }
    __syncthreads();
    // vertical pass: thread (c, rg) takes the outputs of rows MVR rg .. MVR rg + MVR - 1 of column c
    const int c = t & 31, r0 = (t >> 5) * MVR;
    float out[5][MVR];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        float v[MVR + LH];
#pragma unroll
        for (int i = 0; i < MVR + LH; ++i) v[i] = hz[m][r0 + i][c];
#pragma unroll
        for (int j = 0; j < MVR; ++j) {
            float sacc = 0.0f;
#pragma unroll
            for (int k = 0; k < LW; ++k) sacc = __builtin_fmaf(win.g[k], v[j + k], sacc);
            out[m][j] = sacc;
        }
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float local = 0.0f;
    const int x = X0 + c;
#pragma unroll
    for (int j = 0; j < MVR; ++j) {
        const int r = r0 + j, y = Y0 + r;
        if (r >= MTY || y >= Hp || x >= Wp) continue;
        float A = 0.0f, B = 0.0f, D = 0.0f;
        if (y >= LH && y < H && x >= LXS && x < W + (LXS - LH)) {          // map pixel (y - 10, x - 12) exists
            const float m1 = out[0][j], m2 = out[1][j];
            const float s1 = out[2][j] - m1 * m1, s2 = out[3][j] - m2 * m2, s12 = out[4][j] - m1 * m2;
            const float A1 = 2.0f * m1 * m2 + C1, A2 = 2.0f * s12 + C2;
            const float B1 = m1 * m1 + m2 * m2 + C1, B2 = s1 + s2 + C2;
            // S = (A1 / B1) (A2 / B2); dS/dmu1 = 2 mu2 (S/A1 - S/A2) + 2 mu1 S (1/B2 - 1/B1); dS/dE[x^2] = -S/B2; dS/dE[xy] = 2 S/A2
            // with S/A1 = A2 / (B1 B2) and S/A2 = A1 / (B1 B2): nothing is divided by A1 or A2 (A2 passes through zero)
            const float iB1 = __builtin_amdgcn_rcpf(B1), iB2 = __builtin_amdgcn_rcpf(B2);
            const float pq = A1 * iB1, qq = A2 * iB2;
            const float S = pq * qq;
            const float S_A1 = iB1 * qq, S_A2 = pq * iB2;
            local += S;
            A = 2.0f * m2 * (S_A1 - S_A2) + 2.0f * m1 * S * (iB2 - iB1);
            B = -S * iB2;
            D = 2.0f * S_A2;
        }
        const size_t o = ((size_t)ch * Hp + y) * Wp + x;
        mapA[o] = A; mapB[o] = B; mapC[o] = D;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { local += __shfl_xor(local, o, 64); local_l1 += __shfl_xor(local_l1, o, 64); }
    if ((t & 63) == 0) { wsum[0][t >> 6] = local; wsum[1][t >> 6] = local_l1; }
    __syncthreads();
    if (t == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial_ssim[b] = (wsum[0][0] + wsum[0][1]) + (wsum[0][2] + wsum[0][3]);
        partial_l1[b] = (wsum[1][0] + wsum[1][1]) + (wsum[1][2] + wsum[1][3]);
    }
}

// d L / d predicted: pixel (iy, ix) collects map pixels (iy - k, ix - l), k, l in [0, 10] = P rows iy .. iy + 10 and columns
// ix + 2 .. ix + 12, weights g[k] g[l] (the window is symmetric)
// (one block per channel also for a gradient in (H,W,3) memory order: a block that takes the three channels of its tile in turn, so
// that the three 4-byte stores per pixel meet in one L2, measured slower -- 63.8 against 60.5 us; the planar layout runs in 44)
__global__ __launch_bounds__(256) void k_loss_grad(const GsLossImage X, const GsLossImage Y, int H, int W, int Hp, int Wp,
                                                   const GsGaussWin win, const float* __restrict__ mapA, const float* __restrict__ mapB,
                                                   const float* __restrict__ mapC, float lambda, const float* __restrict__ upstream,
                                                   const GsLossImage G)
{
    __shared__ __attribute__((aligned(16))) float hz[3][LHR][LHS];
    const int X0 = blockIdx.x * LT, I0 = blockIdx.y * LTY;
    const int t = threadIdx.x;
    const float* maps[3] = { mapA, mapB, mapC };
    const float up = upstream ? upstream[0] : 1.0f;
    const float k_ssim = lambda / (3.0f * (float)(H - LH) * (float)(W - LH));
    const float k_l1 = (1.0f - lambda) / (3.0f * (float)H * (float)W);
    const int ch = blockIdx.z;
    // this thread's seven pixels of both images, asked for now and used after the two filter passes
    const int ix = X0 + (t & 31), iy_first = I0 + (t >> 5) * LVR;
    float px[LVR], py[LVR];
#pragma unroll
    for (int j = 0; j < LVR; ++j) {
        const int iy = iy_first + j;
        const bool in = (t >> 5) * LVR + j < LTY && iy < H && ix < W;
        px[j] = in ? gs_loss_pixel(X, ch, iy, ix, false) : 0.0f;
        py[j] = in ? gs_loss_pixel(Y, ch, iy, ix, Y.clamp != 0) : 0.0f;
    }
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int item = t + 256 * it;
        const int r = item >> 3, g = item & 7;
        const size_t o = ((size_t)ch * Hp + (size_t)(I0 + r)) * Wp + (size_t)(X0 + 4 * g);       // inside P by construction (gs_loss_hp / gs_loss_wp)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float4* src = reinterpret_cast<const float4*>(maps[m] + o);
            const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
            const float in[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
            float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int i = 2; i < 16; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = i - 2 - j;
                    if (k < 0 || k > LH) continue;
                    acc[j] = __builtin_fmaf(win.g[k], in[i], acc[j]);
                }
            *reinterpret_cast<float4*>(&hz[m][r][4 * g]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();
    const int c = t & 31, r0 = (t >> 5) * LVR;
    float out[3][LVR];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        float v[LVR + LH];
#pragma unroll
        for (int i = 0; i < LVR + LH; ++i) v[i] = hz[m][r0 + i][c];
#pragma unroll
        for (int j = 0; j < LVR; ++j) {
            float sacc = 0.0f;
#pragma unroll
            for (int k = 0; k < LW; ++k) sacc = __builtin_fmaf(win.g[k], v[j + k], sacc);
            out[m][j] = sacc;
        }
    }
#pragma unroll
    for (int j = 0; j < LVR; ++j) {
        const int r = r0 + j, iy = I0 + r;
        if (r >= LTY || iy >= H || ix >= W) continue;
        const float raw = px[j];
        const float x = X.clamp ? (raw < 0.0f ? 0.0f : (raw > 1.0f ? 1.0f : raw)) : raw;
        const float y = py[j];
        const float dS = out[0][j] + 2.0f * x * out[1][j] + y * out[2][j];          // d(sum of the SSIM map)/dx
        const float diff = x - y;
        const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
        float gval = up * (k_l1 * sgn - k_ssim * dS);
        if (X.clamp && !(raw >= 0.0f && raw <= 1.0f)) gval = 0.0f;                  // torch.clamp's backward: the gradient passes where min <= x <= max
        const_cast<float*>(G.p)[(long long)ch * G.sc + (long long)iy * G.sy + (long long)ix * G.sx] = gval;
    }
}

__global__ __launch_bounds__(1024) void k_loss_finish(const float* __restrict__ partial_ssim, const float* __restrict__ partial_l1, int n,
                                                      int H, int W, float lambda, float* __restrict__ terms)
{
    __shared__ double ws[2][16];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = t; i < n; i += 1024) { a += (double)partial_ssim[i]; b += (double)partial_l1[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((t & 63) == 0) { ws[0][t >> 6] = a; ws[1][t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < 16; ++w) { sa += ws[0][w]; sb += ws[1][w]; }
        const double ssim = sa / (3.0 * (double)(H - LH) * (double)(W - LH));
        const double l1 = sb / (3.0 * (double)H * (double)W);
        const double ld = 1.0 - ssim;
        terms[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * ld);
        terms[1] = (float)l1;
        terms[2] = (float)ld;
    }
}

static dim3 gs_loss_grid_maps(int H, int W) { return dim3((unsigned)((gs_loss_wp(W) + LT - 1) / LT), (unsigned)((gs_loss_hp(H) + MTY - 1) / MTY), 3); }

size_t gs_loss_maps_size(int H, int W) { return (size_t)3 * 3 * (size_t)gs_loss_hp(H) * (size_t)gs_loss_wp(W); }
size_t gs_loss_partials_floats(int H, int W) { const dim3 g = gs_loss_grid_maps(H, W); return 2 * (size_t)g.x * g.y * g.z + 64; }

static GsGaussWin gs_loss_window()
{
    GsGaussWin win;
    double g[LW], sum = 0.0;
    for (int i = 0; i < LW; ++i) { const double d = (double)i - (LW / 2); g[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < LW; ++i) win.g[i] = (float)(g[i] / sum);
    return win;
}

static int gs_loss_mode(const GsLossImage& im, int W)
{
    const bool aligned = (W & 3) == 0 && (im.sy & 3) == 0 && ((uintptr_t)im.p & 15u) == 0;
    if (aligned && im.sx == 1 && (im.sc & 3) == 0) return GS_LOSS_VEC;
    if (aligned && im.sx == 3 && im.sc == 1) return GS_LOSS_VEC3;
    return GS_LOSS_SCALAR;
}

void gs_launch_loss_forward(const GsLossImage& X, const GsLossImage& Y, int H, int W, float lambda, float* maps, float* partials, float* terms,
                            hipStream_t s)
{
    const int Hp = gs_loss_hp(H), Wp = gs_loss_wp(W);
    const size_t map = (size_t)3 * Hp * Wp;
    const dim3 grid = gs_loss_grid_maps(H, W);
    const int nb = (int)(grid.x * grid.y * grid.z);
    float* p_ssim = partials; float* p_l1 = partials + nb;
    const int xm = gs_loss_mode(X, W), ym = gs_loss_mode(Y, W);
#define GS_LOSS_MAPS(XM_, YM_) k_loss_ssim_maps<XM_, YM_><<<grid, 256, 0, s>>>(X, Y, H, W, Hp, Wp, gs_loss_window(), maps, maps + map, maps + 2 * map, p_ssim, p_l1)
    if (xm == GS_LOSS_VEC && ym == GS_LOSS_VEC) GS_LOSS_MAPS(GS_LOSS_VEC, GS_LOSS_VEC);
    else if (xm == GS_LOSS_VEC3 && ym == GS_LOSS_VEC) GS_LOSS_MAPS(GS_LOSS_VEC3, GS_LOSS_VEC);          // the trainer's case: rasteriser output against a (3,H,W) photograph
    else GS_LOSS_MAPS(GS_LOSS_SCALAR, GS_LOSS_SCALAR);
#undef GS_LOSS_MAPS
    k_loss_finish<<<1, 1024, 0, s>>>(p_ssim, p_l1, nb, H, W, lambda, terms);
}

void gs_launch_loss_backward(const GsLossImage& X, const GsLossImage& Y, int H, int W, float lambda, const float* maps, const float* upstream,
                             const GsLossImage& G, hipStream_t s)
{
    const int Hp = gs_loss_hp(H), Wp = gs_loss_wp(W);
    const size_t map = (size_t)3 * Hp * Wp;
    const dim3 grid((unsigned)((W + LT - 1) / LT), (unsigned)((H + LTY - 1) / LTY), 3);
    k_loss_grad<<<grid, 256, 0, s>>>(X, Y, H, W, Hp, Wp, gs_loss_window(), maps, maps + map, maps + 2 * map, lambda, upstream, G);
}

// ---------------------------------------------------------------------------------
// Scale regulariser, LossFunction.py:40-51: mean over valid points of || exp(s) ||_2, s = features[:, 4:7].
// In torch the boolean-mask indexing and its backward (a sort-based index_put) cost more than the rasteriser's
// whole backward; here: one pass for the value (per-block partials + fixed-order finish, which also leaves the
// number of valid points on the device) and one pass that writes the dense (N,56) gradient.
__global__ __launch_bounds__(256) void k_reg_partials(const float* __restrict__ feat, const int8_t* __restrict__ invalid, int64_t N,
                                                      float* __restrict__ psum, float* __restrict__ pcnt)
{
    __shared__ float ws[2][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v = 0.0f, c = 0.0f;
    if (i < N && invalid[i] == 0) {
        const float* r = feat + (size_t)GS_NFEAT * i + 4;
        const float e0 = gs_expf(r[0]), e1 = gs_expf(r[1]), e2 = gs_expf(r[2]);
        v = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        c = 1.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = v; ws[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = (ws[0][0] + ws[0][1]) + (ws[0][2] + ws[0][3]);
        pcnt[blockIdx.x] = (ws[1][0] + ws[1][1]) + (ws[1][2] + ws[1][3]);
    }
}

__global__ __launch_bounds__(1024) void k_reg_finish(const float* __restrict__ psum, const float* __restrict__ pcnt, int n,
                                                     float* __restrict__ out /* [0] = mean norm, [1] = number of valid points */)
{
    __shared__ double ws[2][16];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = t; i < n; i += 1024) { a += (double)psum[i]; b += (double)pcnt[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((t & 63) == 0) { ws[0][t >> 6] = a; ws[1][t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < 16; ++w) { sa += ws[0][w]; sb += ws[1][w]; }
        out[0] = (float)(sa / sb);           // 0/0 -> NaN like torch's mean of an empty tensor
        out[1] = (float)sb;
    }
}

// one thread per float4 of the (N,56) gradient (14 per row): the stores of a wave are one contiguous kilobyte (a lane per ROW
// writes 16 bytes of every 224: 2.6 TB/s); the thread that owns floats 4..7 of a valid row computes the three derivatives
__global__ __launch_bounds__(256) void k_reg_grad(const float* __restrict__ feat, const int8_t* __restrict__ invalid, int64_t N,
                                                  const float* __restrict__ value_and_count, const float* __restrict__ upstream,
                                                  float* __restrict__ grad)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * (GS_NFEAT / 4)) return;
    const int64_t i = e / (GS_NFEAT / 4);
    const int c = (int)(e - i * (GS_NFEAT / 4));
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c == 1 && invalid[i] == 0) {
        const float* r = feat + (size_t)GS_NFEAT * i + 4;
        const float e0 = gs_expf(r[0]), e1 = gs_expf(r[1]), e2 = gs_expf(r[2]);
        const float nrm = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        const float k = upstream[0] / (value_and_count[1] * nrm);       // d mean||e|| / d s_j = e_j^2 / (||e|| count)
        out = make_float4(k * e0 * e0, k * e1 * e1, k * e2 * e2, 0.0f);
    }
    reinterpret_cast<float4*>(grad)[e] = out;
}

void gs_launch_reg_value(const float* feat, const int8_t* invalid, int64_t N, float* workspace, float* out, hipStream_t s)
{
    const int nb = (int)((N + 255) / 256);
    if (nb == 0) { (void)hipMemsetAsync(out, 0, 2 * sizeof(float), s); return; }
    k_reg_partials<<<nb, 256, 0, s>>>(feat, invalid, N, workspace, workspace + nb);
    k_reg_finish<<<1, 1024, 0, s>>>(workspace, workspace + nb, nb, out);
}

void gs_launch_reg_grad(const float* feat, const int8_t* invalid, int64_t N, const float* value_and_count, const float* upstream,
                        float* grad, hipStream_t s)
{
    const int64_t nb = (N * (GS_NFEAT / 4) + 255) / 256;
    if (nb == 0) return;
    k_reg_grad<<<(unsigned)nb, 256, 0, s>>>(feat, invalid, N, value_and_count, upstream, grad);
}

// ---------------------------------------------------------------------------------
// torch.optim.Adam(betas, eps, no weight decay, no amsgrad) on a flat f32 tensor: one fused update.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                                              float* __restrict__ exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                                              float bias1, float bias2_sqrt)
{
    // (four elements per thread through 16-byte accesses measured no faster: 0.147 vs 0.141 ms for the two tensors of config 3)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i];
    const float m = beta1 * exp_avg[i] + (1.0f - beta1) * g;          // exp_avg.lerp_(grad, 1 - beta1)
    const float v = beta2 * exp_avg_sq[i] + (1.0f - beta2) * g * g;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    exp_avg[i] = m; exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    param[i] = param[i] - (lr / bias1) * (m / denom);
}

void gs_launch_adam(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int64_t step, hipStream_t s)
{
    if (n <= 0) return;
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    k_adam<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(param, grad, m, v, n, lr, beta1, beta2, eps, (float)b1, (float)sqrt(b2));
}
