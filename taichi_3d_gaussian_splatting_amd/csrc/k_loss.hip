// k_loss.hip -- the step right after the rasteriser in the reference's training loop:
//   L = (1 - lambda) * L1 + lambda * (1 - SSIM)          LossFunction.py:20-38
// with SSIM as pytorch_msssim.ssim(data_range=1, size_average=True) computes it (11-tap Gaussian window,
// sigma 1.5, "valid" separable filtering, K1 = 0.01, K2 = 0.03), forward AND the gradient w.r.t. the predicted
// image in three launches instead of the ~40 small torch kernels of conv2d-based SSIM + autograd:
//   k_loss_ssim_maps  per 32x32 output tile: five filtered maps from an LDS patch, the SSIM value and the
//                     three derivative maps dS/d(mu1), dS/d(E[x^2]), dS/d(E[xy]); per-block SSIM partial sums
//   k_loss_grad       per 32x32 input tile: transposed (full) filtering of the derivative maps, chain to dS/dx,
//                     the L1 term and its sign gradient; per-block L1 partial sums
//   k_loss_finish     fixed-order sum of the partials -> {L, L1, LD_SSIM}
// Images are (3,H,W) f32 contiguous like the trainer's image_pred / image_gt (GaussianPointTrainer.py:173-181).
// Also here: gs_adam_step's kernel (torch.optim.Adam semantics, GaussianPointTrainer.py:131-134,183-184).
// HBM-bound streaming kernels; no float atomics, results reproducible.
#include "gs_common.h"

#define LT 32                 // tile edge
#define LW 11                 // window
#define LH (LW - 1)           // halo
#define LP (LT + LH)          // patch edge = 42

struct GsGaussWin { float g[LW]; };

__global__ __launch_bounds__(256) void k_loss_ssim_maps(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                        GsGaussWin win, float* __restrict__ mapA, float* __restrict__ mapB,
                                                        float* __restrict__ mapC, float* __restrict__ partial_ssim)
{
    __shared__ float sx[LP][LP + 1], sy[LP][LP + 1];
    __shared__ float hz[5][LP][LT + 1];
    __shared__ float wsum[4];
    const int Ho = H - LH, Wo = W - LH;
    const int ch = blockIdx.z;
    const int ox0 = blockIdx.x * LT, oy0 = blockIdx.y * LT;
    const float* Xc = X + (size_t)ch * H * W;
    const float* Yc = Y + (size_t)ch * H * W;
    const int t = threadIdx.x;
    for (int i = t; i < LP * LP; i += 256) {
        const int r = i / LP, c = i % LP;
        const int iy = oy0 + r, ix = ox0 + c;
        const bool in = iy < H && ix < W;
        sx[r][c] = in ? Xc[(size_t)iy * W + ix] : 0.0f;
        sy[r][c] = in ? Yc[(size_t)iy * W + ix] : 0.0f;
    }
    __syncthreads();
    // horizontal pass: 42 rows x 32 columns, five maps
    for (int i = t; i < LP * LT; i += 256) {
        const int r = i / LT, c = i % LT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < LW; ++k) {
            const float a = sx[r][c + k], b = sy[r][c + k], w = win.g[k];
            m1 += w * a; m2 += w * b; e11 += w * (a * a); e22 += w * (b * b); e12 += w * (a * b);
        }
        hz[0][r][c] = m1; hz[1][r][c] = m2; hz[2][r][c] = e11; hz[3][r][c] = e22; hz[4][r][c] = e12;
    }
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float local = 0.0f;
    const int c = t & 31;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = (t >> 5) * 4 + rr;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < LW; ++k) {
            const float w = win.g[k];
            m1 += w * hz[0][r + k][c]; m2 += w * hz[1][r + k][c]; e11 += w * hz[2][r + k][c];
            e22 += w * hz[3][r + k][c]; e12 += w * hz[4][r + k][c];
        }
        const int oy = oy0 + r, ox = ox0 + c;
        if (oy < Ho && ox < Wo) {
            const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
            const float A1 = 2.0f * m1 * m2 + C1, A2 = 2.0f * s12 + C2;
            const float B1 = m1 * m1 + m2 * m2 + C1, B2 = s1 + s2 + C2;
            const float S = (A1 / B1) * (A2 / B2);
            local += S;
            // dS/dmu1 = S (2 mu2/A1 - 2 mu2/A2 - 2 mu1/B1 + 2 mu1/B2); dS/dE[x^2] = -S/B2; dS/dE[xy] = 2 S/A2
            const float iA1 = 1.0f / A1, iA2 = 1.0f / A2, iB1 = 1.0f / B1, iB2 = 1.0f / B2;
            const size_t o = ((size_t)ch * Ho + oy) * Wo + ox;
            mapA[o] = S * (2.0f * m2 * (iA1 - iA2) + 2.0f * m1 * (iB2 - iB1));
            mapB[o] = -S * iB2;
            mapC[o] = 2.0f * S * iA2;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((t & 63) == 0) wsum[t >> 6] = local;
    __syncthreads();
    if (t == 0) partial_ssim[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void k_loss_grad(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                   GsGaussWin win, const float* __restrict__ mapA, const float* __restrict__ mapB,
                                                   const float* __restrict__ mapC, float lambda, float* __restrict__ grad,
                                                   float* __restrict__ partial_l1)
{
    __shared__ float sm[3][LP][LP + 1];
    __shared__ float hz[3][LP][LT + 1];
    __shared__ float wsum[4];
    const int Ho = H - LH, Wo = W - LH;
    const int ch = blockIdx.z;
    const int ix0 = blockIdx.x * LT, iy0 = blockIdx.y * LT;
    const int t = threadIdx.x;
    // input pixel (iy, ix) collects output pixels (iy - k, ix - l), k,l in [0, 10]
    for (int i = t; i < LP * LP; i += 256) {
        const int r = i / LP, c = i % LP;
        const int oy = iy0 + r - LH, ox = ix0 + c - LH;
        const bool in = oy >= 0 && ox >= 0 && oy < Ho && ox < Wo;
        const size_t o = ((size_t)ch * Ho + (in ? oy : 0)) * Wo + (in ? ox : 0);
        sm[0][r][c] = in ? mapA[o] : 0.0f;
        sm[1][r][c] = in ? mapB[o] : 0.0f;
        sm[2][r][c] = in ? mapC[o] : 0.0f;
    }
    __syncthreads();
    for (int i = t; i < LP * LT; i += 256) {
        const int r = i / LT, c = i % LT;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < LW; ++k) {            // patch column c + k <-> output column ix - (LH - k): weight g[LH - k] = g[k] (symmetric)
            const float w = win.g[k];
            a += w * sm[0][r][c + k]; b += w * sm[1][r][c + k]; d += w * sm[2][r][c + k];
        }
        hz[0][r][c] = a; hz[1][r][c] = b; hz[2][r][c] = d;
    }
    __syncthreads();
    const float inv_ssim = 1.0f / (3.0f * (float)Ho * (float)Wo);
    const float inv_l1 = 1.0f / (3.0f * (float)H * (float)W);
    float local = 0.0f;
    const int c = t & 31;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = (t >> 5) * 4 + rr;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < LW; ++k) {
            const float w = win.g[k];
            a += w * hz[0][r + k][c]; b += w * hz[1][r + k][c]; d += w * hz[2][r + k][c];
        }
        const int iy = iy0 + r, ix = ix0 + c;
        if (iy < H && ix < W) {
            const size_t o = ((size_t)ch * H + iy) * W + ix;
            const float x = X[o], y = Y[o];
            const float dS = a + 2.0f * x * b + y * d;                  // d(sum of SSIM map)/dx
            const float diff = x - y;
            local += fabsf(diff);
            const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
            if (grad) grad[o] = (1.0f - lambda) * sgn * inv_l1 - lambda * inv_ssim * dS;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((t & 63) == 0) wsum[t >> 6] = local;
    __syncthreads();
    if (t == 0) partial_l1[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(1024) void k_loss_finish(const float* __restrict__ partial_ssim, int n_ssim,
                                                      const float* __restrict__ partial_l1, int n_l1, int H, int W, float lambda,
                                                      float* __restrict__ terms)
{
    __shared__ double ws[2][16];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = t; i < n_ssim; i += 1024) a += (double)partial_ssim[i];
    for (int i = t; i < n_l1; i += 1024) b += (double)partial_l1[i];
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

size_t gs_loss_workspace_floats(int H, int W)
{
    const size_t Ho = (size_t)(H - LH), Wo = (size_t)(W - LH);
    const size_t blocks_o = 3 * ((Ho + LT - 1) / LT) * ((Wo + LT - 1) / LT);
    const size_t blocks_i = 3 * (((size_t)H + LT - 1) / LT) * (((size_t)W + LT - 1) / LT);
    return 3 * 3 * Ho * Wo + blocks_o + blocks_i + 64;
}

void gs_launch_loss(const float* X, const float* Y, int H, int W, float lambda, float* workspace, float* terms, float* grad,
                    hipStream_t s)
{
    GsGaussWin win;
    double g[LW], sum = 0.0;
    for (int i = 0; i < LW; ++i) { const double d = (double)i - (LW / 2); g[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < LW; ++i) win.g[i] = (float)(g[i] / sum);
    const int Ho = H - LH, Wo = W - LH;
    const dim3 grid_o((Wo + LT - 1) / LT, (Ho + LT - 1) / LT, 3), grid_i((W + LT - 1) / LT, (H + LT - 1) / LT, 3);
    const size_t map = (size_t)3 * Ho * Wo;
    float* mapA = workspace; float* mapB = mapA + map; float* mapC = mapB + map;
    float* p_ssim = mapC + map; float* p_l1 = p_ssim + (size_t)grid_o.x * grid_o.y * 3;
    k_loss_ssim_maps<<<grid_o, 256, 0, s>>>(X, Y, H, W, win, mapA, mapB, mapC, p_ssim);
    k_loss_grad<<<grid_i, 256, 0, s>>>(X, Y, H, W, win, mapA, mapB, mapC, lambda, grad, p_l1);
    k_loss_finish<<<1, 1024, 0, s>>>(p_ssim, (int)(grid_o.x * grid_o.y * 3), p_l1, (int)(grid_i.x * grid_i.y * 3), H, W, lambda, terms);
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

__global__ __launch_bounds__(256) void k_reg_grad(const float* __restrict__ feat, const int8_t* __restrict__ invalid, int64_t N,
                                                  const float* __restrict__ value_and_count, const float* __restrict__ upstream,
                                                  float* __restrict__ grad)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float4* g4 = reinterpret_cast<float4*>(grad + (size_t)GS_NFEAT * i);
    float gs0 = 0.0f, gs1 = 0.0f, gs2 = 0.0f;
    if (invalid[i] == 0) {
        const float* r = feat + (size_t)GS_NFEAT * i + 4;
        const float e0 = gs_expf(r[0]), e1 = gs_expf(r[1]), e2 = gs_expf(r[2]);
        const float nrm = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        const float k = upstream[0] / (value_and_count[1] * nrm);       // d mean||e|| / d s_j = e_j^2 / (||e|| count)
        gs0 = k * e0 * e0; gs1 = k * e1 * e1; gs2 = k * e2 * e2;
    }
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    g4[0] = z;
    g4[1] = make_float4(gs0, gs1, gs2, 0.0f);
#pragma unroll
    for (int k = 2; k < GS_NFEAT / 4; ++k) g4[k] = z;
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
    const int nb = (int)((N + 255) / 256);
    if (nb == 0) return;
    k_reg_grad<<<nb, 256, 0, s>>>(feat, invalid, N, value_and_count, upstream, grad);
}

// ---------------------------------------------------------------------------------
// torch.optim.Adam(betas, eps, no weight decay, no amsgrad) on a flat f32 tensor: one fused update.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                                              float* __restrict__ exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                                              float bias1, float bias2_sqrt)
{
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
