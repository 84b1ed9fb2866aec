/*
 * gs_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See gs_oracle.h for scope, citations and pinning status.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -mfma -fopenmp -shared -fPIC
 * All arithmetic is f32 (UTIL:8-9); products are summed left to right the way
 * the Python expressions associate; fmaf appears only inside gso_expf.
 */
#include "gs_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GSO_ALPHA_EPS 0.00392156862745098f   /* 1./255. (RAST:451, RAST:634) as f32 */

/* ------------------------------------------------------------------------- */
/* scalar helpers                                                            */
/* ------------------------------------------------------------------------- */

float gso_expf(float x)
{
    /* clamp so that 2^n below stays a normal number */
    if (x < -86.0f) x = -86.0f;
    if (x > 88.0f) x = 88.0f;
    float fx = x * 1.44269504088896341f;
    /* round to nearest even through the 1.5*2^23 magic constant */
    float n = (fx + 12582912.0f) - 12582912.0f;
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    union { float f; int32_t i; } u;
    u.f = y;
    u.i += ((int32_t)n) << 23;
    return u.f;
}

/* exp of the Gaussian falloff in the two blend loops (RAST:441-452, RAST:622-634; ti.exp there).  The hot evaluation of
 * the whole path -- once per (pixel, splat) -- so it is the cheapest sequence that stays within 3e-7 of exp on the range
 * that matters (alpha >= 1/255 needs x >= -5.6): 2^(x log2 e) with the integer part split off through the 1.5*2^23
 * constant, the fraction taken with ONE fused multiply-add (so the rounding of x*log2(e) does not enter), a degree-5
 * polynomial for 2^f - 1 on [-0.5, 0.5] (exact at f = 0) and the integer added to the exponent field.  Eleven
 * operations where gso_expf needs seventeen; libgsrast's gs_exp_blend is the same sequence, bit for bit. */
float gso_exp_blend(float x)
{
    if (x < -86.0f) x = -86.0f;
    if (x > 88.0f) x = 88.0f;
    const float L = 1.44269504088896341f;
    float t = x * L;
    float m = t + 12582912.0f;            /* low mantissa bits of m = round-to-nearest-even(t) */
    float n = m - 12582912.0f;
    float f = fmaf(x, L, -n);
    float q = 0.0013264712179079652f;
    q = fmaf(q, f, 0.009671511128544807f);
    q = fmaf(q, f, 0.05550733581185341f);
    q = fmaf(q, f, 0.24022242426872253f);
    q = fmaf(q, f, 0.6931470036506653f);
    float p = fmaf(q, f, 1.0f);
    union { float f; uint32_t u; } a, b;
    a.f = p; b.f = m;
    a.u += b.u << 23;                     /* the bits of 1.5*2^23 shift out; what is left is n << 23 */
    return a.f;
}

/* the exp of the blend loops as gso_config.blend_exp selects it (gs_oracle.h) */
static inline float blend_exp_select(float x, int which)
{
    if (which == GSO_EXP_LIBM) return expf(x);
    if (which == GSO_EXP_FAST2) { float t = x * 1.44269504088896341f; return exp2f(t); }
    if (which == GSO_EXP_ULP2) {                 /* FAST2 moved by -2..+2 ulp, chosen by a hash of the argument's bits */
        float t = x * 1.44269504088896341f;
        union { float f; uint32_t u; int32_t i; } a, r;
        a.f = x; r.f = exp2f(t);
        uint32_t h = a.u * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        if (r.f > 1e-30f) r.i += (int32_t)(h % 5u) - 2;
        return r.f;
    }
    return gso_exp_blend(x);
}

/* UTIL:351-353 ti_sigmoid */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + gso_expf(-x)); }

static inline float max_f(float a, float b) { return a > b ? a : b; }
static inline float min_f(float a, float b) { return a < b ? a : b; }
static inline int32_t max_i(int32_t a, int32_t b) { return a > b ? a : b; }
static inline int32_t min_i(int32_t a, int32_t b) { return a < b ? a : b; }

/* C[r x c] = A[r x k] @ B[k x c], row-major, terms summed k = 0,1,2,... */
static void matmul(const float* A, const float* B, float* C, int r, int k, int c)
{
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            float acc = A[i * k] * B[j];
            for (int t = 1; t < k; ++t) acc = acc + A[i * k + t] * B[t * c + j];
            C[i * c + j] = acc;
        }
}

static void transpose(const float* A, float* At, int r, int c)
{
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) At[j * r + i] = A[i * c + j];
}

/* ------------------------------------------------------------------------- */
/* pose helpers: UTIL:396-432                                                */
/* ------------------------------------------------------------------------- */

/* UTIL:403-413 quaternion_multiply_torch, (x,y,z,w) */
static void quat_mul(const float a[4], const float b[4], float o[4])
{
    float x0 = a[0], y0 = a[1], z0 = a[2], w0 = a[3];
    float x1 = b[0], y1 = b[1], z1 = b[2], w1 = b[3];
    o[0] = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
    o[1] = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
    o[2] = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
    o[3] = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
}

void gso_inverse_se3_qt(const float* q, const float* t, int n, float* q_inv, float* t_inv)
{
    for (int i = 0; i < n; ++i) {
        const float* qi = q + 4 * i;
        /* UTIL:396-400 conjugate */
        float qc[4] = { -qi[0], -qi[1], -qi[2], qi[3] };
        /* UTIL:416-423 quaternion_rotate_torch(q_inv, t): normalises q first */
        float nrm = sqrtf(qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] + qc[3] * qc[3]);
        float qn[4] = { qc[0] / nrm, qc[1] / nrm, qc[2] / nrm, qc[3] / nrm };
        float v4[4] = { t[3 * i], t[3 * i + 1], t[3 * i + 2], 0.0f };
        float qn_conj[4] = { -qn[0], -qn[1], -qn[2], qn[3] };
        float tmp[4], rot[4];
        quat_mul(qn, v4, tmp);
        quat_mul(tmp, qn_conj, rot);
        q_inv[4 * i + 0] = qc[0]; q_inv[4 * i + 1] = qc[1];
        q_inv[4 * i + 2] = qc[2]; q_inv[4 * i + 3] = qc[3];
        t_inv[3 * i + 0] = -rot[0]; t_inv[3 * i + 1] = -rot[1]; t_inv[3 * i + 2] = -rot[2];
    }
}

/* ------------------------------------------------------------------------- */
/* GP3D                                                                      */
/* ------------------------------------------------------------------------- */

/* GP3D:30-48 */
void gso_rotation_matrix_from_quaternion(const float q[4], float R[9])
{
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.0f - 2.0f * (yy + zz); R[1] = 2.0f * (xy - wz);        R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);        R[4] = 1.0f - 2.0f * (xx + zz); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);        R[7] = 2.0f * (yz + wx);        R[8] = 1.0f - 2.0f * (xx + yy);
}

/* GP3D:51-62 */
static void transform_from_qt(const float q[4], const float t[3], float T[16])
{
    float R[9];
    gso_rotation_matrix_from_quaternion(q, R);
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2];  T[3] = t[0];
    T[4] = R[3]; T[5] = R[4]; T[6] = R[5];  T[7] = t[1];
    T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = t[2];
    T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
}

/* GP3D:14-27 project_point_to_camera */
static void project_point_to_camera(const float p[3], const float T[16], const float Km[9],
                                    float uv[2], float pc[3])
{
    float h[4] = { p[0], p[1], p[2], 1.0f };
    float hc[4];
    matmul(T, h, hc, 4, 4, 1);
    pc[0] = hc[0]; pc[1] = hc[1]; pc[2] = hc[2];
    float uv1[3];
    matmul(Km, pc, uv1, 3, 3, 1);
    uv[0] = uv1[0] / pc[2];
    uv[1] = uv1[1] / pc[2];
}

/* UTIL:495-510 taichi_inverse_SE3 -> translation part only */
static void inverse_se3_translation(const float T[16], float o[3])
{
    float RT_neg[9] = { -T[0], -T[4], -T[8], -T[1], -T[5], -T[9], -T[2], -T[6], -T[10] };
    float t[3] = { T[3], T[7], T[11] };
    matmul(RT_neg, t, o, 3, 3, 1);
}

/* GP3D:65-87 */
static void projective_jacobian(const float Km[9], const float xyz[3], float J[6])
{
    float fx = Km[0], fy = Km[4];
    float x = xyz[0], y = xyz[1], z = xyz[2];
    J[0] = fx / z; J[1] = 0.0f;   J[2] = -(fx * x) / (z * z);
    J[3] = 0.0f;   J[4] = fy / z; J[5] = -(fy * y) / (z * z);
}

static void upper3x3(const float T[16], float W[9])
{
    W[0] = T[0]; W[1] = T[1]; W[2] = T[2];
    W[3] = T[4]; W[4] = T[5]; W[5] = T[6];
    W[6] = T[8]; W[7] = T[9]; W[8] = T[10];
}

/* GP3D:161-191 */
void gso_project_to_camera_covariance(const float q_cov[4], const float log_s[3],
                                      const float T[16], const float Km[9],
                                      const float xyz_camera[3], float cov[4])
{
    float J[6], R[9], W[9];
    projective_jacobian(Km, xyz_camera, J);
    gso_rotation_matrix_from_quaternion(q_cov, R);
    float es[3] = { gso_expf(log_s[0]), gso_expf(log_s[1]), gso_expf(log_s[2]) };
    float S[9] = { es[0], 0, 0, 0, es[1], 0, 0, 0, es[2] };
    float St[9], Rt[9], Wt[9], Jt[6];
    transpose(S, St, 3, 3);
    transpose(R, Rt, 3, 3);
    float RS[9], RSS[9], Sigma[9];
    matmul(R, S, RS, 3, 3, 3);
    matmul(RS, St, RSS, 3, 3, 3);
    matmul(RSS, Rt, Sigma, 3, 3, 3);      /* GP3D:182 */
    upper3x3(T, W);
    transpose(W, Wt, 3, 3);
    transpose(J, Jt, 2, 3);
    float JW[6], JWS[6], JWSW[6];
    matmul(J, W, JW, 2, 3, 3);
    matmul(JW, Sigma, JWS, 2, 3, 3);
    matmul(JWS, Wt, JWSW, 2, 3, 3);
    matmul(JWSW, Jt, cov, 2, 3, 2);       /* GP3D:190 */
}

/* GP3D:132-159 */
void gso_project_to_camera_position_jacobian(const float xyz[3], const float T[16],
                                             const float Km[9], float Jout[6])
{
    float W[9];
    upper3x3(T, W);
    float h[4] = { xyz[0], xyz[1], xyz[2], 1.0f }, t[4];
    matmul(T, h, t, 4, 4, 1);
    float tx = t[0], ty = t[1], tz = t[2];
    float d[6] = {
        Km[0] / tz, Km[1] / tz, (-Km[0] * tx - Km[1] * ty) / (tz * tz),
        Km[3] / tz, Km[4] / tz, (-Km[3] * tx - Km[4] * ty) / (tz * tz) };
    matmul(d, W, Jout, 2, 3, 3);
}

/* GP3D:237-331 */
void gso_project_to_camera_covariance_jacobian(const float q_cov[4], const float log_s[3],
                                               const float T[16], const float Km[9],
                                               const float xyz_camera[3],
                                               float dSigma_dq[16], float dSigma_ds[12])
{
    float J[6], R[9], W[9], U[6], M[9];
    projective_jacobian(Km, xyz_camera, J);
    gso_rotation_matrix_from_quaternion(q_cov, R);
    float es[3] = { gso_expf(log_s[0]), gso_expf(log_s[1]), gso_expf(log_s[2]) };
    float S[9] = { es[0], 0, 0, 0, es[1], 0, 0, 0, es[2] };
    matmul(R, S, M, 3, 3, 3);             /* GP3D:257 */
    upper3x3(T, W);
    matmul(J, W, U, 2, 3, 3);             /* GP3D:264 */
#define U_(i, j) U[(i) * 3 + (j)]
#define M_(i, j) M[(i) * 3 + (j)]
#define R_(i, j) R[(i) * 3 + (j)]
    /* GP3D:270-279 */
    float dSp_dS[36] = {
        U_(0,0)*U_(0,0), U_(0,0)*U_(0,1), U_(0,0)*U_(0,2), U_(0,0)*U_(0,1), U_(0,1)*U_(0,1), U_(0,1)*U_(0,2), U_(0,0)*U_(0,2), U_(0,1)*U_(0,2), U_(0,2)*U_(0,2),
        U_(0,0)*U_(1,0), U_(0,0)*U_(1,1), U_(0,0)*U_(1,2), U_(0,1)*U_(1,0), U_(0,1)*U_(1,1), U_(0,1)*U_(1,2), U_(0,2)*U_(1,0), U_(0,2)*U_(1,1), U_(0,2)*U_(1,2),
        U_(0,0)*U_(1,0), U_(0,1)*U_(1,0), U_(0,2)*U_(1,0), U_(0,0)*U_(1,1), U_(0,1)*U_(1,1), U_(0,2)*U_(1,1), U_(0,0)*U_(1,2), U_(0,1)*U_(1,2), U_(0,2)*U_(1,2),
        U_(1,0)*U_(1,0), U_(1,0)*U_(1,1), U_(1,0)*U_(1,2), U_(1,0)*U_(1,1), U_(1,1)*U_(1,1), U_(1,1)*U_(1,2), U_(1,0)*U_(1,2), U_(1,1)*U_(1,2), U_(1,2)*U_(1,2) };
    /* GP3D:282-292 */
    float dS_dM[81] = {
        2*M_(0,0), 2*M_(0,1), 2*M_(0,2), 0, 0, 0, 0, 0, 0,
        M_(1,0), M_(1,1), M_(1,2), M_(0,0), M_(0,1), M_(0,2), 0, 0, 0,
        M_(2,0), M_(2,1), M_(2,2), 0, 0, 0, M_(0,0), M_(0,1), M_(0,2),
        M_(1,0), M_(1,1), M_(1,2), M_(0,0), M_(0,1), M_(0,2), 0, 0, 0,
        0, 0, 0, 2*M_(1,0), 2*M_(1,1), 2*M_(1,2), 0, 0, 0,
        0, 0, 0, M_(2,0), M_(2,1), M_(2,2), M_(1,0), M_(1,1), M_(1,2),
        M_(2,0), M_(2,1), M_(2,2), 0, 0, 0, M_(0,0), M_(0,1), M_(0,2),
        0, 0, 0, M_(2,0), M_(2,1), M_(2,2), M_(1,0), M_(1,1), M_(1,2),
        0, 0, 0, 0, 0, 0, 2*M_(2,0), 2*M_(2,1), 2*M_(2,2) };
    float dSp_dM[36];
    matmul(dSp_dS, dS_dM, dSp_dM, 4, 9, 9);   /* GP3D:294 */
    /* GP3D:297-307 */
    float dM_dS[27] = {
        R_(0,0), 0, 0,  0, R_(0,1), 0,  0, 0, R_(0,2),
        R_(1,0), 0, 0,  0, R_(1,1), 0,  0, 0, R_(1,2),
        R_(2,0), 0, 0,  0, R_(2,1), 0,  0, 0, R_(2,2) };
    float dS_ds[9] = { es[0], 0, 0, 0, es[1], 0, 0, 0, es[2] };  /* GP3D:308-312 */
    float tmp43[12];
    matmul(dSp_dM, dM_dS, tmp43, 4, 9, 3);
    matmul(tmp43, dS_ds, dSigma_ds, 4, 3, 3);  /* GP3D:313 */
    float sx = es[0], sy = es[1], sz = es[2];
    float qx = q_cov[0], qy = q_cov[1], qz = q_cov[2], qw = q_cov[3];
    /* GP3D:319-329 */
    float dM_dq[36] = {
        0, -4*sx*qy, -4*sx*qz, 0,
        2*sy*qy, 2*sy*qx, -2*sy*qw, -2*sy*qz,
        2*sz*qz, 2*sz*qw, 2*sz*qx, 2*sz*qy,
        2*sx*qy, 2*sx*qx, 2*sx*qw, 2*sx*qz,
        -4*sy*qx, 0, -4*sy*qz, 0,
        -2*sz*qw, 2*sz*qz, 2*sz*qy, -2*sz*qx,
        2*sx*qz, -2*sx*qw, 2*sx*qx, -2*sx*qy,
        2*sy*qw, 2*sy*qz, 2*sy*qy, 2*sy*qx,
        -4*sz*qx, -4*sz*qy, 0, 0 };
    matmul(dSp_dM, dM_dq, dSigma_dq, 4, 9, 4);  /* GP3D:330 */
#undef U_
#undef M_
#undef R_
}

/* ------------------------------------------------------------------------- */
/* SH:10-32                                                                  */
/* ------------------------------------------------------------------------- */
void gso_spherical_harmonics(const float d[3], float sh[16])
{
    float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] / nrm, y = d[1] / nrm, z = d[2] / nrm;
    sh[0] = 0.28209479177387814f;
    sh[1] = -0.48860251190291987f * y;
    sh[2] = 0.48860251190291987f * z;
    sh[3] = -0.48860251190291987f * x;
    sh[4] = 1.0925484305920792f * x * y;
    sh[5] = -1.0925484305920792f * y * z;
    sh[6] = 0.94617469575755997f * z * z - 0.31539156525251999f;
    sh[7] = -1.0925484305920792f * x * z;
    sh[8] = 0.54627421529603959f * x * x - 0.54627421529603959f * y * y;
    sh[9] = 0.59004358992664352f * y * (-3.0f * x * x + y * y);
    sh[10] = 2.8906114426405538f * x * y * z;
    sh[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z * z);
    sh[12] = 0.3731763325901154f * z * (5.0f * z * z - 3.0f);
    sh[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z * z);
    sh[14] = 1.4453057213202769f * z * (x * x - y * y);
    sh[15] = 0.59004358992664352f * x * (-x * x + 3.0f * y * y);
}

static float dot16(const float* a, const float* b)
{
    float acc = a[0] * b[0];
    for (int i = 1; i < 16; ++i) acc = acc + a[i] * b[i];
    return acc;
}

/* ------------------------------------------------------------------------- */
/* UTIL:257-272                                                              */
/* ------------------------------------------------------------------------- */
void gso_conic_and_rescale(const float cov_in[4], float out[4])
{
    float c00 = cov_in[0], c01 = cov_in[1], c10 = cov_in[2], c11 = cov_in[3];
    float det_pre = c00 * c11 - c01 * c10;
    c00 = c00 + 0.3f;
    c11 = c11 + 0.3f;
    float det = c00 * c11 - c01 * c10;
    float rescale = sqrtf(max_f(0.0f, det_pre / det));
    float inv_det = 1.0f / det;
    out[0] = inv_det * c11;
    out[1] = inv_det * (-c01);
    out[2] = inv_det * c00;
    out[3] = rescale;
}

/* RAST:81-103 */
static void bounding_box_tiles(float u, float v, float radii, int32_t tiles_u, int32_t tiles_v, int32_t box[4]);

void gso_bounding_box(float u, float v, float radii, int W, int H, int32_t box[4])
{
    bounding_box_tiles(u, v, radii, W / GSO_TILE, H / GSO_TILE, box);
}

static void bounding_box_tiles(float u, float v, float radii, int32_t tiles_u, int32_t tiles_v, int32_t box[4])
{
    radii = max_f(radii, 1.0f);
    float min_u = max_f(0.0f, u - radii);
    float max_u = u + radii;
    float min_v = max_f(0.0f, v - radii);
    float max_v = v + radii;
    int32_t min_tile_u = (int32_t)floorf(min_u / (float)GSO_TILE);
    min_tile_u = min_i(min_tile_u, tiles_u);
    int32_t max_tile_u = (int32_t)floorf(max_u / (float)GSO_TILE) + 1;
    max_tile_u = min_i(max_i(max_tile_u, min_tile_u + 1), tiles_u);
    int32_t min_tile_v = (int32_t)floorf(min_v / (float)GSO_TILE);
    min_tile_v = min_i(min_tile_v, tiles_v);
    int32_t max_tile_v = (int32_t)floorf(max_v / (float)GSO_TILE) + 1;
    max_tile_v = min_i(max_i(max_tile_v, min_tile_v + 1), tiles_v);
    box[0] = min_tile_u; box[1] = max_tile_u; box[2] = min_tile_v; box[3] = max_tile_v;
}

/* RAST:175-193 */
void gso_find_tile_start_and_end(const int64_t* keys, int64_t n, int32_t* ts, int32_t* te)
{
    if (n <= 0) return;
    for (int64_t idx = 0; idx < n - 1; ++idx) {
        int32_t tile_id = (int32_t)(keys[idx] >> 32);
        int32_t next_tile_id = (int32_t)(keys[idx + 1] >> 32);
        if (tile_id != next_tile_id) {
            ts[next_tile_id] = (int32_t)(idx + 1);
            te[tile_id] = (int32_t)(idx + 1);
        }
    }
    te[(int32_t)(keys[n - 1] >> 32)] = (int32_t)n;
}

/* stable LSD radix sort of (i64 key, i32 value); ties keep input order.
 * RAST:947 calls torch sort(stable=False); the adopted contract is the stable
 * order CUB / CPU torch produce in practice (SURVEY 7 "sort tie order"). */
static void stable_sort_pairs(int64_t* keys, int32_t* vals, int64_t n)
{
    if (n <= 1) return;
    int64_t* k2 = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int32_t* v2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int64_t *ka = keys, *kb = k2;
    int32_t *va = vals, *vb = v2;
    for (int pass = 0; pass < 8; ++pass) {
        int shift = pass * 8;
        int64_t hist[257];
        memset(hist, 0, sizeof hist);
        for (int64_t i = 0; i < n; ++i) hist[(((uint64_t)ka[i]) >> shift & 0xff) + 1]++;
        int uniform = 0;
        for (int d = 0; d < 256; ++d) if (hist[d + 1] == n) uniform = 1;
        if (uniform) continue;
        for (int d = 0; d < 256; ++d) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < n; ++i) {
            int64_t pos = hist[((uint64_t)ka[i]) >> shift & 0xff]++;
            kb[pos] = ka[i]; vb[pos] = va[i];
        }
        int64_t* tk = ka; ka = kb; kb = tk;
        int32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        memcpy(keys, ka, sizeof(int64_t) * (size_t)n);
        memcpy(vals, va, sizeof(int32_t) * (size_t)n);
    }
    free(k2); free(v2);
}

/* ------------------------------------------------------------------------- */
/* forward: RAST:830-1023                                                    */
/* ------------------------------------------------------------------------- */

static void* zalloc(size_t bytes) { return calloc(bytes ? bytes : 1, 1); }

void gso_frame_free(gso_frame* f)
{
    if (!f) return;
    free(f->q_camera_pointcloud); free(f->t_camera_pointcloud);
    free(f->point_in_camera_mask); free(f->point_id_in_camera_list);
    free(f->point_uv); free(f->point_in_camera); free(f->point_uv_conic_and_rescale);
    free(f->point_alpha_after_activation); free(f->point_color); free(f->point_radii);
    free(f->num_overlap_tiles); free(f->accumulated_num_overlap_tiles);
    free(f->sort_key_unsorted); free(f->point_offset_unsorted);
    free(f->sort_key); free(f->point_offset_with_sort_key);
    free(f->tile_points_start); free(f->tile_points_end);
    free(f->rasterized_image); free(f->rasterized_depth); free(f->pixel_accumulated_alpha);
    free(f->pixel_offset_of_last_effective_point); free(f->pixel_valid_point_count);
    free(f);
}

int gso_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- per-pixel half of the forward: tile counts are already there; scan, keys, sort, ranges, blend ---- */
static void raster_stage(gso_frame* f, const gso_config* cfg)
{
    const int64_t M = f->M;
    const int32_t H = f->H, W = f->W;
    /* RAST:913-922 exclusive scan */
    int64_t K = 0;
    for (int64_t i = 0; i < M; ++i) { f->accumulated_num_overlap_tiles[i] = K; K += f->num_overlap_tiles[i]; }
    f->K = K;
    f->sort_key_unsorted = (int64_t*)zalloc(sizeof(int64_t) * (size_t)K);
    f->point_offset_unsorted = (int32_t*)zalloc(sizeof(int32_t) * (size_t)K);
    f->sort_key = (int64_t*)zalloc(sizeof(int64_t) * (size_t)K);
    f->point_offset_with_sort_key = (int32_t*)zalloc(sizeof(int32_t) * (size_t)K);

    /* ---- step 4: generate_point_sort_key_by_num_overlap_tiles, RAST:131-172 ---- */
    const float scale = cfg->depth_to_sort_key_scale;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t p = 0; p < M; ++p) {
        int32_t box[4];
        bounding_box_tiles(f->point_uv[2 * p], f->point_uv[2 * p + 1], f->point_radii[p], f->tiles_x, f->tiles_y, box);
        int32_t enc_depth = (int32_t)(f->point_in_camera[3 * p + 2] * scale);
        for (int32_t tu = box[0]; tu < box[1]; ++tu)
            for (int32_t tv = box[2]; tv < box[3]; ++tv) {
                int32_t cnt = (box[3] - box[2]) * (tu - box[0]) + (tv - box[2]);
                int64_t key_idx = f->accumulated_num_overlap_tiles[p] + cnt;
                int32_t tile_id = tu + tv * f->tiles_x;
                f->sort_key_unsorted[key_idx] = (int64_t)enc_depth + (((int64_t)tile_id) << 32);
                f->point_offset_unsorted[key_idx] = (int32_t)p;
            }
    }
    memcpy(f->sort_key, f->sort_key_unsorted, sizeof(int64_t) * (size_t)K);
    memcpy(f->point_offset_with_sort_key, f->point_offset_unsorted, sizeof(int32_t) * (size_t)K);
    stable_sort_pairs(f->sort_key, f->point_offset_with_sort_key, K);   /* RAST:947-949 */

    f->tile_points_start = (int32_t*)zalloc(sizeof(int32_t) * (size_t)f->T);  /* RAST:954-957 zeros */
    f->tile_points_end = (int32_t*)zalloc(sizeof(int32_t) * (size_t)f->T);
    gso_find_tile_start_and_end(f->sort_key, K, f->tile_points_start, f->tile_points_end);

    /* ---- step 5: gaussian_point_rasterisation, RAST:318-485 ----
     * The reference leaves the outputs uninitialised when K == 0 (RAST:967-980);
     * the oracle zero-fills, and tile start/end are all zero then, so the loop
     * below produces the same thing as "no contributors". */
    size_t P = (size_t)H * (size_t)W;
    f->rasterized_image = (float*)zalloc(sizeof(float) * 3 * P);
    f->rasterized_depth = (float*)zalloc(sizeof(float) * P);
    f->pixel_accumulated_alpha = (float*)zalloc(sizeof(float) * P);
    f->pixel_offset_of_last_effective_point = (int32_t*)zalloc(sizeof(int32_t) * P);
    f->pixel_valid_point_count = (int32_t*)zalloc(sizeof(int32_t) * P);
    const int rgb_only = cfg->rgb_only;
    const int which_exp = cfg->blend_exp;
    f->blend_exp = which_exp;
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t tile_id = 0; tile_id < f->T; ++tile_id) {
        int32_t tile_u = tile_id % f->tiles_x, tile_v = tile_id / f->tiles_x;
        int32_t start = f->tile_points_start[tile_id], end = f->tile_points_end[tile_id];
        for (int32_t t = 0; t < GSO_TILE * GSO_TILE; ++t) {
            int32_t pixel_u = tile_u * GSO_TILE + t % GSO_TILE;
            int32_t pixel_v = tile_v * GSO_TILE + t / GSO_TILE;
            if (pixel_u >= W || pixel_v >= H) continue;          /* partial tile (extension) */
            float px = (float)pixel_u + 0.5f, py = (float)pixel_v + 0.5f;
            float T_i = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
            float acc_depth = 0.0f, depth_norm = 0.0f;
            int32_t last = start, count = 0;
            for (int32_t idx = start; idx < end; ++idx) {
                int32_t p = f->point_offset_with_sort_key[idx];
                const float* cn = f->point_uv_conic_and_rescale + 4 * (size_t)p;
                /* UTIL:275-284 */
                float dx = px - f->point_uv[2 * (size_t)p], dy = py - f->point_uv[2 * (size_t)p + 1];
                float exponent = -0.5f * (dx * dx * cn[0] + dy * dy * cn[2]) - dx * dy * cn[1];
                float gaussian_alpha = blend_exp_select(exponent, which_exp) * cn[3];
                float alpha = gaussian_alpha * f->point_alpha_after_activation[p];
                if (alpha < GSO_ALPHA_EPS) continue;            /* RAST:451 */
                alpha = min_f(alpha, 0.99f);                     /* RAST:453 */
                float next_T = T_i * (1.0f - alpha);             /* RAST:457 */
                if (next_T < 0.0001f) break;                     /* RAST:458-460: saturated, nothing after changes state */
                last = idx + 1;
                const float* col = f->point_color + 3 * (size_t)p;
                cr += col[0] * alpha * T_i; cg += col[1] * alpha * T_i; cb += col[2] * alpha * T_i;
                if (!rgb_only) {
                    acc_depth += f->point_in_camera[3 * (size_t)p + 2] * alpha * T_i;
                    depth_norm += alpha * T_i;
                    count += 1;
                }
                T_i = next_T;
            }
            size_t o = (size_t)pixel_v * (size_t)W + (size_t)pixel_u;
            f->rasterized_image[3 * o] = cr; f->rasterized_image[3 * o + 1] = cg; f->rasterized_image[3 * o + 2] = cb;
            if (!rgb_only) {
                f->rasterized_depth[o] = acc_depth / max_f(depth_norm, 1e-6f);
                f->pixel_accumulated_alpha[o] = 1.0f - T_i;
                f->pixel_offset_of_last_effective_point[o] = last;
                f->pixel_valid_point_count[o] = count;
            }
        }
    }
}

gso_frame* gso_forward(const float* pc, float* feat, const int8_t* invalid, const int32_t* obj,
                       int64_t N, const float* q_pc, const float* t_pc, int32_t n_obj,
                       const float* Km, int32_t H, int32_t W, const gso_config* cfg)
{
    if (W <= 0 || H <= 0 || n_obj <= 0) return NULL;
    if (!cfg->allow_partial_tiles && (W % GSO_TILE != 0 || H % GSO_TILE != 0)) return NULL;  /* RAST:1193-1194 */
    gso_frame* f = (gso_frame*)zalloc(sizeof(gso_frame));
    f->N = N; f->H = H; f->W = W; f->n_objects = n_obj;
    f->tiles_x = (W + GSO_TILE - 1) / GSO_TILE; f->tiles_y = (H + GSO_TILE - 1) / GSO_TILE;   /* = W/16, H/16 for the reference's sizes */
    f->T = f->tiles_x * f->tiles_y;
    f->q_camera_pointcloud = (float*)zalloc(sizeof(float) * 4 * n_obj);
    f->t_camera_pointcloud = (float*)zalloc(sizeof(float) * 3 * n_obj);
    gso_inverse_se3_qt(q_pc, t_pc, n_obj, f->q_camera_pointcloud, f->t_camera_pointcloud); /* RAST:845 */

    /* ---- step 1: filter_point_in_camera, RAST:31-78 ---- */
    f->point_in_camera_mask = (int8_t*)zalloc((size_t)N);
    const float near_plane = cfg->near_plane, far_plane = cfg->far_plane;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        if (invalid[i] == 1) { f->point_in_camera_mask[i] = 0; continue; }
        float T[16], uv[2], pcam[3];
        transform_from_qt(f->q_camera_pointcloud + 4 * obj[i], f->t_camera_pointcloud + 3 * obj[i], T);
        project_point_to_camera(pc + 3 * i, T, Km, uv, pcam);
        float z = pcam[2];
        int in = z > near_plane && z < far_plane &&
                 uv[0] >= (float)(-GSO_TILE * GSO_BOUNDARY_TILES) &&
                 uv[0] < (float)(W + GSO_TILE * GSO_BOUNDARY_TILES) &&
                 uv[1] >= (float)(-GSO_TILE * GSO_BOUNDARY_TILES) &&
                 uv[1] < (float)(H + GSO_TILE * GSO_BOUNDARY_TILES);
        f->point_in_camera_mask[i] = in ? 1 : 0;
    }
    /* ---- compaction, RAST:861-870 (ascending ids) ---- */
    int64_t M = 0;
    for (int64_t i = 0; i < N; ++i) M += f->point_in_camera_mask[i];
    f->M = M;
    f->point_id_in_camera_list = (int32_t*)zalloc(sizeof(int32_t) * (size_t)M);
    { int64_t m = 0; for (int64_t i = 0; i < N; ++i) if (f->point_in_camera_mask[i]) f->point_id_in_camera_list[m++] = (int32_t)i; }

    f->point_uv = (float*)zalloc(sizeof(float) * 2 * (size_t)M);
    f->point_in_camera = (float*)zalloc(sizeof(float) * 3 * (size_t)M);
    f->point_uv_conic_and_rescale = (float*)zalloc(sizeof(float) * 4 * (size_t)M);
    f->point_alpha_after_activation = (float*)zalloc(sizeof(float) * (size_t)M);
    f->point_color = (float*)zalloc(sizeof(float) * 3 * (size_t)M);
    f->point_radii = (float*)zalloc(sizeof(float) * (size_t)M);
    f->num_overlap_tiles = (int32_t*)zalloc(sizeof(int32_t) * (size_t)M);
    f->accumulated_num_overlap_tiles = (int64_t*)zalloc(sizeof(int64_t) * (size_t)M);

    /* ---- step 2: generate_point_attributes_in_camera_plane, RAST:239-315 ---- */
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < M; ++idx) {
        int32_t pid = f->point_id_in_camera_list[idx];
        float* row = feat + (size_t)GSO_FEAT * pid;
        /* RAST:196-205 normalise the rotation quaternion in place */
        {
            float n = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2] + row[3] * row[3]);
            row[0] = row[0] / n; row[1] = row[1] / n; row[2] = row[2] / n; row[3] = row[3] / n;
        }
        const float* xyz = pc + 3 * (size_t)pid;
        float T[16], ray_origin[3];
        transform_from_qt(f->q_camera_pointcloud + 4 * obj[pid], f->t_camera_pointcloud + 3 * obj[pid], T);
        inverse_se3_translation(T, ray_origin);                       /* RAST:280-282 */
        float uv[2], pcam[3], cov[4], conic[4];
        project_point_to_camera(xyz, T, Km, uv, pcam);               /* RAST:284-287 */
        gso_project_to_camera_covariance(row, row + 4, T, Km, pcam, cov); /* RAST:288-292 */
        gso_conic_and_rescale(cov, conic);                            /* RAST:293 */
        f->point_uv[2 * idx] = uv[0]; f->point_uv[2 * idx + 1] = uv[1];
        f->point_in_camera[3 * idx] = pcam[0]; f->point_in_camera[3 * idx + 1] = pcam[1]; f->point_in_camera[3 * idx + 2] = pcam[2];
        memcpy(f->point_uv_conic_and_rescale + 4 * idx, conic, sizeof conic);
        f->point_alpha_after_activation[idx] = 1.0f / (1.0f + gso_expf(-row[7]));  /* RAST:299-300 */
        float dir[3] = { xyz[0] - ray_origin[0], xyz[1] - ray_origin[1], xyz[2] - ray_origin[2] };
        float sh[16];
        gso_spherical_harmonics(dir, sh);
        f->point_color[3 * idx + 0] = sigmoidf_(dot16(row + 8, sh));   /* GP3D:333-349 */
        f->point_color[3 * idx + 1] = sigmoidf_(dot16(row + 24, sh));
        f->point_color[3 * idx + 2] = sigmoidf_(dot16(row + 40, sh));
        /* RAST:311-315: radius from the covariance as the caller still holds it.
         * Taichi passes the mat2 to get_point_conic_and_rescale by value, so the
         * +0.3 blur does not leak back (switch kept for the open point in SURVEY 8a). */
        float c00 = cov[0], c01 = cov[1], c10 = cov[2], c11 = cov[3];
        if (!cfg->radius_from_preblur_cov) { c00 = c00 + 0.3f; c11 = c11 + 0.3f; }
        float large_eigen = (c00 + c11 + sqrtf((c00 - c11) * (c00 - c11) + 4.0f * c01 * c10)) / 2.0f;
        float radii = sqrtf(large_eigen) * 3.0f;
        f->point_radii[idx] = radii;
        /* ---- step 3: generate_num_overlap_tiles, RAST:106-128 ---- */
        int32_t box[4];
        bounding_box_tiles(uv[0], uv[1], radii, f->tiles_x, f->tiles_y, box);
        f->num_overlap_tiles[idx] = (box[1] - box[0]) * (box[3] - box[2]);
    }
    raster_stage(f, cfg);
    return f;
}

/* ------------------------------------------------------------------------- */
/* The same path cut at the projected records (test stand-in for the staged   */
/* entry points of libgsrast: gs_project_shard / gs_forward_projected /       */
/* gs_backward_projected / gs_backward_shard, include/gs_rasterizer.h).       */
/* A record = the per-point arrays of RAST:873-911 for one in-camera point:   */
/* u v conic_a conic_b | conic_c rescale opacity depth | r g b (unused) | x y z(camera) radius */
/* ------------------------------------------------------------------------- */
void gso_pack_records(const gso_frame* f, float* out)
{
    for (int64_t i = 0; i < f->M; ++i) {
        float* r = out + 16 * i;
        const float* cn = f->point_uv_conic_and_rescale + 4 * i;
        r[0] = f->point_uv[2 * i]; r[1] = f->point_uv[2 * i + 1]; r[2] = cn[0]; r[3] = cn[1];
        r[4] = cn[2]; r[5] = cn[3]; r[6] = f->point_alpha_after_activation[i]; r[7] = f->point_in_camera[3 * i + 2];
        r[8] = f->point_color[3 * i]; r[9] = f->point_color[3 * i + 1]; r[10] = f->point_color[3 * i + 2]; r[11] = 0.0f;
        r[12] = f->point_in_camera[3 * i]; r[13] = f->point_in_camera[3 * i + 1]; r[14] = f->point_in_camera[3 * i + 2];
        r[15] = f->point_radii[i];
    }
}

gso_frame* gso_forward_from_projected(int64_t M, const float* records, int32_t H, int32_t W, const gso_config* cfg)
{
    if (W <= 0 || H <= 0 || M < 0) return NULL;
    if (!cfg->allow_partial_tiles && (W % GSO_TILE != 0 || H % GSO_TILE != 0)) return NULL;
    gso_frame* f = (gso_frame*)zalloc(sizeof(gso_frame));
    f->N = M; f->M = M; f->H = H; f->W = W; f->n_objects = 0;
    f->tiles_x = (W + GSO_TILE - 1) / GSO_TILE; f->tiles_y = (H + GSO_TILE - 1) / GSO_TILE;
    f->T = f->tiles_x * f->tiles_y;
    f->point_uv = (float*)zalloc(sizeof(float) * 2 * (size_t)M);
    f->point_in_camera = (float*)zalloc(sizeof(float) * 3 * (size_t)M);
    f->point_uv_conic_and_rescale = (float*)zalloc(sizeof(float) * 4 * (size_t)M);
    f->point_alpha_after_activation = (float*)zalloc(sizeof(float) * (size_t)M);
    f->point_color = (float*)zalloc(sizeof(float) * 3 * (size_t)M);
    f->point_radii = (float*)zalloc(sizeof(float) * (size_t)M);
    f->num_overlap_tiles = (int32_t*)zalloc(sizeof(int32_t) * (size_t)M);
    f->accumulated_num_overlap_tiles = (int64_t*)zalloc(sizeof(int64_t) * (size_t)M);
    for (int64_t i = 0; i < M; ++i) {
        const float* r = records + 16 * i;
        f->point_uv[2 * i] = r[0]; f->point_uv[2 * i + 1] = r[1];
        float* cn = f->point_uv_conic_and_rescale + 4 * i;
        cn[0] = r[2]; cn[1] = r[3]; cn[2] = r[4]; cn[3] = r[5];
        f->point_alpha_after_activation[i] = r[6];
        f->point_color[3 * i] = r[8]; f->point_color[3 * i + 1] = r[9]; f->point_color[3 * i + 2] = r[10];
        f->point_in_camera[3 * i] = r[12]; f->point_in_camera[3 * i + 1] = r[13]; f->point_in_camera[3 * i + 2] = r[14];
        f->point_radii[i] = r[15];
        int32_t box[4];
        bounding_box_tiles(r[0], r[1], r[15], f->tiles_x, f->tiles_y, box);          /* RAST:106-128 */
        f->num_overlap_tiles[i] = (box[1] - box[0]) * (box[3] - box[2]);
    }
    raster_stage(f, cfg);
    return f;
}

/* ------------------------------------------------------------------------- */
/* backward: RAST:488-772 + RAST:1025-1163                                   */
/* ------------------------------------------------------------------------- */
int gso_backward(const gso_frame* f, const float* pc, const float* feat, const int32_t* obj,
                 const float* q_pc, const float* t_pc, const float* Km,
                 const float* grad_image, int32_t sh_band, const gso_config* cfg,
                 float* g_pc, float* g_feat, float* g_uv, float* mag, float* mag_img,
                 int32_t* n_affected, float* out_cov_buf, float* out_color_buf)
{
    return gso_backward_ex(f, pc, feat, obj, q_pc, t_pc, Km, grad_image, sh_band, cfg, g_pc, g_feat, g_uv, mag, mag_img,
                           n_affected, out_cov_buf, out_color_buf, NULL, NULL);
}

/* The same, plus (optional) for every gradient element the MAGNITUDE OF WHAT WAS SUMMED to produce it: the sums of the
 * absolute values of the loop-1 contributions (RAST:674-696), carried through the absolute values of the loop-2
 * Jacobians and the grad factors.  A float accumulation can only be judged against that magnitude (an element whose
 * terms cancel has no relative accuracy to speak of); the parity tests use it as the floor of their per-element bar. */
/* loop 1 (RAST:531-705) into the reference's accumulators; doubles stand in for ti.atomic_add (RAST:674-696) */
typedef struct { double *a_uv, *a_cov, *a_col, *a_alpha, *a_mag, *a_abs; int64_t* a_cnt; } gso_acc;

static void backward_loop1(const gso_frame* f, const float* grad_image, float* mag_img, gso_acc acc, int strict_dpdcov)
{
    const int32_t W = f->W;
    const int which_exp = f->blend_exp;                          /* the exp the forward of this frame used */
    double *a_uv = acc.a_uv, *a_cov = acc.a_cov, *a_col = acc.a_col, *a_alpha = acc.a_alpha, *a_mag = acc.a_mag, *a_abs = acc.a_abs;
    int64_t* a_cnt = acc.a_cnt;
    /* ---- loop 1, RAST:531-705; tiles in parallel, sums via atomic double adds ---- */
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t tile_id = 0; tile_id < f->T; ++tile_id) {
        int32_t tile_u = tile_id % f->tiles_x, tile_v = tile_id / f->tiles_x;
        int32_t start = f->tile_points_start[tile_id], end = f->tile_points_end[tile_id];
        for (int32_t t = 0; t < GSO_TILE * GSO_TILE; ++t) {
            int32_t pixel_u = tile_u * GSO_TILE + t % GSO_TILE;
            int32_t pixel_v = tile_v * GSO_TILE + t / GSO_TILE;
            if (pixel_u >= f->W || pixel_v >= f->H) continue;          /* partial tile (extension) */
            size_t o = (size_t)pixel_v * (size_t)W + (size_t)pixel_u;
            int32_t last = f->pixel_offset_of_last_effective_point[o];
            float accumulated_alpha = f->pixel_accumulated_alpha[o];
            float T_i = 1.0f - accumulated_alpha;
            float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f;
            float gr = grad_image[3 * o], gg = grad_image[3 * o + 1], gb = grad_image[3 * o + 2];
            float tot0 = 0.0f, tot1 = 0.0f;
            float px = (float)pixel_u + 0.5f, py = (float)pixel_v + 0.5f;
            for (int32_t idx = end - 1; idx >= start; --idx) {
                if (idx >= last) continue;                       /* RAST:609-610 */
                int32_t p = f->point_offset_with_sort_key[idx];
                const float* cn = f->point_uv_conic_and_rescale + 4 * (size_t)p;
                float a = cn[0], b = cn[1], c = cn[2];
                /* UTIL:331-348 */
                float dx = px - f->point_uv[2 * (size_t)p], dy = py - f->point_uv[2 * (size_t)p + 1];
                float cix = a * dx + b * dy, ciy = b * dx + c * dy;
                float quad = dx * cix + dy * ciy;
                float exponent = -0.5f * quad;
                float gaussian_alpha = blend_exp_select(exponent, which_exp) * cn[3];
                float dpm0 = gaussian_alpha * cix, dpm1 = gaussian_alpha * ciy;
                /* UTIL:343-345: 0.5 p (Sigma^-1 (d d^T) Sigma^-1), two 2x2 products in f32 */
                float oxx = dx * dx, oxy = dx * dy, oyx = dy * dx, oyy = dy * dy;
                float io00 = a * oxx + b * oyx, io01 = a * oxy + b * oyy;
                float io10 = b * oxx + c * oyx, io11 = b * oxy + c * oyy;
                float m00 = io00 * a + io01 * b, m01 = io00 * b + io01 * c, m11 = io10 * b + io11 * c;
                if (!strict_dpdcov) {        /* diagnostic only (gso_config.bwd_strict_dpdcov = 0): the same matrix as v v^T, v = Sigma^-1 d */
                    m00 = cix * cix; m01 = cix * ciy; m11 = ciy * ciy;
                }
                float hp = 0.5f * gaussian_alpha;
                float dpc00 = hp * m00, dpc01 = hp * m01, dpc11 = hp * m11;
                float apt = f->point_alpha_after_activation[p];
                float prod_alpha = gaussian_alpha * apt;
                if (prod_alpha >= GSO_ALPHA_EPS) {               /* RAST:634 */
                    float alpha = min_f(prod_alpha, 0.99f);
                    const float* col = f->point_color + 3 * (size_t)p;
                    T_i = T_i / (1.0f - alpha);                  /* RAST:643 */
                    accumulated_alpha = 1.0f - T_i; (void)accumulated_alpha;  /* RAST:644 */
                    float d_rgb_d_color = alpha * T_i;
                    float gcol0 = d_rgb_d_color * gr, gcol1 = d_rgb_d_color * gg, gcol2 = d_rgb_d_color * gb;
                    float one_m = 1.0f - alpha;
                    float ag0 = (col[0] * T_i - w0 / one_m) * gr;   /* RAST:653-654 */
                    float ag1 = (col[1] * T_i - w1 / one_m) * gg;
                    float ag2 = (col[2] * T_i - w2 / one_m) * gb;
                    const float w0_before = w0, w1_before = w1, w2_before = w2;
                    w0 += col[0] * alpha * T_i; w1 += col[1] * alpha * T_i; w2 += col[2] * alpha * T_i;
                    float alpha_grad = ag0 + ag1 + ag2;
                    float pa_grad = alpha_grad * gaussian_alpha;
                    float opacity_grad = pa_grad * (1.0f - apt) * apt;   /* RAST:659-661 */
                    float g_alpha_grad = alpha_grad * apt;
                    float vs0 = g_alpha_grad * dpm0, vs1 = g_alpha_grad * dpm1;
                    tot0 += fabsf(vs0); tot1 += fabsf(vs1);
                    float cg00 = g_alpha_grad * dpc00, cg01 = g_alpha_grad * dpc01, cg11 = g_alpha_grad * dpc11;
                    float mg = sqrtf(vs0 * vs0 + vs1 * vs1);   /* RAST:691-694 */
#define GSO_ADD(dst, v) _Pragma("omp atomic") dst += (v)
                    GSO_ADD(a_uv[2 * (size_t)p], vs0); GSO_ADD(a_uv[2 * (size_t)p + 1], vs1);
                    GSO_ADD(a_cov[3 * (size_t)p], cg00); GSO_ADD(a_cov[3 * (size_t)p + 1], cg01); GSO_ADD(a_cov[3 * (size_t)p + 2], cg11);
                    GSO_ADD(a_col[3 * (size_t)p], gcol0); GSO_ADD(a_col[3 * (size_t)p + 1], gcol1); GSO_ADD(a_col[3 * (size_t)p + 2], gcol2);
                    GSO_ADD(a_alpha[p], opacity_grad);
                    GSO_ADD(a_mag[p], mg);
                    GSO_ADD(a_cnt[p], 1);
                    if (a_abs) {
                        /* the same expressions with every product replaced by its absolute value, all the way down:
                         * a contribution is itself a sum that can cancel (colour*T against w/(1-alpha) in alpha_grad,
                         * a*dx against b*dy in Sigma^-1 d, the terms of Sigma^-1 d d^T Sigma^-1) */
                        double* ab = a_abs + 9 * (size_t)p;
                        double Aag = ((double)fabsf(col[0] * T_i) + fabsf(w0_before / one_m)) * fabsf(gr)
                                   + ((double)fabsf(col[1] * T_i) + fabsf(w1_before / one_m)) * fabsf(gg)
                                   + ((double)fabsf(col[2] * T_i) + fabsf(w2_before / one_m)) * fabsf(gb);
                        double Ag = Aag * fabsf(apt) * fabsf(gaussian_alpha);          /* |g_alpha_grad| * p, un-cancelled */
                        double fa = fabsf(a), fb = fabsf(b), fc = fabsf(c), adx = fabsf(dx), ady = fabsf(dy);
                        double Acix = fa * adx + fb * ady, Aciy = fb * adx + fc * ady;
                        double Aio00 = fa * adx * adx + fb * ady * adx, Aio01 = fa * adx * ady + fb * ady * ady;
                        double Aio10 = fb * adx * adx + fc * ady * adx, Aio11 = fb * adx * ady + fc * ady * ady;
                        GSO_ADD(ab[0], Ag * Acix); GSO_ADD(ab[1], Ag * Aciy);
                        GSO_ADD(ab[2], 0.5 * Ag * (Aio00 * fa + Aio01 * fb));
                        GSO_ADD(ab[3], 0.5 * Ag * (Aio00 * fb + Aio01 * fc));
                        GSO_ADD(ab[4], 0.5 * Ag * (Aio10 * fb + Aio11 * fc));
                        GSO_ADD(ab[5], fabsf(gcol0)); GSO_ADD(ab[6], fabsf(gcol1)); GSO_ADD(ab[7], fabsf(gcol2));
                        GSO_ADD(ab[8], Aag * fabsf(gaussian_alpha) * fabsf((1.0f - apt) * apt));
                    }
#undef GSO_ADD
                }
            }
            mag_img[2 * o] = tot0; mag_img[2 * o + 1] = tot1;    /* RAST:700-704 */
        }
    }
}

/* the accumulators as one float row per in-camera point -- the tensors the reference hands from loop 1 to loop 2
 * (grad_uv, the cov / colour buffers, opacity gradient, magnitude, pixel count), rounded to f32 exactly where
 * RAST:716-721 reads them: d uv (2) | d cov xx xy yy (3) | d colour (3) | d opacity | sum |d uv| | count (i32 bits) | 0 */
static void pack_sums(int64_t M, gso_acc acc, float* sums)
{
    for (int64_t i = 0; i < M; ++i) {
        float* r = sums + 12 * i;
        r[0] = (float)acc.a_uv[2 * i]; r[1] = (float)acc.a_uv[2 * i + 1];
        r[2] = (float)acc.a_cov[3 * i]; r[3] = (float)acc.a_cov[3 * i + 1]; r[4] = (float)acc.a_cov[3 * i + 2];
        r[5] = (float)acc.a_col[3 * i]; r[6] = (float)acc.a_col[3 * i + 1]; r[7] = (float)acc.a_col[3 * i + 2];
        r[8] = (float)acc.a_alpha[i]; r[9] = (float)acc.a_mag[i];
        int32_t cnt = (int32_t)acc.a_cnt[i];
        memcpy(r + 10, &cnt, sizeof cnt);
        r[11] = 0.0f;
    }
}

static gso_acc acc_alloc(int64_t M, int with_abs)
{
    gso_acc a;
    a.a_uv = (double*)zalloc(sizeof(double) * 2 * (size_t)M);
    a.a_cov = (double*)zalloc(sizeof(double) * 3 * (size_t)M);
    a.a_col = (double*)zalloc(sizeof(double) * 3 * (size_t)M);
    a.a_alpha = (double*)zalloc(sizeof(double) * (size_t)M);
    a.a_mag = (double*)zalloc(sizeof(double) * (size_t)M);
    a.a_cnt = (int64_t*)zalloc(sizeof(int64_t) * (size_t)M);
    a.a_abs = with_abs ? (double*)zalloc(sizeof(double) * 9 * (size_t)M) : NULL;   /* sums of |products|: uv 0..1, cov 2..4, colour 5..7, opacity 8 */
    return a;
}

static void acc_free(gso_acc a) { free(a.a_uv); free(a.a_cov); free(a.a_col); free(a.a_alpha); free(a.a_mag); free(a.a_cnt); free(a.a_abs); }

/* loop 2 (RAST:708-772) + masking/scaling (RAST:1102-1125) from the packed rows.  `f` supplies the per-point arrays of
 * the shard whose rows these are (ids, point_in_camera, camera pose); N rows of gradients are fully written. */
static void backward_loop2(const gso_frame* f, const float* pc, const float* feat, const int32_t* obj, const float* t_pc,
                           const float* Km, const float* sums, const double* a_abs, int32_t sh_band, const gso_config* cfg,
                           float* g_pc, float* g_feat, float* g_uv, float* mag, int32_t* n_affected,
                           float* out_cov_buf, float* out_color_buf, float* summed_pc, float* summed_feat)
{
    const int64_t N = f->N, M = f->M;
    memset(g_pc, 0, sizeof(float) * 3 * (size_t)N);           /* RAST:1051-1058 */
    memset(g_feat, 0, sizeof(float) * GSO_FEAT * (size_t)N);
    memset(g_uv, 0, sizeof(float) * 2 * (size_t)N);
    memset(mag, 0, sizeof(float) * (size_t)N);
    memset(n_affected, 0, sizeof(int32_t) * (size_t)M);
    if (summed_pc) memset(summed_pc, 0, sizeof(float) * 3 * (size_t)N);
    if (summed_feat) memset(summed_feat, 0, sizeof(float) * GSO_FEAT * (size_t)N);
    /* ---- loop 2, RAST:708-772, then masking/scaling RAST:1102-1125 ---- */
    int keep = sh_band <= 0 ? 1 : sh_band == 1 ? 4 : sh_band == 2 ? 9 : 16;  /* RAST:1167-1182 */
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < M; ++idx) {
        int32_t pid = f->point_id_in_camera_list[idx];
        const float* row = feat + (size_t)GSO_FEAT * pid;
        const float* xyz = pc + 3 * (size_t)pid;
        const float* row12 = sums + 12 * (size_t)idx;
        float guv[2] = { row12[0], row12[1] };
        float gc0 = row12[2], gc1 = row12[3], gc2 = row12[4];
        float gcov[4] = { gc0, gc1, gc1, gc2 };                 /* RAST:716-721 */
        float gcol[3] = { row12[5], row12[6], row12[7] };
        float T[16];
        transform_from_qt(f->q_camera_pointcloud + 4 * obj[pid], f->t_camera_pointcloud + 3 * obj[pid], T);
        const float* ray_origin = t_pc + 3 * obj[pid];          /* RAST:731-732 */
        float Jpos[6], dSq[16], dSs[12];
        gso_project_to_camera_position_jacobian(xyz, T, Km, Jpos);
        gso_project_to_camera_covariance_jacobian(row, row + 4, T, Km, f->point_in_camera + 3 * idx, dSq, dSs);
        float dir[3] = { xyz[0] - ray_origin[0], xyz[1] - ray_origin[1], xyz[2] - ray_origin[2] };
        float sh[16];
        gso_spherical_harmonics(dir, sh);
        float jac[3];
        for (int ch = 0; ch < 3; ++ch) {                        /* GP3D:351-373 */
            float s = sigmoidf_(dot16(row + 8 + 16 * ch, sh));
            jac[ch] = s * (1.0f - s);
        }
        float gt[3], gq[4], gs[3];
        matmul(guv, Jpos, gt, 1, 2, 3);                         /* RAST:757 */
        matmul(gcov, dSq, gq, 1, 4, 4);                         /* RAST:760 */
        matmul(gcov, dSs, gs, 1, 4, 3);                         /* RAST:761 */
        float* gp = g_pc + 3 * (size_t)pid;
        float* gf = g_feat + (size_t)GSO_FEAT * pid;
        gp[0] = gt[0]; gp[1] = gt[1]; gp[2] = gt[2];
        for (int i = 0; i < 4; ++i) gf[i] = gq[i] * cfg->grad_q_factor;
        for (int i = 0; i < 3; ++i) gf[4 + i] = gs[i] * cfg->grad_s_factor;
        gf[7] = row12[8] * cfg->grad_alpha_factor;
        for (int ch = 0; ch < 3; ++ch)
            for (int i = 0; i < 16; ++i) {
                float v = gcol[ch] * (jac[ch] * sh[i]);          /* RAST:754-756, GP3D:368-370 */
                if (i >= keep) v = 0.0f;
                else v = v * (i == 0 ? cfg->grad_color_factor : cfg->grad_high_order_color_factor);
                gf[8 + 16 * ch + i] = v;
            }
        if (a_abs) {
            const double* ab = a_abs + 9 * (size_t)idx;
            const double acov[4] = { ab[2], ab[3], ab[3], ab[4] };
            if (summed_pc)
                for (int j = 0; j < 3; ++j) summed_pc[3 * (size_t)pid + j] = (float)(ab[0] * fabsf(Jpos[j]) + ab[1] * fabsf(Jpos[3 + j]));
            if (summed_feat) {
                float* sf = summed_feat + (size_t)GSO_FEAT * pid;
                for (int k = 0; k < 4; ++k) {
                    double v = 0.0;
                    for (int i = 0; i < 4; ++i) v += acov[i] * fabsf(dSq[4 * i + k]);
                    sf[k] = (float)(v * fabsf(cfg->grad_q_factor));
                }
                for (int k = 0; k < 3; ++k) {
                    double v = 0.0;
                    for (int i = 0; i < 4; ++i) v += acov[i] * fabsf(dSs[3 * i + k]);
                    sf[4 + k] = (float)(v * fabsf(cfg->grad_s_factor));
                }
                sf[7] = (float)(ab[8] * fabsf(cfg->grad_alpha_factor));
                for (int ch = 0; ch < 3; ++ch)
                    for (int i = 0; i < 16; ++i)
                        sf[8 + 16 * ch + i] = i >= keep ? 0.0f
                            : (float)(ab[5 + ch] * fabsf(jac[ch] * sh[i]) * fabsf(i == 0 ? cfg->grad_color_factor : cfg->grad_high_order_color_factor));
            }
        }
        g_uv[2 * (size_t)pid] = guv[0]; g_uv[2 * (size_t)pid + 1] = guv[1];
        mag[pid] = row12[9];
        { int32_t cnt; memcpy(&cnt, row12 + 10, sizeof cnt); n_affected[idx] = cnt; }
        if (out_cov_buf) { out_cov_buf[3 * idx] = gc0; out_cov_buf[3 * idx + 1] = gc1; out_cov_buf[3 * idx + 2] = gc2; }
        if (out_color_buf) { out_color_buf[3 * idx] = gcol[0]; out_color_buf[3 * idx + 1] = gcol[1]; out_color_buf[3 * idx + 2] = gcol[2]; }
    }
}

int gso_backward_ex(const gso_frame* f, const float* pc, const float* feat, const int32_t* obj,
                    const float* q_pc, const float* t_pc, const float* Km,
                    const float* grad_image, int32_t sh_band, const gso_config* cfg,
                    float* g_pc, float* g_feat, float* g_uv, float* mag, float* mag_img,
                    int32_t* n_affected, float* out_cov_buf, float* out_color_buf,
                    float* summed_pc, float* summed_feat)
{
    (void)q_pc;
    const int64_t M = f->M;
    gso_acc acc = acc_alloc(M, summed_pc || summed_feat);
    backward_loop1(f, grad_image, mag_img, acc, cfg->bwd_strict_dpdcov);
    float* sums = (float*)zalloc(sizeof(float) * 12 * (size_t)M);
    pack_sums(M, acc, sums);
    backward_loop2(f, pc, feat, obj, t_pc, Km, sums, acc.a_abs, sh_band, cfg, g_pc, g_feat, g_uv, mag, n_affected,
                   out_cov_buf, out_color_buf, summed_pc, summed_feat);
    free(sums);
    acc_free(acc);
    return 0;
}

/* ---- the two halves on their own (stand-ins for gs_backward_projected / gs_backward_shard) ---- */
int gso_backward_sums(const gso_frame* f, const float* grad_image, float* sums, float* mag_img)
{
    gso_acc acc = acc_alloc(f->M, 0);
    backward_loop1(f, grad_image, mag_img, acc, 1);
    pack_sums(f->M, acc, sums);
    acc_free(acc);
    return 0;
}

int gso_backward_points(const gso_frame* f, const float* pc, const float* feat, const int32_t* obj, const float* t_pc,
                        const float* Km, const float* sums, int32_t sh_band, const gso_config* cfg,
                        float* g_pc, float* g_feat, float* g_uv, float* mag, int32_t* n_affected)
{
    backward_loop2(f, pc, feat, obj, t_pc, Km, sums, NULL, sh_band, cfg, g_pc, g_feat, g_uv, mag, n_affected, NULL, NULL, NULL, NULL);
    return 0;
}

