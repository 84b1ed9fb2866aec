// k_project.hip -- per-point stages of the forward:
//   k_filter         filter_point_in_camera, RAST:31-78, with the host prologue RAST:841-846 (inverse_SE3_qt_torch,
//                    UTIL:426-432) folded in: every block derives the pose records itself
//   k_scan_tiles_publish   hand-over of M, K and the depth-code range to the host when it has to wait for them (exact sizing);
//                    with predicted sizing the last block of k_keygen does it and this launch does not exist
//   k_project        point_id[mask] (RAST:861-870, ascending ids) + generate_point_attributes_in_camera_plane RAST:239-315
//                    + generate_num_overlap_tiles RAST:106-128 in one kernel
// All HBM-bound streaming kernels; layouts in DESIGN.md.
#include "gs_common.h"

#ifndef GS_RADIUS_FROM_PREBLUR_COV
#define GS_RADIUS_FROM_PREBLUR_COV 1
#endif

// ---------------------------------------------------------------------------------
__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4])
{   // UTIL:403-413
    float x0 = a[0], y0 = a[1], z0 = a[2], w0 = a[3];
    float x1 = b[0], y1 = b[1], z1 = b[2], w1 = b[3];
    o[0] = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
    o[1] = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
    o[2] = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
    o[3] = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
}

__device__ __forceinline__ void rotation_from_quaternion(const float q[4], float R[9])
{   // GP3D:30-48
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.0f - 2.0f * (yy + zz); R[1] = 2.0f * (xy - wz);        R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);        R[4] = 1.0f - 2.0f * (xx + zz); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);        R[7] = 2.0f * (yz + wx);        R[8] = 1.0f - 2.0f * (xx + yy);
}

// Pose record of one object: inverse_SE3_qt_torch (UTIL:426-432) + rotation_matrix_from_quaternion (GP3D:30-48)
// + the camera centre as taichi_inverse_SE3 computes it (UTIL:495-510).
__device__ __forceinline__ GsPose make_pose(const float* __restrict__ q_pc, const float* __restrict__ t_pc, int i)
{
    float qc[4] = { -q_pc[4 * i], -q_pc[4 * i + 1], -q_pc[4 * i + 2], q_pc[4 * i + 3] };
    float nrm = sqrtf(qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] + qc[3] * qc[3]);
    float qn[4] = { qc[0] / nrm, qc[1] / nrm, qc[2] / nrm, qc[3] / nrm };
    float v4[4] = { t_pc[3 * i], t_pc[3 * i + 1], t_pc[3 * i + 2], 0.0f };
    float qn_conj[4] = { -qn[0], -qn[1], -qn[2], qn[3] };
    float tmp[4], rot[4];
    quat_mul(qn, v4, tmp);
    quat_mul(tmp, qn_conj, rot);
    GsPose p;
    rotation_from_quaternion(qc, p.R);
    p.t[0] = -rot[0]; p.t[1] = -rot[1]; p.t[2] = -rot[2];
    float RTn[9] = { -p.R[0], -p.R[3], -p.R[6], -p.R[1], -p.R[4], -p.R[7], -p.R[2], -p.R[5], -p.R[8] };
    gs_mm<3, 3, 1>(RTn, p.t, p.origin_fwd);
    p.origin_bwd[0] = t_pc[3 * i]; p.origin_bwd[1] = t_pc[3 * i + 1]; p.origin_bwd[2] = t_pc[3 * i + 2];
    p.q_cp[0] = qc[0]; p.q_cp[1] = qc[1]; p.q_cp[2] = qc[2]; p.q_cp[3] = qc[3];
    p.pad[0] = p.pad[1] = 0.0f;
    return p;
}

// project_point_to_camera, GP3D:14-27 (T = [R|t; 0 0 0 1])
__device__ __forceinline__ void project_point(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ Km,
                                              float x, float y, float z, float uv[2], float pc[3])
{
    pc[0] = ((R[0] * x + R[1] * y) + R[2] * z) + t[0] * 1.0f;
    pc[1] = ((R[3] * x + R[4] * y) + R[5] * z) + t[1] * 1.0f;
    pc[2] = ((R[6] * x + R[7] * y) + R[8] * z) + t[2] * 1.0f;
    float u1 = (Km[0] * pc[0] + Km[1] * pc[1]) + Km[2] * pc[2];
    float v1 = (Km[3] * pc[0] + Km[4] * pc[1]) + Km[5] * pc[2];
    uv[0] = u1 / pc[2];
    uv[1] = v1 / pc[2];
}

// ---------------------------------------------------------------------------------
#define FILTER_POSE_CACHE 8
__global__ __launch_bounds__(256) void k_filter(const float* __restrict__ pc, const int8_t* __restrict__ invalid,
                                                const int32_t* __restrict__ obj, const float* __restrict__ Kmat,
                                                const float* __restrict__ q_pc, const float* __restrict__ t_pc, int n_objects,
                                                GsPose* __restrict__ pose, GsCounters* __restrict__ counters,
                                                int64_t N, int W, int H, float near_plane, float far_plane,
                                                int8_t* __restrict__ mask, int32_t* __restrict__ block_counts,
                                                int32_t* __restrict__ tile_arrays, int tile_ints)
{
    __shared__ int wave_cnt[4];
    __shared__ GsPose sp[FILTER_POSE_CACHE];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // the frame's first kernel also clears tile_start | tile_end (RAST:954-957 zero-init) | tile_work for the binning and the blend
    for (int64_t k = i; k < tile_ints; k += (int64_t)gridDim.x * 256) tile_arrays[k] = 0;
    // every block derives the (few) pose records itself -- same arithmetic as the stored ones -- so that no separate
    // pose launch has to finish first; block 0 stores them for the later kernels
    const bool cached = n_objects <= FILTER_POSE_CACHE;
    if (cached && threadIdx.x < n_objects) sp[threadIdx.x] = make_pose(q_pc, t_pc, threadIdx.x);
    if (blockIdx.x == 0) {            // (the frame counters were reset when the previous frame's were published)
        for (int o = threadIdx.x; o < n_objects; o += 256) pose[o] = make_pose(q_pc, t_pc, o);
    }
    __syncthreads();
    bool in = false;
    // an object id outside [0, n_objects) would index past the pose rows: such a row is left out and counted, and
    // gs_forward reports it (the reference is unchecked here and reads whatever lies behind the array)
    const int oid = (i < N && invalid[i] != 1) ? obj[i] : 0;
    const bool bad_id = oid < 0 || oid >= n_objects;
    if (bad_id) atomicAdd(&counters->bad_object_ids, 1);
    if (i < N && invalid[i] != 1 && !bad_id) {
        const GsPose P = cached ? sp[oid] : make_pose(q_pc, t_pc, oid);
        float Km[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Km[k] = Kmat[k];
        float uv[2], p3[3];
        project_point(P.R, P.t, Km, pc[3 * i], pc[3 * i + 1], pc[3 * i + 2], uv, p3);
        float z = p3[2];
        in = z > near_plane && z < far_plane &&
             uv[0] >= (float)(-GS_TILE_SZ * GS_BOUNDARY_TILES) && uv[0] < (float)(W + GS_TILE_SZ * GS_BOUNDARY_TILES) &&
             uv[1] >= (float)(-GS_TILE_SZ * GS_BOUNDARY_TILES) && uv[1] < (float)(H + GS_TILE_SZ * GS_BOUNDARY_TILES);
    }
    if (i < N) mask[i] = in ? 1 : 0;
    unsigned long long b = gs_ballot(in);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// ---------------------------------------------------------------------------------
// Compaction (point_id[mask], RAST:861-870, ascending ids) fused with the projection: a block takes 256 consecutive
// rows of the point cloud, sums the in-camera counts of the blocks before it (a few thousand L2-resident ints, instead
// of waiting for a scan launch), compacts its own in-camera rows through LDS and then projects them with its first
// `cnt` threads -- dense within the block, and the records of block b start at in-camera offset block_offsets[b].
__global__ __launch_bounds__(256) void k_project(const float* __restrict__ pc, float* __restrict__ feat,
                                                 const int32_t* __restrict__ obj, const float* __restrict__ Kmat,
                                                 const GsPose* __restrict__ pose, const int8_t* __restrict__ mask,
                                                 const int32_t* __restrict__ block_counts, int64_t N,
                                                 int32_t* __restrict__ ids, int32_t* __restrict__ cam_index,
                                                 int32_t* __restrict__ block_offsets,
                                                 int W, int H, float depth_scale,
                                                 float4* __restrict__ PA, float4* __restrict__ PB, float4* __restrict__ PC,
                                                 float4* __restrict__ PD, ushort4* __restrict__ boxes,
                                                 int32_t* __restrict__ ntiles, uint32_t* __restrict__ tile_block_sums,
                                                 GsCounters* counters, int32_t* __restrict__ depth_codes, int32_t* __restrict__ max_tiles_hint)
{
    __shared__ int wave_sum[4];
    __shared__ int wave_max[4];
    __shared__ int wave_maxn[4];
    // the four float4 of a record leave through LDS: a lane-per-record store writes 16 bytes out of every 64 per
    // instruction; staged, each of the wave's four store instructions writes 1 KB of consecutive bytes
    __shared__ float4 sOut[4][4 * 64];
    __shared__ int sIds[256];
    __shared__ int wave_cnt[4];
    __shared__ int wave_pre[4];
    int block_offset, cnt;
    {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const bool in = i < N && mask[i] != 0;
        const unsigned long long b = gs_ballot(in);
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        int pre = 0;
        for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) pre += block_counts[j];
        pre = gs_wave_sum_i(pre);
        if (lane_ == 0) { wave_cnt[wave_] = __popcll(b); wave_pre[wave_] = pre; }
        __syncthreads();
        block_offset = wave_pre[0] + wave_pre[1] + wave_pre[2] + wave_pre[3];
        cnt = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        int woff = 0;
        for (int w = 0; w < wave_; ++w) woff += wave_cnt[w];
        const int r = woff + __popcll(b & ((1ull << lane_) - 1ull));
        if (in) { sIds[r] = (int)i; ids[block_offset + r] = (int32_t)i; }
        if (i < N) cam_index[i] = in ? block_offset + r : -1;
        if (threadIdx.x == 0) {
            block_offsets[blockIdx.x] = block_offset;
            if (blockIdx.x == gridDim.x - 1) counters->M = block_offset + cnt;
        }
        __syncthreads();
    }
    const int idx = block_offset + threadIdx.x;                      // in-camera offset of this thread's point
    int count = 0, depth_code = 0;
    float4 recA = make_float4(0.f, 0.f, 0.f, 0.f), recB = recA, recC = recA, recD = recA;
    if ((int)threadIdx.x < cnt) {
        int pid = sIds[threadIdx.x];
        float4* row4 = reinterpret_cast<float4*>(feat + (size_t)GS_NFEAT * pid);
        float row[GS_NFEAT];
#pragma unroll
        for (int k = 0; k < GS_NFEAT / 4; ++k) {
            float4 v = row4[k];
            row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
        }
        // RAST:196-205 normalise the rotation in place
        {
            float n = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2] + row[3] * row[3]);
            const float q0 = row[0] / n, q1 = row[1] / n, q2 = row[2] / n, q3 = row[3] / n;
            // written back only when the division changed a bit: a static scene (inference, a frozen parameter) is
            // already normalised and costs no partial-line writes into the caller's rows
            const bool changed = __float_as_int(q0) != __float_as_int(row[0]) || __float_as_int(q1) != __float_as_int(row[1]) ||
                                 __float_as_int(q2) != __float_as_int(row[2]) || __float_as_int(q3) != __float_as_int(row[3]);
            row[0] = q0; row[1] = q1; row[2] = q2; row[3] = q3;
            if (changed) row4[0] = make_float4(q0, q1, q2, q3);
        }
        const GsPose& P = pose[obj[pid]];
        float Km[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Km[k] = Kmat[k];
        float x = pc[3 * (size_t)pid], y = pc[3 * (size_t)pid + 1], z = pc[3 * (size_t)pid + 2];
        float uv[2], pcam[3];
        project_point(P.R, P.t, Km, x, y, z, uv, pcam);
        // ---- project_to_camera_covariance, GP3D:161-191 (same product chain as the Python) ----
        float fx = Km[0], fy = Km[4];
        float J[6] = { fx / pcam[2], 0.0f, -(fx * pcam[0]) / (pcam[2] * pcam[2]),
                       0.0f, fy / pcam[2], -(fy * pcam[1]) / (pcam[2] * pcam[2]) };
        float R[9];
        rotation_from_quaternion(row, R);
        float es0 = gs_expf(row[4]), es1 = gs_expf(row[5]), es2 = gs_expf(row[6]);
        float S[9] = { es0, 0.0f, 0.0f, 0.0f, es1, 0.0f, 0.0f, 0.0f, es2 };   // S == S^T
        float Rt[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
        float RS[9], RSS[9], Sigma[9];
        gs_mm<3, 3, 3>(R, S, RS);
        gs_mm<3, 3, 3>(RS, S, RSS);
        gs_mm<3, 3, 3>(RSS, Rt, Sigma);
        float Wt[9] = { P.R[0], P.R[3], P.R[6], P.R[1], P.R[4], P.R[7], P.R[2], P.R[5], P.R[8] };
        float Jt[6] = { J[0], J[3], J[1], J[4], J[2], J[5] };
        float Wm[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Wm[k] = P.R[k];
        float JW[6], JWS[6], JWSW[6], cov[4];
        gs_mm<2, 3, 3>(J, Wm, JW);
        gs_mm<2, 3, 3>(JW, Sigma, JWS);
        gs_mm<2, 3, 3>(JWS, Wt, JWSW);
        gs_mm<2, 3, 2>(JWSW, Jt, cov);
        // ---- get_point_conic_and_rescale, UTIL:257-272 ----
        float c00 = cov[0], c01 = cov[1], c10 = cov[2], c11 = cov[3];
        float det_pre = c00 * c11 - c01 * c10;
        float b00 = c00 + 0.3f, b11 = c11 + 0.3f;
        float det = b00 * b11 - c01 * c10;
        float ratio = det_pre / det;
        float rescale = sqrtf(0.0f > ratio ? 0.0f : ratio);
        float inv_det = 1.0f / det;
        float conic_a = inv_det * b11, conic_b = inv_det * (-c01), conic_c = inv_det * b00;
        float alpha = 1.0f / (1.0f + gs_expf(-row[7]));                      // RAST:299-300
        // ---- colour, GP3D:333-349 + SH:10-53 ----
        float dx = x - P.origin_fwd[0], dy = y - P.origin_fwd[1], dz = z - P.origin_fwd[2];
        float dn = sqrtf(dx * dx + dy * dy + dz * dz);
        float sx = dx / dn, sy = dy / dn, sz = dz / dn;
        float sh[16];
        sh[0] = 0.28209479177387814f;
        sh[1] = -0.48860251190291987f * sy;
        sh[2] = 0.48860251190291987f * sz;
        sh[3] = -0.48860251190291987f * sx;
        sh[4] = 1.0925484305920792f * sx * sy;
        sh[5] = -1.0925484305920792f * sy * sz;
        sh[6] = 0.94617469575755997f * sz * sz - 0.31539156525251999f;
        sh[7] = -1.0925484305920792f * sx * sz;
        sh[8] = 0.54627421529603959f * sx * sx - 0.54627421529603959f * sy * sy;
        sh[9] = 0.59004358992664352f * sy * (-3.0f * sx * sx + sy * sy);
        sh[10] = 2.8906114426405538f * sx * sy * sz;
        sh[11] = 0.45704579946446572f * sy * (1.0f - 5.0f * sz * sz);
        sh[12] = 0.3731763325901154f * sz * (5.0f * sz * sz - 3.0f);
        sh[13] = 0.45704579946446572f * sx * (1.0f - 5.0f * sz * sz);
        sh[14] = 1.4453057213202769f * sz * (sx * sx - sy * sy);
        sh[15] = 0.59004358992664352f * sx * (-sx * sx + 3.0f * sy * sy);
        float col[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* f = row + 8 + 16 * ch;
            float acc = f[0] * sh[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) acc = acc + f[k] * sh[k];
            col[ch] = gs_sigmoid(acc);
        }
        // ---- radius, RAST:311-315.  Taichi passes the mat2 to get_point_conic_and_rescale by value, so the radius
        // sees the covariance BEFORE the +0.3 blur.  The switch mirrors the oracle's
        // radius_from_preblur_cov in case a comparison against real Taichi output ever says otherwise. ----
#if GS_RADIUS_FROM_PREBLUR_COV
        const float e00 = c00, e11 = c11;
#else
        const float e00 = b00, e11 = b11;
#endif
        float large_eigen = (e00 + e11 + sqrtf((e00 - e11) * (e00 - e11) + 4.0f * c01 * c10)) / 2.0f;
        float radii = sqrtf(large_eigen) * 3.0f;
        // ---- tile box + count, RAST:81-128 ----
        int box[4];
        gs_tile_box(uv[0], uv[1], radii, (W + GS_TILE_SZ - 1) / GS_TILE_SZ, (H + GS_TILE_SZ - 1) / GS_TILE_SZ, box);   // = W/16, H/16 at the reference's sizes
        count = (box[1] - box[0]) * (box[3] - box[2]);
        depth_code = (int)(pcam[2] * depth_scale);                            // RAST:159-160
        // Conservative log-domain cut for the blend kernels: alpha = exp(e)*rescale*opacity < 1/255
        // whenever e < cut (margins are applied where it is used).  Not an index-determining value.
        float ra = rescale * alpha;
        float cut = ra > 0.0f ? __logf(GS_ALPHA_EPS / ra) : (ra == 0.0f ? 3.0e38f : -3.0e38f);
        recA = make_float4(uv[0], uv[1], conic_a, conic_b);
        recB = make_float4(conic_c, rescale, alpha, pcam[2]);
        recC = make_float4(col[0], col[1], col[2], cut);
        recD = make_float4(pcam[0], pcam[1], pcam[2], radii);
        boxes[idx] = make_ushort4((unsigned short)box[0], (unsigned short)box[1], (unsigned short)box[2], (unsigned short)box[3]);
        ntiles[idx] = count;
        depth_codes[idx] = depth_code;          // for the key build: 4 bytes instead of a 64-byte record row per point
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if GS_RS == 4
    {
        float4* mine = sOut[wave];
        mine[lane] = recA; mine[64 + lane] = recB; mine[128 + lane] = recC; mine[192 + lane] = recD;
        __builtin_amdgcn_wave_barrier();
        const int wave_first = block_offset + wave * 64;                  // first record of this wave
        const int n_rec = cnt - wave * 64 < 64 ? cnt - wave * 64 : 64;    // wave-uniform; <= 0 past the block's last in-camera point
        float4* dst = PA + (size_t)wave_first * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = k * 64 + lane;                                  // float4 e of the wave's 4 KB: record e / 4, quarter e % 4
            if ((e >> 2) < n_rec) dst[e] = mine[(e & 3) * 64 + (e >> 2)];
        }
    }
#else
    if ((int)threadIdx.x < cnt) { GS_REC(PA, idx) = recA; GS_REC(PB, idx) = recB; GS_REC(PC, idx) = recC; GS_REC(PD, idx) = recD; }
#endif
    int s = gs_wave_sum_i(count);
    int mx = gs_wave_max_i(depth_code);
    int mn = gs_wave_max_i(count);
    if (lane == 0) { wave_sum[wave] = s; wave_max[wave] = mx; wave_maxn[wave] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_block_sums[blockIdx.x] = (uint32_t)(wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3]);
        int m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
        if (m > 0) atomicMax(&counters->max_depth_code, m);
        // the frame's largest tile count of one point (a word of the tile arrays, cleared by k_filter): the backward's row sum looks for
        // points with thousands of rows only in a frame that has any (k_backward.hip: SUM_ROWS_GIANT)
        const int mt = max(max(wave_maxn[0], wave_maxn[1]), max(wave_maxn[2], wave_maxn[3]));
        if (max_tiles_hint && mt > 64) atomicMax(max_tiles_hint, mt);
    }
}


// ---------------------------------------------------------------------------------
// gs_forward_projected: the records arrive from elsewhere (another rank's k_project); what the binning needs besides
// them -- tile box, tile count, per-block count sums, depth-code range -- is recomputed from u, v, radius and depth with
// the same expressions k_project uses (RAST:81-128, 159-160), so the result is the one k_project would have stored.
__global__ __launch_bounds__(256) void k_boxes_from_records(const float4* __restrict__ PA, const float4* __restrict__ PB,
                                                            const float4* __restrict__ PD, int M, int W, int H, float depth_scale,
                                                            ushort4* __restrict__ boxes, int32_t* __restrict__ ntiles,
                                                            uint32_t* __restrict__ tile_block_sums, GsCounters* counters,
                                                            int32_t* __restrict__ depth_codes, int32_t* __restrict__ tile_arrays, int tile_ints)
{
    __shared__ int wave_sum[4];
    __shared__ int wave_max[4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    for (int k = idx; k < tile_ints; k += (int)gridDim.x * 256) tile_arrays[k] = 0;       // as k_filter does
    int count = 0, depth_code = 0;
    if (idx < M) {
        const float4 A = GS_REC(PA, idx);
        int box[4];
        gs_tile_box(A.x, A.y, GS_REC(PD, idx).w, (W + GS_TILE_SZ - 1) / GS_TILE_SZ, (H + GS_TILE_SZ - 1) / GS_TILE_SZ, box);
        count = (box[1] - box[0]) * (box[3] - box[2]);
        depth_code = (int)(GS_REC(PB, idx).w * depth_scale);
        boxes[idx] = make_ushort4((unsigned short)box[0], (unsigned short)box[1], (unsigned short)box[2], (unsigned short)box[3]);
        ntiles[idx] = count;
        depth_codes[idx] = depth_code;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sm = gs_wave_sum_i(count), mx = gs_wave_max_i(depth_code);
    if (lane == 0) { wave_sum[wave] = sm; wave_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_block_sums[blockIdx.x] = (uint32_t)(wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3]);
        const int m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
        if (m > 0) atomicMax(&counters->max_depth_code, m);
        if (blockIdx.x == 0) counters->M = M;
    }
}

// ---------------------------------------------------------------------------------
// Hand-over of the frame counters to the host: M, K (= the sum of the per-block tile counts), the depth-code range and the
// bad-object-id count go straight into pinned host memory, followed by the ticket the host waits for (spinning on it costs
// a few microseconds where a copy + stream synchronisation left the GPU idle for tens).  Called by ONE thread that already
// holds K.  Two places do it: k_scan_tiles_publish when the host needs the counters before it can queue the binning (exact
// sizing), and the last block of k_keygen when the binning was queued on predicted sizes (gs_api.hip: run_forward_tail) --
// then the frame has one launch fewer.
__global__ __launch_bounds__(1024) void k_scan_tiles_publish(const uint32_t* __restrict__ in, int n, GsCounters* __restrict__ counters,
                                                             volatile GsCounters* host_mirror, int32_t ticket)
{
    __shared__ uint32_t wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = 0;
    for (int i = threadIdx.x; i < n; i += 1024) v += in[i];
    v = (uint32_t)gs_wave_sum_i((int)v);
    if (lane == 0) wave_tot[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t K = 0;
        for (int w = 0; w < 16; ++w) K += wave_tot[w];
        gs_publish_counters(counters, K, host_mirror, ticket);
    }
}

void gs_launch_publish(const GsProjectArgs& a, int n_blocks, hipStream_t s)
{
    if (n_blocks <= 0) return;
    GS_TIMED(a.prof, KID_PUBLISH, s, k_scan_tiles_publish<<<1, 1024, 0, s>>>(a.tile_block_sums, n_blocks, a.counters, a.host_mirror, a.ticket));
}

void gs_launch_project(const GsProjectArgs& a, hipStream_t s, bool publish)
{
    const int nb = (int)((a.N + 255) / 256);
    if (nb == 0) {
        (void)hipMemsetAsync(a.tile_arrays, 0, sizeof(int32_t) * (size_t)a.tile_ints, s);     // empty scene: nothing else clears them
        return;
    }
    GS_TIMED(a.prof, KID_FILTER, s, k_filter<<<nb, 256, 0, s>>>(a.point_cloud, a.invalid, a.object_id, a.Kmat, a.q_pc, a.t_pc, a.n_objects,
                                                            a.pose, a.counters, a.N, a.W, a.H, a.near_plane, a.far_plane, a.mask,
                                                            a.block_counts, a.tile_arrays, a.tile_ints));
    GS_TIMED(a.prof, KID_PROJECT, s, k_project<<<nb, 256, 0, s>>>(a.point_cloud, a.features, a.object_id, a.Kmat, a.pose, a.mask, a.block_counts, a.N,
                                                              a.ids, a.cam_index, a.block_offsets, a.W, a.H,
                                                              a.depth_scale, a.PA, a.PB, a.PC, a.PD, a.box, a.ntiles,
                                                              a.tile_block_sums, a.counters, a.depth_codes,
                                                              a.tile_arrays ? a.tile_arrays + a.tile_ints - GS_TILE_SPARE_MAX_TILES : nullptr));
    if (publish) gs_launch_publish(a, nb, s);
}

void gs_launch_boxes_from_records(const GsProjectArgs& a, int M, hipStream_t s, bool publish)
{
    const int nb = (M + 255) / 256;
    if (nb == 0) {
        (void)hipMemsetAsync(a.tile_arrays, 0, sizeof(int32_t) * (size_t)a.tile_ints, s);
        return;
    }
    GS_TIMED(a.prof, KID_PROJECT, s, k_boxes_from_records<<<nb, 256, 0, s>>>(a.PA, a.PB, a.PD, M, a.W, a.H, a.depth_scale, a.box, a.ntiles,
                                                                          a.tile_block_sums, a.counters, a.depth_codes, a.tile_arrays, a.tile_ints));
    if (publish) gs_launch_publish(a, nb, s);
}
