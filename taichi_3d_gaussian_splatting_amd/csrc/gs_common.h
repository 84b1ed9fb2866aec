// gs_common.h -- shared declarations of libgsrast (HIP, gfx950 only).
//
// Arithmetic contract: every expression that decides an INDEX (frustum test, tile
// box, depth code, the 1/255 and 1e-4 blend thresholds) is evaluated with the same
// IEEE f32 operation sequence on every build of this library: the sources are
// compiled with -ffp-contract=off, division and sqrt are correctly rounded (hipcc
// default), and exp is gs_expf below (add/mul/fma only).  See DESIGN.md "Numerics".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_TILE_SZ 16

// Wave-wide vote.  The intrinsic folds into the v_cmp that produced the predicate (an SGPR-pair mask); HIP's __ballot()
// goes through v_cndmask + v_cmp_ne, two half-rate VALU instructions per vote.
#define gs_ballot(pred) __builtin_amdgcn_ballot_w64(pred)

// Diagnostic builds only, never loaded by the product:
//   `make stats` (-DGS_STATS=2) -> libgsrast_stats.so: event counters of the two blend kernels (tools/blend_stats.py) + wave times;
//   `make times` (-DGS_STATS=1) -> libgsrast_times.so: start / end time of every blend wave ONLY (tools/*_wave_timeline.py).
// The counters are global atomics in the inner loops (~100 ns each on this part): a build that counts runs tens of times
// slower and its wave times say nothing about the real launch, hence the separate timing-only build.
#ifdef GS_STATS
extern __device__ unsigned long long gs_stats_counters[32];
extern __device__ unsigned long long gs_stats_wave_times[2 * 65536];   // start, end (wall clock ticks) of each blend wave
#endif
#if defined(GS_STATS) && GS_STATS >= 2
#define GS_STAT(i, n) do { const unsigned long long gs_stat_n = (unsigned long long)(n); if (threadIdx.x % 64 == 0) atomicAdd(&gs_stats_counters[i], gs_stat_n); } while (0)
#else
#define GS_STAT(i, n) do { } while (0)
#endif
// Projected splat records: four float4 per in-camera point -- A {u, v, conic a, conic b}, B {conic c, rescale, opacity, depth},
// C {r, g, b, log-domain alpha cut}, D {x, y, z in camera, radius} -- kept as ONE 64-byte row per point (GS_RS = 4: the
// pointers PA..PD are rec, rec+1, rec+2, rec+3), so that the blend kernels' gather by sorted index touches one
// 64-byte segment per splat instead of three cache lines, and so that a shard's records are one contiguous
// (M,16) float array for the Gaussian-parallel exchange.  GS_RS = 1 selects four separate planes (A/B measurement).
#ifndef GS_RS
#define GS_RS 4
#endif
#define GS_REC(ptr, i) (ptr)[(size_t)(i) * GS_RS]
#define GS_BOUNDARY_TILES 3          // reference RAST:26
#define GS_ALPHA_EPS 0.00392156862745098f   // 1./255. (RAST:451, RAST:634)
#define GS_ALPHA_MAX 0.99f           // RAST:453
#define GS_T_STOP 0.0001f            // RAST:458
#define GS_NFEAT 56
// Cuts of long tile lists (k_blend_fwd writes them, the backward blend of a HEAVY tile starts its segments from them): every
// GS_SEG entries of a list longer than GS_CUT_MIN_LEN the forward stores each pixel's transmittance and accumulated colour.
// Every list longer than one segment is cut: a dense list of 600 - 1000 entries, left whole, is twice the longest segment and was
// what the backward of the clustered 976x544 workload waited for (0.224 -> 0.179 ms; 1024 was the threshold while cutting cost
// an atomic claim and a barrier at the head of the list -- k_blend_fwd.hip).  Segments of 256 entries: no further gain (0.183).
#ifndef GS_SEG
#define GS_SEG 512
#endif
#ifndef GS_CUT_MIN_LEN
#define GS_CUT_MIN_LEN 512
#endif
#define GS_HEAVY_CAP 1024            // most tiles a backward treats as heavy
// tile arrays of a frame, cleared by its first kernel: tile_start | tile_end | tile_work | tile_cut (first cut record + 1, 0 = none) | cut_alloc
#define GS_TILE_INTS(T) (4 * (size_t)(T) + 4)
// the four trailing ints: one unused (the cut records' claim counter until their positions became a closed form), then the largest tile count of one point of the frame (k_project -> k_sum_rows), two spare
#define GS_TILE_SPARE_MAX_TILES 3          // index from the END of the tile arrays
// tile_order buffer: order (T) | n_heavy | n_items | pad pad | item_base (GS_HEAVY_CAP + 1)
#define GS_ORDER_INTS(T) ((size_t)(T) + 4 + GS_HEAVY_CAP + 4 + GS_HEAVY_CAP)
#define GS_ORDER_REDO_OFFSET (4 + GS_HEAVY_CAP + 4)      // from n_heavy: one flag per heavy tile, "walk this tile again in one piece" (k_backward.hip)

// Per-object pose record built once per frame (every k_filter block derives it, block 0 stores it).
struct GsPose {
    float R[9];          // rotation_matrix_from_quaternion(conj(q_pointcloud_camera)), GP3D:30-48
    float t[3];          // t_camera_pointcloud, UTIL:426-432
    float origin_fwd[3]; // camera centre as forward computes it: taichi_inverse_SE3, RAST:280-282
    float origin_bwd[3]; // camera centre as backward reads it: t_pointcloud_camera, RAST:731-732
    float q_cp[4];
    float pad[2];
};

// Device-side frame counters, mirrored to pinned host memory once per forward.
struct GsCounters {
    int32_t M;               // points in camera
    uint32_t K;              // sort pairs
    int32_t max_depth_code;  // max over visible points of i32(depth * scale)
    int32_t reserved;        // host mirror only: ticket of the forward that published these values
    int32_t bad_object_ids;  // valid rows whose point_object_id is outside [0, n_objects): treated as not in camera, reported
    int32_t pad[3];
};

// One thread hands the frame counters to the host (see k_project.hip: k_scan_tiles_publish, k_binning.hip: k_keygen).
__device__ __forceinline__ void gs_publish_counters(GsCounters* __restrict__ counters, uint32_t K, volatile GsCounters* host_mirror, int32_t ticket)
{
    counters->K = K;
    host_mirror->M = counters->M;
    host_mirror->K = K;
    host_mirror->max_depth_code = counters->max_depth_code;
    host_mirror->bad_object_ids = counters->bad_object_ids;
    __threadfence_system();
    host_mirror->reserved = ticket;              // the host waits for this value
    __threadfence_system();
    // the accumulating counters start the next frame at zero (zero at gs_create for the first one): no clearing
    // launch, and no block of the next k_filter / k_project can run ahead of a clear
    counters->max_depth_code = 0;
    counters->bad_object_ids = 0;
}

// ---- device math -------------------------------------------------------------
__device__ __forceinline__ float gs_expf(float x)
{
    x = x < -86.0f ? -86.0f : x;                     // the oracle's two compares, so that a NaN parameter stays a NaN as in the reference
    x = x > 88.0f ? 88.0f : x;                       // (v_med3_f32 / v_min / v_max would return the finite operand)
    float fx = x * 1.44269504088896341f;
    float n = (fx + 12582912.0f) - 12582912.0f;      // round to nearest even
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    float y = __builtin_fmaf(p, z, r);
    y = y + 1.0f;
    int32_t bits = __float_as_int(y) + (((int32_t)n) << 23);
    return __int_as_float(bits);
}

// exp of the Gaussian falloff in the blend kernels (oracle: gso_exp_blend, same sequence bit for bit): eleven VALU
// instructions where gs_expf takes seventeen.  2^(x log2 e): integer part through the 1.5*2^23 constant, fraction with one
// fused multiply-add (the rounding of x*log2(e) does not enter), degree-5 polynomial for 2^f on [-0.5, 0.5], the integer
// added to the exponent field (v_lshl_add_u32).  Within 3e-7 of exp for x in [-10, 0].
__device__ __forceinline__ float gs_exp_blend(float x)
{
    x = __builtin_amdgcn_fmed3f(x, -86.0f, 88.0f);   // (a NaN exponent -- NaN position, rotation or scale -- becomes -86: such a splat has no
                                                     // defined tile box in the reference either; NaN colour and opacity do propagate)
    const float L = 1.44269504088896341f;
    const float t = x * L;
    const float m = t + 12582912.0f;
    const float n = m - 12582912.0f;
    const float f = __builtin_fmaf(x, L, -n);
    float q = 0.0013264712179079652f;
    q = __builtin_fmaf(q, f, 0.009671511128544807f);
    q = __builtin_fmaf(q, f, 0.05550733581185341f);
    q = __builtin_fmaf(q, f, 0.24022242426872253f);
    q = __builtin_fmaf(q, f, 0.6931470036506653f);
    const float p = __builtin_fmaf(q, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(m) << 23));
}

__device__ __forceinline__ float gs_sigmoid(float x) { return 1.0f / (1.0f + gs_expf(-x)); }

// C[r x c] = A[r x k] @ B[k x c]; terms summed k = 0,1,2,... like the Python matmul chain.
template <int R, int K, int C>
__device__ __forceinline__ void gs_mm(const float* A, const float* B, float* Cout)
{
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float acc = A[i * K] * B[j];
#pragma unroll
            for (int t = 1; t < K; ++t) acc = acc + A[i * K + t] * B[t * C + j];
            Cout[i * C + j] = acc;
        }
}

// Tile box of a splat, RAST:81-103.  box = {min_tile_u, max_tile_u, min_tile_v, max_tile_v}
__device__ __forceinline__ void gs_tile_box(float u, float v, float radii, int tiles_u, int tiles_v, int box[4])
{
    radii = radii > 1.0f ? radii : 1.0f;
    float min_u = u - radii; min_u = min_u > 0.0f ? min_u : 0.0f;
    float max_u = u + radii;
    float min_v = v - radii; min_v = min_v > 0.0f ? min_v : 0.0f;
    float max_v = v + radii;
    int a = (int)floorf(min_u / 16.0f); a = a < tiles_u ? a : tiles_u;
    int b = (int)floorf(max_u / 16.0f) + 1; b = b > a + 1 ? b : a + 1; b = b < tiles_u ? b : tiles_u;
    int c = (int)floorf(min_v / 16.0f); c = c < tiles_v ? c : tiles_v;
    int d = (int)floorf(max_v / 16.0f) + 1; d = d > c + 1 ? d : c + 1; d = d < tiles_v ? d : tiles_v;
    box[0] = a; box[1] = b; box[2] = c; box[3] = d;
}

// wave64 helpers -----------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float gs_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

// Sum of eleven values over the 64 lanes (totals valid in lanes 48..63, all of row 3): six DPP steps
// (quad_perm x2, row_half_mirror, row_mirror, row_bcast:15, row_bcast:31), written as v_add_f32_dpp so that every step is ONE
// instruction per value (hipcc lowers the builtin form to v_mov_dpp + add, and to three
// instructions for the row_bcast steps).  A DPP source written by the previous VALU instruction
// needs two wait states: each step starts with s_nop 1; inside a step the eleven registers
// are independent and eleven instructions apart from their next use.
#define GS_DPP11(ctrl)                                                                              \
    asm volatile("s_nop 1\n\t"                                                                      \
                 "v_add_f32_dpp %0, %0, %0 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %2, %2, %2 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %3, %3, %3 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %4, %4, %4 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %5, %5, %5 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %6, %6, %6 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %7, %7, %7 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %8, %8, %8 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %9, %9, %9 " ctrl "\n\t"                                            \
                 "v_add_f32_dpp %10, %10, %10 " ctrl "\n\t"                                         \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), \
                   "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]))

__device__ __forceinline__ void gs_wave_sum11_row3(float (&v)[11])
{
    GS_DPP11("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    GS_DPP11("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    GS_DPP11("row_half_mirror row_mask:0xf bank_mask:0xf");
    GS_DPP11("row_mirror row_mask:0xf bank_mask:0xf");
    GS_DPP11("row_bcast:15 row_mask:0xa bank_mask:0xf");     // rows 1,3 += lane 15 of rows 0,2; rows 0,2 untouched
    GS_DPP11("row_bcast:31 row_mask:0xc bank_mask:0xf");     // rows 2,3 += lane 31
    asm volatile("s_nop 1");
}

__device__ __forceinline__ int gs_wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int gs_wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int w = __shfl_xor(v, o, 64); v = v > w ? v : w; }
    return v;
}

// ---- optional per-kernel timing with HIP events on the launch stream -------------
// Kernel ids index the comma-separated list returned by gs_kernel_names().
enum GsKernelId { KID_FILTER = 0, KID_PUBLISH, KID_PROJECT, KID_KEYGEN,
                  KID_SORT_HIST, KID_SORT_ROWSCAN, KID_SORT_SCATTER, KID_BLEND_FWD,
                  KID_BLEND_BWD, KID_BWD_POINTS, KID_SUM_ROWS, KID_TILE_ORDER, KID_BLEND_BWD_REPAIR, KID_COUNT_ };
struct GsProf;
int gs_prof_begin(GsProf* p, int kid, hipStream_t s);     // returns a record index or -1
void gs_prof_end(GsProf* p, int rec, hipStream_t s);
#define GS_TIMED(prof, kid, stream, ...)                     \
    do {                                                     \
        int rec__ = gs_prof_begin((prof), (kid), (stream));  \
        __VA_ARGS__;                                         \
        gs_prof_end((prof), rec__, (stream));                \
    } while (0)

// ---- host-side launch wrappers (implemented in the k_*.hip files) --------------
struct GsProjectArgs {
    GsProf* prof;
    const float* point_cloud; float* features; const int8_t* invalid; const int32_t* object_id;
    int64_t N; const float* q_pc; const float* t_pc; int n_objects; const float* Kmat;
    int H, W; float near_plane, far_plane, depth_scale;
    GsPose* pose; int8_t* mask; int32_t* block_counts; int32_t* block_offsets; int32_t* ids; int32_t* cam_index;
    float4 *PA, *PB, *PC, *PD; ushort4* box; int32_t* ntiles; uint32_t* tile_block_sums;
    int32_t* depth_codes;                       // (M) i32(depth * scale) per in-camera point, for the key build
    GsCounters* counters;
    int32_t* tile_arrays; int tile_ints;        // tile_start | tile_end | tile_work, cleared before the binning
    GsCounters* host_mirror; int32_t ticket;    // pinned host copy of the counters; .reserved = ticket once they are valid
};
void gs_launch_project(const GsProjectArgs& a, hipStream_t s, bool publish);
void gs_launch_publish(const GsProjectArgs& a, int n_blocks, hipStream_t s);    // the hand-over of the counters as a launch of its own
// the per-pixel half started from records (gs_forward_projected): tile boxes, counts, block sums and the depth-code range
// from the records, then the same scan + publication as gs_launch_project
void gs_launch_boxes_from_records(const GsProjectArgs& a, int M, hipStream_t s, bool publish);

struct GsBinArgs {
    GsProf* prof;
    // M and K are BOUNDS here: the per-pixel half may be queued before the host has read the frame's counters (gs_api.hip,
    // "predicted sizing").  M bounds the in-camera offsets (N rows when unknown); K is the pair capacity the launch geometry and
    // the buffers were sized for -- every kernel works on min(counters->K, K) pairs, read on the device.
    int64_t N; int M; uint32_t K; const GsCounters* counters; int H, W, tiles_x; float depth_scale; int depth_bits; int key_bits;
    const float4 *PA, *PB; const ushort4* box; const int32_t* ntiles; const int32_t* depth_codes; const uint32_t* tile_block_sums;
    GsCounters* counters_rw; GsCounters* host_mirror; int32_t ticket;   // host_mirror != NULL: the last k_keygen block publishes the frame counters
    const int32_t *block_offsets, *block_counts;   // k_project's blocks (first in-camera offset, count); NULL: 256 consecutive records per block
    uint32_t* offsets;                          // (M) exclusive scan of ntiles, written by keygen
    void *keys_a, *keys_b; int32_t *vals_a, *vals_b;       // ping-pong (K); keys are u32, or u64 when key64
    int key64;                                             // depth bits + tile bits > 32
    uint32_t* hist;                             // (256 * sort_blocks) + scratch
    uint32_t* scan_tmp;                         // 256 digit totals of the current pass
    int32_t *tile_start, *tile_end; int T;
    void** keys_sorted; int32_t** vals_sorted;       // out: which of a/b holds the result
};
void gs_launch_binning(const GsBinArgs& a, hipStream_t s);
size_t gs_sort_hist_elems(uint32_t K);
size_t gs_scan_tmp_elems(size_t n);

struct GsBlendFwdArgs {
    GsProf* prof;
    int H, W, tiles_x, T; int rgb_only;
    int32_t *tile_start, *tile_end;      // written by the blend kernel itself (each tile's block finds its range in the sorted keys)
    const void* keys_sorted; int key64, depth_bits; uint32_t K; const GsCounters* counters;   // min(counters->K, K) pairs (see GsBinArgs)
    const int32_t* vals_sorted;
    const float4 *PA, *PB, *PC;
    float* image; float* depth; float* acc_alpha; int32_t* last; int32_t* count;
    int32_t* tile_work;            // (T) zeroed together with the tile ranges; max over the tile's pixels of last - start
    float4* cuts; int32_t* tile_cut; int32_t* cut_alloc; int cut_cap;    // list cuts for the backward (cut_cap == 0: none wanted)
    const int32_t* order_hint;     // (T) or NULL: a permutation of the tiles, heaviest first, from an earlier frame of this ctx (scheduling only)
};
void gs_launch_blend_fwd(const GsBlendFwdArgs& a, hipStream_t s);

struct GsBackwardArgs {
    GsProf* prof;
    int64_t N; int M; uint32_t K; int H, W, tiles_x, T;
    const int32_t *tile_start, *tile_end; const int32_t* vals_sorted;
    const int32_t* tile_work; int32_t* tile_order;   // scheduling: heaviest tiles first
    int32_t* order_hint;                             // (T) or NULL: a second copy of tile_order that outlives the frame (the next forward's dispatch order)
    const float4 *PA, *PB, *PC, *PD; const ushort4* box; const uint32_t* offsets; const int32_t* ntiles;
    const int32_t* ids; const int32_t* cam_index;
    const float* grad_image; const float* acc_alpha; const int32_t* last;
    int G;                          // waves per tile in k_blend_bwd_tile (1, 2 or 4) = rows of `partial` per (point, tile) pair
    int32_t* n_heavy;               // device: number of heavy tiles at the head of tile_order (k_tile_order -> k_blend_bwd_tile); n_items and item_base follow it
    const float4* cuts; float2* cut_mag; const int32_t* tile_cut;    // list cuts of the forward (NULL: none), per-segment |d uv| partial sums
    int item_cap;                   // work items of heavy tiles the launch has room for when there are cuts
    int heavy_factor_x2;            // a tile is heavy from this many half-means of work on; 0: no tile is (GS_BWD_SPLIT_HEAVY=0)
    int strict;                     // gs_config.bwd_reference_order: loop 1's UTIL:331-348 in the reference's own operation order
    float* partial;                 // (K*G,12) per (point,tile[,quadrant group]) sums in slot order
    uint8_t* visited;               // (K*G) == gen where the row of `partial` was written by this backward
    uint8_t gen;                    // this backward's tag (1..255): flags are never cleared per backward, a stale one just does not match
    uint8_t* touched;               // (M) == gen where some pixel took a contribution from the point
    const float4* zero_row;         // that row
    const int32_t* max_tiles_hint;  // device: the frame's largest tile count of one point (k_project), or NULL when nobody computed it
    float4* sums;                   // (M,3) per-point sums of the visited rows (count as int32 bits in [10])
    const float* point_cloud; const float* features; const int32_t* object_id; const float* Kmat; const GsPose* pose;
    int sh_band; float f_color, f_high, f_s, f_q, f_alpha;
    float* grad_pc; float* grad_feat; float* grad_uv; float* mag; float* mag_image; int32_t* n_affected;
    float* hook_gpc; float* hook_gfeat; float* hook_guv; float* hook_mag;
    int32_t* hook_ids; int32_t* hook_ntiles; float* hook_depth; float* hook_uv;
    // adaptive-controller accumulators (CTRL:114-141), all nullable together
    int32_t* c_num_in_camera; int32_t* c_num_pixels; float* c_vs_grad; float* c_vs_grad_avg; float* c_pos_grad; float* c_pos_grad_norm;
};
void gs_launch_backward_blend(const GsBackwardArgs& a, hipStream_t s);     // tile order, blend backward, per-splat sums -> a.sums
void gs_launch_backward_points(const GsBackwardArgs& a, hipStream_t s);    // a.sums -> every gradient / hook array

struct GsExportArgs { int what; int64_t N; int M; uint32_t K; int T; int depth_bits; int key64;
    const int32_t* ids; const float4 *PA, *PB, *PC, *PD; const int32_t* ntiles; const uint32_t* offsets;
    const void* keys_sorted; const int32_t* vals_sorted; const int32_t *tile_start, *tile_end; const int8_t* mask; void* dst; };
void gs_launch_export(const GsExportArgs& a, hipStream_t s);

// a (3,H,W) f32 image as the loss kernels see it: element (c, y, x) at p[c * sc + y * sy + x * sx] (strides in floats), optionally
// passed through torch.clamp(., 0, 1) on the fly
struct GsLossImage { const float* p; long long sc, sy, sx; int clamp; };
size_t gs_loss_maps_size(int H, int W);          // the three padded derivative maps the forward leaves for the backward
size_t gs_loss_partials_floats(int H, int W);      // per-block partial sums of the forward
void gs_launch_loss_forward(const GsLossImage& X, const GsLossImage& Y, int H, int W, float lambda, float* maps, float* partials, float* terms,
                            hipStream_t s);
void gs_launch_loss_backward(const GsLossImage& X, const GsLossImage& Y, int H, int W, float lambda, const float* maps, const float* upstream,
                             const GsLossImage& G, hipStream_t s);
void gs_launch_adam(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int64_t step, hipStream_t s);
void gs_launch_reg_value(const float* feat, const int8_t* invalid, int64_t N, float* workspace, float* out, hipStream_t s);
void gs_launch_reg_grad(const float* feat, const int8_t* invalid, int64_t N, const float* value_and_count, const float* upstream,
                        float* grad, hipStream_t s);
