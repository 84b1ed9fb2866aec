// k_blend_fwd.hip -- front-to-back alpha blend, gaussian_point_rasterisation RAST:318-485.
//
// CDNA4 mapping (not the reference's 256-thread/one-barrier-per-batch scheme):
//   * one workgroup (256 threads = 4 waves) per 16x16 tile, but each WAVE owns an 8x8
//     pixel quadrant and walks the tile's sorted list on its own: no __syncthreads at all,
//     so a quadrant that saturates stops while its neighbours continue;
//   * per 64-entry batch every lane fetches one splat record (3 x float4, gathered by the
//     sorted index) and tests it against the wave's 8x8 rectangle with a conservative
//     bound on the Gaussian exponent; a 64-bit ballot keeps only splats that can reach
//     alpha >= 1/255 somewhere in the quadrant, and the wave iterates the set bits;
//   * survivors are re-read from a wave-private LDS slab with broadcast ds_read_b128.
// Culling is invisible in the results: a culled splat would have been skipped by the
// alpha < 1/255 test for every pixel of the quadrant (RAST:451-452).
// Bound: FP32 VALU (exp polynomial + blend), not HBM: see DESIGN.md.
#include "gs_common.h"
#include "gs_cull.h"
#ifndef GS_FWD_INT_MIN
#define GS_FWD_INT_MIN 1
#endif
#ifndef GS_WORK_PER_BATCH
#define GS_WORK_PER_BATCH 8      // a walked batch in units of (splat, quadrant) evaluations, for the tile ordering / heavy-tile choice
#endif

// find_tile_start_and_end (RAST:175-193) without a launch of its own: the first index of the sorted, compact keys
// (tile << depth_bits | depth code) whose tile is >= `tile`, by a 64-ary search -- every step the wave's 64 lanes probe
// 64 positions of the remaining span and a ballot narrows it 64-fold (four or five dependent loads for millions of keys).
template <typename KeyT>
__device__ __forceinline__ int gs_first_key_of_tile(const KeyT* __restrict__ keys, uint32_t K, int depth_bits, uint32_t tile, int lane)
{
    uint32_t lo = 0, hi = K;                       // answer in [lo, hi]; every index < lo is below the tile, index hi (if < K) is not
    while (lo < hi) {
        const uint32_t span = hi - lo, step = (span + 63u) / 64u;
        const uint32_t mine = (uint32_t)(lane + 1) * step;
        const uint32_t p = lo + (mine < span ? mine : span) - 1u;
        const unsigned long long b = gs_ballot((uint32_t)(keys[p] >> depth_bits) >= tile);
        if (b == 0ull) { lo = hi; break; }
        const uint32_t f = (uint32_t)__builtin_ctzll(b);
        const uint32_t pf = lo + ((f + 1u) * step < span ? (f + 1u) * step : span) - 1u;
        if (f > 0u) lo = lo + (f * step < span ? f * step : span);      // = p_(f-1) + 1
        hi = pf;
    }
    return (int)lo;
}

template <bool RGB_ONLY>
__global__ __launch_bounds__(256) void k_blend_fwd(int32_t* __restrict__ tile_start, int32_t* __restrict__ tile_end,
                                                   const void* __restrict__ sorted_keys, int key64, int depth_bits,
                                                   const GsCounters* __restrict__ ctr, uint32_t K_cap,
                                                   const int32_t* __restrict__ sorted_vals,
                                                   const float4* __restrict__ PA, const float4* __restrict__ PB,
                                                   const float4* __restrict__ PC, int W, int H, int tiles_x,
                                                   float* __restrict__ image, float* __restrict__ depth_out,
                                                   float* __restrict__ acc_alpha, int32_t* __restrict__ last_out,
                                                   int32_t* __restrict__ count_out, int32_t* __restrict__ tile_work,
                                                   const int32_t* __restrict__ order_hint,
                                                   float4* __restrict__ cuts, int32_t* __restrict__ tile_cut, int32_t* __restrict__ cut_alloc, int cut_cap)
{
    __shared__ float4 sRec[4][64][3];          // the batch's splat records, one slab per wave
    const uint32_t K = min(ctr->K, K_cap);     // pairs of this frame, read on the device (gs_api.hip: predicted sizing)
    // Dispatch order: heaviest tiles first when an earlier frame of this context left its ordering (same tile count) -- the
    // launch otherwise ends on the long walks of the image's dense region, started late.  Any permutation gives the same results.
    const int tile = order_hint ? order_hint[blockIdx.x] : (int)blockIdx.x;
    // the wave index as a scalar: LDS addresses of the wave's slab are then SGPR arithmetic + one v_mov instead of a 64-bit VALU mad per splat
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int tile_u = tile % tiles_x, tile_v = tile / tiles_x;
    const int qx = tile_u * 16 + (wave & 1) * 8, qy = tile_v * 16 + (wave >> 1) * 8;
    const int pixel_u = qx + (lane & 7), pixel_v = qy + (lane >> 3);
    const float px = (float)pixel_u + 0.5f, py = (float)pixel_v + 0.5f;
    const float rx0 = (float)qx + 0.5f, ry0 = (float)qy + 0.5f;
    // the tile's range in the sorted list: wave 0 finds where the tile starts, wave 1 where the next one does; an empty
    // tile keeps the reference's zero-initialised 0, 0 (RAST:954-957).  The ranges are stored for the backward.
    __shared__ int sRange[2];
    if (wave < 2) {
        const uint32_t want = (uint32_t)tile + (uint32_t)wave;
        const int r = K == 0u ? 0 : (key64 ? gs_first_key_of_tile(reinterpret_cast<const uint64_t*>(sorted_keys), K, depth_bits, want, lane)
                                           : gs_first_key_of_tile(reinterpret_cast<const uint32_t*>(sorted_keys), K, depth_bits, want, lane));
        if (lane == 0) sRange[wave] = r;
    }
    __syncthreads();
    const bool occupied = sRange[1] > sRange[0];
    const int start = occupied ? sRange[0] : 0, end = occupied ? sRange[1] : 0;
    if (threadIdx.x == 0) { tile_start[tile] = start; tile_end[tile] = end; }
    // A LONG list (more than GS_CUT_MIN_LEN entries) is blended in SEGMENTS of GS_SEG entries: at every cut the colour gathered
    // since the last cut is added to a running total and starts again from zero -- in every mode, so that the image does not
    // depend on whether a backward is wanted (T, and with it every index, is untouched; a short list has no cut and its colour is
    // the plain left-to-right sum as before).  When the frame is kept for a backward the cut is also RECORDED: each pixel's T at
    // the cut and the colour of the segment that just ended (16 bytes), once more at the end of the walk -- the backward can
    // then start anywhere in the list (k_backward.hip: heavy tiles are handed out segment by segment), and what it needs there,
    // the colour BEHIND the cut, is a sum of later segments' colours: small terms added to small terms, as in the reference's own
    // back-to-front accumulation (a difference of front sums would lose exactly the digits a dim pixel lives on).  A list's
    // records sit at start / GS_SEG + tile: the lists are disjoint ranges of the sorted pairs, so the next tile's first record
    // lies at least (length / GS_SEG) + 1 further on -- room for all of this one's, without an atomic claim (and the barrier
    // behind it) at the head of every long list; K / GS_SEG + T + 1 records hold them all.
    const bool long_list = end - start > GS_CUT_MIN_LEN;                               // (workgroup-uniform)
    int cut_base = -1;
    if (!RGB_ONLY && cut_cap > 0 && long_list) {
        const int n_rec = (end - start - 1) / GS_SEG + 1;
        const int b = start / GS_SEG + tile;
        if (b + n_rec <= cut_cap) cut_base = b;
        if (threadIdx.x == 0) tile_cut[tile] = cut_base + 1;
    }
    int next_rec = 0;                                                                  // cut records this wave has written
    float tot_r = 0.0f, tot_g = 0.0f, tot_b = 0.0f;                                    // colour of the segments behind the last cut
#ifdef GS_STATS
    const unsigned long long gs_t0 = wall_clock64();
#endif

    float T_i = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, acc_d = 0.0f, norm = 0.0f;
    int last = start, count = 0;
    int evals = 0;                 // work the backward will repeat for this quadrant (wave-uniform): GS_WORK_PER_BATCH per 64-entry batch
                                   // walked (index load, record gather, cull: latency a lone wave pays in full) + 1 per splat that survived the cull
    // a pixel of a partial edge tile that lies outside the image does not exist (extension; W,H % 16 == 0 in the reference)
    const bool inside = pixel_u < W && pixel_v < H;
    // Lane predicates live as wave-uniform 64-bit masks in SGPRs (votes fold into the v_cmp that made them, and
    // inverse_ballot turns a mask back into exec without VALU work): `alive` = lanes that have not saturated yet.
    unsigned long long alive = gs_ballot(inside);

    for (int base = start; base < end; base += 64) {
        if (alive == 0ull) break;
        if (long_list && base > start && ((base - start) & (GS_SEG - 1)) == 0) {             // a cut: the state BEFORE entry `base`
            if (cut_base >= 0) { cuts[(size_t)(cut_base + next_rec) * 256 + threadIdx.x] = make_float4(T_i, cr, cg, cb); next_rec += 1; }
            tot_r += cr; tot_g += cg; tot_b += cb;
            cr = 0.0f; cg = 0.0f; cb = 0.0f;
        }
        evals += GS_WORK_PER_BATCH;
        const int i = base + lane;
        const bool valid = i < end;
        const int p = valid ? sorted_vals[i] : 0;
        float4 A = GS_REC(PA, p), B = GS_REC(PB, p), C = GS_REC(PC, p);
        const CullRect lr = gs_live_rect(alive, rx0, ry0);          // the pixels that have not saturated yet
        bool keep = valid && !gs_cull(gs_cull_prepare(A, B, C), lr.x0, lr.y0, lr.wx, lr.wy);
        unsigned long long mask = gs_ballot(keep);
        GS_STAT(0, 1); GS_STAT(1, __popcll(mask));
        if (mask == 0ull) continue;
        evals += __popcll(mask);
        sRec[wave][lane][0] = A; sRec[wave][lane][1] = B; sRec[wave][lane][2] = C;
        // The loop below takes the survivors two at a time.  An odd count is made even with a filler: the record of some lane
        // that did not survive is overwritten with a splat of opacity 0 (alpha = 0 fails the 1/255 test for every pixel), so the
        // loop needs no "is there a second one" logic.  (64 survivors are even already; an odd count leaves a lane free.)
        if (__popcll(mask) & 1) {
            const int jz = __builtin_ctzll(~mask);
            if (lane == jz) {
                sRec[wave][lane][0] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
                sRec[wave][lane][1] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
                sRec[wave][lane][2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            mask |= 1ull << jz;
        }
        __builtin_amdgcn_wave_barrier();
        // Two splats per trip: their alphas are independent (two dependency chains to interleave: a quadrant's walk is
        // otherwise one long chain of dependent instructions, and the end of the launch runs with few waves per SIMD
        // to hide it behind), the two blend steps then run in list order.  Same operations per splat as one at a time.
        int lj = -1;                                // batch position of this pixel's latest contributor, if any
        while (mask) {
            // (s_bitset0_b64 clears the bit in one scalar instruction; `mask &= mask - 1` is a 64-bit subtract and an and: three)
            const int j0 = __builtin_ctzll(mask);
            asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(j0));
            const int j1 = __builtin_ctzll(mask);                               // (an even number of bits: see the filler above)
            asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(j1));
            const float4 a40 = sRec[wave][j0][0], b40 = sRec[wave][j0][1], c40 = sRec[wave][j0][2];
            const float4 a41 = sRec[wave][j1][0], b41 = sRec[wave][j1][1], c41 = sRec[wave][j1][2];
            // get_point_probability_density_from_conic_and_rescale, UTIL:275-284 (same op order)
            const float dx0 = px - a40.x, dy0 = py - a40.y, dx1 = px - a41.x, dy1 = py - a41.y;
            const float e0 = -0.5f * (dx0 * dx0 * a40.z + dy0 * dy0 * b40.x) - dx0 * dy0 * a40.w;
            const float e1 = -0.5f * (dx1 * dx1 * a41.z + dy1 * dy1 * b41.x) - dx1 * dy1 * a41.w;
            float alpha0 = gs_exp_blend(e0) * b40.y * b40.z;
            float alpha1 = gs_exp_blend(e1) * b41.y * b41.z;
            // both alphas are complete HERE: without this the compiler sinks the second chain below the first blend step
            // (next to its only use) and the two run one after the other
            asm volatile("" : "+v"(alpha0), "+v"(alpha1));
#define GS_FWD_STEP(ALPHA, A4, B4, C4, J, LIVE)                                                                                        \
            {                                                                                                                    \
                GS_STAT(2, 1);                                                                                                   \
                float alpha = (ALPHA);                                                                                           \
                unsigned long long use_m = gs_ballot(!(alpha < GS_ALPHA_EPS)) & (LIVE);   /* RAST:451 */                          \
                /* min(alpha, 0.99), RAST:453, as the UNSIGNED minimum of the bit patterns: one instruction (the float form costs a  */ \
                /* canonicalising v_max first).  Same value for every alpha >= 0; a NaN of either sign gives 0.99 like the compare- */ \
                /* and-select of the oracle; a negative alpha (never blended: it fails the 1/255 test) may come out as 0.99         */ \
                alpha = GS_FWD_INT_MIN ? __uint_as_float(min(__float_as_uint(alpha), __float_as_uint(GS_ALPHA_MAX)))                \
                                       : (alpha < GS_ALPHA_MAX ? alpha : GS_ALPHA_MAX);                                             \
                const float next_T = T_i * (1.0f - alpha);                                /* RAST:457 */                          \
                const unsigned long long sat_m = gs_ballot(next_T < GS_T_STOP) & use_m;   /* RAST:458-460 */                      \
                alive &= ~sat_m;                                                                                                 \
                use_m &= ~sat_m;                                                                                                 \
                GS_STAT(3, __popcll(use_m)); GS_STAT(5, __popcll(alive));                                                        \
                if (__builtin_amdgcn_inverse_ballot_w64(use_m)) {                                                                \
                    lj = (J);                                  /* RAST:461: position in the batch; the base is added once per batch */ \
                    /* alpha and T (and with them every index the forward returns) follow the reference operation sequence   */ \
                    /* bit for bit; the weighted sums use one shared weight and fused multiply-adds (float outputs, 1e-4 bar) */ \
                    const float w = alpha * T_i;                                                                                 \
                    cr = __builtin_fmaf((C4).x, w, cr); cg = __builtin_fmaf((C4).y, w, cg); cb = __builtin_fmaf((C4).z, w, cb);  /* RAST:462 */ \
                    if (!RGB_ONLY) { acc_d = __builtin_fmaf((B4).w, w, acc_d); norm += w; count += 1; }   /* RAST:464-469 */      \
                    T_i = next_T;                                                                                                \
                }                                                                                                                \
            }
            GS_FWD_STEP(alpha0, a40, b40, c40, j0, alive)
            // no branch around the second step (a missing second splat just has no live lanes): its alpha is then needed
            // unconditionally and the compiler keeps the two alpha chains in one block, where it interleaves them
            GS_FWD_STEP(alpha1, a41, b41, c41, j1, alive)
#undef GS_FWD_STEP
            if (alive == 0ull) break;
        }
        last = lj >= 0 ? base + lj + 1 : last;
        __builtin_amdgcn_wave_barrier();
    }
#ifdef GS_STATS
    if (lane == 0 && blockIdx.x * 4 + wave < 65536) { gs_stats_wave_times[2 * (blockIdx.x * 4 + wave)] = gs_t0; gs_stats_wave_times[2 * (blockIdx.x * 4 + wave) + 1] = wall_clock64(); }
#endif
    if (cut_base >= 0) {
        // a wave that stopped early (every pixel saturated) leaves the cuts it never reached as empty segments; the colour gathered
        // since its last cut belongs to the last record, which also carries the final T
        const int n_rec = (end - start - 1) / GS_SEG + 1;
        for (int k = next_rec; k < n_rec - 1; ++k) cuts[(size_t)(cut_base + k) * 256 + threadIdx.x] = make_float4(T_i, 0.0f, 0.0f, 0.0f);
        cuts[(size_t)(cut_base + n_rec - 1) * 256 + threadIdx.x] = make_float4(T_i, cr, cg, cb);
    }
    cr += tot_r; cg += tot_g; cb += tot_b;          // (exact for a list without cuts: the totals are zero)
    if (!inside) return;
    const size_t o = (size_t)pixel_v * (size_t)W + (size_t)pixel_u;
    image[3 * o] = cr; image[3 * o + 1] = cg; image[3 * o + 2] = cb;
    if (!RGB_ONLY) {
        depth_out[o] = acc_d / (norm > 1e-6f ? norm : 1e-6f);                  // RAST:479-480
        acc_alpha[o] = 1.0f - T_i;
        last_out[o] = last;
        count_out[o] = count;
        // work of this tile in (splat, quadrant) evaluations, batches included: the backward repeats it (same cull, same stop), so it
        // predicts the backward's cost per tile far better than the list length does.  Scheduling hint only.
        if (lane == 0 && evals > 0) atomicAdd(&tile_work[tile], evals);
    }
}

void gs_launch_blend_fwd(const GsBlendFwdArgs& a, hipStream_t s)
{
    if (a.T <= 0) return;
    if (a.rgb_only)
        GS_TIMED(a.prof, KID_BLEND_FWD, s, k_blend_fwd<true><<<a.T, 256, 0, s>>>(a.tile_start, a.tile_end, a.keys_sorted, a.key64, a.depth_bits, a.counters, a.K, a.vals_sorted, a.PA, a.PB, a.PC, a.W, a.H,
                                                                             a.tiles_x, a.image, a.depth, a.acc_alpha, a.last, a.count, a.tile_work, a.order_hint, a.cuts, a.tile_cut, a.cut_alloc, a.cut_cap));
    else
        GS_TIMED(a.prof, KID_BLEND_FWD, s, k_blend_fwd<false><<<a.T, 256, 0, s>>>(a.tile_start, a.tile_end, a.keys_sorted, a.key64, a.depth_bits, a.counters, a.K, a.vals_sorted, a.PA, a.PB, a.PC, a.W, a.H,
                                                                              a.tiles_x, a.image, a.depth, a.acc_alpha, a.last, a.count, a.tile_work, a.order_hint, a.cuts, a.tile_cut, a.cut_alloc, a.cut_cap));
}
