// k_backward.hip -- gaussian_point_rasterisation_backward, RAST:488-772, plus the torch
// post-processing RAST:1102-1140, as three kernels (+ a tile ordering) and NO global atomics:
//
//   k_tile_order  tiles by decreasing work (scheduling only) and the number of HEAVY tiles at the head of that order.
//   k_blend_bwd_tile  loop 1 (RAST:531-705).  One wave per tile walks the tile's list back to
//                 front (a heavy tile: a workgroup of four waves, a quadrant each).  The 11 per-contribution quantities the
//                 reference sends to HBM with ti.atomic_add (RAST:674-696) are summed over the tile's 256 pixels in
//                 registers + an LDS transpose and stored ONCE per (point, tile) pair as a 12-float
//                 row of `partial`, at the pair's pre-sort slot (offsets[p] + position of the tile
//                 in p's tile box), so all rows of a point are contiguous; a byte of `visited`
//                 tags the rows this backward wrote.  Factors that are constant per splat (opacity,
//                 0.5, (1 - opacity) * opacity) are left out of the rows and applied once per point.
//   k_sum_rows    sums a point's visited rows in slot order (deterministic).
//   k_bwd_points  loop 2 (RAST:708-772) over all N rows: chains the Jacobians (GP3D:132-159, 237-331, 351-373),
//                 applies band masks and grad factors (RAST:1102-1125, 1167-1182) and writes
//                 every output row exactly once (zero for rows outside the frustum), including
//                 the BackwardValidPointHookInput gathers (RAST:1128-1140).
// k_blend_bwd_tile is VALU/latency bound, k_bwd_points HBM bound: see DESIGN.md.
#include "gs_common.h"
#include "gs_cull.h"
#include <cstdlib>

#define RED_STRIDE 68   // floats per value row of the LDS transpose (64 lanes + 4 pad: conflict-free b128 reads)
#define PW 12     // floats per partial row: vs0 vs1 | cov00 cov01 cov11 | col r g b | opacity | |vs| | count (int32 bits) | pad

// ---------------------------------------------------------------------------------
// Loop 1: ONE WAVE PER TILE, four pixels per lane (one per 8x8 quadrant).
//   * no __syncthreads, no cross-wave combine: the wave sums a splat's contributions over its
//     own 256 pixels (in-lane over the quadrants, then an LDS transpose) and twelve lanes store the
//     48-byte row with one instruction;
//   * the four quadrants give every lane four independent T/w recurrences to interleave;
//   * culling stays per quadrant: each lane tests its splat against the four 8x8 rectangles,
//     four ballots, and a quadrant body runs only for the (splat, quadrant) pairs that survive.
// Pairs that are never visited (beyond every pixel's last index -- about three quarters of a
// saturated tile's list -- or culled in all four quadrants) cost nothing: their rows are not
// written and their `visited` byte does not carry this backward's tag (the flags are tagged, never cleared per backward).
// Tiles in order of decreasing backward work (entries to walk), so that the heaviest tiles are
// dispatched first and the launch does not end on a few long-running waves.  Counting sort in one
// workgroup: 2048 bins of width ORDER_BIN_WIDTH (more work shares the first bin).  Pure scheduling: results do
// not depend on it.  Work = what the forward counted for the tile: 8 per walked batch + 1 per (splat, quadrant) evaluation.
#define ORDER_BINS 2048
#define ORDER_BIN_WIDTH 4
#define ORDER_PER 8
// HEAVY tiles: the first n_heavy entries of the order -- tiles whose work is at least heavy_factor_x2 / 2 times the mean (the host's
// choice: twice the mean) and at least HEAVY_MIN_WORK evaluations (at most T / 8 of them, HEAVY_CAP in all).  k_blend_bwd_tile gives each of them a whole
// workgroup (four cooperating waves, one per quadrant) instead of one wave: on a clustered scene a few tiles carry lists
// ten to twenty times the mean and their single waves ARE the launch (profiles/r03_*_wave_timeline*.txt); on the uniform
// generator (max / mean = 2) there are none.  Which tiles are heavy changes the ORDER in which a pair's contributions are added
// (quadrant sums, then their sum), hence the set is whole bins of the histogram: a deterministic function of the work counts.
#ifndef HEAVY_MIN_WORK
#define HEAVY_MIN_WORK 512       // (1024 until the lists over 512 entries were cut: cfg2_clustered backward 0.180 -> 0.164 ms, nothing elsewhere; 256: the same)
#endif
#define HEAVY_CAP GS_HEAVY_CAP
__host__ __device__ inline int gs_heavy_cap(int T) { return T / 8 < HEAVY_CAP ? T / 8 : HEAVY_CAP; }
// A heavy tile whose list the forward CUT (k_blend_fwd: every GS_SEG entries each pixel's T and the colour since the last cut) is handed out
// as one work item per segment: item_base[h] .. item_base[h + 1] are the items of heavy tile h (n_heavy_out[1] = their number,
// n_heavy_out[4 ..] = item_base).  A heavy tile without cuts is one item.
__global__ __launch_bounds__(1024) void k_tile_order(const int32_t* __restrict__ tile_work, int T, int32_t* __restrict__ order, int32_t* __restrict__ hint,
                                                     int32_t* __restrict__ n_heavy_out, int heavy_factor_x2,
                                                     const int32_t* __restrict__ tile_start, const int32_t* __restrict__ tile_end,
                                                     const int32_t* __restrict__ tile_cut, int item_cap)
{
    __shared__ uint32_t bins[ORDER_BINS];
    __shared__ int32_t sHeavy[HEAVY_CAP];
    __shared__ int32_t sNHeavy;
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long wtot[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < ORDER_BINS; i += 1024) bins[i] = 0;
    if (t < HEAVY_CAP) n_heavy_out[GS_ORDER_REDO_OFFSET + t] = 0;          // no heavy tile has asked to be walked again yet
    __syncthreads();
    unsigned long long total = 0ull;
    // ORDER_PER loads in flight per thread (the kernel is one workgroup deep in load latency, not in work)
    for (int base = 0; base < T; base += 1024 * ORDER_PER) {
        int work[ORDER_PER];
#pragma unroll
        for (int k = 0; k < ORDER_PER; ++k) { const int i = base + k * 1024 + t; work[k] = i < T ? tile_work[i] : -1; }
#pragma unroll
        for (int k = 0; k < ORDER_PER; ++k) {
            if (base + k * 1024 + t >= T) continue;
            total += (unsigned long long)(work[k] > 0 ? work[k] : 0);
            int w = work[k] / ORDER_BIN_WIDTH; w = w < 0 ? 0 : (w > ORDER_BINS - 1 ? ORDER_BINS - 1 : w);
            atomicAdd(&bins[ORDER_BINS - 1 - w], 1u);           // bin 0 = heaviest
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if (lane == 0) wtot[wave] = total;
    __syncthreads();
    // exclusive scan of the 2048 bins (2 per thread)
    uint32_t a = bins[2 * t], b = bins[2 * t + 1];
    uint32_t incl = a + b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const uint32_t excl = woff + incl - (a + b);
    __syncthreads();
    bins[2 * t] = excl; bins[2 * t + 1] = excl + a;
    __syncthreads();
    if (t == 0) {
        // tiles in bins of work >= thr: bin index < ORDER_BINS - thr / width; their number = the scanned start of the first bin below
        unsigned long long sum = 0ull;
        for (int w = 0; w < 16; ++w) sum += wtot[w];
        const unsigned long long mean = T > 0 ? sum / (unsigned long long)T : 0ull;
        unsigned long long thr = (unsigned long long)heavy_factor_x2 * mean / 2ull;
        if (thr < HEAVY_MIN_WORK) thr = HEAVY_MIN_WORK;
        unsigned long long bw = (thr + ORDER_BIN_WIDTH - 1) / ORDER_BIN_WIDTH;              // first bin (by work) that counts as heavy
        if (bw > ORDER_BINS - 1) bw = ORDER_BINS - 1;                                       // (everything beyond the last bin edge shares it)
        int n = 0;
        if (heavy_factor_x2 > 0) n = (int)bins[ORDER_BINS - (int)bw];                               // start of the heaviest light bin = number of heavier tiles
        // at most gs_heavy_cap(T) of them, and always WHOLE bins: which tiles of a bin come first in the order is up to the LDS atomics
        // below, and a heavy tile's sums are added up in another order than an ordinary one's -- the set must not depend on that
        const int cap = gs_heavy_cap(T);
        if (n > cap) {
            int lo = 0, hi = ORDER_BINS - (int)bw;                       // bins[] is the exclusive scan: bins[k] = tiles in bins < k
            while (lo < hi) { const int mid = (lo + hi + 1) / 2; if ((int)bins[mid] <= cap) lo = mid; else hi = mid - 1; }
            n = (int)bins[lo];
        }
        sNHeavy = n;
        *n_heavy_out = sNHeavy;
    }
    __syncthreads();
    for (int base = 0; base < T; base += 1024 * ORDER_PER) {
        int work[ORDER_PER];
#pragma unroll
        for (int k = 0; k < ORDER_PER; ++k) { const int i = base + k * 1024 + t; work[k] = i < T ? tile_work[i] : -1; }
#pragma unroll
        for (int k = 0; k < ORDER_PER; ++k) {
            const int i = base + k * 1024 + t;
            if (i >= T) continue;
            int w = work[k] / ORDER_BIN_WIDTH; w = w < 0 ? 0 : (w > ORDER_BINS - 1 ? ORDER_BINS - 1 : w);
            const uint32_t pos = atomicAdd(&bins[ORDER_BINS - 1 - w], 1u);
            order[pos] = i;
            if (hint) hint[pos] = i;
            if (pos < HEAVY_CAP) sHeavy[pos] = i;
        }
    }
    __syncthreads();
    // work items of the heavy tiles: one per segment of a cut list (exclusive scan over at most 1024 tiles, one per thread)
    const int n_heavy = sNHeavy;
    int nseg = 0;
    if (t < n_heavy) {
        const int tile = sHeavy[t];
        const int L = tile_end[tile] - tile_start[tile];
        nseg = (tile_cut && tile_cut[tile] > 0 && L > 0) ? (L - 1) / GS_SEG + 1 : 1;
    }
    // Segments are handed out heaviest tile first for as long as the grid's item capacity lasts (every later tile still needs one item):
    // inclusive scan of the wanted counts; a tile keeps its segments while (items up to and including it) + (tiles after it) fits.
    auto block_scan = [&](uint32_t x, uint32_t& excl, uint32_t& total_out) {
        uint32_t inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t wo = 0, tt = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) wo += wsum[w]; tt += wsum[w]; }
        excl = wo + inc - x; total_out = tt;
    };
    uint32_t ibase = 0, total_items = 0;
    block_scan((uint32_t)nseg, ibase, total_items);
    if (t < n_heavy && (int)(ibase + (uint32_t)nseg) + (n_heavy - 1 - t) > item_cap) nseg = 1;     // (never with the host's capacity: see gs_api.hip)
    block_scan((uint32_t)nseg, ibase, total_items);
    if (t < n_heavy) n_heavy_out[4 + t] = (int32_t)ibase;
    if (t == 0) { n_heavy_out[4 + n_heavy] = (int32_t)total_items; n_heavy_out[1] = (int32_t)total_items; }
}

// W = sum over the splats behind of (colour . pixel gradient) * alpha * T: the reference's three accumulated colours
// (RAST:656) only ever enter through this dot product (RAST:653-657), so one scalar per pixel carries them.
struct QuadState { float T, W, gr, gg, gb, tot0, tot1; int last; };

// NQ = quadrants per wave: 4 (one wave per tile), 2 (two waves per tile, upper / lower half) or 1.
// With G = 4/NQ waves per tile every (point, tile) pair owns G consecutive rows of `partial`.
// STRICT (gs_config.bwd_reference_order): grad_point_probability_density_from_conic_and_rescale in the reference's own f32
// operation order (UTIL:331-348) -- Sigma^-1 d without fused multiply-adds, the falloff through the reference polynomial for
// every lane, and d p / d Sigma' as 0.5 p (Sigma^-1 (d d^T) Sigma^-1) with two 2x2 matrix products, like
// the CPU oracle (test infrastructure).  The default (fast) form takes the same matrix as v v^T with v = Sigma^-1 d: three fused
// multiply-adds instead of 24 operations, and for a long thin conic -- where a*dx and b*dy cancel in Sigma^-1 d -- more
// accurate than the reference's own rounding, which is why three ill-conditioned soak scenes sit 1.05e-4 .. 1.5e-4 of the tensor
// maximum from the oracle in the fast form and at the summation-order floor in this one
// (tests/test_gpu_parity.py::test_soak_seeds_with_ill_conditioned_splats, profiles/r03_strict_vs_fast.json).
// COOP: the wave is one of the FOUR waves of a workgroup that share a HEAVY tile (NQ = 1: a quadrant each).  They walk the
// list batch by batch in step; a wave leaves the sum of a splat over its own quadrant in its record slab (a processed splat's
// 48-byte record is dead, and twelve floats is what a row holds), and after the batch the workgroup adds the four slabs up and
// stores ONE row per pair -- the same rows, flags and per-point sums as the single-wave form, a heavy tile's critical path
// cut to a quarter of the quadrant evaluations.  (Summation order within a pair: quadrants 0..3, each summed as before.)
struct BwdCoop { unsigned long long* done; int32_t* slot; int32_t* point; int32_t* tile_last; float (*slab)[64][12];
                 // the SEGMENT of the tile's list this workgroup walks (seg of nseg; nseg == 1: the whole list), the forward's cut
                 // records of the tile (256 float4 each; record k: T before entry start + (k + 1) GS_SEG and the colour blended in
                 // segment k; record nseg - 1: the final T and the last segment's colour) and where a segment leaves its |d uv| sums
                 int seg, nseg; const float4* cut_rec; float2* mag_part;
                 // A segment starts from the FORWARD's transmittance at its cut.  The reference decides "does this splat contribute"
                 // (alpha >= 1/255) once in its forward and again, from another expression, in its backward (UTIL:257-284 against
                 // UTIL:331-348): for an alpha within an ulp of 1/255 the two can disagree, and the reference's backward chain then
                 // differs from its forward chain by that splat's factor 1 - 1/255.  A walk that starts from 1 - accumulated_alpha
                 // and decides for itself reproduces that; a start taken from the forward's T does not (0.39 % on the contributions of
                 // the segments in front of such a splat: one soak scene in 300 once every list over 512 entries was cut).  So every
                 // segment compares the T it arrives at, at its front edge, with the forward's record there; a difference of 1e-3
                 // (rounding stays below 1e-4 over 512 steps, the disagreement is 3.9e-3) sets this flag and k_blend_bwd_repair
                 // walks the tile again in one piece, over the same rows.
                 int32_t* redo; };

template <int NQ, bool STRICT, bool COOP>
__device__ __forceinline__ void gs_bwd_tile_body(const int tile, const int grp, const int G_rows, float4 (*sRec)[3], float* sRed, const BwdCoop coop,
                                                 const int32_t* __restrict__ tile_start, const int32_t* __restrict__ tile_end,
                                                 const int32_t* __restrict__ sorted_vals,
                                                 const float4* __restrict__ PA, const float4* __restrict__ PB,
                                                 const float4* __restrict__ PC, const ushort4* __restrict__ boxes,
                                                 const uint32_t* __restrict__ offsets,
                                                 const float* __restrict__ grad_image, const float* __restrict__ acc_alpha,
                                                 const int32_t* __restrict__ last_in, int W, int H, int tiles_x,
                                                 float* __restrict__ partial, uint8_t* __restrict__ visited, uint8_t* __restrict__ touched,
                                                 const uint8_t gen, float* __restrict__ mag_image)
{
#ifdef GS_STATS
    const unsigned long long gs_t0 = wall_clock64();
#endif
    const int lane = threadIdx.x & 63;
    const int tile_u = tile % tiles_x, tile_v = tile / tiles_x;
    const int start = tile_start[tile], end = tile_end[tile];
    // COOP with a cut list: this workgroup owns entries [seg_lo, seg_hi) only
    const int seg_lo = (COOP && coop.nseg > 1) ? start + coop.seg * GS_SEG : start;
    const int seg_hi = (COOP && coop.nseg > 1 && coop.seg < coop.nseg - 1) ? start + (coop.seg + 1) * GS_SEG : end;
    const int lx = lane & 7, ly = lane >> 3;
    QuadState Q[NQ];
    float t_end[COOP ? NQ : 1];                  // COOP: the pixel's 1 - accumulated_alpha (the segment check at the end)
    int qlast[NQ];
    float rx0[NQ], ry0[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const int q = grp * NQ + qi;
        const int pu = tile_u * 16 + (q & 1) * 8 + lx, pv = tile_v * 16 + (q >> 1) * 8 + ly;
        // pixels of a partial edge tile outside the image do not exist: nothing is in range for them
        const bool inside = pu < W && pv < H;
        const size_t o = inside ? (size_t)pv * (size_t)W + (size_t)pu : 0;
        Q[qi].last = inside ? last_in[o] : start;                    // RAST:558
        Q[qi].T = 1.0f - acc_alpha[o];                               // RAST:559-560
        if constexpr (COOP) t_end[qi] = Q[qi].T;
        Q[qi].W = 0.0f;
        Q[qi].gr = grad_image[3 * o]; Q[qi].gg = grad_image[3 * o + 1]; Q[qi].gb = grad_image[3 * o + 2];
        Q[qi].tot0 = Q[qi].tot1 = 0.0f;
        if constexpr (COOP) {
            // A segment that does not end the list starts from the forward's records: the pixel's transmittance at the cut, and
            // W = (colour blended BEHIND the cut) . (pixel gradient), the later segments' colours added up from the back -- small
            // terms first, as the reference's own back-to-front accumulation does (RAST:656).  A pixel whose last contributor
            // lies before the cut takes nothing from beyond it: its final state stands.
            if (coop.nseg > 1 && coop.seg < coop.nseg - 1 && Q[qi].last >= seg_hi) {
                const float4* rec = coop.cut_rec + grp * 64 + lane;
                float br = 0.0f, bg = 0.0f, bb = 0.0f, t_final = 1.0f;
                for (int k = coop.nseg - 1; k > coop.seg; --k) {
                    const float4 r = rec[(size_t)k * 256];
                    if (k == coop.nseg - 1) t_final = r.x;
                    br += r.y; bg += r.z; bb += r.w;
                }
                // The reference's backward does not know the forward's T: it starts from 1 - accumulated_alpha (RAST:559-560), which
                // has lost up to 6e-8 / T of a dim pixel's final T, and every T and every accumulated colour of that pixel's walk
                // carries that factor.  The records hold the forward's exact values; scaled by (1 - accumulated_alpha) / (final T)
                // they are what the reference's division chain arrives at.
                const float ratio = t_final > 0.0f ? Q[qi].T / t_final : 1.0f;
                Q[qi].T = rec[(size_t)coop.seg * 256].x * ratio;
                Q[qi].W = ((Q[qi].gr * br + Q[qi].gg * bg) + Q[qi].gb * bb) * ratio;
            }
        }
        qlast[qi] = gs_wave_max_i(Q[qi].last);
        rx0[qi] = (float)(tile_u * 16 + (q & 1) * 8) + 0.5f;
        ry0[qi] = (float)(tile_v * 16 + (q >> 1) * 8) + 0.5f;
    }
    int tile_last = qlast[0];
#pragma unroll
    for (int qi = 1; qi < NQ; ++qi) tile_last = max(tile_last, qlast[qi]);
    if constexpr (COOP) {                        // the four waves walk the same batches: the tile's last index over all of them
        if (lane == 0) coop.tile_last[grp] = tile_last;
        __syncthreads();
        tile_last = max(max(coop.tile_last[0], coop.tile_last[1]), max(coop.tile_last[2], coop.tile_last[3]));
    }
    bool pixels_finite;                          // every pixel gradient and final T of this wave is a finite number
    {
        float t = 0.0f;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) t += ((Q[qi].gr + Q[qi].gg) + Q[qi].gb) + Q[qi].T;
        pixels_finite = gs_ballot(!(t - t == 0.0f)) == 0ull;
    }
    float pxq[NQ], pyq[NQ];                      // pixel centres of this lane in its NQ quadrants
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { pxq[qi] = rx0[qi] + (float)lx; pyq[qi] = ry0[qi] + (float)ly; }

    // entries at or beyond tile_last are dead for every pixel of the tile (RAST:609-610)
    for (int hi = min(seg_hi, tile_last); hi > seg_lo; hi -= 64) {
        const int lo = max(seg_lo, hi - 64);
        const int i = lo + lane;
        const bool valid = i < hi;
        const int p = valid ? sorted_vals[i] : 0;
        {
            const float4 A = GS_REC(PA, p), B = GS_REC(PB, p), C = GS_REC(PC, p);
            const CullSplat cs = gs_cull_prepare(A, B, C);
            unsigned long long mq[NQ];
            unsigned long long U = 0ull;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                // pixels of the quadrant some entry of this batch is still in range for (RAST:609-610)
                const unsigned long long inq = gs_ballot(Q[qi].last > lo);
                if (inq == 0ull) { mq[qi] = 0ull; continue; }
                const CullRect lr = gs_live_rect(inq, rx0[qi], ry0[qi]);
                mq[qi] = gs_ballot(valid && i < qlast[qi] && !gs_cull(cs, lr.x0, lr.y0, lr.wx, lr.wy));
                U |= mq[qi];
            }
            GS_STAT(8, 1);
            // `clean`: nothing that enters the arithmetic below is NaN or infinite (the batch's records, the running sums W; this
            // wave's pixel gradients, checked once).  Only then may a lane that takes nothing from a splat run the same instructions on
            // alpha = 0 (0 * NaN would not be 0); otherwise the exec-masked form is used for the whole batch.
            float fsum = ((A.x + A.y) + (A.z + A.w)) + ((B.x + B.y) + B.z) + ((C.x + C.y) + C.z);
            fsum = valid ? fsum : 0.0f;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) fsum += Q[qi].W;              // (a NaN colour met in an earlier batch lives on in W)
            const bool clean = pixels_finite && gs_ballot(!(fsum - fsum == 0.0f)) == 0ull;
            unsigned long long done = 0ull;                                   // COOP: splats this wave left a sum for in its slab
            if (U) {
                uint32_t slot = 0;
                if ((U >> lane) & 1ull) {                                     // pre-sort slot of this (point, tile) pair
                    const ushort4 bx = boxes[p];
                    slot = (offsets[p] + (uint32_t)(((int)bx.w - (int)bx.z) * (tile_u - (int)bx.x) + (tile_v - (int)bx.z))) * (uint32_t)G_rows
                           + (COOP ? 0u : (uint32_t)grp);
                    if constexpr (COOP) { coop.slot[lane] = (int32_t)slot; coop.point[lane] = p; }   // (every wave that keeps the splat writes the same two values)
                }
                sRec[lane][0] = A; sRec[lane][1] = B; sRec[lane][2] = C;
                __builtin_amdgcn_wave_barrier();
                while (U) {
                    const int j = 63 - __builtin_clzll(U);                    // back to front, RAST:605-608
                    U &= ~(1ull << j);
                    GS_STAT(9, 1);
                    const float4 a4 = sRec[j][0], b4 = sRec[j][1], c4 = sRec[j][2];
                    const float a = a4.z, b = a4.w, c = b4.x, apt = b4.z;
                    float v[10];
#pragma unroll
                    for (int k = 0; k < 10; ++k) v[k] = 0.0f;
                    int n_use = 0;                                            // contributions of this splat in this tile (wave-uniform)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        if (!((mq[qi] >> j) & 1ull)) continue;                // wave-uniform
                        GS_STAT(10, 1);
                        // lane predicates as wave-uniform SGPR masks (see k_blend_fwd)
                        const unsigned long long inr_m = gs_ballot((lo + j) < Q[qi].last);   // RAST:609-610 (never empty here: the batch cull saw a pixel in range)
                        GS_STAT(11, 1);
                        // grad_point_probability_density_from_conic_and_rescale, UTIL:331-348.  Fast form (fused multiply-adds,
                        // hardware v_exp_f32): alpha is within 2e-6 (relative) of the reference operation sequence, so unless
                        // some lane sits within 1e-5 of the 1/255 threshold every keep/skip decision is already the oracle's;
                        // otherwise the strict sequence and the reference polynomial decide.
                        const float dx = pxq[qi] - a4.x, dy = pyq[qi] - a4.y;
                        float cix, ciy, g, prod_alpha;
                        float m00 = 0.0f, m01 = 0.0f, m11 = 0.0f;                 // Sigma^-1 (d d^T) Sigma^-1 (STRICT only)
                        unsigned long long use_m;
                        if (STRICT) {
                            // UTIL:331-348 as written (no contraction: file default)
                            cix = a * dx + b * dy; ciy = b * dx + c * dy;
                            g = gs_exp_blend(-0.5f * (dx * cix + dy * ciy)) * b4.y;
                            const float oxx = dx * dx, oxy = dx * dy, oyx = dy * dx, oyy = dy * dy;
                            const float io00 = a * oxx + b * oyx, io01 = a * oxy + b * oyy;
                            const float io10 = b * oxx + c * oyx, io11 = b * oxy + c * oyy;
                            m00 = io00 * a + io01 * b; m01 = io00 * b + io01 * c; m11 = io10 * b + io11 * c;
                            prod_alpha = g * apt;
                            use_m = gs_ballot(prod_alpha >= GS_ALPHA_EPS) & inr_m;       // RAST:634
                        } else {
                            cix = __builtin_fmaf(a, dx, b * dy); ciy = __builtin_fmaf(b, dx, c * dy);
                            // exp(-0.5 * q) = 2^(q * (-0.5 * log2 e)): one multiply instead of two
                            g = __builtin_amdgcn_exp2f(__builtin_fmaf(dx, cix, dy * ciy) * -0.72134752044448170f) * b4.y;
                            prod_alpha = g * apt;
                            // RAST:634 against both edges of the band: where the two votes agree no lane is inside it (two compares, no subtraction)
                            use_m = gs_ballot(prod_alpha >= GS_ALPHA_EPS + 4.0e-8f) & inr_m;
                            if (use_m != (gs_ballot(prod_alpha >= GS_ALPHA_EPS - 4.0e-8f) & inr_m)) {
                                const float sx = a * dx + b * dy, sy = b * dx + c * dy;      // no contraction here (file default)
                                g = gs_exp_blend(-0.5f * (dx * sx + dy * sy)) * b4.y;
                                prod_alpha = g * apt;
                                use_m = gs_ballot(prod_alpha >= GS_ALPHA_EPS) & inr_m;
                                GS_STAT(14, 1);
                            }
                        }
                        n_use += __popcll(use_m);
                        GS_STAT(12, __popcll(use_m)); GS_STAT(15, use_m != 0ull ? 1 : 0);
                        if (qi == 0 && clean) {                               // wave-uniform
                            // The first quadrant WRITES the ten sums, and all 64 lanes run the block: a lane that does not contribute
                            // computes on alpha = 0 and g = 0 (two selects), which makes each of its terms an exact zero and leaves its
                            // T and W as they were.  Clearing ten registers first and masking exec around the block (scalar
                            // instructions and a branch, in a kernel whose waves mostly wait on each other's latencies) costs more:
                            // 0.242 -> 0.228 ms.  Contributing lanes execute exactly the masked form's operations.
#pragma clang fp contract(fast)
                            const bool use = __builtin_amdgcn_inverse_ballot_w64(use_m);
                            const float alpha = use ? __int_as_float(min(__float_as_int(prod_alpha), __float_as_int(GS_ALPHA_MAX))) : 0.0f;
                            const float gg = use ? g : 0.0f;
                            const float one_m = 1.0f - alpha;
                            const float inv = __builtin_amdgcn_rcpf(one_m);
                            const float Tn = Q[qi].T * inv;
                            const float cg = c4.x * Q[qi].gr + c4.y * Q[qi].gg + c4.z * Q[qi].gb;
                            const float ag = Tn * cg - inv * Q[qi].W;
                            const float d_rgb = alpha * Tn;
                            const float agg = ag * gg;
                            const float vs0 = agg * cix, vs1 = agg * ciy;
                            v[0] = vs0; v[1] = vs1;
                            if (STRICT) { v[2] = agg * m00; v[3] = agg * m01; v[4] = agg * m11; }
                            else { v[2] = vs0 * cix; v[3] = vs0 * ciy; v[4] = vs1 * ciy; }
                            v[5] = d_rgb * Q[qi].gr; v[6] = d_rgb * Q[qi].gg; v[7] = d_rgb * Q[qi].gb;
                            v[8] = agg;
                            v[9] = __builtin_amdgcn_sqrtf(vs0 * vs0 + vs1 * vs1);
                            Q[qi].T = Tn;
                            Q[qi].W = __builtin_fmaf(cg, d_rgb, Q[qi].W);
                            Q[qi].tot0 = __builtin_fmaf(fabsf(vs0), apt, Q[qi].tot0);
                            Q[qi].tot1 = __builtin_fmaf(fabsf(vs1), apt, Q[qi].tot1);
                        } else if (__builtin_amdgcn_inverse_ballot_w64(use_m)) {                // exec-masked: idle lanes add nothing
                            // float outputs only from here on: let the compiler fuse multiply-adds
#pragma clang fp contract(fast)
                            // min(prod_alpha, 0.99), RAST:636: both positive, so the integer minimum of the bit patterns (no canonicalise)
                            const float alpha = __int_as_float(min(__float_as_int(prod_alpha), __float_as_int(GS_ALPHA_MAX)));
                            const float one_m = 1.0f - alpha;
                            const float inv = __builtin_amdgcn_rcpf(one_m);
                            const float Tn = Q[qi].T * inv;                        // RAST:643 (v_rcp_f32: 1 ulp)
                            // d alpha: sum_c (colour_c*T - accumulated_c/(1-alpha)) * g_c, RAST:653-657, with the sums over c taken first
                            const float cg = c4.x * Q[qi].gr + c4.y * Q[qi].gg + c4.z * Q[qi].gb;
                            const float ag = Tn * cg - inv * Q[qi].W;
                            const float d_rgb = alpha * Tn;                     // RAST:649
                            // Per-splat constant factors are applied once per point in k_bwd_points instead of once per
                            // contribution: opacity on v[0..4] and v[9] (RAST:662), 0.5 on v[2..4], (1-opacity)*opacity on v[8]
                            const float agg = ag * g;
                            const float vs0 = agg * cix, vs1 = agg * ciy;       // RAST:664-665 without the opacity factor
                            v[0] += vs0; v[1] += vs1;
                            if (STRICT) {
                                v[2] = __builtin_fmaf(agg, m00, v[2]); v[3] = __builtin_fmaf(agg, m01, v[3]); v[4] = __builtin_fmaf(agg, m11, v[4]);
                            } else {
                                v[2] = __builtin_fmaf(vs0, cix, v[2]);
                                v[3] = __builtin_fmaf(vs0, ciy, v[3]);
                                v[4] = __builtin_fmaf(vs1, ciy, v[4]);
                            }
                            v[5] = __builtin_fmaf(d_rgb, Q[qi].gr, v[5]);         // RAST:650
                            v[6] = __builtin_fmaf(d_rgb, Q[qi].gg, v[6]);
                            v[7] = __builtin_fmaf(d_rgb, Q[qi].gb, v[7]);
                            v[8] += agg;                                         // RAST:658-661
                            v[9] += __builtin_amdgcn_sqrtf(vs0 * vs0 + vs1 * vs1);   // RAST:691-694
                            Q[qi].T = Tn;
                            Q[qi].W = __builtin_fmaf(cg, d_rgb, Q[qi].W);        // RAST:656
                            Q[qi].tot0 = __builtin_fmaf(fabsf(vs0), apt, Q[qi].tot0);   // RAST:666-667
                            Q[qi].tot1 = __builtin_fmaf(fabsf(vs1), apt, Q[qi].tot1);
                        }
                    }
                    if (n_use == 0) continue;
                    GS_STAT(13, 1);
                    // Sum the ten values over the 64 lanes through a wave-private LDS transpose: 10 conflict-free
                    // ds_write_b32, then lane 4k+s adds 16 floats of value k (4 ds_read_b128, row stride 68 floats
                    // keeps the reads conflict-free) and two quad DPP adds fold s.  About 30 VALU issue slots
                    // instead of 130+ for six half-rate DPP steps on eleven registers.  The eleventh value, the
                    // number of contributions (RAST:695-696), is the population count of the vote masks.
#pragma unroll
                    for (int k = 0; k < 10; ++k) sRed[k * RED_STRIDE + lane] = v[k];
                    __builtin_amdgcn_wave_barrier();
                    float t = 0.0f;
                    if (lane < 40) {
                        const float4* src = reinterpret_cast<const float4*>(sRed + (lane >> 2) * RED_STRIDE + 16 * (lane & 3));
                        const float4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
                        t = (((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w))) +
                            (((x2.x + x2.y) + (x2.z + x2.w)) + ((x3.x + x3.y) + (x3.z + x3.w)));
                    }
                    t += gs_dpp<0xB1>(t);              // quad_perm [1,0,3,2]
                    t += gs_dpp<0x4E>(t);              // quad_perm [2,3,0,1]
                    if (lane == 40) t = __int_as_float(n_use);      // the count travels as an integer end to end (exact for any image size)
                    if constexpr (COOP) {
                        // the splat's record has been consumed: its 48 bytes of the slab take this quadrant's twelve sums
                        if ((lane & 3) == 0 && lane < 48) reinterpret_cast<float*>(&sRec[j][0])[lane >> 2] = t;
                        done |= 1ull << j;
                    } else {
                        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
                        float* row = partial + (size_t)sj * PW;
                        if ((lane & 3) == 0 && lane < 48) row[lane >> 2] = t;              // 12 floats (pad = 0), one store
                        if (lane == 63) { visited[sj] = gen; touched[__builtin_amdgcn_readlane(p, j)] = gen; }   // row written; point has a contribution (this backward's tag)
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (COOP) {
                // the workgroup adds its four slabs up: thread (j, part) takes floats 3 part .. 3 part + 2 of splat j, quadrants in
                // order 0..3 (column 10, the contribution count, as the integer it is), and stores the pair's row
                if (lane == 0) coop.done[grp] = done;
                __syncthreads();
                const int tj = (int)(threadIdx.x >> 2), part = (int)(threadIdx.x & 3);
                float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
                int cnt = 0;
                bool any = false;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if ((coop.done[w] >> tj) & 1ull) {
                        const float* src = &coop.slab[w][tj][3 * part];
                        r0 += src[0];
                        if (part == 3) cnt += __float_as_int(src[1]);                       // column 10: an integer
                        else { r1 += src[1]; r2 += src[2]; }
                        any = true;
                    }
                }
                if (any) {
                    const uint32_t sj = (uint32_t)coop.slot[tj];
                    float* row = partial + (size_t)sj * PW + 3 * part;
                    row[0] = r0; row[1] = part == 3 ? __int_as_float(cnt) : r1; row[2] = r2;      // (column 11 is padding: 0)
                    if (part == 0) { visited[sj] = gen; touched[coop.point[tj]] = gen; }
                }
                __syncthreads();                                                            // the slabs are free for the next batch
            }
        }
    }
    if constexpr (COOP) {
        if (coop.redo && coop.nseg > 1 && coop.seg > 0) {
            const float4* rec = coop.cut_rec + grp * 64 + lane;
            bool differs = false;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                if (Q[qi].last > seg_lo) {                                       // the pixel walked (part of) this segment: Q.T is its T before entry seg_lo
                    const float t_final = rec[(size_t)(coop.nseg - 1) * 256].x;
                    const float expect = rec[(size_t)(coop.seg - 1) * 256].x * (t_final > 0.0f ? t_end[qi] / t_final : 1.0f);
                    differs = differs || fabsf(Q[qi].T - expect) > 1.0e-3f * expect;
                }
            }
            if (gs_ballot(differs) != 0ull && lane == 0) *coop.redo = 1;
        }
    }
#ifdef GS_STATS
    {
        const unsigned wid = blockIdx.x * 4u + (threadIdx.x >> 6);
        if (lane == 0 && wid < 65536u) { gs_stats_wave_times[2 * wid] = gs_t0; gs_stats_wave_times[2 * wid + 1] = wall_clock64(); }
    }
#endif
    if (mag_image) {                                                            // RAST:700-704
        // pixel coordinates derived afresh (the asm hides that they equal the prologue's): five registers would otherwise stay
        // live across the whole walk, and the kernel sits exactly at the 96-VGPR boundary of five waves per SIMD
        int lane_e = threadIdx.x & 63;
        asm volatile("" : "+v"(lane_e));
        const int lx_e = lane_e & 7, ly_e = lane_e >> 3;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const int q = grp * NQ + qi;
            const int pu = tile_u * 16 + (q & 1) * 8 + lx_e, pv = tile_v * 16 + (q >> 1) * 8 + ly_e;
            if (pu >= W || pv >= H) continue;
            if (COOP && coop.nseg > 1) {          // a segment's share: summed over the segments by the fold blocks of k_sum_rows
                coop.mag_part[(size_t)coop.seg * 256 + grp * 64 + lane_e] = make_float2(Q[qi].tot0, Q[qi].tot1);
                continue;
            }
            const size_t o = (size_t)pv * (size_t)W + (size_t)pu;
            mag_image[2 * o] = Q[qi].tot0; mag_image[2 * o + 1] = Q[qi].tot1;
        }
    }
}

// One launch for the whole backward blend.  Workgroups of four waves: the first n_items workgroups take one work item of a HEAVY
// tile each (k_tile_order put those tiles at the head of the order and counted their items: a segment of the tile's cut list, or
// the whole list) and share it cooperatively, a quadrant per wave; every later workgroup takes four (tile, quadrant group) work
// items of the ordinary kind -- NQ = 4: four tiles, one wave each; NQ = 2: two tiles, two waves each; NQ = 1: one tile.  The grid
// is sized for the largest possible number of heavy items (the host does not know it); surplus workgroups leave at once.
#ifndef GS_BWD_MIN_WAVES
#define GS_BWD_MIN_WAVES 4        // waves per SIMD the register allocator must leave room for (6 spills 17 registers: measured slower)
#endif
template <int NQ, bool STRICT>
__global__ __launch_bounds__(256, GS_BWD_MIN_WAVES) void k_blend_bwd_tile(const int32_t* __restrict__ tile_order, const int32_t* __restrict__ n_heavy_ptr, int T,
                                                        const int32_t* __restrict__ tile_start, const int32_t* __restrict__ tile_end,
                                                        const int32_t* __restrict__ sorted_vals,
                                                        const float4* __restrict__ PA, const float4* __restrict__ PB,
                                                        const float4* __restrict__ PC, const ushort4* __restrict__ boxes,
                                                        const uint32_t* __restrict__ offsets,
                                                        const float* __restrict__ grad_image, const float* __restrict__ acc_alpha,
                                                        const int32_t* __restrict__ last_in, int W, int H, int tiles_x,
                                                        float* __restrict__ partial, uint8_t* __restrict__ visited, uint8_t* __restrict__ touched,
                                                        const uint8_t gen, float* __restrict__ mag_image,
                                                        const float4* __restrict__ cuts, float2* __restrict__ cut_mag, const int32_t* __restrict__ tile_cut)
{
    __shared__ float4 sRecAll[4][64][3];             // per wave: the batch's splat records (COOP: then the per-quadrant sums)
    __shared__ __attribute__((aligned(16))) float sRedAll[4][11 * RED_STRIDE];
    __shared__ unsigned long long sDone[4];
    __shared__ int32_t sSlot[64], sPoint[64], sTileLast[4];
    constexpr int G = 4 / NQ;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_heavy = n_heavy_ptr[0], n_items = n_heavy_ptr[1];
    const int32_t* item_base = n_heavy_ptr + 4;
    BwdCoop coop;
    coop.done = sDone; coop.slot = sSlot; coop.point = sPoint; coop.tile_last = sTileLast;
    coop.slab = reinterpret_cast<float (*)[64][12]>(&sRecAll[0][0][0]);
    coop.seg = 0; coop.nseg = 1; coop.cut_rec = nullptr; coop.mag_part = nullptr; coop.redo = nullptr;
    // (heavy items first: handing them out BEHIND the ordinary tiles, as the short jobs they are, measured slower -- DESIGN.md section 5)
    const int hb = (int)blockIdx.x < n_items ? (int)blockIdx.x : -1;
    const int lb = (int)blockIdx.x - n_items;
    if (hb >= 0) {
        int h = 0;                                                     // the heavy tile that owns item hb: last h with item_base[h] <= hb
        for (int step = HEAVY_CAP / 2; step > 0; step >>= 1)
            if (h + step < n_heavy && item_base[h + step] <= hb) h += step;
        const int tile = tile_order[h];
        coop.nseg = item_base[h + 1] - item_base[h];
        coop.seg = hb - item_base[h];
        if (coop.nseg > 1) {
            const size_t first = (size_t)(tile_cut[tile] - 1) * 256;
            coop.cut_rec = cuts + first; coop.mag_part = cut_mag + first;
            coop.redo = const_cast<int32_t*>(n_heavy_ptr) + GS_ORDER_REDO_OFFSET + h;
        }
        gs_bwd_tile_body<1, STRICT, true>(tile, wave, G, sRecAll[wave], sRedAll[wave], coop, tile_start, tile_end, sorted_vals,
                                          PA, PB, PC, boxes, offsets, grad_image, acc_alpha, last_in, W, H, tiles_x, partial, visited, touched, gen, mag_image);
        return;
    }
    const int item = lb * 4 + wave;                                    // work item among the ordinary (tile, quadrant group) pairs
    const int ti = n_heavy + item / G;
    if (ti >= T) return;
    gs_bwd_tile_body<NQ, STRICT, false>(tile_order[ti], item % G, G, sRecAll[wave], sRedAll[wave], coop, tile_start, tile_end, sorted_vals,
                                        PA, PB, PC, boxes, offsets, grad_image, acc_alpha, last_in, W, H, tiles_x, partial, visited, touched, gen, mag_image);
}

// Heavy tiles whose segments found the forward's transmittance at a cut at odds with their own walk (BwdCoop::redo): the whole
// list again, in one piece, from 1 - accumulated_alpha, over the same rows, flags and pixels.  One workgroup per heavy tile; all
// but the flagged ones (normally none) leave at once.  Launched only for frames with cut lists.
template <bool STRICT>
__global__ __launch_bounds__(256) void k_blend_bwd_repair(const int32_t* __restrict__ tile_order, const int32_t* __restrict__ n_heavy_ptr, int G,
                                                          const int32_t* __restrict__ tile_start, const int32_t* __restrict__ tile_end,
                                                          const int32_t* __restrict__ sorted_vals,
                                                          const float4* __restrict__ PA, const float4* __restrict__ PB,
                                                          const float4* __restrict__ PC, const ushort4* __restrict__ boxes,
                                                          const uint32_t* __restrict__ offsets,
                                                          const float* __restrict__ grad_image, const float* __restrict__ acc_alpha,
                                                          const int32_t* __restrict__ last_in, int W, int H, int tiles_x,
                                                          float* __restrict__ partial, uint8_t* __restrict__ visited, uint8_t* __restrict__ touched,
                                                          const uint8_t gen, float* __restrict__ mag_image)
{
    const int h = (int)blockIdx.x;
    if (h >= n_heavy_ptr[0] || n_heavy_ptr[GS_ORDER_REDO_OFFSET + h] == 0) return;
    __shared__ float4 sRecAll[4][64][3];
    __shared__ __attribute__((aligned(16))) float sRedAll[4][11 * RED_STRIDE];
    __shared__ unsigned long long sDone[4];
    __shared__ int32_t sSlot[64], sPoint[64], sTileLast[4];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    BwdCoop coop;
    coop.done = sDone; coop.slot = sSlot; coop.point = sPoint; coop.tile_last = sTileLast;
    coop.slab = reinterpret_cast<float (*)[64][12]>(&sRecAll[0][0][0]);
    coop.seg = 0; coop.nseg = 1; coop.cut_rec = nullptr; coop.mag_part = nullptr; coop.redo = nullptr;
    gs_bwd_tile_body<1, STRICT, true>(tile_order[h], wave, G, sRecAll[wave], sRedAll[wave], coop, tile_start, tile_end, sorted_vals,
                                      PA, PB, PC, boxes, offsets, grad_image, acc_alpha, last_in, W, H, tiles_x, partial, visited, touched, gen, mag_image);
}

// ---------------------------------------------------------------------------------
// Per-point sum of the visited rows of `partial` (fixed order => bitwise reproducible).  FOUR lanes per
// in-camera point: lane q of the quad takes rows q, q+4, q+8, ... so a point with at most 32 rows has all its
// loads in flight at once (the kernel is latency bound: flag byte -> row), unvisited rows read a shared all-zero
// row, and two quad DPP steps fold the four partial sums.  A point with more rows is summed by its whole wave
// (lanes stride over the rows, then a DPP reduction), so one huge splat cannot become the critical path.
#define SUM_ROWS_SMALL 32
#define SUM_ROWS_GIANT 1024
// rows first, first + STRIDE, ... of one point: four per thread in flight (flags, then the rows; an unvisited row reads the shared zero
// row), added in index order
template <int STRIDE>
__device__ __forceinline__ void gs_sum_rows_strided(const float4* __restrict__ rows, const uint8_t* __restrict__ vis, const uint8_t gen,
                                                    const float4* __restrict__ zero_row, const int first, const int cnt, float (&w)[11], int& wpix)
{
    for (int i0 = first; i0 < cnt; i0 += 4 * STRIDE) {
        const float4* r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + STRIDE * k;
            const bool on = i < cnt && vis[i] == gen;
            r[k] = on ? rows + 3 * i : zero_row;
        }
        float4 a[4], b[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = r[k][0]; b[k] = r[k][1]; c[k] = r[k][2]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[0] += a[k].x; w[1] += a[k].y; w[2] += a[k].z; w[3] += a[k].w; w[4] += b[k].x; w[5] += b[k].y; w[6] += b[k].z; w[7] += b[k].w;
            w[8] += c[k].x; w[9] += c[k].y; wpix += __float_as_int(c[k].z);
        }
    }
}
struct GsMagFold { unsigned first_block; const int32_t* n_heavy; const int32_t* tile_order; const int32_t* tile_cut; const float2* cut_mag;
                   float* mag_image; int W, H, tiles_x; };
// fold blocks of k_sum_rows (behind the summing ones)
__device__ __forceinline__ void gs_fold_mag(const GsMagFold& fold)
{
        // heavy tile h was walked in segments, each of which left its share of the pixels'
        // sum |d uv| (RAST:666-667, 700-704) in the cut records' side array; one thread per pixel adds them up in segment order
        const int h = (int)(blockIdx.x - fold.first_block);
        if (!fold.mag_image || h >= fold.n_heavy[0]) return;
        const int32_t* item_base = fold.n_heavy + 4;
        const int nseg = item_base[h + 1] - item_base[h];
        if (nseg <= 1) return;
        if (fold.n_heavy[GS_ORDER_REDO_OFFSET + h] != 0) return;              // walked again in one piece (k_blend_bwd_repair): its pixels' sums are final
        const int tile = fold.tile_order[h];
        const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int pu = (tile % fold.tiles_x) * 16 + (q & 1) * 8 + (lane & 7), pv = (tile / fold.tiles_x) * 16 + (q >> 1) * 8 + (lane >> 3);
        if (pu >= fold.W || pv >= fold.H) return;
        const float2* part = fold.cut_mag + (size_t)(fold.tile_cut[tile] - 1) * 256 + threadIdx.x;
        float a = 0.0f, b = 0.0f;
        for (int sgm = nseg - 1; sgm >= 0; --sgm) { const float2 v = part[(size_t)sgm * 256]; a += v.x; b += v.y; }   // back to front, like the walk
        const size_t o = (size_t)pv * (size_t)fold.W + (size_t)pu;
        fold.mag_image[2 * o] = a; fold.mag_image[2 * o + 1] = b;
}

__global__ __launch_bounds__(256, 7) void k_sum_rows(int M, int G, const uint32_t* __restrict__ offsets, const int32_t* __restrict__ ntiles,
                                                  const float* __restrict__ partial, const uint8_t* __restrict__ visited, const uint8_t* __restrict__ touched,
                                                  const uint8_t gen, const float4* __restrict__ zero_row, float4* __restrict__ sums, const GsMagFold fold,
                                                  const int32_t* __restrict__ max_tiles_hint)
{
    if (blockIdx.x >= fold.first_block) { gs_fold_mag(fold); return; }
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t >> 2, q = t & 3;
    const int lane = threadIdx.x & 63;
    const bool valid = m < M;
    // a point no pixel took a contribution from (nine in ten at the headline config) has no visited row: its byte of
    // `touched` says so and none of its flags or rows is looked at
    const bool live = valid && touched[m] == gen;
    const uint32_t off = live ? offsets[m] * (uint32_t)G : 0u;       // G rows per (point, tile) pair
    const int cnt = live ? ntiles[m] * G : 0;
    // the block's 64 points seen by every wave: which of them are GIANT (summed by the whole block at the end) -- looked for only in a
    // frame whose largest point can be one (k_project left its tile count; the two extra loads cost 2 us of the launch at the headline config)
    // A hint that says "none" only moves such a point back to its wave: giant_rows is the one threshold both passes use.
    unsigned long long giants = 0ull;
    int giant_rows = 0x7fffffff;
    if (!max_tiles_hint || max_tiles_hint[0] * G > SUM_ROWS_GIANT) {              // (uniform: a scalar load)
        giant_rows = SUM_ROWS_GIANT;
        const int bl = (int)blockIdx.x * 64 + lane;
        giants = gs_ballot(bl < M && touched[bl] == gen && ntiles[bl] * G > SUM_ROWS_GIANT);
    }
    float v[11];
    int npix = 0;                                                     // column 10 is an integer count: summed as one
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] = 0.0f;
    if (cnt <= SUM_ROWS_SMALL) {
        const float4* rows = reinterpret_cast<const float4*>(partial + (size_t)off * PW);
        const uint8_t* vis = visited + off;
        const float4* r[SUM_ROWS_SMALL / 4];
#pragma unroll
        for (int k = 0; k < SUM_ROWS_SMALL / 4; ++k) {
            const int i = q + 4 * k;
            const bool on = i < cnt && vis[i] == gen;
            r[k] = on ? rows + 3 * i : zero_row;
        }
#pragma unroll
        for (int h = 0; h < SUM_ROWS_SMALL / 4; h += 4) {
            if (4 * h >= cnt) break;                                  // wave-divergent but cheap: most points have few rows
            float4 a[4], b[4], c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = r[h + k][0]; b[k] = r[h + k][1]; c[k] = r[h + k][2]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[0] += a[k].x; v[1] += a[k].y; v[2] += a[k].z; v[3] += a[k].w;
                v[4] += b[k].x; v[5] += b[k].y; v[6] += b[k].z; v[7] += b[k].w;
                v[8] += c[k].x; v[9] += c[k].y; npix += __float_as_int(c[k].z);
            }
        }
    }
    GS_DPP11("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    GS_DPP11("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    asm volatile("s_nop 1");
    npix += __builtin_amdgcn_update_dpp(0, npix, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    npix += __builtin_amdgcn_update_dpp(0, npix, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
    if (valid && q == 0 && cnt <= SUM_ROWS_SMALL) {
        sums[3 * (size_t)m] = make_float4(v[0], v[1], v[2], v[3]);
        sums[3 * (size_t)m + 1] = make_float4(v[4], v[5], v[6], v[7]);
        sums[3 * (size_t)m + 2] = make_float4(v[8], v[9], __int_as_float(npix), 0.0f);
    }
    // wave-cooperative pass over the large points of this wave (one vote per quad leader); the GIANT ones (a background splat over
    // the whole image: thousands of rows, 350 KB) are left to the whole block below
    const bool leader = valid && q == 0;
    unsigned long long big = gs_ballot(leader && cnt > SUM_ROWS_SMALL && cnt <= giant_rows);
    while (big) {
        const int j = __builtin_ctzll(big);
        big &= big - 1ull;
        const uint32_t boff = (uint32_t)__builtin_amdgcn_readlane((int)off, j);
        const int bcnt = __builtin_amdgcn_readlane(cnt, j);
        const int bm = __builtin_amdgcn_readlane(m, j);
        const float4* rows = reinterpret_cast<const float4*>(partial + (size_t)boff * PW);
        const uint8_t* vis = visited + boff;
        float w[11];
        int wpix = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) w[k] = 0.0f;
        if (bcnt <= 256) {
            for (int i = lane; i < bcnt; i += 64) {
                if (vis[i] == gen) {
                    const float4 a = rows[3 * i], b = rows[3 * i + 1], c = rows[3 * i + 2];
                    w[0] += a.x; w[1] += a.y; w[2] += a.z; w[3] += a.w; w[4] += b.x; w[5] += b.y; w[6] += b.z; w[7] += b.w;
                    w[8] += c.x; w[9] += c.y; wpix += __float_as_int(c.z);
                }
            }
        } else {
            gs_sum_rows_strided<64>(rows, vis, gen, zero_row, lane, bcnt, w, wpix);
        }
        gs_wave_sum11_row3(w);
        wpix = gs_wave_sum_i(wpix);
        if (lane == 63) {
            sums[3 * (size_t)bm] = make_float4(w[0], w[1], w[2], w[3]);
            sums[3 * (size_t)bm + 1] = make_float4(w[4], w[5], w[6], w[7]);
            sums[3 * (size_t)bm + 2] = make_float4(w[8], w[9], __int_as_float(wpix), 0.0f);
        }
    }
    // block-cooperative pass over the giant points of this block: its 256 threads stride over the rows, four per thread in flight; the
    // four waves' sums are added in wave order.  Which points are giant is a function of their tile counts alone: fixed order, fixed bits.
    // Every wave looks at all 64 points of the block itself (one more coalesced load), so a block without giants -- nearly all of
    // them -- leaves without a barrier (waiting for the slowest wave there cost 3 us of the launch at the headline config).
    __shared__ float sPart[4][12];
    const int wave = threadIdx.x >> 6;
    unsigned long long gm = giants;
    {
        while (gm) {
            const int j = __builtin_ctzll(gm);
            gm &= gm - 1ull;
            const int bm = (int)blockIdx.x * 64 + j;
            const uint32_t boff = offsets[bm] * (uint32_t)G;
            const int bcnt = ntiles[bm] * G;
            float w[11];
            int wpix = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) w[k] = 0.0f;
            gs_sum_rows_strided<256>(reinterpret_cast<const float4*>(partial + (size_t)boff * PW), visited + boff, gen, zero_row, (int)threadIdx.x, bcnt, w, wpix);
            gs_wave_sum11_row3(w);
            wpix = gs_wave_sum_i(wpix);
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 10; ++k) sPart[wave][k] = w[k];
                sPart[wave][10] = __int_as_float(wpix);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                float r[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) r[k] = ((sPart[0][k] + sPart[1][k]) + sPart[2][k]) + sPart[3][k];
                const int np = (__float_as_int(sPart[0][10]) + __float_as_int(sPart[1][10])) + (__float_as_int(sPart[2][10]) + __float_as_int(sPart[3][10]));
                sums[3 * (size_t)bm] = make_float4(r[0], r[1], r[2], r[3]);
                sums[3 * (size_t)bm + 1] = make_float4(r[4], r[5], r[6], r[7]);
                sums[3 * (size_t)bm + 2] = make_float4(r[8], r[9], __int_as_float(np), 0.0f);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------
#define ROW_LDS 60
__global__ __launch_bounds__(256) void k_bwd_points(
    int64_t N, const int32_t* __restrict__ cam_index, const float4* __restrict__ sums, const float4* __restrict__ PD,
    const float* __restrict__ pc, const float* __restrict__ feat, const int32_t* __restrict__ obj,
    const float* __restrict__ Kmat, const GsPose* __restrict__ pose,
    int keep, float f_color, float f_high, float f_s, float f_q, float f_alpha,
    float* __restrict__ grad_pc, float* __restrict__ grad_feat, float* __restrict__ grad_uv, float* __restrict__ mag,
    int32_t* __restrict__ n_affected,
    float* __restrict__ hook_gpc, float* __restrict__ hook_gfeat, float* __restrict__ hook_guv, float* __restrict__ hook_mag,
    int32_t* __restrict__ hook_ids, int32_t* __restrict__ hook_ntiles, float* __restrict__ hook_depth, float* __restrict__ hook_uv,
    const float4* __restrict__ PA, const float4* __restrict__ PB, const int32_t* __restrict__ ntiles,
    int32_t* __restrict__ c_num_in_camera, int32_t* __restrict__ c_num_pixels, float* __restrict__ c_vs_grad,
    float* __restrict__ c_vs_grad_avg, float* __restrict__ c_pos_grad, float* __restrict__ c_pos_grad_norm)
{
    // The 56-float gradient rows leave through LDS: a lane-per-row store reaches ~3.2 TB/s, the same rows written
    // as one contiguous 14 KB run per wave ~6.2 TB/s (tools/ubench_rows.hip).  Row stride 60 floats keeps both the
    // lane-wise float4 staging writes and the row-wise reads spread over the banks.
    __shared__ __attribute__((aligned(16))) float sRows[4][64 * ROW_LDS + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = n < N;
    const int m = valid ? cam_index[n] : -1;
    float out[GS_NFEAT];
#pragma unroll
    for (int k = 0; k < GS_NFEAT; ++k) out[k] = 0.0f;               // RAST:1051-1058 zero rows
    if (valid && m < 0) {
        grad_pc[3 * n] = 0.0f; grad_pc[3 * n + 1] = 0.0f; grad_pc[3 * n + 2] = 0.0f;
        if (grad_uv) { grad_uv[2 * n] = 0.0f; grad_uv[2 * n + 1] = 0.0f; }
        if (mag) mag[n] = 0.0f;
    }
    float s[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) s[k] = 0.0f;
    if (m >= 0) {
        const float4 r0 = sums[3 * (size_t)m], r1 = sums[3 * (size_t)m + 1], r2 = sums[3 * (size_t)m + 2];
        s[0] = r0.x; s[1] = r0.y; s[2] = r0.z; s[3] = r0.w; s[4] = r1.x; s[5] = r1.y; s[6] = r1.z; s[7] = r1.w;
        s[8] = r2.x; s[9] = r2.y; s[10] = r2.z; s[11] = r2.w;
    }
    // An in-camera point no pixel took a contribution from (hidden behind saturated pixels, or below 1/255 everywhere:
    // nine in ten at the headline config) has all-zero sums, hence all-zero gradients: its 224-byte feature row is not
    // read and the Jacobian chain is not evaluated.  (The reference multiplies those zeros through the chain, which
    // gives the same zeros unless a Jacobian entry is not finite.)
    const bool touched = m >= 0 && __float_as_int(s[10]) != 0;
    if (m >= 0 && !touched) {
        grad_pc[3 * n] = 0.0f; grad_pc[3 * n + 1] = 0.0f; grad_pc[3 * n + 2] = 0.0f;
        if (grad_uv) { grad_uv[2 * n] = 0.0f; grad_uv[2 * n + 1] = 0.0f; }
        if (mag) mag[n] = 0.0f;
        if (n_affected) n_affected[m] = 0;
        if (hook_gpc) { hook_gpc[3 * (size_t)m] = 0.0f; hook_gpc[3 * (size_t)m + 1] = 0.0f; hook_gpc[3 * (size_t)m + 2] = 0.0f; }
        if (hook_guv) { hook_guv[2 * (size_t)m] = 0.0f; hook_guv[2 * (size_t)m + 1] = 0.0f; }
        if (hook_mag) hook_mag[m] = 0.0f;
        if (hook_ids) hook_ids[m] = (int32_t)n;
        if (hook_ntiles) hook_ntiles[m] = ntiles[m];
        if (hook_depth) hook_depth[m] = GS_REC(PB, m).w;
        if (hook_uv) { const float4 pa = GS_REC(PA, m); hook_uv[2 * (size_t)m] = pa.x; hook_uv[2 * (size_t)m + 1] = pa.y; }
        if (c_num_in_camera) c_num_in_camera[n] += 1;                   // CTRL:133; the other five accumulators get += 0
    }
    if (touched) {
        {   // the per-splat factors k_blend_bwd_tile left out (see there): opacity, 0.5, (1 - opacity) * opacity
            const float apt = GS_REC(PB, m).z;
            s[0] *= apt; s[1] *= apt; s[9] *= apt;
            const float h = 0.5f * apt;
            s[2] *= h; s[3] *= h; s[4] *= h;
            s[8] *= (1.0f - apt) * apt;
        }
        const float guv0 = s[0], guv1 = s[1];
        const float g00 = s[2], g01 = s[3], g11 = s[4];
        const float4* row4 = reinterpret_cast<const float4*>(feat + (size_t)GS_NFEAT * n);
        float row[GS_NFEAT];
#pragma unroll
        for (int k = 0; k < GS_NFEAT / 4; ++k) {
            float4 v = row4[k];
            row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
        }
        const GsPose& P = pose[obj[n]];
        float Km[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Km[k] = Kmat[k];
        const float x = pc[3 * n], y = pc[3 * n + 1], z = pc[3 * n + 2];
        // ---- d uv / d xyz, GP3D:132-159 ----
        float tx = ((P.R[0] * x + P.R[1] * y) + P.R[2] * z) + P.t[0];
        float ty = ((P.R[3] * x + P.R[4] * y) + P.R[5] * z) + P.t[1];
        float tz = ((P.R[6] * x + P.R[7] * y) + P.R[8] * z) + P.t[2];
        float d[6] = { Km[0] / tz, Km[1] / tz, (-Km[0] * tx - Km[1] * ty) / (tz * tz),
                       Km[3] / tz, Km[4] / tz, (-Km[3] * tx - Km[4] * ty) / (tz * tz) };
        float gt[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                                   // (d uv/d p_cam . W) first, GP3D:157-159, then RAST:757
            const float j0 = (d[0] * P.R[j] + d[1] * P.R[3 + j]) + d[2] * P.R[6 + j];
            const float j1 = (d[3] * P.R[j] + d[4] * P.R[3 + j]) + d[5] * P.R[6 + j];
            gt[j] = guv0 * j0 + guv1 * j1;
        }
        // ---- d Sigma' / d(q, s), GP3D:237-331, contracted with (g00 g01; g01 g11) ----
        const float4 pd = GS_REC(PD, m);                                        // translation_camera, RAST:737-738
        const float fx = Km[0], fy = Km[4];
        float J[6] = { fx / pd.z, 0.0f, -(fx * pd.x) / (pd.z * pd.z), 0.0f, fy / pd.z, -(fy * pd.y) / (pd.z * pd.z) };
        float U[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            U[j] = J[0] * P.R[j] + J[2] * P.R[6 + j];
            U[3 + j] = J[4] * P.R[3 + j] + J[5] * P.R[6 + j];
        }
        const float qx = row[0], qy = row[1], qz = row[2], qw = row[3];
        float R[9];
        {
            float xx = qx * qx, yy = qy * qy, zz = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
            R[0] = 1.0f - 2.0f * (yy + zz); R[1] = 2.0f * (xy - wz); R[2] = 2.0f * (xz + wy);
            R[3] = 2.0f * (xy + wz); R[4] = 1.0f - 2.0f * (xx + zz); R[5] = 2.0f * (yz - wx);
            R[6] = 2.0f * (xz - wy); R[7] = 2.0f * (yz + wx); R[8] = 1.0f - 2.0f * (xx + yy);
        }
        const float es[3] = { gs_expf(row[4]), gs_expf(row[5]), gs_expf(row[6]) };
        float Mm[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Mm[3 * i + j] = R[3 * i + j] * es[j];       // M = R S, GP3D:257
        // d Sigma'/d M = (U (x) U)(4x9) . d Sigma/d M (9x9), then . dM/dq (9x4) and . dM/dS . dS/ds (GP3D:270-330), and only
        // then the contraction with the accumulated (g00 g01; g01 g11) (RAST:716-721, 760-761): the reference's own
        // product order, every sum left to right over the inner index with the structural zeros of the tables left out
        // (adding an exact zero changes nothing).  A closed form (dL/dM = 2 U^T g U M) is shorter and, for a huge,
        // strongly anisotropic splat, more accurate -- but then it is not the reference's f32 result: the 4x4 and 4x3
        // Jacobians of such a splat hold large entries that cancel in the final contraction, and the reference rounds them first.
        const float sx = es[0], sy = es[1], sz = es[2];
        // dM/dq, GP3D:319-329 (rows = M entries 00 01 02 10 11 12 20 21 22; columns = q x y z w)
        const float dMdq[36] = {
            0.0f, -4 * sx * qy, -4 * sx * qz, 0.0f,
            2 * sy * qy, 2 * sy * qx, -2 * sy * qw, -2 * sy * qz,
            2 * sz * qz, 2 * sz * qw, 2 * sz * qx, 2 * sz * qy,
            2 * sx * qy, 2 * sx * qx, 2 * sx * qw, 2 * sx * qz,
            -4 * sy * qx, 0.0f, -4 * sy * qz, 0.0f,
            -2 * sz * qw, 2 * sz * qz, 2 * sz * qy, -2 * sz * qx,
            2 * sx * qz, -2 * sx * qw, 2 * sx * qx, -2 * sx * qy,
            2 * sy * qw, 2 * sy * qz, 2 * sy * qy, 2 * sy * qx,
            -4 * sz * qx, -4 * sz * qy, 0.0f, 0.0f };
        const float gcov[4] = { g00, g01, g01, g11 };
        float gq[4], gs_[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // row i of d Sigma'/d Sigma (GP3D:270-279): entry 3a+b is U[i>>1][a] * U[i&1][b] for rows 0, 1, 3 and
            // U[0][b] * U[1][a] for row 2
            float D[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb)
                    D[3 * a + bb] = i == 2 ? U[bb] * U[3 + a] : U[3 * (i >> 1) + a] * U[3 * (i & 1) + bb];
            float Pm[9];                                          // row i of d Sigma'/d M, GP3D:294
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
                const float m0 = Mm[bb], m1 = Mm[3 + bb], m2 = Mm[6 + bb];
                Pm[bb] = (((D[0] * (2.0f * m0) + D[1] * m1) + D[2] * m2) + D[3] * m1) + D[6] * m2;
                Pm[3 + bb] = (((D[1] * m0 + D[3] * m0) + D[4] * (2.0f * m1)) + D[5] * m2) + D[7] * m2;
                Pm[6 + bb] = (((D[2] * m0 + D[5] * m1) + D[6] * m0) + D[7] * m1) + D[8] * (2.0f * m2);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                         // row i of d Sigma'/dq, GP3D:330
                float acc = 0.0f;
                bool first = true;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const bool zero = (e == 0 && (k == 0 || k == 3)) || (e == 4 && (k == 1 || k == 3)) || (e == 8 && k >= 2);
                    if (zero) continue;
                    const float term = Pm[e] * dMdq[4 * e + k];
                    acc = first ? term : acc + term;
                    first = false;
                }
                gq[k] = i == 0 ? gcov[0] * acc : gq[k] + gcov[i] * acc;     // RAST:760
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {                         // row i of d Sigma'/ds, GP3D:297-313
                const float acc = ((Pm[c] * R[c] + Pm[3 + c] * R[3 + c]) + Pm[6 + c] * R[6 + c]) * es[c];
                gs_[c] = i == 0 ? gcov[0] * acc : gs_[c] + gcov[i] * acc;   // RAST:761
            }
        }
        // ---- colour, GP3D:351-373 with the backward's ray origin (RAST:731-732, 749) ----
        float dx = x - P.origin_bwd[0], dy = y - P.origin_bwd[1], dz = z - P.origin_bwd[2];
        float dn = sqrtf(dx * dx + dy * dy + dz * dz);
        float ux = dx / dn, uy = dy / dn, uz = dz / dn;
        float sh[16];
        sh[0] = 0.28209479177387814f;
        sh[1] = -0.48860251190291987f * uy;
        sh[2] = 0.48860251190291987f * uz;
        sh[3] = -0.48860251190291987f * ux;
        sh[4] = 1.0925484305920792f * ux * uy;
        sh[5] = -1.0925484305920792f * uy * uz;
        sh[6] = 0.94617469575755997f * uz * uz - 0.31539156525251999f;
        sh[7] = -1.0925484305920792f * ux * uz;
        sh[8] = 0.54627421529603959f * ux * ux - 0.54627421529603959f * uy * uy;
        sh[9] = 0.59004358992664352f * uy * (-3.0f * ux * ux + uy * uy);
        sh[10] = 2.8906114426405538f * ux * uy * uz;
        sh[11] = 0.45704579946446572f * uy * (1.0f - 5.0f * uz * uz);
        sh[12] = 0.3731763325901154f * uz * (5.0f * uz * uz - 3.0f);
        sh[13] = 0.45704579946446572f * ux * (1.0f - 5.0f * uz * uz);
        sh[14] = 1.4453057213202769f * uz * (ux * ux - uy * uy);
        sh[15] = 0.59004358992664352f * ux * (-ux * ux + 3.0f * uy * uy);
        out[0] = gq[0] * f_q; out[1] = gq[1] * f_q; out[2] = gq[2] * f_q; out[3] = gq[3] * f_q;      // RAST:1105-1106
        out[4] = gs_[0] * f_s; out[5] = gs_[1] * f_s; out[6] = gs_[2] * f_s;                          // RAST:1107-1108
        out[7] = s[8] * f_alpha;                                                                      // RAST:1109-1110
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* f = row + 8 + 16 * ch;
            float accd = f[0] * sh[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) accd = accd + f[k] * sh[k];
            float sg = gs_sigmoid(accd);
            float jac = sg * (1.0f - sg);
            float gc = s[5 + ch];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float v = gc * (jac * sh[k]);                           // RAST:754-756
                v = k < keep ? v * (k == 0 ? f_color : f_high) : 0.0f;  // RAST:1167-1182, 1112-1125
                out[8 + 16 * ch + k] = v;
            }
        }
        grad_pc[3 * n] = gt[0]; grad_pc[3 * n + 1] = gt[1]; grad_pc[3 * n + 2] = gt[2];
        if (grad_uv) { grad_uv[2 * n] = guv0; grad_uv[2 * n + 1] = guv1; }
        if (mag) mag[n] = s[9];
        if (n_affected) n_affected[m] = __float_as_int(s[10]);
        if (hook_gpc) { hook_gpc[3 * (size_t)m] = gt[0]; hook_gpc[3 * (size_t)m + 1] = gt[1]; hook_gpc[3 * (size_t)m + 2] = gt[2]; }
        if (hook_guv) { hook_guv[2 * (size_t)m] = guv0; hook_guv[2 * (size_t)m + 1] = guv1; }
        if (hook_mag) hook_mag[m] = s[9];
        if (hook_ids) hook_ids[m] = (int32_t)n;                         // RAST:1129, 1136-1139
        if (hook_ntiles) hook_ntiles[m] = ntiles[m];
        if (hook_depth) hook_depth[m] = GS_REC(PB, m).w;
        if (hook_uv) { const float4 pa = GS_REC(PA, m); hook_uv[2 * (size_t)m] = pa.x; hook_uv[2 * (size_t)m + 1] = pa.y; }
        if (c_num_in_camera) {                                          // GaussianPointAdaptiveController.update, CTRL:133-141
            const int32_t npix = __float_as_int(s[10]);
            c_num_in_camera[n] += 1;
            c_num_pixels[n] += npix;
            c_vs_grad[n] += s[9];
            const float avg = s[9] / (float)npix;                       // 0/0 -> NaN -> 0 (CTRL:138-139); x/0 -> inf is kept
            c_vs_grad_avg[n] += (avg != avg) ? 0.0f : avg;
            c_pos_grad[3 * n] += gt[0]; c_pos_grad[3 * n + 1] += gt[1]; c_pos_grad[3 * n + 2] += gt[2];
            c_pos_grad_norm[n] += sqrtf(gt[0] * gt[0] + gt[1] * gt[1] + gt[2] * gt[2]);
        }
    }   // m >= 0

    // ---- rows out: stage, then the wave writes its 64 consecutive rows of grad_feat as one contiguous run ----
    float* mine = sRows[wave] + lane * ROW_LDS;
#pragma unroll
    for (int k = 0; k < GS_NFEAT / 4; ++k)
        *reinterpret_cast<float4*>(mine + 4 * k) = make_float4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    const unsigned long long cam_m = gs_ballot(m >= 0);
    int* rank_to_lane = reinterpret_cast<int*>(sRows[wave] + 64 * ROW_LDS);
    if (m >= 0) rank_to_lane[__popcll(cam_m & ((1ull << lane) - 1ull))] = lane;
    __builtin_amdgcn_wave_barrier();
    const int64_t n0 = (int64_t)blockIdx.x * 256 + wave * 64;
    const int rows = (int)(N - n0 < 64 ? N - n0 : 64);               // wave-uniform; <= 0 for a wave past the end
    if (rows > 0) {
        float4* dst = reinterpret_cast<float4*>(grad_feat + (size_t)GS_NFEAT * n0);
        for (int e = lane; e < rows * (GS_NFEAT / 4); e += 64) {
            const int r = e / (GS_NFEAT / 4), c = e - r * (GS_NFEAT / 4);
            dst[e] = *reinterpret_cast<const float4*>(sRows[wave] + r * ROW_LDS + 4 * c);
        }
    }
    // the hook's gather (RAST:1131): the in-camera lanes of a wave own consecutive rows m of it
    if (hook_gfeat && cam_m != 0ull) {
        const int m_first = __builtin_amdgcn_readlane(m, __builtin_ctzll(cam_m));
        const int cnt = __popcll(cam_m);
        float4* dst = reinterpret_cast<float4*>(hook_gfeat + (size_t)GS_NFEAT * m_first);
        for (int e = lane; e < cnt * (GS_NFEAT / 4); e += 64) {
            const int r = e / (GS_NFEAT / 4), c = e - r * (GS_NFEAT / 4);
            dst[e] = *reinterpret_cast<const float4*>(sRows[wave] + rank_to_lane[r] * ROW_LDS + 4 * c);
        }
    }
}

void gs_launch_backward_blend(const GsBackwardArgs& a, hipStream_t s)
{
    if (a.T > 0 && a.K > 0) {
        // (the `visited` / `touched` flags are not cleared per backward: a flag counts only if it holds THIS backward's tag, a.gen)
        GS_TIMED(a.prof, KID_TILE_ORDER, s, k_tile_order<<<1, 1024, 0, s>>>(a.tile_work, a.T, a.tile_order, a.order_hint, a.n_heavy, a.heavy_factor_x2,
                                                                              a.tile_start, a.tile_end, a.cuts ? a.tile_cut : nullptr, a.cuts ? a.item_cap : gs_heavy_cap(a.T)));
        // workgroups: room for every segment of every heavy tile (no cuts: one item per heavy tile) + the ordinary work items four to a workgroup
        const unsigned groups = (unsigned)(a.cuts ? a.item_cap : gs_heavy_cap(a.T)) + (unsigned)(((size_t)a.T * (size_t)a.G + 3) / 4);
#define GS_BWD_LAUNCH(NQ_, STRICT_)                                                                                                    \
        GS_TIMED(a.prof, KID_BLEND_BWD, s, k_blend_bwd_tile<NQ_, STRICT_><<<groups, 256, 0, s>>>(a.tile_order, a.n_heavy, a.T, a.tile_start,  \
                 a.tile_end, a.vals_sorted, a.PA, a.PB, a.PC, a.box, a.offsets, a.grad_image, a.acc_alpha, a.last, a.W, a.H, a.tiles_x,   \
                 a.partial, a.visited, a.touched, a.gen, a.mag_image, a.cuts, a.cut_mag, a.tile_cut))
        if (a.G == 1) { if (a.strict) GS_BWD_LAUNCH(4, true); else GS_BWD_LAUNCH(4, false); }
        else if (a.G == 2) { if (a.strict) GS_BWD_LAUNCH(2, true); else GS_BWD_LAUNCH(2, false); }
        else { if (a.strict) GS_BWD_LAUNCH(1, true); else GS_BWD_LAUNCH(1, false); }
#undef GS_BWD_LAUNCH
        if (a.cuts && gs_heavy_cap(a.T) > 0) {                 // (fewer than eight tiles: no tile can be heavy)
#define GS_BWD_REPAIR(STRICT_)                                                                                                          \
            GS_TIMED(a.prof, KID_BLEND_BWD_REPAIR, s, k_blend_bwd_repair<STRICT_><<<(unsigned)gs_heavy_cap(a.T), 256, 0, s>>>(a.tile_order, a.n_heavy, a.G, \
                     a.tile_start, a.tile_end, a.vals_sorted, a.PA, a.PB, a.PC, a.box, a.offsets, a.grad_image, a.acc_alpha, a.last, a.W, a.H,  \
                     a.tiles_x, a.partial, a.visited, a.touched, a.gen, a.mag_image))
            if (a.strict) GS_BWD_REPAIR(true); else GS_BWD_REPAIR(false);
#undef GS_BWD_REPAIR
        }
    }
    else if (a.mag_image)
        (void)hipMemsetAsync(a.mag_image, 0, sizeof(float) * 2 * (size_t)a.H * (size_t)a.W, s);
    if (a.M > 0 && a.T > 0 && a.K > 0) {
        GsMagFold fold{};
        fold.first_block = (unsigned)(((size_t)a.M * 4 + 255) / 256);
        fold.n_heavy = a.n_heavy; fold.tile_order = a.tile_order; fold.tile_cut = a.tile_cut; fold.cut_mag = a.cut_mag;
        fold.mag_image = a.cuts ? a.mag_image : nullptr; fold.W = a.W; fold.H = a.H; fold.tiles_x = a.tiles_x;
        const unsigned fold_blocks = fold.mag_image ? (unsigned)gs_heavy_cap(a.T) : 0u;
        GS_TIMED(a.prof, KID_SUM_ROWS, s, k_sum_rows<<<fold.first_block + fold_blocks, 256, 0, s>>>(a.M, a.G, a.offsets, a.ntiles, a.partial, a.visited, a.touched,
                                                                                  a.gen, a.zero_row, a.sums, fold, a.max_tiles_hint));
    }
    else if (a.M > 0)
        (void)hipMemsetAsync(a.sums, 0, sizeof(float) * PW * (size_t)a.M, s);          // no pairs at all: every sum is zero
}

void gs_launch_backward_points(const GsBackwardArgs& a, hipStream_t s)
{
    const int nb = (int)((a.N + 255) / 256);
    if (nb == 0) return;
    int keep = a.sh_band <= 0 ? 1 : a.sh_band == 1 ? 4 : a.sh_band == 2 ? 9 : 16;
    GS_TIMED(a.prof, KID_BWD_POINTS, s, k_bwd_points<<<nb, 256, 0, s>>>(a.N, a.cam_index, a.sums, a.PD, a.point_cloud, a.features,
                                                                    a.object_id, a.Kmat, a.pose, keep, a.f_color, a.f_high, a.f_s, a.f_q, a.f_alpha,
                                                                    a.grad_pc, a.grad_feat, a.grad_uv, a.mag, a.n_affected,
                                                                    a.hook_gpc, a.hook_gfeat, a.hook_guv, a.hook_mag,
                                                                    a.hook_ids, a.hook_ntiles, a.hook_depth, a.hook_uv, a.PA, a.PB, a.ntiles,
                                                                    a.c_num_in_camera, a.c_num_pixels, a.c_vs_grad, a.c_vs_grad_avg, a.c_pos_grad, a.c_pos_grad_norm));
}
