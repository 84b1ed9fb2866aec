// k_binning.hip -- tile binning: key build, stable LSD radix sort, per-tile ranges.
//   k_keygen        generate_point_sort_key_by_num_overlap_tiles, RAST:131-172, with the
//                   exclusive scan of RAST:913-922 folded in (block-local scan + block offset)
//   k_sort_hist / k_sort_scatter   Tensor.sort() + gather, RAST:947-949 (stable: ties keep
//                   ascending in-camera offset)
//   (find_tile_start_and_end, RAST:175-193: a 64-ary search in k_blend_fwd's prologue, no kernel here)
//
// Keys are stored compactly: (tile_id << depth_bits) | depth_code, depth_bits = bits of the
// largest depth code in THIS frame, so only the significant bits are radix-sorted.  The order
// is identical to sorting the reference's 64-bit (tile_id << 32) + depth_code keys; exports
// rebuild those.  HBM-bound integer work: no LDS reshaping beyond per-wave digit counters.
#include "gs_common.h"
#include <type_traits>

#ifndef SORT_ROUNDS
#define SORT_ROUNDS 16
#endif
#define SORT_TILE (256 * SORT_ROUNDS)          // keys per tile: 256 threads x SORT_ROUNDS

// One lane per PAIR, not per point: a block scans its 256 points' tile counts, keeps {first pair, tile box, depth code} of
// every point in a small LDS table, and then walks the BLOCK's pairs e = 0, 1, 2 ... 256 at a time -- a thread finds the
// point that owns pair e (binary search over the 256 scanned counts), derives the tile from the pair's position in that
// point's box (RAST:163-168: tile_u outer, tile_v inner) and stores key and value at slot block_base + e.  The slots of a
// block are consecutive, so every store instruction writes 256 (or 512) consecutive bytes per wave whatever the splat sizes
// are; a splat covering thousands of tiles is simply many iterations in which all threads find the same owner -- shared by the
// block's four waves (walked by its own wave alone, one background splat over the whole image was 120 iterations of one wave
// against 8 for the others: k_keygen 51 us on the clustered workload with fewer pairs than the uniform one's 22 us).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_keygen(const int32_t* __restrict__ depth_codes, const ushort4* __restrict__ boxes,
                                                const int32_t* __restrict__ ntiles,
                                                const uint32_t* __restrict__ tile_block_sums,
                                                const int32_t* __restrict__ block_offsets, const int32_t* __restrict__ block_counts,
                                                int M, int tiles_x,
                                                float depth_scale, int depth_bits, uint32_t K_cap,
                                                uint32_t* __restrict__ offsets, KeyT* __restrict__ keys,
                                                int32_t* __restrict__ vals,
                                                GsCounters* __restrict__ counters, volatile GsCounters* host_mirror, int32_t ticket)
{
    __shared__ uint32_t ws[4];
    __shared__ uint32_t wpre[4];
    __shared__ uint32_t sExcl[256 + 1];
    __shared__ ushort4 sBox[256];
    __shared__ KeyT sCode[256];
    // same blocks as the kernel that produced ntiles / tile_block_offsets: k_project's (256 rows of the point cloud each, in-camera
    // points dense from block_offsets[b]) or, for records that arrived from elsewhere, 256 consecutive records
    const int first = block_offsets ? block_offsets[blockIdx.x] : (int)blockIdx.x * 256;
    const int mine = block_offsets ? block_counts[blockIdx.x] : min(256, M - first);
    const int idx = (int)threadIdx.x < mine ? first + (int)threadIdx.x : M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool valid = idx < M;
    const uint32_t n = valid ? (uint32_t)ntiles[idx] : 0u;
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) ws[wave] = incl;
    sBox[threadIdx.x] = valid ? boxes[idx] : make_ushort4(0, 1, 0, 1);
    sCode[threadIdx.x] = valid ? (KeyT)(uint32_t)depth_codes[idx] : (KeyT)0;     // i32(depth * scale), RAST:159-160, from k_project
    // first pair of this block = the tile counts of all blocks before it (RAST:913-922 across blocks): a few thousand L2-resident
    // counters summed by the block itself, instead of a scan launch in between
    uint32_t pre = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) pre += tile_block_sums[j];
    pre = (uint32_t)gs_wave_sum_i((int)pre);
    if (lane == 0) wpre[wave] = pre;
    __syncthreads();
    const uint32_t block_base = wpre[0] + wpre[1] + wpre[2] + wpre[3];
    const uint32_t block_total = ws[0] + ws[1] + ws[2] + ws[3];
    // the last block knows K = its base + its own count: it hands the frame counters to the host when the launch was queued
    // before the host had them (predicted sizing; k_project.hip: gs_publish_counters)
    if (host_mirror && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        gs_publish_counters(counters, block_base + block_total, host_mirror, ticket);
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += ws[w];
    const uint32_t excl = woff + incl - n;                                        // first pair of this point inside the block
    sExcl[threadIdx.x] = excl;
    if (threadIdx.x == 0) sExcl[256] = 0xffffffffu;
    if (valid) offsets[idx] = block_base + excl;                                  // RAST:913-922 (also the backward's row slots)
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < block_total; e += 256u) {
        int j = 0;                                                                // owner: the last point whose first pair is <= e
#pragma unroll
        for (int step = 128; step > 0; step >>= 1) if (sExcl[j + step] <= e) j += step;
        const uint32_t t = e - sExcl[j];                                          // position in the owner's box
        const ushort4 bx = sBox[j];
        const uint32_t dv = (uint32_t)((int)bx.w - (int)bx.z);
        // t / dv: float estimate + one correction step either way (exact below 2^22; a box has at most tiles_x * tiles_y entries)
        uint32_t tq, tr;
        if (t < (1u << 22)) {
            tq = (uint32_t)((float)t * __builtin_amdgcn_rcpf((float)dv));
            int r = (int)t - (int)(tq * dv);
            if (r < 0) { tq -= 1u; r += (int)dv; }
            if (r >= (int)dv) { tq += 1u; r -= (int)dv; }
            tr = (uint32_t)r;
        } else { tq = t / dv; tr = t - tq * dv; }
        const KeyT tile_id = (KeyT)(((uint32_t)bx.x + tq) + ((uint32_t)bx.z + tr) * (uint32_t)tiles_x);   // RAST:163-168
        const uint32_t slot = block_base + e;
        if (slot < K_cap) {
            keys[slot] = (tile_id << depth_bits) | sCode[j];                     // RAST:169-170, compact form
            vals[slot] = first + j;
        }
    }
    (void)depth_scale;
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_sort_hist(const KeyT* __restrict__ keys, const GsCounters* __restrict__ ctr, uint32_t n_cap, int shift,
                                                   uint32_t* __restrict__ hist, int nblocks, int tiles_per_block)
{
    __shared__ uint32_t lh[256];
    const uint32_t n = min(ctr->K, n_cap);      // the frame's pair count, on the device (the launch may have been sized before the host knew it)
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t begin = (uint64_t)blockIdx.x * tiles_per_block * SORT_TILE;     // multiple of 4096: 16-byte aligned
    uint64_t end = begin + (uint64_t)tiles_per_block * SORT_TILE;
    if (end > n) end = n;
    // four keys per load (16 or 32 bytes per lane), then four LDS atomics
    constexpr int V = 4;
    struct alignas(sizeof(KeyT) * V) KeyVec { KeyT k[V]; };
    const uint64_t nvec = begin < end ? (end - begin) / V : 0;
    const KeyVec* kv = reinterpret_cast<const KeyVec*>(keys + begin);
    for (uint64_t i = threadIdx.x; i < nvec; i += 256) {
        const KeyVec x = kv[i];
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&lh[(uint32_t)(x.k[j] >> shift) & 255u], 1u);
    }
    for (uint64_t i = begin + nvec * V + threadIdx.x; i < end; i += 256)
        atomicAdd(&lh[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = lh[threadIdx.x];
}

// The [digit][block] table of per-block counts becomes global offsets in ONE small launch plus a few instructions of
// the scatter's prologue: k_sort_rowscan (one block per digit) scans the digit's own row exclusively and leaves the
// row's total in `totals`; every scatter block then adds, for each digit, the totals of all smaller digits (a
// 256-element block scan).  No atomics, no cross-block waiting, fixed order.
__global__ __launch_bounds__(1024) void k_sort_rowscan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ totals,
                                                       uint32_t* __restrict__ offsets_out, int nblocks)
{
    __shared__ uint32_t wsum[16];
    const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t carry = 0;
    const uint32_t* row = hist + (size_t)d * nblocks;
    uint32_t* orow = offsets_out + (size_t)d * nblocks;
    for (int start = 0; start < nblocks; start += 1024) {
        const int i = start + t;
        const uint32_t x = i < nblocks ? row[i] : 0u;
        uint32_t incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (i < nblocks) orow[i] = carry + woff + incl - x;
        uint32_t tot = 0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        carry += tot;
    }
    if (t == 0) totals[d] = carry;
}

// Stable scatter of one radix pass.  Per 4096-key tile: (1) every wave ranks its 1024 keys with
// ballot matching (8 ballots -> lanes with the same digit -> rank = popcount of lower lanes) and
// per-wave digit counters in LDS; (2) the tile's pairs are written to LDS in their sorted order;
// (3) the block copies them out, so that each digit's run is written to HBM by consecutive lanes
// (coalesced runs instead of 4-byte scatters).  Tiles of a block are processed in order with a
// running global offset per digit, which keeps the pass stable.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_sort_scatter(const KeyT* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                      KeyT* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                      const GsCounters* __restrict__ ctr, uint32_t n_cap, int shift, const uint32_t* __restrict__ hist_scanned,
                                                      const uint32_t* __restrict__ totals, int nblocks, int tiles_per_block)
{
    const uint32_t n = min(ctr->K, n_cap);
    __shared__ uint32_t cnt[4][256];      // per-wave digit counts, then per-wave start inside the tile
    __shared__ uint32_t gbase[256];       // running global offset of each digit for this block
    __shared__ uint32_t dstart[256];      // start of each digit inside the sorted tile
    __shared__ uint32_t wtot[4];
    __shared__ KeyT skeys[SORT_TILE];
    __shared__ int32_t svals[SORT_TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    {   // global start of digit t = totals of all smaller digits (exclusive scan of the 256 row totals) + this block's offset in the row
        const uint32_t tot = totals[t];
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        gbase[t] = (woff + incl - tot) + hist_scanned[t * nblocks + blockIdx.x];
        __syncthreads();
    }
    // one 4096-pair tile; FULL (every tile but the last of the array) drops all bounds predicates
    auto do_tile = [&](auto full_tag, const uint64_t tile_base, const uint32_t tile_n) {
        constexpr bool FULL = decltype(full_tag)::value;
        cnt[0][t] = 0; cnt[1][t] = 0; cnt[2][t] = 0; cnt[3][t] = 0;
        __syncthreads();
        KeyT k[SORT_ROUNDS];
        int32_t v[SORT_ROUNDS];
        uint32_t rank[SORT_ROUNDS];
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; ++r) {
            const uint32_t li = (uint32_t)wave * (SORT_ROUNDS * 64) + r * 64 + lane;
            const bool valid = FULL || li < tile_n;
            k[r] = valid ? keys_in[tile_base + li] : (KeyT)0;
            v[r] = valid ? vals_in[tile_base + li] : 0;
        }
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; ++r) {
            const uint32_t li = (uint32_t)wave * (SORT_ROUNDS * 64) + r * 64 + lane;
            const bool valid = FULL || li < tile_n;
            const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
            // lanes with the same digit, as two 32-bit halves per lane: per bit one vote, then same &= ~(vote ^ m) with m = all
            // ones where the lane's bit is set (v_bfe_i32, v_xnor_b32, v_and_b32 per half)
            const unsigned long long vm = gs_ballot(valid);
            uint32_t slo = (uint32_t)vm, shi = (uint32_t)(vm >> 32);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const int32_t mb = (int32_t)(d << (31 - bit)) >> 31;
                const unsigned long long bal = gs_ballot(mb < 0);
                slo &= ~((uint32_t)bal ^ (uint32_t)mb);
                shi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)mb);
            }
            const uint32_t in_round = (uint32_t)__popc(slo & (uint32_t)lt_mask) + (uint32_t)__popc(shi & (uint32_t)(lt_mask >> 32));
            const uint32_t before = valid ? cnt[wave][d] : 0u;
            if (valid && in_round == 0) cnt[wave][d] = before + (uint32_t)__popc(slo) + (uint32_t)__popc(shi);
            rank[r] = before + in_round;
        }
        __syncthreads();
        // thread t owns digit t: tile total, exclusive scan over digits, per-wave starts
        const uint32_t c0 = cnt[0][t], c1 = cnt[1][t], c2 = cnt[2][t], c3 = cnt[3][t];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        const uint32_t ds = woff + incl - tot;
        dstart[t] = ds;
        cnt[0][t] = ds; cnt[1][t] = ds + c0; cnt[2][t] = ds + c0 + c1; cnt[3][t] = ds + c0 + c1 + c2;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; ++r) {
            const uint32_t li = (uint32_t)wave * (SORT_ROUNDS * 64) + r * 64 + lane;
            if (FULL || li < tile_n) {
                const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
                const uint32_t ipos = cnt[wave][d] + rank[r];
                skeys[ipos] = k[r];
                svals[ipos] = v[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; ++r) {
            const uint32_t idx = (uint32_t)r * 256u + t;
            if (FULL || idx < tile_n) {
                const KeyT kk = skeys[idx];
                const uint32_t d = (uint32_t)(kk >> shift) & 255u;
                const uint32_t pos = gbase[d] + (idx - dstart[d]);
                keys_out[pos] = kk;
                vals_out[pos] = svals[idx];
            }
        }
        __syncthreads();
        gbase[t] += tot;
        __syncthreads();
    };
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const uint64_t tile_base = ((uint64_t)blockIdx.x * tiles_per_block + tile) * SORT_TILE;
        if (tile_base >= n) break;
        const uint32_t tile_n = (uint32_t)((n - tile_base) < SORT_TILE ? (n - tile_base) : SORT_TILE);
        if (tile_n == SORT_TILE) do_tile(std::true_type{}, tile_base, tile_n);
        else do_tile(std::false_type{}, tile_base, tile_n);
    }
}

static void sort_geometry(uint32_t K, int* nblocks, int* tiles_per_block)
{
    uint32_t tiles = (K + SORT_TILE - 1) / SORT_TILE;
    uint32_t target = tiles < 1024u ? tiles : 1024u;
    if (target == 0) target = 1;
    *tiles_per_block = (int)((tiles + target - 1) / target);
    if (*tiles_per_block < 1) *tiles_per_block = 1;
    *nblocks = (int)((tiles + *tiles_per_block - 1) / *tiles_per_block);
    if (*nblocks < 1) *nblocks = 1;
}

size_t gs_sort_hist_elems(uint32_t K)
{
    int nb, tpb;
    sort_geometry(K, &nb, &tpb);
    return (size_t)2 * 256 * nb;      // raw counts + scanned offsets
}

size_t gs_scan_tmp_elems(size_t) { return 256; }      // the 256 row totals of k_sort_rowscan

template <typename KeyT>
static void launch_binning_t(const GsBinArgs& a, hipStream_t s)
{
    KeyT* keys_a = reinterpret_cast<KeyT*>(a.keys_a);
    KeyT* keys_b = reinterpret_cast<KeyT*>(a.keys_b);
    *a.keys_sorted = a.keys_a;
    *a.vals_sorted = a.vals_a;
    if (a.N == 0 || a.M == 0) return;
    const unsigned kg_blocks = a.block_offsets ? (unsigned)((a.N + 255) / 256) : (unsigned)((a.M + 255) / 256);
    GS_TIMED(a.prof, KID_KEYGEN, s, k_keygen<KeyT><<<kg_blocks, 256, 0, s>>>(
        a.depth_codes, a.box, a.ntiles, a.tile_block_sums, a.block_offsets, a.block_counts, a.M, a.tiles_x, a.depth_scale, a.depth_bits, a.K, a.offsets, keys_a, a.vals_a,
        a.counters_rw, a.host_mirror, a.ticket));
    if (a.K == 0) return;
    int nb, tpb;
    sort_geometry(a.K, &nb, &tpb);
    KeyT *kin = keys_a, *kout = keys_b;
    int32_t *vin = a.vals_a, *vout = a.vals_b;
    for (int shift = 0; shift < a.key_bits; shift += 8) {
        uint32_t* offs = a.hist + (size_t)256 * nb;                 // second half of the table: scanned offsets
        GS_TIMED(a.prof, KID_SORT_HIST, s, k_sort_hist<KeyT><<<nb, 256, 0, s>>>(kin, a.counters, a.K, shift, a.hist, nb, tpb));
        GS_TIMED(a.prof, KID_SORT_ROWSCAN, s, k_sort_rowscan<<<256, 1024, 0, s>>>(a.hist, a.scan_tmp, offs, nb));
        GS_TIMED(a.prof, KID_SORT_SCATTER, s, k_sort_scatter<KeyT><<<nb, 256, 0, s>>>(kin, vin, kout, vout, a.counters, a.K, shift, offs, a.scan_tmp, nb, tpb));
        KeyT* tk = kin; kin = kout; kout = tk;
        int32_t* tv = vin; vin = vout; vout = tv;
    }
    *a.keys_sorted = kin;
    *a.vals_sorted = vin;
    // (find_tile_start_and_end, RAST:175-193, has no launch of its own: k_blend_fwd's blocks look their range up in these keys)
}

void gs_launch_binning(const GsBinArgs& a, hipStream_t s)
{
    // tile_start | tile_end (RAST:954-957 zero-init) | tile_work were cleared by the frame's first kernel (k_filter / k_boxes_from_records)
    if (a.key64) launch_binning_t<uint64_t>(a, s);
    else launch_binning_t<uint32_t>(a, s);
}
