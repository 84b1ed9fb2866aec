// gs_api.hip -- the C ABI of libgsrast.so (include/gs_rasterizer.h): context, device
// arena, frame pool and the forward / backward orchestration that replaces
// _module_function.forward / .backward of the reference (RAST:830-1163).
#include "../../include/gs_rasterizer.h"
#include "gs_common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>
#include <cstdlib>

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(e__ == hipErrorOutOfMemory ? GS_ERR_OUT_OF_MEMORY : GS_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                       \
    } while (0)

// Grow-only device buffer.  Growth (hipFree + hipMalloc) happens only when a frame is larger
// than anything seen before; steady-state frames allocate nothing.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, int64_t* total)
    {
        if (bytes <= cap) return hipSuccess;
        size_t want = bytes + bytes / 4 + 256;     // 25 % slack: K drifts from frame to frame
        if (p) { (void)hipFree(p); *total -= (int64_t)cap; p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want; *total += (int64_t)want;
        return hipSuccess;
    }
    void release(int64_t* total) { if (p) { (void)hipFree(p); *total -= (int64_t)cap; } p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// What RAST:998-1019 saves for backward, in this library's layouts (DESIGN.md "HBM layout").
struct FrameBufs {
    DevBuf mask, ids, cam_index, rec, box, ntiles, depth_codes, offsets, keys_a, keys_b, vals_a, vals_b, tile_start, pose, tile_order, cuts, cut_mag;
    bool in_use = false;
    void release(int64_t* total)
    {
        DevBuf* all[] = { &mask, &ids, &cam_index, &rec, &box, &ntiles, &depth_codes, &offsets, &keys_a, &keys_b, &vals_a, &vals_b,
                          &tile_start, &pose, &tile_order, &cuts, &cut_mag };
        for (DevBuf* b : all) b->release(total);
    }
};

// A frame as the library sees it.  Callers hold an opaque ticket (index + generation, encoded in the gs_frame* value)
// that is resolved under the ctx mutex on every use, so a released, recycled or foreign handle is an error, never a
// dereference of freed memory.
struct Frame {
    FrameBufs* bufs = nullptr;
    gs_frame_info info{};
    int depth_bits = 0;
    void* keys_sorted = nullptr;
    int key64 = 0;
    int32_t* vals_sorted = nullptr;
    bool live = false;
    int cut_cap = 0;                    // list-cut records the forward of this frame could claim (0: it wrote none)
    int bwd_reference_order = 0;        // gs_config.bwd_reference_order of the forward that made the frame (gs_backward_projected has no config)
    bool max_tiles_known = false;       // k_project of this frame left the largest tile count of one point in the tile arrays (frames from records: no)
    uint32_t generation = 0;
    // gs_project_shard_begin: the hand-over of M (and the object-id check) has not been read yet; slot of the pinned counters
    // the frame's publish kernel writes, its ticket and the stream to fall back on
    int pending_slot = -1;
    int32_t pending_ticket = 0;
    hipStream_t pending_stream = nullptr;
};

struct GsProf {
    uint64_t mask = 0;
    struct Rec { int kid; hipEvent_t a, b; };
    std::vector<Rec> recs;          // records in flight since the last read
    std::vector<hipEvent_t> spare;  // recycled events
    double total_ms[KID_COUNT_] = {};
    int64_t launches[KID_COUNT_] = {};
};

int gs_prof_begin(GsProf* p, int kid, hipStream_t s)
{
    if (!p || !((p->mask >> kid) & 1ull)) return -1;
    GsProf::Rec r; r.kid = kid;
    hipEvent_t* ev[2] = { &r.a, &r.b };
    for (hipEvent_t* e : ev) {
        if (!p->spare.empty()) { *e = p->spare.back(); p->spare.pop_back(); }
        else if (hipEventCreate(e) != hipSuccess) return -1;
    }
    (void)hipEventRecord(r.a, s);
    p->recs.push_back(r);
    return (int)p->recs.size() - 1;
}

void gs_prof_end(GsProf* p, int rec, hipStream_t s)
{
    if (!p || rec < 0) return;
    (void)hipEventRecord(p->recs[rec].b, s);
}

#define GS_COUNTER_SLOTS 64
struct gs_ctx {
    int device = 0;
    GsProf prof;
    std::mutex mu;
    int64_t device_bytes = 0;
    std::vector<FrameBufs*> pool;
    std::vector<Frame*> frames;         // slot i of the ticket space (recycled, generation-tagged)
    int transient = -1;                 // slot of the frame of the last keep_for_backward == 0 call
    // scratch shared by all frames (stream ordered)
    DevBuf block_counts, block_offsets, tile_block_sums, hist, scan_tmp, counters, partial, visited, zero_row, sums, loss_ws;
    uint8_t visit_gen = 0;                 // tag of the last backward's flags in `visited` (0: the buffer is all zero)
    GsCounters* host_counters = nullptr;   // pinned, device-visible, GS_COUNTER_SLOTS of them; written by gs_publish_counters (k_keygen's last block or k_scan_tiles_publish)
    GsCounters* host_counters_dev = nullptr;   // the device's address of it
    uint64_t slots_busy = 1ull;            // slot 0 serves the calls that wait at once; the others belong to frames begun and not yet read
    int32_t ticket = 0;                    // sequence number of the last forward
    int64_t counter_wait_ns = 0;           // host time spent waiting for frame counters so far (gs_ctx_counter_wait_ns)
    // Predicted sizing of the per-pixel half (run_forward_tail): what the last frame of this ctx needed, for an image of this
    // size.  The next frame's binning, sort and blend are queued on it without waiting for the frame's own counters.
    struct { bool valid = false; int H = 0, W = 0; uint32_t K = 0; int max_code = 0; } seen;
    // dispatch order for the next forward blend: the last backward's tile order (heaviest first).  Only ever a complete
    // permutation of [0, order_hint_T) written by k_tile_order; 0 = none.  A hint only moves work in time.
    DevBuf order_hint;
    int order_hint_T = 0;
    // stream hand-over: the scratch above is recycled in stream order, so work arriving on another stream waits for
    // everything issued on the previous one
    bool has_stream = false;
    hipStream_t last_stream = nullptr;
    hipEvent_t switch_event = nullptr;
};

extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }
extern "C" const char* gs_last_error(void) { return g_last_error.c_str(); }

extern "C" const char* gs_kernel_names(void)
{
    return "k_filter,k_scan_tiles_publish,k_project,k_keygen,k_sort_hist,"
           "k_sort_rowscan,k_sort_scatter,k_blend_fwd,k_blend_bwd_tile,k_bwd_points,k_sum_rows,k_tile_order,k_blend_bwd_repair";
}

extern "C" int gs_create(int32_t device, gs_ctx** out)
{
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: out is NULL");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    gs_ctx* c = new gs_ctx();
    c->device = device;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->host_counters), sizeof(GsCounters) * GS_COUNTER_SLOTS, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { delete c; return fail(GS_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    std::memset(c->host_counters, 0, sizeof(GsCounters) * GS_COUNTER_SLOTS);
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->host_counters_dev), c->host_counters, 0);
    if (e != hipSuccess) { (void)hipHostFree(c->host_counters); delete c; return fail(GS_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e)); }
    e = c->counters.ensure(sizeof(GsCounters), &c->device_bytes);
    if (e == hipSuccess) e = hipMemset(c->counters.p, 0, sizeof(GsCounters));      // every later frame leaves them reset (gs_publish_counters)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->switch_event, hipEventDisableTiming);
    if (e != hipSuccess) { c->counters.release(&c->device_bytes); (void)hipHostFree(c->host_counters); delete c; return fail(GS_ERR_OUT_OF_MEMORY, "gs_create: counters"); }
    *out = c;
    return GS_OK;
}

extern "C" int gs_destroy(gs_ctx* c)
{
    if (!c) return GS_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (FrameBufs* b : c->pool) { b->release(&c->device_bytes); delete b; }
    for (Frame* f : c->frames) delete f;
    DevBuf* all[] = { &c->block_counts, &c->block_offsets, &c->tile_block_sums, &c->hist, &c->scan_tmp,
                      &c->counters, &c->partial, &c->visited, &c->zero_row, &c->sums, &c->loss_ws, &c->order_hint };
    for (DevBuf* b : all) b->release(&c->device_bytes);
    for (GsProf::Rec& r : c->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : c->prof.spare) (void)hipEventDestroy(e);
    if (c->switch_event) (void)hipEventDestroy(c->switch_event);
    if (c->host_counters) (void)hipHostFree(c->host_counters);
    delete c;
    return GS_OK;
}

extern "C" int gs_profile_enable(gs_ctx* c, uint64_t kernel_mask)
{
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_profile_enable: ctx is NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    c->prof.mask = kernel_mask;
    return GS_OK;
}

extern "C" int gs_profile_read(gs_ctx* c, double* total_ms, int64_t* launches, int32_t n, int32_t reset)
{
    if (!c || !total_ms || !launches) return fail(GS_ERR_INVALID_ARGUMENT, "gs_profile_read: NULL argument");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    GsProf& p = c->prof;
    for (GsProf::Rec& r : p.recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        p.total_ms[r.kid] += ms; p.launches[r.kid] += 1;
        p.spare.push_back(r.a); p.spare.push_back(r.b);
    }
    p.recs.clear();
    for (int i = 0; i < n && i < KID_COUNT_; ++i) { total_ms[i] = p.total_ms[i]; launches[i] = p.launches[i]; }
    if (reset) for (int i = 0; i < KID_COUNT_; ++i) { p.total_ms[i] = 0.0; p.launches[i] = 0; }
    return GS_OK;
}

extern "C" int64_t gs_ctx_device_bytes(const gs_ctx* c) { return c ? c->device_bytes : 0; }
extern "C" int64_t gs_ctx_counter_wait_ns(const gs_ctx* c) { return c ? c->counter_wait_ns : 0; }

// ---- frame tickets ------------------------------------------------------------------------------------------------
static gs_frame* ticket_of(int slot, uint32_t generation)
{
    return reinterpret_cast<gs_frame*>((uintptr_t)(((uint64_t)generation << 32) | (uint64_t)(uint32_t)(slot + 1)));
}

// resolves a ticket (mutex held); nullptr for anything that is not a live frame of THIS ctx
static Frame* resolve(gs_ctx* c, const gs_frame* h, int* slot_out = nullptr)
{
    const uint64_t v = (uint64_t)(uintptr_t)h;
    const uint32_t lo = (uint32_t)v, gen = (uint32_t)(v >> 32);
    if (lo == 0 || lo > c->frames.size()) return nullptr;
    Frame* f = c->frames[lo - 1];
    if (!f->live || f->generation != gen) return nullptr;
    if (slot_out) *slot_out = (int)lo - 1;
    return f;
}

static FrameBufs* acquire_bufs(gs_ctx* c)
{
    for (FrameBufs* b : c->pool) if (!b->in_use) { b->in_use = true; return b; }
    FrameBufs* b = new FrameBufs();
    b->in_use = true;
    c->pool.push_back(b);
    return b;
}

static Frame* acquire_frame(gs_ctx* c, int* slot)
{
    for (size_t i = 0; i < c->frames.size(); ++i)
        if (!c->frames[i]->live) { Frame* f = c->frames[i]; f->live = true; f->generation += 1; if (f->generation == 0) f->generation = 1; *slot = (int)i; return f; }
    Frame* f = new Frame();
    f->live = true; f->generation = 1;
    c->frames.push_back(f);
    *slot = (int)c->frames.size() - 1;
    return f;
}

static void drop_frame(gs_ctx* c, Frame* f)
{
    if (!f || !f->live) return;
    if (f->pending_slot > 0) c->slots_busy &= ~(1ull << f->pending_slot);      // a late write into a freed slot is harmless: tickets are unique
    f->pending_slot = -1;
    if (f->bufs) f->bufs->in_use = false;
    f->bufs = nullptr; f->live = false;
    for (size_t i = 0; i < c->frames.size(); ++i) if (c->frames[i] == f && c->transient == (int)i) c->transient = -1;
}

// Entering a call that launches on stream s (mutex held).
static hipError_t enter_stream(gs_ctx* c, hipStream_t s)
{
    if (c->has_stream && c->last_stream != s) {
        hipError_t e = hipEventRecord(c->switch_event, c->last_stream);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(s, c->switch_event, 0);
        if (e != hipSuccess) return e;
    }
    c->has_stream = true; c->last_stream = s;
    return hipSuccess;
}

static int waves_per_tile(int n_tiles);
static int bits_for(uint32_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b < 1 ? 1 : b; }

#define ENSURE(buf, bytes)                                                                 \
    do {                                                                                   \
        hipError_t e__ = (buf).ensure((bytes), &c->device_bytes);                          \
        if (e__ != hipSuccess) { drop_frame(c, f); return fail(GS_ERR_OUT_OF_MEMORY, std::string("device allocation failed: " #buf)); } \
    } while (0)

#define HIP_TRY_F(expr)                                                                            \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            drop_frame(c, f);                                                                      \
            return fail(GS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));          \
        }                                                                                          \
    } while (0)

static void set_records(GsProjectArgs& pa, const FrameBufs& B, size_t rows)
{   // GS_RS == 4: one 64-byte row per point; GS_RS == 1: four planes of `rows` records
    float4* rec = B.rec.as<float4>();
    const size_t plane = GS_RS == 4 ? 1 : rows;
    pa.PA = rec; pa.PB = rec + plane; pa.PC = rec + 2 * plane; pa.PD = rec + 3 * plane;
}

// The one device->host hand-over of a frame: M, K, the depth-code range (and the bad-object-id count).  The last
// prologue kernel writes them into pinned host memory and then the ticket; spinning on it costs a few microseconds
// where a copy + stream synchronisation left the GPU idle for ~30.
static int wait_counters(gs_ctx* c, hipStream_t s, int32_t ticket, int slot = 0)
{
    static const bool wait_on_stream = []{ const char* e = getenv("GS_COUNTERS_WAIT"); return e && std::strcmp(e, "stream") == 0; }();
    if (wait_on_stream) {                   // diagnostic alternative: block in the runtime instead of spinning
        const auto t0s = std::chrono::steady_clock::now();
        HIP_TRY(hipStreamSynchronize(s));
        c->counter_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0s).count();
        return GS_OK;
    }
    volatile GsCounters* hc = c->host_counters + slot;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0; hc->reserved != ticket; ++spin) {
        if ((spin & 0xfffu) == 0xfffu) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) {                         // stream drained: the ticket must be there now
                if (hc->reserved != ticket) return fail(GS_ERR_HIP, "frame counters were not published");
                break;
            }
            if (q != hipErrorNotReady) return fail(GS_ERR_HIP, std::string("waiting for the frame counters: ") + hipGetErrorString(q));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                return fail(GS_ERR_HIP, "timed out waiting for the frame counters");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    c->counter_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return GS_OK;
}

static int check_geometry(const gs_camera* cam, const gs_config* cfg, const char* who, int* tiles_x, int* tiles_y)
{
    const int H = cam->camera_height, W = cam->camera_width;
    if (W <= 0 || H <= 0 || (!cfg->allow_partial_tiles && (W % GS_TILE != 0 || H % GS_TILE != 0)))        // RAST:1193-1194
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": camera_width and camera_height must be positive multiples of 16 "
                                             "(or set gs_config.allow_partial_tiles)");
    *tiles_x = (W + GS_TILE - 1) / GS_TILE; *tiles_y = (H + GS_TILE - 1) / GS_TILE;
    if (*tiles_x > 65535 || *tiles_y > 65535) return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": image too large");
    return GS_OK;
}

static int check_scene(const gs_scene* sc, const gs_camera* cam, const char* who)
{
    const int64_t N = sc->n_points;
    if (N < 0 || N >= (int64_t)1 << 31) return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": n_points out of range");
    if (cam->n_objects <= 0) return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": n_objects must be >= 1");
    if (N > 0 && (!sc->point_cloud || !sc->point_cloud_features || !sc->point_invalid_mask || !sc->point_object_id))
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL scene array");
    if (!cam->q_pointcloud_camera || !cam->t_pointcloud_camera || !cam->camera_intrinsics)
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL camera array");
    if (((uintptr_t)sc->point_cloud_features & 15u) != 0)
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": point_cloud_features must be 16-byte aligned");
    return GS_OK;
}

static int check_forward_out(const gs_forward_out* out, const gs_config* cfg, int keep, const char* who)
{
    if (!out->rasterized_image) return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": rasterized_image is NULL");
    if (!cfg->rgb_only && (!out->rasterized_depth || !out->pixel_accumulated_alpha ||
                           !out->pixel_offset_of_last_effective_point || !out->pixel_valid_point_count))
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": an output is NULL although rgb_only is false");
    if (keep && cfg->rgb_only) return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": rgb_only frames cannot be kept for backward (RAST:478-484)");
    return GS_OK;
}

// ---- per-point half: filter, compaction, projection (+ tile counts, scan, publication) ----------------------------
// On success the frame's buffers hold mask / ids / cam_index / records / box / ntiles.  The counters (M, K, depth-code range,
// bad object ids) are handed over through pinned memory; `wait` says when the host reads them:
//   WAIT_NOW    before this function returns (M, K, max_code are filled in);
//   WAIT_LATER  the caller reads slot 0 itself (read_counters) after it has queued more work -- nothing else may publish before;
//   WAIT_FRAME  the frame keeps a slot of its own and whoever needs M first reads it (resolve_pending, gs_project_shard_begin).
enum CounterWait { WAIT_NOW, WAIT_LATER, WAIT_FRAME };

static int read_counters(gs_ctx* c, Frame* f, hipStream_t s, int32_t ticket, int* M_out, uint32_t* K_out, int* max_code_out)
{
    const int rc = wait_counters(c, s, ticket);
    if (rc != GS_OK) { drop_frame(c, f); return rc; }
    if (c->host_counters->bad_object_ids != 0) {
        drop_frame(c, f);
        return fail(GS_ERR_INVALID_ARGUMENT, "point_object_id holds " + std::to_string(c->host_counters->bad_object_ids) +
                                             " value(s) outside [0, n_objects) on valid rows");
    }
    *M_out = c->host_counters->M; *K_out = c->host_counters->K; *max_code_out = c->host_counters->max_depth_code;
    return GS_OK;
}

static int run_project_stage(gs_ctx* c, Frame* f, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg, int T,
                             hipStream_t s, GsProjectArgs* pa_out, int* M_out, uint32_t* K_out, int* max_code_out, CounterWait wait = WAIT_NOW)
{
    FrameBufs& B = *f->bufs;
    const int64_t N = sc->n_points;
    const int H = cam->camera_height, W = cam->camera_width;
    const size_t nb = (size_t)((N + 255) / 256);
    const size_t Np = (size_t)(N > 0 ? N : 1);
    ENSURE(B.mask, Np); ENSURE(B.ids, 4 * Np); ENSURE(B.cam_index, 4 * Np);
    ENSURE(B.rec, 64 * Np);
    ENSURE(B.box, 8 * Np); ENSURE(B.ntiles, 4 * Np); ENSURE(B.depth_codes, 4 * Np); ENSURE(B.offsets, 4 * Np);
    ENSURE(B.tile_start, 4 * GS_TILE_INTS(T));    // tile_start | tile_end | tile_work | tile_cut | cut_alloc, cleared together
    ENSURE(B.tile_order, 4 * GS_ORDER_INTS(T));   // + heavy-tile count, item count and item bases behind the order
    ENSURE(B.pose, sizeof(GsPose) * (size_t)cam->n_objects);
    ENSURE(c->block_counts, 4 * (nb + 1)); ENSURE(c->block_offsets, 4 * (nb + 1));
    ENSURE(c->tile_block_sums, 4 * (nb + 1));

    GsProjectArgs pa{};
    pa.prof = &c->prof;
    pa.point_cloud = sc->point_cloud; pa.features = sc->point_cloud_features; pa.invalid = sc->point_invalid_mask;
    pa.object_id = sc->point_object_id; pa.N = N; pa.q_pc = cam->q_pointcloud_camera; pa.t_pc = cam->t_pointcloud_camera;
    pa.n_objects = cam->n_objects; pa.Kmat = cam->camera_intrinsics; pa.H = H; pa.W = W;
    pa.near_plane = cfg->near_plane; pa.far_plane = cfg->far_plane; pa.depth_scale = cfg->depth_to_sort_key_scale;
    pa.pose = B.pose.as<GsPose>(); pa.mask = B.mask.as<int8_t>();
    pa.block_counts = c->block_counts.as<int32_t>(); pa.block_offsets = c->block_offsets.as<int32_t>();
    pa.ids = B.ids.as<int32_t>(); pa.cam_index = B.cam_index.as<int32_t>();
    set_records(pa, B, Np);
    pa.box = B.box.as<ushort4>(); pa.ntiles = B.ntiles.as<int32_t>(); pa.depth_codes = B.depth_codes.as<int32_t>();
    pa.tile_block_sums = c->tile_block_sums.as<uint32_t>();
    pa.counters = c->counters.as<GsCounters>();
    pa.tile_arrays = B.tile_start.as<int32_t>(); pa.tile_ints = (int)GS_TILE_INTS(T);
    int slot = 0;
    if (wait == WAIT_FRAME && N > 0) {
        const uint64_t free_slots = ~c->slots_busy;
        if (free_slots != 0ull) { slot = __builtin_ctzll(free_slots); c->slots_busy |= 1ull << slot; }     // none left: wait at once
    }
    pa.host_mirror = c->host_counters_dev + slot; pa.ticket = ++c->ticket;
    if (c->ticket == 0x7fffffff) c->ticket = 0;
    gs_launch_project(pa, s, wait != WAIT_LATER);      // WAIT_LATER: the caller decides who publishes the counters (run_forward_tail)
    f->max_tiles_known = true;
    HIP_TRY_F(hipGetLastError());
    *pa_out = pa;
    *M_out = 0; *K_out = 0u; *max_code_out = 0;
    if (slot > 0) {
        f->pending_slot = slot; f->pending_ticket = pa.ticket; f->pending_stream = s;
        return GS_OK;
    }
    if (N > 0 && wait != WAIT_LATER) return read_counters(c, f, s, pa.ticket, M_out, K_out, max_code_out);
    return GS_OK;
}

// ---- per-pixel half: key build, sort, tile ranges, blend ----------------------------------------------------------
// K_bound: pair capacity the buffers and the launch geometry are sized for; depth_bits: width of the depth field of the keys.
// Both may be PREDICTIONS (run_forward_tail): the kernels take the frame's real pair count from the device counters and stay
// inside K_bound whatever it is, and any depth_bits >= the real width sorts into the same order.
static int run_raster_stage(gs_ctx* c, Frame* f, const GsProjectArgs& pa, int64_t n_rows, int M_bound, uint32_t K_bound, int depth_bits,
                            int H, int W, int tiles_x, int T, const gs_config* cfg, const gs_forward_out* out, hipStream_t s, bool publish)
{
    FrameBufs& B = *f->bufs;
    // list cuts for the backward's heavy tiles (k_blend_fwd): only a frame that will be back-propagated wants them.
    // Policy (measured): segments pay where the ordinary waves do not fill the chip anyway (T * G waves for 5120
    // slots: cfg2_clustered 0.38 -> 0.22 ms) and cost where they do (cfg3_clustered 0.31 -> 0.34 ms: more, shorter work items in a
    // launch that was already full).  GS_BWD_SEGMENTS=0 / 1 forces never / always.
    static const int seg_env = []{ const char* e = getenv("GS_BWD_SEGMENTS"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool want_cuts = seg_env >= 0 ? seg_env == 1 : (int64_t)T * waves_per_tile(T) < 6144;
    int cut_cap = 0;
    if (want_cuts && f->info.kept_for_backward && !cfg->rgb_only && K_bound > 0) {
        // every long list has its records at start / GS_SEG + tile (k_blend_fwd.hip): K / GS_SEG + T + 1 of them hold all lists'.  (Nothing
        // is claimed at run time: a capacity that could run out would make WHICH lists get cuts, and with it the last bits of the
        // gradients, depend on the order of the claims.)
        cut_cap = (int)std::min<uint64_t>((uint64_t)K_bound / GS_SEG + (uint64_t)T + 2, 0x7fffffffu);
        ENSURE(B.cuts, (size_t)cut_cap * 256 * sizeof(float4)); ENSURE(B.cut_mag, (size_t)cut_cap * 256 * sizeof(float2));
    }
    f->cut_cap = cut_cap;
    const int tile_bits = bits_for((uint32_t)(T > 1 ? T - 1 : 1));
    const int key64 = depth_bits + tile_bits > 32 ? 1 : 0;      // compact 32-bit keys whenever they fit
    if (depth_bits + tile_bits > 63) { drop_frame(c, f); return fail(GS_ERR_INVALID_ARGUMENT, "sort key needs more than 63 bits"); }
    const size_t Kp = K_bound > 0 ? K_bound : 1;
    const size_t key_bytes = key64 ? 8 : 4;
    ENSURE(B.keys_a, key_bytes * Kp); ENSURE(B.keys_b, key_bytes * Kp); ENSURE(B.vals_a, 4 * Kp); ENSURE(B.vals_b, 4 * Kp);
    const size_t hist_elems = gs_sort_hist_elems(K_bound);
    ENSURE(c->hist, 4 * hist_elems); ENSURE(c->scan_tmp, 4 * gs_scan_tmp_elems(hist_elems));

    GsBinArgs ba{};
    ba.prof = &c->prof;
    ba.N = n_rows; ba.M = M_bound; ba.K = K_bound; ba.counters = c->counters.as<GsCounters>();
    ba.H = H; ba.W = W; ba.tiles_x = tiles_x; ba.depth_scale = cfg->depth_to_sort_key_scale;
    ba.depth_bits = depth_bits; ba.key_bits = depth_bits + tile_bits;
    ba.PA = pa.PA; ba.PB = pa.PB; ba.box = pa.box; ba.ntiles = pa.ntiles; ba.depth_codes = pa.depth_codes; ba.tile_block_sums = pa.tile_block_sums;
    ba.counters_rw = pa.counters; ba.host_mirror = publish ? pa.host_mirror : nullptr; ba.ticket = pa.ticket;   // publish: k_keygen's last block hands the counters over
    ba.block_offsets = pa.block_offsets; ba.block_counts = pa.block_counts;     // NULL for records that did not come from k_project
    ba.offsets = B.offsets.as<uint32_t>();
    ba.keys_a = B.keys_a.p; ba.keys_b = B.keys_b.p; ba.key64 = key64;
    ba.vals_a = B.vals_a.as<int32_t>(); ba.vals_b = B.vals_b.as<int32_t>();
    ba.hist = c->hist.as<uint32_t>(); ba.scan_tmp = c->scan_tmp.as<uint32_t>();
    ba.tile_start = B.tile_start.as<int32_t>(); ba.tile_end = B.tile_start.as<int32_t>() + T; ba.T = T;
    ba.keys_sorted = &f->keys_sorted; ba.vals_sorted = &f->vals_sorted;
    gs_launch_binning(ba, s);
    HIP_TRY_F(hipGetLastError());

    GsBlendFwdArgs fa{};
    fa.prof = &c->prof;
    fa.H = H; fa.W = W; fa.tiles_x = tiles_x; fa.T = T; fa.rgb_only = cfg->rgb_only;
    fa.tile_start = ba.tile_start; fa.tile_end = ba.tile_end; fa.vals_sorted = f->vals_sorted;
    fa.keys_sorted = f->keys_sorted; fa.key64 = key64; fa.depth_bits = depth_bits; fa.K = K_bound; fa.counters = ba.counters;
    fa.PA = pa.PA; fa.PB = pa.PB; fa.PC = pa.PC;
    fa.image = out->rasterized_image; fa.depth = out->rasterized_depth; fa.acc_alpha = out->pixel_accumulated_alpha;
    fa.last = out->pixel_offset_of_last_effective_point; fa.count = out->pixel_valid_point_count;
    fa.tile_work = B.tile_start.as<int32_t>() + 2 * (size_t)T;
    fa.cuts = cut_cap > 0 ? B.cuts.as<float4>() : nullptr; fa.tile_cut = B.tile_start.as<int32_t>() + 3 * (size_t)T;
    fa.cut_alloc = B.tile_start.as<int32_t>() + 4 * (size_t)T; fa.cut_cap = cut_cap;
    static const bool use_hint = []{ const char* e = getenv("GS_FWD_ORDER_HINT"); return !(e && e[0] == '0'); }();
    fa.order_hint = (use_hint && c->order_hint_T == T && T > 0) ? c->order_hint.as<int32_t>() : nullptr;
    // tile ranges are all zero when K == 0, so the kernel writes the "no contributor" values itself
    gs_launch_blend_fwd(fa, s);
    HIP_TRY_F(hipGetLastError());
    f->depth_bits = depth_bits;
    f->key64 = key64;
    f->info.sort_key_bits = depth_bits + tile_bits;
    return GS_OK;
}

// Everything of a forward after the per-point kernels have been QUEUED (their counters not yet read): the per-pixel half and
// the one device->host hand-over of the frame.
//
// The reference stops twice per forward to learn M and K on the host (RAST:870, 916-931).  Here the host needs them only to
// size buffers and grids, so in steady state it does not stop in the middle at all: binning, sort and blend are queued at
// once on PREDICTED sizes -- pair capacity = what the last frame of this ctx needed + 25 %, key width from the last frame's
// depth-code range + 25 % -- and the hand-over is read AFTER the last launch,
// when the GPU has the whole forward in its queue instead of nothing.  The kernels read the real pair count on the device and
// never leave the predicted capacity, so a wrong prediction is harmless: the host sees it in the counters (K beyond the
// capacity, or depth codes wider than the key field), grows the buffers and queues the per-pixel half again with the exact
// sizes before the call returns (GS_SIZING_REDONE; the caller's outputs are simply written a second time, in stream order).
// The first frame of a ctx, a new image size and GS_PREDICT_SIZES=0 take the exact path: wait, then queue (GS_SIZING_EXACT).
static int run_forward_tail(gs_ctx* c, Frame* f, const GsProjectArgs& pa, int64_t n_rows, int M_known, int32_t ticket,
                            int H, int W, int tiles_x, int T, const gs_config* cfg, const gs_forward_out* out, hipStream_t s,
                            int* M_out, uint32_t* K_out)
{
    static const bool predict = []{ const char* e = getenv("GS_PREDICT_SIZES"); return !(e && e[0] == '0'); }();
    int rc, M = 0, max_code = 0; uint32_t K = 0;
    f->info.sizing = GS_SIZING_EXACT;
    if (n_rows == 0) {                                     // nothing was published: an empty frame
        if ((rc = run_raster_stage(c, f, pa, 0, 0, 0u, 1, H, W, tiles_x, T, cfg, out, s, false)) != GS_OK) return rc;
        *M_out = 0; *K_out = 0u;
        return GS_OK;
    }
    const int tile_bits = bits_for((uint32_t)(T > 1 ? T - 1 : 1));
    if (predict && c->seen.valid && c->seen.H == H && c->seen.W == W) {
        const int bits_p = bits_for((uint32_t)(c->seen.max_code + c->seen.max_code / 4));
        const uint64_t want = (uint64_t)c->seen.K + c->seen.K / 4 + 4096;      // buffers grow to this if they have to (once)
        const uint32_t K_p = (uint32_t)std::min<uint64_t>(want, 0x7fffffffu);
        if (K_p > 0 && bits_p + tile_bits <= 63) {
            if ((rc = run_raster_stage(c, f, pa, n_rows, M_known >= 0 ? M_known : (int)n_rows, K_p, bits_p, H, W, tiles_x, T, cfg, out, s, true)) != GS_OK) return rc;
            if ((rc = read_counters(c, f, s, ticket, &M, &K, &max_code)) != GS_OK) return rc;
            if (K >= (1u << 31)) { drop_frame(c, f); return fail(GS_ERR_INVALID_ARGUMENT, "more than 2^31 sort pairs (tile ranges are int32, RAST:954-957)"); }
            const int bits = bits_for((uint32_t)(max_code > 0 ? max_code : 0));
            if (K <= K_p && bits <= bits_p) {
                f->info.sizing = GS_SIZING_PREDICTED;
            } else {
                // the per-pixel half ran on sizes that did not hold: its results are void (not out of bounds).  Again, exactly.
                HIP_TRY_F(hipMemsetAsync(pa.tile_arrays, 0, sizeof(int32_t) * (size_t)pa.tile_ints, s));
                f->max_tiles_known = false;                 // (k_project's word went with them: the backward's row sum then looks for giant points itself)
                if ((rc = run_raster_stage(c, f, pa, n_rows, M, K, bits, H, W, tiles_x, T, cfg, out, s, false)) != GS_OK) return rc;
                f->info.sizing = GS_SIZING_REDONE;
            }
            c->seen.valid = true; c->seen.H = H; c->seen.W = W; c->seen.K = K; c->seen.max_code = max_code;
            *M_out = M; *K_out = K;
            return GS_OK;
        }
    }
    gs_launch_publish(pa, (int)((n_rows + 255) / 256), s);            // exact sizing: the hand-over is a launch of its own, and the host waits for it here
    HIP_TRY_F(hipGetLastError());
    if ((rc = read_counters(c, f, s, ticket, &M, &K, &max_code)) != GS_OK) return rc;
    if (K >= (1u << 31)) { drop_frame(c, f); return fail(GS_ERR_INVALID_ARGUMENT, "more than 2^31 sort pairs (tile ranges are int32, RAST:954-957)"); }
    if ((rc = run_raster_stage(c, f, pa, n_rows, M, K, bits_for((uint32_t)(max_code > 0 ? max_code : 0)), H, W, tiles_x, T, cfg, out, s, false)) != GS_OK) return rc;
    c->seen.valid = true; c->seen.H = H; c->seen.W = W; c->seen.K = K; c->seen.max_code = max_code;
    *M_out = M; *K_out = K;
    return GS_OK;
}

static void finish_frame(gs_ctx* c, Frame* f, int slot, int64_t N, int M, uint32_t K, int T, int H, int W, int keep, int stages,
                         gs_frame** frame_out)
{
    f->info.n_points = N; f->info.n_points_in_camera = M; f->info.n_keys = K; f->info.n_tiles = T;
    f->info.camera_height = H; f->info.camera_width = W;
    f->info.kept_for_backward = keep ? 1 : 0;
    f->info.stages = stages;
    if (!keep) c->transient = slot;
    *frame_out = ticket_of(slot, f->generation);
}

extern "C" int gs_forward(gs_ctx* c, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg,
                          const gs_forward_out* out, int32_t keep, gs_frame** frame_out, gs_stream stream_)
{
    if (!c || !sc || !cam || !cfg || !out || !frame_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_forward: NULL argument");
    int tiles_x = 0, tiles_y = 0, rc;
    if ((rc = check_geometry(cam, cfg, "gs_forward", &tiles_x, &tiles_y)) != GS_OK) return rc;
    if ((rc = check_scene(sc, cam, "gs_forward")) != GS_OK) return rc;
    if ((rc = check_forward_out(out, cfg, keep, "gs_forward")) != GS_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    if (c->transient >= 0) drop_frame(c, c->frames[c->transient]);
    int slot = -1;
    Frame* f = acquire_frame(c, &slot);
    f->bufs = acquire_bufs(c);
    f->info = gs_frame_info{};
    f->info.kept_for_backward = keep ? 1 : 0;      // (known before the stages run: a kept frame gets list cuts for its backward)
    const int T = tiles_x * tiles_y;
    GsProjectArgs pa{};
    int M = 0, max_code = 0; uint32_t K = 0;
    if ((rc = run_project_stage(c, f, sc, cam, cfg, T, s, &pa, &M, &K, &max_code, WAIT_LATER)) != GS_OK) return rc;
    if ((rc = run_forward_tail(c, f, pa, sc->n_points, -1, pa.ticket, cam->camera_height, cam->camera_width, tiles_x, T, cfg, out, s, &M, &K)) != GS_OK) return rc;
    f->bwd_reference_order = cfg->bwd_reference_order;
    finish_frame(c, f, slot, sc->n_points, M, K, T, cam->camera_height, cam->camera_width, keep, GS_STAGE_PROJECT | GS_STAGE_RASTER, frame_out);
    return GS_OK;
}

extern "C" int gs_project_shard(gs_ctx* c, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg,
                                float* records_out, int32_t* ids_out, int32_t keep, gs_frame** frame_out, gs_stream stream_)
{
    if (!c || !sc || !cam || !cfg || !frame_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_project_shard: NULL argument");
    if (sc->n_points > 0 && !records_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_project_shard: records_out is NULL");
    if (((uintptr_t)records_out & 15u) != 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_project_shard: records_out must be 16-byte aligned");
    int tiles_x = 0, tiles_y = 0, rc;
    if ((rc = check_geometry(cam, cfg, "gs_project_shard", &tiles_x, &tiles_y)) != GS_OK) return rc;
    if ((rc = check_scene(sc, cam, "gs_project_shard")) != GS_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    if (c->transient >= 0) drop_frame(c, c->frames[c->transient]);
    int slot = -1;
    Frame* f = acquire_frame(c, &slot);
    f->bufs = acquire_bufs(c);
    f->info = gs_frame_info{};
    f->info.kept_for_backward = keep ? 1 : 0;      // (known before the stages run: a kept frame gets list cuts for its backward)
    const int T = tiles_x * tiles_y;
    GsProjectArgs pa{};
    int M = 0, max_code = 0; uint32_t K = 0;
    if ((rc = run_project_stage(c, f, sc, cam, cfg, T, s, &pa, &M, &K, &max_code)) != GS_OK) return rc;
    static_assert(GS_RS == 4, "the staged entry points hand records over as (M,16) rows");
    if (M > 0) {
        HIP_TRY_F(hipMemcpyAsync(records_out, f->bufs->rec.p, (size_t)M * 64, hipMemcpyDeviceToDevice, s));
        if (ids_out) HIP_TRY_F(hipMemcpyAsync(ids_out, f->bufs->ids.p, (size_t)M * 4, hipMemcpyDeviceToDevice, s));
    }
    finish_frame(c, f, slot, sc->n_points, M, K, T, cam->camera_height, cam->camera_width, keep, GS_STAGE_PROJECT, frame_out);
    return GS_OK;
}

// Reads the hand-over of a frame begun with gs_project_shard_begin (mutex held): M, K and the object-id verdict.
static int resolve_pending(gs_ctx* c, Frame* f)
{
    if (!f || f->pending_slot <= 0) return GS_OK;
    const int slot = f->pending_slot;
    HIP_TRY(hipSetDevice(c->device));
    const int rc = wait_counters(c, f->pending_stream, f->pending_ticket, slot);
    const GsCounters hc = c->host_counters[slot];
    c->slots_busy &= ~(1ull << slot);
    f->pending_slot = -1;
    if (rc != GS_OK) { drop_frame(c, f); return rc; }
    if (hc.bad_object_ids != 0) {
        drop_frame(c, f);
        return fail(GS_ERR_INVALID_ARGUMENT, "point_object_id holds " + std::to_string(hc.bad_object_ids) + " value(s) outside [0, n_objects) on valid rows");
    }
    f->info.n_points_in_camera = hc.M;
    f->info.n_keys = hc.K;
    return GS_OK;
}

extern "C" int gs_project_shard_begin(gs_ctx* c, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg,
                                      int32_t keep, gs_frame** frame_out, gs_stream stream_)
{
    if (!c || !sc || !cam || !cfg || !frame_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_project_shard_begin: NULL argument");
    int tiles_x = 0, tiles_y = 0, rc;
    if ((rc = check_geometry(cam, cfg, "gs_project_shard_begin", &tiles_x, &tiles_y)) != GS_OK) return rc;
    if ((rc = check_scene(sc, cam, "gs_project_shard_begin")) != GS_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    if (c->transient >= 0) drop_frame(c, c->frames[c->transient]);
    int slot = -1;
    Frame* f = acquire_frame(c, &slot);
    f->bufs = acquire_bufs(c);
    f->info = gs_frame_info{};
    f->info.kept_for_backward = keep ? 1 : 0;      // (known before the stages run: a kept frame gets list cuts for its backward)
    const int T = tiles_x * tiles_y;
    GsProjectArgs pa{};
    int M = 0, max_code = 0; uint32_t K = 0;
    if ((rc = run_project_stage(c, f, sc, cam, cfg, T, s, &pa, &M, &K, &max_code, WAIT_FRAME)) != GS_OK) return rc;
    finish_frame(c, f, slot, sc->n_points, M, K, T, cam->camera_height, cam->camera_width, keep, GS_STAGE_PROJECT, frame_out);
    return GS_OK;
}

extern "C" int gs_forward_projected(gs_ctx* c, const float* records, int64_t m, const gs_camera* cam, const gs_config* cfg,
                                    const gs_forward_out* out, int32_t keep, gs_frame** frame_out, gs_stream stream_)
{
    if (!c || !cam || !cfg || !out || !frame_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_forward_projected: NULL argument");
    if (m < 0 || m >= (int64_t)1 << 31) return fail(GS_ERR_INVALID_ARGUMENT, "gs_forward_projected: m out of range");
    if (m > 0 && !records) return fail(GS_ERR_INVALID_ARGUMENT, "gs_forward_projected: records is NULL");
    int tiles_x = 0, tiles_y = 0, rc;
    if ((rc = check_geometry(cam, cfg, "gs_forward_projected", &tiles_x, &tiles_y)) != GS_OK) return rc;
    if ((rc = check_forward_out(out, cfg, keep, "gs_forward_projected")) != GS_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    if (c->transient >= 0) drop_frame(c, c->frames[c->transient]);
    int slot = -1;
    Frame* f = acquire_frame(c, &slot);
    f->bufs = acquire_bufs(c);
    f->info = gs_frame_info{};
    f->info.kept_for_backward = keep ? 1 : 0;      // (known before the stages run: a kept frame gets list cuts for its backward)
    FrameBufs& B = *f->bufs;
    const int T = tiles_x * tiles_y, H = cam->camera_height, W = cam->camera_width;
    const size_t Mp = (size_t)(m > 0 ? m : 1);
    const size_t nb = (size_t)((m + 255) / 256);
    ENSURE(B.rec, 64 * Mp); ENSURE(B.box, 8 * Mp); ENSURE(B.ntiles, 4 * Mp); ENSURE(B.depth_codes, 4 * Mp); ENSURE(B.offsets, 4 * Mp);
    ENSURE(B.tile_start, 4 * GS_TILE_INTS(T)); ENSURE(B.tile_order, 4 * GS_ORDER_INTS(T));
    ENSURE(c->tile_block_sums, 4 * (nb + 1));
    if (m > 0) HIP_TRY_F(hipMemcpyAsync(B.rec.p, records, (size_t)m * 64, hipMemcpyDeviceToDevice, s));   // the frame keeps its own copy for backward
    GsProjectArgs pa{};
    pa.prof = &c->prof; pa.N = m; pa.H = H; pa.W = W; pa.depth_scale = cfg->depth_to_sort_key_scale;
    set_records(pa, B, Mp);
    pa.box = B.box.as<ushort4>(); pa.ntiles = B.ntiles.as<int32_t>(); pa.depth_codes = B.depth_codes.as<int32_t>();
    pa.tile_block_sums = c->tile_block_sums.as<uint32_t>();
    pa.counters = c->counters.as<GsCounters>();
    pa.tile_arrays = B.tile_start.as<int32_t>(); pa.tile_ints = (int)GS_TILE_INTS(T);
    pa.host_mirror = c->host_counters_dev; pa.ticket = ++c->ticket;
    if (c->ticket == 0x7fffffff) c->ticket = 0;
    gs_launch_boxes_from_records(pa, (int)m, s, false);       // the counters are published from run_forward_tail
    f->max_tiles_known = false;
    HIP_TRY_F(hipGetLastError());
    uint32_t K = 0; int M_seen = 0;
    if ((rc = run_forward_tail(c, f, pa, m, (int)m, pa.ticket, H, W, tiles_x, T, cfg, out, s, &M_seen, &K)) != GS_OK) return rc;
    f->bwd_reference_order = cfg->bwd_reference_order;
    finish_frame(c, f, slot, m, (int)m, K, T, H, W, keep, GS_STAGE_RASTER, frame_out);
    return GS_OK;
}

extern "C" int gs_frame_get_info(gs_ctx* c, const gs_frame* h, gs_frame_info* info)
{
    if (!c || !info) return fail(GS_ERR_INVALID_ARGUMENT, "gs_frame_get_info: NULL argument");
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f) return fail(GS_ERR_STATE, "gs_frame_get_info: not a live frame of this context");
    if (const int rc = resolve_pending(c, f)) return rc;
    *info = f->info;
    return GS_OK;
}

static int64_t export_count(const Frame* f, gs_export what)
{
    const int64_t M = f->info.n_points_in_camera, K = f->info.n_keys, T = f->info.n_tiles, N = f->info.n_points;
    const bool proj = (f->info.stages & GS_STAGE_PROJECT) != 0, rast = (f->info.stages & GS_STAGE_RASTER) != 0;
    switch (what) {
    case GS_X_POINT_ID_IN_CAMERA_LIST: return proj ? M : -1;
    case GS_X_POINT_IN_CAMERA_MASK: return proj ? N : -1;
    case GS_X_POINT_ALPHA_AFTER_ACTIVATION: case GS_X_POINT_RADII: case GS_X_NUM_OVERLAP_TILES: case GS_X_POINT_DEPTH: return M;
    case GS_X_ACCUMULATED_NUM_OVERLAP_TILES: return rast ? M : -1;
    case GS_X_POINT_UV: return 2 * M;
    case GS_X_POINT_IN_CAMERA: case GS_X_POINT_COLOR: return 3 * M;
    case GS_X_POINT_UV_CONIC_AND_RESCALE: return 4 * M;
    case GS_X_RECORDS: return proj ? 16 * M : -1;
    case GS_X_SORT_KEY: case GS_X_POINT_OFFSET_WITH_SORT_KEY: return rast ? K : -1;
    case GS_X_TILE_POINTS_START: case GS_X_TILE_POINTS_END: return rast ? T : -1;
    default: return -1;
    }
}

extern "C" int64_t gs_frame_export_count(gs_ctx* c, const gs_frame* h, gs_export what)
{
    if (!c) return -1;
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (f && resolve_pending(c, f) != GS_OK) return -1;
    return f ? export_count(f, what) : -1;
}

static void set_records(GsBackwardArgs& a, const FrameBufs& B, size_t rows)
{
    const float4* rec = B.rec.as<float4>();
    const size_t plane = GS_RS == 4 ? 1 : rows;
    a.PA = rec; a.PB = rec + plane; a.PC = rec + 2 * plane; a.PD = rec + 3 * plane;
}

extern "C" int gs_frame_export(gs_ctx* c, const gs_frame* h, gs_export what, void* dst, gs_stream stream_)
{
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_frame_export: ctx is NULL");
    if (!dst) return fail(GS_ERR_INVALID_ARGUMENT, "gs_frame_export: dst is NULL");
    if (what < 0 || what >= GS_X_COUNT_) return fail(GS_ERR_INVALID_ARGUMENT, "gs_frame_export: unknown export id");
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f || !f->bufs) return fail(GS_ERR_STATE, "gs_frame_export: not a live frame of this context");
    if (const int rc = resolve_pending(c, f)) return rc;
    if (export_count(f, what) < 0) return fail(GS_ERR_STATE, "gs_frame_export: this frame does not hold that stage");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(enter_stream(c, s));
    const FrameBufs& B = *f->bufs;
    if (what == GS_X_RECORDS) {                  // the (M,16) rows as they are: what the owner of a shard sends to the renderers
        static_assert(GS_RS == 4, "records are exported as (M,16) rows");
        if (f->info.n_points_in_camera > 0)
            HIP_TRY(hipMemcpyAsync(dst, B.rec.p, (size_t)f->info.n_points_in_camera * 64, hipMemcpyDeviceToDevice, s));
        return GS_OK;
    }
    GsExportArgs a{};
    a.what = (int)what; a.N = f->info.n_points; a.M = (int)f->info.n_points_in_camera; a.K = (uint32_t)f->info.n_keys;
    a.T = f->info.n_tiles; a.depth_bits = f->depth_bits; a.key64 = f->key64;
    a.ids = B.ids.as<int32_t>();
    {
        const float4* rec = B.rec.as<float4>();
        const size_t plane = GS_RS == 4 ? 1 : (size_t)(f->info.n_points > 0 ? f->info.n_points : 1);
        a.PA = rec; a.PB = rec + plane; a.PC = rec + 2 * plane; a.PD = rec + 3 * plane;
    }
    a.ntiles = B.ntiles.as<int32_t>(); a.offsets = B.offsets.as<uint32_t>();
    a.keys_sorted = f->keys_sorted; a.vals_sorted = f->vals_sorted;
    a.tile_start = B.tile_start.as<int32_t>(); a.tile_end = B.tile_start.as<int32_t>() + f->info.n_tiles; a.mask = B.mask.as<int8_t>();
    a.dst = dst;
    gs_launch_export(a, s);
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

// ---- backward -----------------------------------------------------------------------------------------------------
// waves per tile of the backward blend: one wave per tile is the cheapest in instructions, but a small image
// has too few tiles to fill 1024 SIMDs, so tiles are split into 2 or 4 quadrant groups (more rows of `partial`)
static int waves_per_tile(int n_tiles)
{
    int G = 1;
    if (const char* e = getenv("GS_BWD_WAVES_PER_TILE")) G = atoi(e);
    else if (n_tiles < 2000) G = 4;      // measured at 976x544 (2074 tiles): G = 2 0.167 ms, G = 4 0.203, G = 1 0.235
    else if (n_tiles < 6144) G = 2;      // at 1920x1088 (8160 tiles): G = 1 0.259 ms, G = 2 0.285
    if (G != 1 && G != 2 && G != 4) G = 1;
    return G;
}

// fills the blend half of the arguments; sums_out = where the per-splat sums go
static int prepare_backward_blend(gs_ctx* c, const Frame* f, const float* grad_image, const float* acc_alpha, const int32_t* last,
                                  float* mag_image, float4* sums_out, int strict, hipStream_t stream, GsBackwardArgs* a_out)
{
    const uint32_t K = (uint32_t)f->info.n_keys;
    const int G = waves_per_tile(f->info.n_tiles);
    const size_t rows = (size_t)(K > 0 ? K : 1) * (size_t)G;
    const size_t flag_bytes = (rows + 15) / 16 * 16;
    hipError_t e = c->partial.ensure(rows * 12 * sizeof(float), &c->device_bytes);
    if (e != hipSuccess) return fail(GS_ERR_OUT_OF_MEMORY, "backward: partial-sum buffer");
    const size_t Mp = (size_t)(f->info.n_points_in_camera > 0 ? f->info.n_points_in_camera : 1);
    // Row flags (K*G bytes) and per-point `touched` bytes behind them.  They are TAGGED, not cleared: a backward writes its own tag
    // (1..255) and reads a flag as set only if it holds that tag, so the 6 MB clear per backward is gone; the buffer is zeroed when it
    // is (re)allocated and when the tags wrap round.
    const void* before = c->visited.p;
    e = c->visited.ensure(flag_bytes + Mp + 16, &c->device_bytes);
    if (e != hipSuccess) return fail(GS_ERR_OUT_OF_MEMORY, "backward: visited buffer");
    if (c->visited.p != before || c->visit_gen == 255) {
        HIP_TRY(hipMemsetAsync(c->visited.p, 0, c->visited.cap, stream));
        c->visit_gen = 0;
    }
    c->visit_gen += 1;
    if (!c->zero_row.p) {
        if (c->zero_row.ensure(64, &c->device_bytes) != hipSuccess) return fail(GS_ERR_OUT_OF_MEMORY, "backward: zero row");
        HIP_TRY(hipMemsetAsync(c->zero_row.p, 0, c->zero_row.cap, stream));
    }
    if (c->order_hint_T != f->info.n_tiles) {          // another tile grid: the old ordering is void from here on
        c->order_hint_T = 0;
        if (c->order_hint.ensure(4 * (size_t)(f->info.n_tiles > 0 ? f->info.n_tiles : 1), &c->device_bytes) != hipSuccess)
            return fail(GS_ERR_OUT_OF_MEMORY, "backward: tile order buffer");
    }
    const FrameBufs& B = *f->bufs;
    GsBackwardArgs a{};
    a.prof = &c->prof;
    a.order_hint = c->order_hint.as<int32_t>();
    a.N = f->info.n_points; a.M = (int)f->info.n_points_in_camera; a.K = K;
    a.H = f->info.camera_height; a.W = f->info.camera_width; a.T = f->info.n_tiles;
    a.tiles_x = (a.W + GS_TILE - 1) / GS_TILE;
    a.tile_start = B.tile_start.as<int32_t>(); a.tile_end = B.tile_start.as<int32_t>() + f->info.n_tiles; a.vals_sorted = f->vals_sorted;
    a.tile_work = B.tile_start.as<int32_t>() + 2 * (size_t)f->info.n_tiles; a.tile_order = B.tile_order.as<int32_t>();
    set_records(a, B, (size_t)(f->info.n_points > 0 ? f->info.n_points : 1));
    a.box = B.box.as<ushort4>(); a.offsets = B.offsets.as<uint32_t>(); a.ntiles = B.ntiles.as<int32_t>();
    a.grad_image = grad_image; a.acc_alpha = acc_alpha; a.last = last;
    a.partial = c->partial.as<float>();
    a.visited = c->visited.as<uint8_t>();
    a.G = G;
    a.n_heavy = B.tile_order.as<int32_t>() + f->info.n_tiles;
    a.cuts = f->cut_cap > 0 ? B.cuts.as<float4>() : nullptr; a.cut_mag = B.cut_mag.as<float2>();
    a.item_cap = f->cut_cap;                          // >= the segments of all heavy tiles together (never binds: deterministic)
    a.tile_cut = B.tile_start.as<int32_t>() + 3 * (size_t)f->info.n_tiles;
    // heavy-tile threshold in half-means of work: sharing a tile among four waves costs more work in total and shortens the launch
    // only where long walks are what the launch waits for; twice the mean measured best on both clustered workloads and changes
    // nothing on the uniform ones (max / mean = 2).  GS_BWD_HEAVY_X2 overrides; GS_BWD_SPLIT_HEAVY=0 = no heavy tiles.
    static const int heavy_env = []{ const char* e = getenv("GS_BWD_SPLIT_HEAVY"); if (e && e[0] == '0') return 0;
                                     const char* x = getenv("GS_BWD_HEAVY_X2"); return x ? atoi(x) : -1; }();
    a.heavy_factor_x2 = heavy_env >= 0 ? heavy_env : 4;
    a.strict = strict ? 1 : 0;
    a.gen = c->visit_gen;
    a.touched = c->visited.as<uint8_t>() + flag_bytes;
    a.zero_row = c->zero_row.as<float4>();
    a.max_tiles_hint = f->max_tiles_known ? B.tile_start.as<int32_t>() + GS_TILE_INTS(f->info.n_tiles) - GS_TILE_SPARE_MAX_TILES : nullptr;
    a.sums = sums_out;
    a.mag_image = mag_image;
    *a_out = a;
    return GS_OK;
}

static int prepare_backward_points(const Frame* f, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg,
                                   int32_t sh_band, const gs_backward_out* out, const float4* sums, GsBackwardArgs* a)
{
    const FrameBufs& B = *f->bufs;
    a->N = f->info.n_points; a->M = (int)f->info.n_points_in_camera;
    set_records(*a, B, (size_t)(f->info.n_points > 0 ? f->info.n_points : 1));
    a->ntiles = B.ntiles.as<int32_t>();
    a->ids = B.ids.as<int32_t>(); a->cam_index = B.cam_index.as<int32_t>();
    a->sums = const_cast<float4*>(sums);
    a->point_cloud = sc->point_cloud; a->features = sc->point_cloud_features; a->object_id = sc->point_object_id;
    a->Kmat = cam->camera_intrinsics; a->pose = B.pose.as<GsPose>();
    a->sh_band = sh_band; a->f_color = cfg->grad_color_factor; a->f_high = cfg->grad_high_order_color_factor;
    a->f_s = cfg->grad_s_factor; a->f_q = cfg->grad_q_factor; a->f_alpha = cfg->grad_alpha_factor;
    a->grad_pc = out->grad_pointcloud; a->grad_feat = out->grad_pointcloud_features; a->grad_uv = out->grad_viewspace;
    a->mag = out->magnitude_grad_viewspace;
    a->n_affected = out->num_affected_pixels;
    a->hook_gpc = out->hook_grad_point_in_camera; a->hook_gfeat = out->hook_grad_pointfeatures_in_camera;
    a->hook_guv = out->hook_grad_viewspace; a->hook_mag = out->hook_magnitude_grad_viewspace;
    a->hook_ids = out->hook_point_id_in_camera_list; a->hook_ntiles = out->hook_num_overlap_tiles;
    a->hook_depth = out->hook_point_depth; a->hook_uv = out->hook_point_uv_in_camera;
    if (const gs_controller_accumulators* ca = out->controller) {
        if (!ca->accumulated_num_in_camera || !ca->accumulated_num_pixels || !ca->accumulated_view_space_position_gradients ||
            !ca->accumulated_view_space_position_gradients_avg || !ca->accumulated_position_gradients || !ca->accumulated_position_gradients_norm)
            return fail(GS_ERR_INVALID_ARGUMENT, "backward: controller accumulators must all be given");
        a->c_num_in_camera = ca->accumulated_num_in_camera; a->c_num_pixels = ca->accumulated_num_pixels;
        a->c_vs_grad = ca->accumulated_view_space_position_gradients; a->c_vs_grad_avg = ca->accumulated_view_space_position_gradients_avg;
        a->c_pos_grad = ca->accumulated_position_gradients; a->c_pos_grad_norm = ca->accumulated_position_gradients_norm;
    }
    return GS_OK;
}

static int check_backward_points_args(const Frame* f, const gs_scene* sc, const gs_camera* cam, const gs_backward_out* out, const char* who)
{
    if (sc->n_points > 0 && (!out->grad_pointcloud || !out->grad_pointcloud_features))
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": grad_pointcloud / grad_pointcloud_features are mandatory");
    if (sc->n_points != f->info.n_points || cam->camera_height != f->info.camera_height || cam->camera_width != f->info.camera_width)
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": scene/camera do not match the frame");
    if (((uintptr_t)out->grad_pointcloud_features & 15u) != 0 || ((uintptr_t)sc->point_cloud_features & 15u) != 0)
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": feature arrays must be 16-byte aligned");
    if (out->hook_grad_pointfeatures_in_camera && ((uintptr_t)out->hook_grad_pointfeatures_in_camera & 15u) != 0)
        return fail(GS_ERR_INVALID_ARGUMENT, std::string(who) + ": hook feature array must be 16-byte aligned");
    return GS_OK;
}

extern "C" int gs_backward(gs_ctx* c, gs_frame* h, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg,
                           const float* grad_image, const float* acc_alpha, const int32_t* last,
                           int32_t sh_band, const gs_backward_out* out, gs_stream stream_)
{
    if (!c || !sc || !cam || !cfg || !out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward: NULL argument");
    if (!grad_image || !acc_alpha || !last) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward: NULL image-sized input");
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f || !f->bufs) return fail(GS_ERR_STATE, "gs_backward: not a live frame of this context");
    if (!f->info.kept_for_backward) return fail(GS_ERR_STATE, "gs_backward: frame was not kept for backward");
    if (f->info.stages != (GS_STAGE_PROJECT | GS_STAGE_RASTER)) return fail(GS_ERR_STATE, "gs_backward: frame does not come from gs_forward");
    int rc;
    if ((rc = check_backward_points_args(f, sc, cam, out, "gs_backward")) != GS_OK) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    if (c->sums.ensure((size_t)(f->info.n_points_in_camera > 0 ? f->info.n_points_in_camera : 1) * 48, &c->device_bytes) != hipSuccess)
        return fail(GS_ERR_OUT_OF_MEMORY, "gs_backward: per-point sums buffer");
    GsBackwardArgs a{};
    if ((rc = prepare_backward_blend(c, f, grad_image, acc_alpha, last, out->magnitude_grad_viewspace_on_image, c->sums.as<float4>(), cfg->bwd_reference_order, s, &a)) != GS_OK) return rc;
    if ((rc = prepare_backward_points(f, sc, cam, cfg, sh_band, out, c->sums.as<float4>(), &a)) != GS_OK) return rc;
    gs_launch_backward_blend(a, s);
    if (a.T > 0 && a.K > 0) c->order_hint_T = a.T;          // k_tile_order ran: the hint is a complete permutation
    gs_launch_backward_points(a, s);
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_backward_projected(gs_ctx* c, gs_frame* h, const float* grad_image, const float* acc_alpha, const int32_t* last,
                                     float* splat_sums_out, float* mag_image, gs_stream stream_)
{
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_projected: ctx is NULL");
    if (!grad_image || !acc_alpha || !last) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_projected: NULL image-sized input");
    if (((uintptr_t)splat_sums_out & 15u) != 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_projected: splat_sums_out must be 16-byte aligned");
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f || !f->bufs) return fail(GS_ERR_STATE, "gs_backward_projected: not a live frame of this context");
    if (!f->info.kept_for_backward) return fail(GS_ERR_STATE, "gs_backward_projected: frame was not kept for backward");
    if (!(f->info.stages & GS_STAGE_RASTER)) return fail(GS_ERR_STATE, "gs_backward_projected: frame holds no raster stage");
    if (f->info.n_points_in_camera > 0 && !splat_sums_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_projected: splat_sums_out is NULL");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    GsBackwardArgs a{};
    int rc;
    if ((rc = prepare_backward_blend(c, f, grad_image, acc_alpha, last, mag_image, reinterpret_cast<float4*>(splat_sums_out), f->bwd_reference_order, s, &a)) != GS_OK) return rc;
    gs_launch_backward_blend(a, s);
    if (a.T > 0 && a.K > 0) c->order_hint_T = a.T;          // k_tile_order ran: the hint is a complete permutation
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_backward_shard(gs_ctx* c, gs_frame* h, const gs_scene* sc, const gs_camera* cam, const gs_config* cfg,
                                 const float* splat_sums, int32_t sh_band, const gs_backward_out* out, gs_stream stream_)
{
    if (!c || !sc || !cam || !cfg || !out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_shard: NULL argument");
    if (((uintptr_t)splat_sums & 15u) != 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_shard: splat_sums must be 16-byte aligned");
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f || !f->bufs) return fail(GS_ERR_STATE, "gs_backward_shard: not a live frame of this context");
    if (const int rp = resolve_pending(c, f)) return rp;
    if (!f->info.kept_for_backward) return fail(GS_ERR_STATE, "gs_backward_shard: frame was not kept for backward");
    if (!(f->info.stages & GS_STAGE_PROJECT)) return fail(GS_ERR_STATE, "gs_backward_shard: frame holds no projection stage");
    if (f->info.n_points_in_camera > 0 && !splat_sums) return fail(GS_ERR_INVALID_ARGUMENT, "gs_backward_shard: splat_sums is NULL");
    int rc;
    if ((rc = check_backward_points_args(f, sc, cam, out, "gs_backward_shard")) != GS_OK) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, s));
    GsBackwardArgs a{};
    a.prof = &c->prof;
    if ((rc = prepare_backward_points(f, sc, cam, cfg, sh_band, out, reinterpret_cast<const float4*>(splat_sums), &a)) != GS_OK) return rc;
    gs_launch_backward_points(a, s);
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

static GsLossImage loss_image(const gs_loss_image* im, int clamp)
{
    GsLossImage r; r.p = im->data; r.sc = im->stride_channel; r.sy = im->stride_row; r.sx = im->stride_column; r.clamp = clamp;
    return r;
}
static int loss_check(const char* who, const gs_loss_image* a, const gs_loss_image* b, int32_t H, int32_t W)
{
    if (!a || !b || !a->data || !b->data) return fail(GS_ERR_INVALID_ARGUMENT, (std::string(who) + ": NULL image"));
    if (H < 11 || W < 11) return fail(GS_ERR_INVALID_ARGUMENT, (std::string(who) + ": image smaller than the 11x11 SSIM window"));
    return GS_OK;
}

extern "C" int64_t gs_loss_maps_floats(int32_t H, int32_t W) { return (H < 1 || W < 1) ? 0 : (int64_t)gs_loss_maps_size((int)H, (int)W); }

extern "C" int gs_loss_l1_ssim_forward(gs_ctx* c, const gs_loss_image* pred, const gs_loss_image* gt, int32_t H, int32_t W, int32_t clamp_pred,
                                       float lambda_value, float* maps, float* loss_terms, gs_stream stream_)
{
    if (!c || !maps || !loss_terms) return fail(GS_ERR_INVALID_ARGUMENT, "gs_loss_l1_ssim_forward: NULL argument");
    if (int rc = loss_check("gs_loss_l1_ssim_forward", pred, gt, H, W)) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = c->loss_ws.ensure(gs_loss_partials_floats(H, W) * sizeof(float), &c->device_bytes);
    if (e != hipSuccess) return fail(GS_ERR_OUT_OF_MEMORY, "gs_loss_l1_ssim_forward: workspace");
    HIP_TRY(enter_stream(c, reinterpret_cast<hipStream_t>(stream_)));
    gs_launch_loss_forward(loss_image(pred, clamp_pred != 0), loss_image(gt, 0), H, W, lambda_value, maps, c->loss_ws.as<float>(), loss_terms,
                           reinterpret_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_loss_l1_ssim_backward(gs_ctx* c, const gs_loss_image* pred, const gs_loss_image* gt, int32_t H, int32_t W, int32_t clamp_pred,
                                        float lambda_value, const float* maps, const float* upstream, const gs_loss_image* grad_pred,
                                        gs_stream stream_)
{
    if (!c || !maps || !grad_pred || !grad_pred->data) return fail(GS_ERR_INVALID_ARGUMENT, "gs_loss_l1_ssim_backward: NULL argument");
    if (int rc = loss_check("gs_loss_l1_ssim_backward", pred, gt, H, W)) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, reinterpret_cast<hipStream_t>(stream_)));
    gs_launch_loss_backward(loss_image(pred, clamp_pred != 0), loss_image(gt, 0), H, W, lambda_value, maps, upstream, loss_image(grad_pred, 0),
                            reinterpret_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

// the one-call form: contiguous (3,H,W) images, the maps in the context's own workspace, upstream = 1
extern "C" int gs_loss_l1_ssim(gs_ctx* c, const float* pred, const float* gt, int32_t H, int32_t W, float lambda_value,
                               float* loss_terms, float* grad_pred, gs_stream stream_)
{
    if (!c || !pred || !gt || !loss_terms) return fail(GS_ERR_INVALID_ARGUMENT, "gs_loss_l1_ssim: NULL argument");
    if (H < 11 || W < 11) return fail(GS_ERR_INVALID_ARGUMENT, "gs_loss_l1_ssim: image smaller than the 11x11 SSIM window");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const size_t maps_floats = gs_loss_maps_size(H, W);
    hipError_t e = c->loss_ws.ensure((maps_floats + gs_loss_partials_floats(H, W)) * sizeof(float), &c->device_bytes);
    if (e != hipSuccess) return fail(GS_ERR_OUT_OF_MEMORY, "gs_loss_l1_ssim: workspace");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(enter_stream(c, s));
    GsLossImage X{ pred, (long long)H * W, (long long)W, 1, 0 }, Y{ gt, (long long)H * W, (long long)W, 1, 0 };
    float* maps = c->loss_ws.as<float>();
    gs_launch_loss_forward(X, Y, H, W, lambda_value, maps, maps + maps_floats, loss_terms, s);
    if (grad_pred) {
        GsLossImage G{ grad_pred, (long long)H * W, (long long)W, 1, 0 };
        gs_launch_loss_backward(X, Y, H, W, lambda_value, maps, nullptr, G, s);
    }
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_scale_regulariser(gs_ctx* c, const float* feat, const int8_t* mask, int64_t n, float* out, gs_stream stream_)
{
    if (!c || !out || (n > 0 && (!feat || !mask))) return fail(GS_ERR_INVALID_ARGUMENT, "gs_scale_regulariser: NULL argument");
    if (n < 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_scale_regulariser: n_points < 0");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = c->loss_ws.ensure((size_t)(2 * ((n + 255) / 256) + 16) * sizeof(float), &c->device_bytes);
    if (e != hipSuccess) return fail(GS_ERR_OUT_OF_MEMORY, "gs_scale_regulariser: workspace");
    HIP_TRY(enter_stream(c, reinterpret_cast<hipStream_t>(stream_)));
    gs_launch_reg_value(feat, mask, n, c->loss_ws.as<float>(), out, reinterpret_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_scale_regulariser_grad(gs_ctx* c, const float* feat, const int8_t* mask, int64_t n, const float* value_and_count,
                                         const float* upstream, float* grad, gs_stream stream_)
{
    if (!c || (n > 0 && (!feat || !mask || !value_and_count || !upstream || !grad)))
        return fail(GS_ERR_INVALID_ARGUMENT, "gs_scale_regulariser_grad: NULL argument");
    if (grad && ((uintptr_t)grad & 15u) != 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_scale_regulariser_grad: grad must be 16-byte aligned");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, reinterpret_cast<hipStream_t>(stream_)));
    gs_launch_reg_grad(feat, mask, n, value_and_count, upstream, grad, reinterpret_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_adam_step(gs_ctx* c, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                            float lr, float beta1, float beta2, float eps, int64_t step, gs_stream stream_)
{
    if (!c || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return fail(GS_ERR_INVALID_ARGUMENT, "gs_adam_step: NULL argument");
    if (n < 0 || step < 1) return fail(GS_ERR_INVALID_ARGUMENT, "gs_adam_step: n must be >= 0 and step >= 1");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(enter_stream(c, reinterpret_cast<hipStream_t>(stream_)));
    gs_launch_adam(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, reinterpret_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_frame_heavy_tiles(gs_ctx* c, const gs_frame* h, int32_t* n_out, gs_stream stream_)
{
    if (!c || !n_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_frame_heavy_tiles: NULL argument");
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f || !f->bufs) return fail(GS_ERR_STATE, "gs_frame_heavy_tiles: not a live frame of this context");
    if (!(f->info.stages & GS_STAGE_RASTER) || f->info.n_tiles <= 0 || f->info.n_keys <= 0 || !f->bufs->tile_order.p) { n_out[0] = n_out[1] = 0; return GS_OK; }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(enter_stream(c, s));
    HIP_TRY(hipMemcpyAsync(n_out, f->bufs->tile_order.as<int32_t>() + f->info.n_tiles, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GS_OK;
}

extern "C" int gs_frame_release(gs_ctx* c, gs_frame* h)
{
    if (!c || !h) return GS_OK;
    std::lock_guard<std::mutex> lock(c->mu);
    Frame* f = resolve(c, h);
    if (!f) return fail(GS_ERR_STATE, "gs_frame_release: not a live frame of this context (already released?)");
    drop_frame(c, f);
    return GS_OK;
}

// ---- diagnostic build only (make stats): counters of the blend kernels, tools/blend_stats.py ----
#ifdef GS_STATS
__device__ unsigned long long gs_stats_counters[32];
__device__ unsigned long long gs_stats_wave_times[2 * 65536];
extern "C" int gs_debug_wave_times_read(unsigned long long* out, int n_waves)
{
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (n_waves > 65536) n_waves = 65536;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gs_stats_wave_times), sizeof(unsigned long long) * 2 * (size_t)n_waves) != hipSuccess) return -2;
    return 0;
}
extern "C" int gs_debug_stats_read(unsigned long long* out32, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(gs_stats_counters), sizeof(unsigned long long) * 32) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gs_stats_counters), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#endif
