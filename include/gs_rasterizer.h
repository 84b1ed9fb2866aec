/*
 * gs_rasterizer.h -- C ABI of libgsrast.so, the MI355X (gfx950) drop-in for the
 * rasterisation hot path of Wenri/taichi_3d_gaussian_splatting.
 *
 * The entry points are what the reference's Python operator would bind in place of
 * its Taichi kernels; each one names the reference interface it replaces
 * (paths under the reference repo, RAST = taichi_3d_gaussian_splatting/
 * GaussianPointCloudRasterisation.py).  Plain C, no torch types: every pointer
 * marked "device" is a HIP device pointer on the context's GPU, owned by the caller
 * unless stated otherwise; the library never frees caller memory.  All functions
 * return 0 on success and a negative gs_status otherwise, never throw, and set a
 * thread-local message readable through gs_last_error().
 *
 * Threading: a gs_ctx may be used from any host thread; calls on one ctx are
 * serialised by a mutex (PyTorch calls forward on the main thread and backward on an
 * autograd worker).  Streams: the scratch buffers of a ctx are recycled in stream order.
 * When a call arrives on a different stream than the previous call of the same ctx, the
 * library makes the new stream wait (hipStreamWaitEvent, no host block) for everything the
 * ctx has issued on the old one, so switching streams is safe; use one ctx per stream for
 * work that is meant to overlap.
 *
 * Frame handles (gs_frame*) are opaque tickets, never dereferenced by the caller and
 * checked on every use: a released, recycled or foreign handle gives GS_ERR_STATE.
 */
#ifndef GS_RASTERIZER_H
#define GS_RASTERIZER_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 8
#define GS_TILE 16              /* RAST:27-28 TILE_WIDTH = TILE_HEIGHT */
#define GS_FEATURES 56          /* RAST:208-236 row layout */

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_INVALID_ARGUMENT = -1,   /* the reference's Python asserts / Taichi TypeErrors (RAST:1193-1194) */
    GS_ERR_HIP = -2,                /* a HIP runtime call failed; message carries hipGetErrorString */
    GS_ERR_OUT_OF_MEMORY = -3,
    GS_ERR_STATE = -4               /* e.g. backward on a frame that was not kept, or a stale frame handle */
} gs_status;

typedef struct gs_ctx gs_ctx;       /* one per operator instance per device (RAST:819-826) */
typedef struct gs_frame gs_frame;   /* what RAST:998-1021 ctx.save_for_backward keeps alive */
typedef void* gs_stream;            /* hipStream_t; NULL = the legacy default stream */

/* GaussianPointCloudRasterisationConfig, RAST:776-786 (same defaults) */
typedef struct gs_config {
    float   near_plane;                      /* 0.8   */
    float   far_plane;                       /* 1000  */
    float   depth_to_sort_key_scale;         /* 100   */
    int32_t rgb_only;                        /* 0     */
    float   grad_color_factor;               /* 5     */
    float   grad_high_order_color_factor;    /* 1     */
    float   grad_s_factor;                   /* 0.5   */
    float   grad_q_factor;                   /* 1     */
    float   grad_alpha_factor;               /* 20    */
    int32_t allow_partial_tiles;             /* 0; EXTENSION: 1 lifts the W,H % 16 == 0 requirement (RAST:1193-1194): tile
                                                counts are rounded up and pixels outside the image do not exist */
    int32_t bwd_reference_order;             /* 0; 1 = the backward's per-contribution Gaussian gradient
                                                (grad_point_probability_density_from_conic_and_rescale, utils.py:331-348) in the
                                                reference's own f32 operation order: Sigma^-1 d without fused multiply-adds and
                                                d p / d Sigma' as 0.5 p (Sigma^-1 (d d^T) Sigma^-1) with two 2x2 products.  The
                                                default forms the same matrix as v v^T, v = Sigma^-1 d (three fused multiply-adds;
                                                for long thin conics closer to exact arithmetic than the reference's rounding,
                                                and up to 1.5e-4 of the tensor maximum away from it on such scenes).  Read by
                                                gs_backward from the config passed to it, by gs_backward_projected from the
                                                config its frame was made with.  ~25 % slower blend backward. */
} gs_config;

/* The point-cloud half of GaussianPointCloudRasterisationInput, RAST:788-804. */
typedef struct gs_scene {
    const float*   point_cloud;             /* device (N,3) f32 */
    float*         point_cloud_features;    /* device (N,56) f32; [:,0:4] is normalised IN PLACE (RAST:264-266) */
    const int8_t*  point_invalid_mask;      /* device (N) i8, 1 = skip */
    const int32_t* point_object_id;         /* device (N) i32 in [0, n_objects); a valid row with an id outside that range is
                                               treated as not in camera and gs_forward returns GS_ERR_INVALID_ARGUMENT */
    int64_t        n_points;                /* N */
} gs_scene;

/* The camera half: CameraInfo (Camera.py:6-11) + the pose rows, RAST:799-803. */
typedef struct gs_camera {
    const float* q_pointcloud_camera;       /* device (n_objects,4) xyzw */
    const float* t_pointcloud_camera;       /* device (n_objects,3) */
    int32_t      n_objects;
    const float* camera_intrinsics;         /* device (3,3) row-major */
    int32_t      camera_height;             /* multiples of 16 (RAST:1193-1194) unless gs_config.allow_partial_tiles */
    int32_t      camera_width;
} gs_camera;

/* Image-sized outputs of the blend kernel, RAST:967-976.  With rgb_only only
 * rasterized_image is written (RAST:478-484) and the others may be NULL. */
typedef struct gs_forward_out {
    float*   rasterized_image;                       /* device (H,W,3) f32 */
    float*   rasterized_depth;                       /* device (H,W) f32 */
    float*   pixel_accumulated_alpha;                /* device (H,W) f32 */
    int32_t* pixel_offset_of_last_effective_point;   /* device (H,W) i32 */
    int32_t* pixel_valid_point_count;                /* device (H,W) i32 */
} gs_forward_out;

/* Data-dependent sizes of a frame (the reference learns them through two device->host syncs in the middle of its forward,
 * RAST:870 and RAST:916-931; here the host reads them once, after it has queued the whole forward -- see `sizing`). */
typedef struct gs_frame_info {
    int64_t n_points;               /* N */
    int64_t n_points_in_camera;     /* M */
    int64_t n_keys;                 /* K = sum of num_overlap_tiles */
    int32_t n_tiles;                /* (W/16)*(H/16) */
    int32_t camera_height, camera_width;
    int32_t sort_key_bits;          /* significant bits actually radix-sorted */
    int32_t kept_for_backward;
    int32_t stages;                 /* GS_STAGE_PROJECT | GS_STAGE_RASTER: which halves of the path the frame holds */
    int32_t sizing;                 /* how the per-pixel half of the forward was sized (see gs_forward): GS_SIZING_EXACT,
                                       GS_SIZING_PREDICTED or GS_SIZING_REDONE */
} gs_frame_info;

#define GS_SIZING_EXACT     0       /* the host waited for M, K and the depth-code range before it queued binning, sort and blend */
#define GS_SIZING_PREDICTED 1       /* queued without waiting, on the buffer capacities and key width the context already had; held */
#define GS_SIZING_REDONE    2       /* the prediction did not hold (K beyond the buffers, or wider depth codes): queued again */

#define GS_STAGE_PROJECT 1          /* filter + compaction + projection (gs_forward, gs_project_shard) */
#define GS_STAGE_RASTER  2          /* binning + sort + blend (gs_forward, gs_forward_projected) */
#define GS_RECORD_FLOATS 16         /* floats per projected splat record */
#define GS_SPLAT_SUM_FLOATS 12      /* floats per per-splat backward sum row */

/* Optional: the six per-point accumulators of the adaptive density controller
 * (GaussianPointAdaptiveController.py:114-127), updated IN PLACE (+=) for every in-camera point exactly as
 * GaussianPointAdaptiveController.update() does from the hook payload (:133-141), so that the statistics
 * need neither the hook gathers nor six torch scatter-adds.  All device arrays with N rows. */
typedef struct gs_controller_accumulators {
    int32_t* accumulated_num_in_camera;                      /* (N)   += 1 */
    int32_t* accumulated_num_pixels;                         /* (N)   += num_affected_pixels */
    float*   accumulated_view_space_position_gradients;      /* (N)   += magnitude_grad_viewspace */
    float*   accumulated_view_space_position_gradients_avg;  /* (N)   += magnitude / num_affected_pixels (0 when 0/0) */
    float*   accumulated_position_gradients;                 /* (N,3) += grad_pointcloud row */
    float*   accumulated_position_gradients_norm;            /* (N)   += |grad_pointcloud row|_2 */
} gs_controller_accumulators;

/* Outputs of backward, RAST:1051-1067 + RAST:1127-1140.  The first two are
 * mandatory; every other pointer may be NULL (need_extra_info = false).
 * All N-row arrays are fully written (rows outside the frustum = 0). */
typedef struct gs_backward_out {
    float*   grad_pointcloud;                   /* device (N,3)  */
    float*   grad_pointcloud_features;          /* device (N,56) band-masked and factor-scaled (RAST:1102-1125) */
    float*   grad_viewspace;                    /* device (N,2)  */
    float*   magnitude_grad_viewspace;          /* device (N)    */
    float*   magnitude_grad_viewspace_on_image; /* device (H,W,2) */
    int32_t* num_affected_pixels;               /* device (M)    */
    /* BackwardValidPointHookInput gathers (RAST:1128-1140), written directly: */
    float*   hook_grad_point_in_camera;         /* device (M,3)  */
    float*   hook_grad_pointfeatures_in_camera; /* device (M,56) */
    float*   hook_grad_viewspace;               /* device (M,2)  */
    float*   hook_magnitude_grad_viewspace;     /* device (M)    */
    const gs_controller_accumulators* controller; /* host pointer to a struct of device arrays, or NULL */
    /* the remaining four fields of BackwardValidPointHookInput (RAST:1128-1140), each nullable; written by the same
     * kernel as the gathers above so that a hook costs no extra launches (gs_frame_export gives the same arrays): */
    int32_t* hook_point_id_in_camera_list;      /* device (M)    */
    int32_t* hook_num_overlap_tiles;            /* device (M)    */
    float*   hook_point_depth;                  /* device (M)    */
    float*   hook_point_uv_in_camera;           /* device (M,2)  */
} gs_backward_out;

/* Intermediates a frame can copy out, in the reference's layouts (saved tensors
 * RAST:998-1019 and the locals of RAST:873-964). */
typedef enum gs_export {
    GS_X_POINT_ID_IN_CAMERA_LIST = 0,   /* (M) i32 */
    GS_X_POINT_UV = 1,                  /* (M,2) f32 */
    GS_X_POINT_IN_CAMERA = 2,           /* (M,3) f32 */
    GS_X_POINT_UV_CONIC_AND_RESCALE = 3,/* (M,4) f32 */
    GS_X_POINT_ALPHA_AFTER_ACTIVATION = 4, /* (M) f32 */
    GS_X_POINT_COLOR = 5,               /* (M,3) f32 */
    GS_X_POINT_RADII = 6,               /* (M) f32 */
    GS_X_NUM_OVERLAP_TILES = 7,         /* (M) i32 */
    GS_X_ACCUMULATED_NUM_OVERLAP_TILES = 8, /* (M) i64, exclusive */
    GS_X_SORT_KEY = 9,                  /* (K) i64 sorted: (tile_id << 32) + i32(depth*scale) */
    GS_X_POINT_OFFSET_WITH_SORT_KEY = 10, /* (K) i32 sorted */
    GS_X_TILE_POINTS_START = 11,        /* (T) i32 */
    GS_X_TILE_POINTS_END = 12,          /* (T) i32 */
    GS_X_POINT_DEPTH = 13,              /* (M) f32 = point_in_camera[:,2] */
    GS_X_POINT_IN_CAMERA_MASK = 14,     /* (N) i8 */
    GS_X_RECORDS = 15,                  /* (M,16) f32: the projected records A|B|C|D of a frame with the projection stage */
    GS_X_COUNT_
} gs_export;

int gs_abi_version(void);
const char* gs_last_error(void);

/* Replaces ti.init + module construction (GaussianPointTrainer.py:124, RAST:819-826). */
int gs_create(int32_t device, gs_ctx** out);
int gs_destroy(gs_ctx* ctx);

/* Replaces _module_function.forward, RAST:830-1023 (kernels RAST:31-485).
 * keep_for_backward != 0: *frame_out receives a handle that must be passed to
 * gs_backward and/or gs_frame_release; 0: nothing is kept (torch.no_grad path,
 * benchmark/inference_benchmark.py:110-156) and *frame_out still receives a handle
 * valid for gs_frame_get_info/gs_frame_export until the next call on this ctx.
 * When K == 0 the outputs are zero-filled (the reference leaves torch.empty
 * garbage, RAST:967-980).
 * Sizing: the per-pixel half (binning, sort, blend) needs M, K and the depth-code range only to size buffers and grids.  From
 * the second frame of a ctx at an image size on, it is queued on PREDICTED sizes (the last frame's + 25 %) before the host has
 * read the frame's counters; the kernels take the real pair count on the device and never leave the predicted capacity; the
 * host reads the counters after its last launch and, if the prediction did not hold, queues the per-pixel half again with
 * exact sizes before this call returns (gs_frame_info.sizing says which happened; GS_PREDICT_SIZES=0 disables prediction).
 * Either way the outputs the caller sees are the exact ones, in stream order. */
int gs_forward(gs_ctx* ctx, const gs_scene* scene, const gs_camera* camera, const gs_config* config,
               const gs_forward_out* out, int32_t keep_for_backward, gs_frame** frame_out, gs_stream stream);

int gs_frame_get_info(gs_ctx* ctx, const gs_frame* frame, gs_frame_info* info);

/* Element count of an export (so the caller can size dst); -1 for a stale handle or an export the frame does not hold. */
int64_t gs_frame_export_count(gs_ctx* ctx, const gs_frame* frame, gs_export what);
int gs_frame_export(gs_ctx* ctx, const gs_frame* frame, gs_export what, void* dst_device, gs_stream stream);

/* Replaces _module_function.backward, RAST:1025-1163 (kernel RAST:488-772 and the
 * torch post-processing RAST:1102-1140).  pixel_accumulated_alpha and
 * pixel_offset_of_last_effective_point are the forward's outputs (saved tensors
 * RAST:1006-1007).  grad_rasterized_image is (H,W,3) contiguous. */
int gs_backward(gs_ctx* ctx, gs_frame* frame, const gs_scene* scene, const gs_camera* camera,
                const gs_config* config, const float* grad_rasterized_image,
                const float* pixel_accumulated_alpha, const int32_t* pixel_offset_of_last_effective_point,
                int32_t color_max_sh_band, const gs_backward_out* out, gs_stream stream);

/* May be called any number of times between forward and release (backward(retain_graph=True) in PyTorch terms). */

/* Diagnostic: n_out[0] = how many tiles the last backward blend of this frame treated as HEAVY (a workgroup of four cooperating
 * waves instead of one wave; k_backward.hip), n_out[1] = how many work items they were handed out as (a heavy tile whose list the
 * forward cut is walked in segments of 512 entries, one item each).  n_out: host int32[2].  Synchronises the stream. */
int gs_frame_heavy_tiles(gs_ctx* ctx, const gs_frame* frame, int32_t* n_out, gs_stream stream);

int gs_frame_release(gs_ctx* ctx, gs_frame* frame);

/* ---- the same path cut at the projected records and at the per-splat sums -----------------
 * For Gaussian-parallel multi-GPU rendering: the rank that OWNS a shard of the Gaussians runs the per-point halves
 * (gs_project_shard, gs_backward_shard), the rank that RENDERS a view runs the per-pixel halves (gs_forward_projected,
 * gs_backward_projected); between them travel the projected splat records (16 floats per in-camera point) and the
 * per-splat backward sums (12 floats), instead of 59 gradient floats per Gaussian of the scene.  Chained on one GPU
 * the four calls give bit for bit what gs_forward + gs_backward give.  The reference has no counterpart (it is
 * single-GPU); the cut points are its own intermediate tensors: the records are RAST:873-911's per-point arrays, the
 * sums are the accumulators of RAST:674-696 (grad_uv, cov buffer, colour buffer, opacity, magnitude, pixel count).
 *
 * Record layout (GS_RECORD_FLOATS): u v conic_a conic_b | conic_c rescale opacity depth | r g b alpha_cut | x y z(camera) radius.
 * Sum row layout (GS_SPLAT_SUM_FLOATS): d uv (2) | d cov xx xy yy (3) | d colour (3) | d opacity | sum |d uv| | pixel count (i32 bits) | 0;
 * rows are pre-factor sums exactly as k_blend_bwd_tile leaves them (opacity, 0.5 and (1-o)o are applied by the shard half). */

/* Per-point half of the forward for a shard of the scene: frustum filter, compaction (ascending ids), projection.
 * records_out: device (capacity >= scene->n_points rows, 16 floats each), receives M rows in ascending point id;
 * ids_out: device (n_points) i32 or NULL, receives the M in-camera point ids.  M = n_points_in_camera of the frame. */
int gs_project_shard(gs_ctx* ctx, const gs_scene* shard, const gs_camera* camera, const gs_config* config,
                     float* records_out, int32_t* ids_out, int32_t keep_for_backward, gs_frame** frame_out, gs_stream stream);

/* The same stage without the host waiting for M: the kernels are queued and the call returns.  An owner that projects its
 * shard for W views calls this W times back to back -- the GPU runs the W projections without idling in between -- and only
 * then asks for the counts: the first gs_frame_get_info / gs_frame_export_count / gs_frame_export / gs_backward_shard on a
 * frame reads its hand-over (blocking until its kernels have run; an out-of-range point_object_id is reported there).
 * The records are fetched with gs_frame_export(GS_X_RECORDS), the ids with GS_X_POINT_ID_IN_CAMERA_LIST.  At most 63
 * frames can be begun and unread at a time per context; further ones wait at once like gs_project_shard. */
int gs_project_shard_begin(gs_ctx* ctx, const gs_scene* shard, const gs_camera* camera, const gs_config* config,
                           int32_t keep_for_backward, gs_frame** frame_out, gs_stream stream);

/* Per-pixel half of the forward from m records (any concatenation of shards' records; ties in the depth sort are
 * broken by position in this array, so concatenating shards in ascending point-id order reproduces gs_forward).
 * Only camera_height / camera_width of *camera are read. */
int gs_forward_projected(gs_ctx* ctx, const float* records, int64_t m, const gs_camera* camera, const gs_config* config,
                         const gs_forward_out* out, int32_t keep_for_backward, gs_frame** frame_out, gs_stream stream);

/* Per-pixel half of the backward: blend backward + per-splat sums.  splat_sums_out: device (m, 12) f32, every row
 * written; magnitude_grad_viewspace_on_image: device (H,W,2) or NULL. */
int gs_backward_projected(gs_ctx* ctx, gs_frame* frame, const float* grad_rasterized_image,
                          const float* pixel_accumulated_alpha, const int32_t* pixel_offset_of_last_effective_point,
                          float* splat_sums_out, float* magnitude_grad_viewspace_on_image, gs_stream stream);

/* Per-point half of the backward for the shard a gs_project_shard frame came from: Jacobian chain, band masks, grad
 * factors, hook payload (as gs_backward; out->magnitude_grad_viewspace_on_image is not written here).
 * splat_sums: device (M, 12), the rows of this shard's in-camera points in the order of records_out. */
int gs_backward_shard(gs_ctx* ctx, gs_frame* frame, const gs_scene* shard, const gs_camera* camera, const gs_config* config,
                      const float* splat_sums, int32_t color_max_sh_band, const gs_backward_out* out, gs_stream stream);

/* ---- the step either side of the operator in the reference's training loop ---------------- */

/* L = (1 - lambda) * L1 + lambda * (1 - SSIM), LossFunction.py:20-38, with SSIM as pytorch_msssim.ssim(
 * data_range=1, size_average=True): forward value AND d L / d predicted_image in three launches.
 * Both images are device (3,H,W) f32 contiguous, the layout GaussianPointTrainer.py:173-181 hands to the loss
 * (already clamped to [0,1] and permuted).  loss_terms: device float[3] = {L, L1, LD_SSIM}.
 * grad_predicted: device (3,H,W) or NULL.  H and W must be at least 11.  The scale regulariser of
 * LossFunction.py:40-51 is not part of this call. */
int gs_loss_l1_ssim(gs_ctx* ctx, const float* predicted_image, const float* ground_truth_image, int32_t height, int32_t width,
                    float lambda_value, float* loss_terms, float* grad_predicted, gs_stream stream);

/* The same loss as two calls, for an autograd node (forward now, gradient when and if backward runs) and for images that are not
 * contiguous (3,H,W) arrays: an image is described by its base pointer and the strides, in floats, of (channel, row, column)
 * -- {H*W, W, 1} for a contiguous (3,H,W) array, {1, 3*W, 3} for the rasteriser's (H,W,3) output read in place, which saves
 * the permuted copy GaussianPointTrainer.py:173 makes.  clamp_predicted != 0 applies torch.clamp(., 0, 1) to the predicted
 * image on the fly (same line of the trainer): same value, and a gradient of zero where the raw value lies outside [0, 1].
 * gs_loss_maps_floats(H, W) floats of device memory, owned by the caller, carry the forward's derivative maps to the backward. */
typedef struct gs_loss_image {
    const float* data;
    int64_t stride_channel, stride_row, stride_column;
} gs_loss_image;
int64_t gs_loss_maps_floats(int32_t height, int32_t width);
int gs_loss_l1_ssim_forward(gs_ctx* ctx, const gs_loss_image* predicted_image, const gs_loss_image* ground_truth_image,
                            int32_t height, int32_t width, int32_t clamp_predicted, float lambda_value,
                            float* maps, float* loss_terms, gs_stream stream);
/* upstream: device scalar d(final loss)/dL, or NULL for 1.  grad_predicted: where to write d/d predicted_image, with its own
 * strides (data is written despite the const of the shared struct). */
int gs_loss_l1_ssim_backward(gs_ctx* ctx, const gs_loss_image* predicted_image, const gs_loss_image* ground_truth_image,
                             int32_t height, int32_t width, int32_t clamp_predicted, float lambda_value,
                             const float* maps, const float* upstream, const gs_loss_image* grad_predicted, gs_stream stream);

/* Scale regulariser of LossFunction.py:40-51: mean over valid points (point_invalid_mask == 0) of
 * || exp(features[:, 4:7]) ||_2.  value_and_count: device float[2] = {mean, number of valid points}. */
int gs_scale_regulariser(gs_ctx* ctx, const float* point_cloud_features, const int8_t* point_invalid_mask, int64_t n_points,
                         float* value_and_count, gs_stream stream);
/* Its gradient: writes the whole (N,56) array (zero outside columns 4:7 and for invalid rows), scaled by the
 * device scalar *upstream; value_and_count is the output of gs_scale_regulariser for the same inputs. */
int gs_scale_regulariser_grad(gs_ctx* ctx, const float* point_cloud_features, const int8_t* point_invalid_mask, int64_t n_points,
                              const float* value_and_count, const float* upstream, float* grad_features, gs_stream stream);

/* One torch.optim.Adam step (betas, eps; no weight decay, no amsgrad) on a flat device f32 tensor of n
 * elements, replacing optimizer.step() / position_optimizer.step() of GaussianPointTrainer.py:131-134,183-184.
 * `step` is the 1-based step count used for the bias corrections. */
int gs_adam_step(gs_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, int64_t step, gs_stream stream);

/* Bytes of device memory the context currently owns (arena + frames). */
int64_t gs_ctx_device_bytes(const gs_ctx* ctx);

/* Diagnostic: nanoseconds the host has spent so far, over all calls on this ctx, waiting for a frame's counters (M, K,
 * depth-code range) to arrive from the device -- the one host dependency of a forward (tools/host_timeline.py). */
int64_t gs_ctx_counter_wait_ns(const gs_ctx* ctx);

/* Names of the kernels one forward+backward launches, comma separated; the position of a
 * name is its kernel id in the two calls below. */
const char* gs_kernel_names(void);

/* Per-kernel timing with HIP events recorded on the launch stream around the selected
 * kernels (bit i of kernel_mask selects kernel id i; 0 switches timing off).  Stands in for the
 * Taichi kernel profiler of the reference (GaussianPointTrainer.py:49,124,225-227). */
int gs_profile_enable(gs_ctx* ctx, uint64_t kernel_mask);
/* Waits for the recorded events, then copies accumulated milliseconds and launch counts of
 * kernel ids [0, n) out; reset != 0 clears the accumulators. */
int gs_profile_read(gs_ctx* ctx, double* total_ms, int64_t* launches, int32_t n, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
