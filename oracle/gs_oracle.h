/*
 * gs_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the rasterisation hot path of
 * Wenri/taichi_3d_gaussian_splatting, used only as the checker for the HIP
 * implementation:  only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The shipped path (libgsrast.so) never
 * includes, links or calls anything in this directory.
 *
 * Citation shorthand (paths under the reference repository):
 *   RAST = taichi_3d_gaussian_splatting/GaussianPointCloudRasterisation.py
 *   GP3D = taichi_3d_gaussian_splatting/GaussianPoint3D.py
 *   SH   = taichi_3d_gaussian_splatting/SphericalHarmonics.py
 *   UTIL = taichi_3d_gaussian_splatting/utils.py
 *
 * Pinning status: see oracle/README.md.  Pinned by the reference's own
 * fixtures where they exist (tile ranges RAST tests:19-42, single-point
 * alpha + Jacobians tests:353-548, covariance projection GP3D tests:12-54,
 * quaternion->R GP3D tests:56-67, pose inversion UTIL tests:127-157) and by a
 * float64 torch.autograd restatement (tests/torch_ref.py) for everything the
 * reference's tests leave open.  Whole-frame outputs of the Taichi kernels
 * themselves cannot be produced here (Taichi is not installable; the kernels
 * are CUDA-only): those rows are "parity unpinned" against real Taichi output.
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSO_TILE 16            /* RAST:27-28 */
#define GSO_BOUNDARY_TILES 3   /* RAST:26    */
#define GSO_FEAT 56            /* row layout RAST:208-236 */

typedef struct gso_config {
    float near_plane;                /* RAST:778 */
    float far_plane;                 /* RAST:779 */
    float depth_to_sort_key_scale;   /* RAST:780 */
    int   rgb_only;                  /* RAST:781 */
    float grad_color_factor;             /* RAST:782 (5)   */
    float grad_high_order_color_factor;  /* RAST:783 (1)   */
    float grad_s_factor;                 /* RAST:784 (0.5) */
    float grad_q_factor;                 /* RAST:785 (1)   */
    float grad_alpha_factor;             /* RAST:786 (20)  */
    int   radius_from_preblur_cov;   /* SURVEY 8a a5-vii switch; 1 = Taichi by-value semantics */
    int   allow_partial_tiles;       /* EXTENSION (not in the reference, which asserts W,H % 16 == 0): the last tile
                                        row/column may be partly outside the image; tile counts are rounded up */
    int   blend_exp;                 /* which exp the two blend loops use for the Gaussian falloff (RAST:441-452, UTIL:275-284,
                                        UTIL:331-348; `ti.exp` there, compiled by Taichi with fast-math to the GPU's fast exp):
                                        GSO_EXP_POLY (0, default) = gso_exp_blend, the sequence libgsrast shares bit for bit;
                                        GSO_EXP_LIBM (1) = libm expf (correctly rounded to < 1 ulp);
                                        GSO_EXP_FAST2 (2) = exp2f(x * log2(e)) with the product rounded to f32 first, the shape of
                                        CUDA's __expf; GSO_EXP_ULP2 (3) = that result moved by a pseudo-random -2..+2 ulp, the error
                                        a hardware approximation unit (ex2.approx: 2 ulp) is allowed.  1..3 exist to MEASURE how much the choice of exp moves the index outputs
                                        (tests/test_oracle_exp_sensitivity.py); the HIP path is compared against 0 only. */
    int   bwd_strict_dpdcov;         /* loop 1's d p / d Sigma' (UTIL:343-345).  1 (default) = the reference's own f32 operation
                                        order, 0.5 p (Sigma^-1 (d d^T) Sigma^-1) with two 2x2 products; 0 = the algebraically equal
                                        v v^T with v = Sigma^-1 d that libgsrast's default (fast) backward evaluates.  The 0 form
                                        exists to show, on the CPU alone, that this one expression is what separates the two
                                        (tests/test_oracle_exp_sensitivity.py::test_dpdcov_order_is_the_soak_gap). */
} gso_config;
#define GSO_EXP_POLY 0
#define GSO_EXP_LIBM 1
#define GSO_EXP_FAST2 2
#define GSO_EXP_ULP2 3

/* Everything the forward produces, in the reference's own layouts. */
typedef struct gso_frame {
    int64_t N, M, K;
    int32_t H, W, tiles_x, tiles_y, T, n_objects;
    float*   q_camera_pointcloud;   /* (Kobj,4)  RAST:845 */
    float*   t_camera_pointcloud;   /* (Kobj,3) */
    int8_t*  point_in_camera_mask;  /* (N)   RAST:848-861 */
    int32_t* point_id_in_camera_list; /* (M) RAST:864 */
    float*   point_uv;              /* (M,2) */
    float*   point_in_camera;       /* (M,3) */
    float*   point_uv_conic_and_rescale; /* (M,4) */
    float*   point_alpha_after_activation; /* (M) */
    float*   point_color;           /* (M,3) */
    float*   point_radii;           /* (M) */
    int32_t* num_overlap_tiles;     /* (M)  RAST:904-911 */
    int64_t* accumulated_num_overlap_tiles; /* (M) exclusive RAST:913-922 */
    int64_t* sort_key_unsorted;     /* (K)  RAST:934-945 */
    int32_t* point_offset_unsorted; /* (K) */
    int64_t* sort_key;              /* (K) sorted RAST:947 */
    int32_t* point_offset_with_sort_key; /* (K) sorted RAST:948 */
    int32_t* tile_points_start;     /* (T)  RAST:952-964 */
    int32_t* tile_points_end;       /* (T) */
    float*   rasterized_image;      /* (H,W,3) */
    float*   rasterized_depth;      /* (H,W) */
    float*   pixel_accumulated_alpha; /* (H,W) */
    int32_t* pixel_offset_of_last_effective_point; /* (H,W) */
    int32_t* pixel_valid_point_count; /* (H,W) */
    int32_t  blend_exp;             /* the gso_config.blend_exp this frame was blended with: its backward uses the same one */
} gso_frame;

/* f32 exp used everywhere the reference writes ti.exp / ti.math.exp.
 * Cody-Waite reduction + Cephes degree-5 polynomial, only IEEE add/mul/fma:
 * the HIP kernels implement the same operation sequence so index-determining
 * thresholds agree bit for bit. */
float gso_expf(float x);
float gso_exp_blend(float x);

/* RAST:845 + UTIL:396-432  (inverse_SE3_qt_torch) */
void gso_inverse_se3_qt(const float* q, const float* t, int n, float* q_inv, float* t_inv);

/* GP3D:30-48 */
void gso_rotation_matrix_from_quaternion(const float q[4], float R[9]);

/* GP3D:161-191 : 2x2 projected covariance (row-major c[4]) */
void gso_project_to_camera_covariance(const float q_cov[4], const float log_s[3],
                                      const float T_camera_world[16], const float Kmat[9],
                                      const float xyz_camera[3], float cov[4]);

/* GP3D:132-159 : d uv / d xyz  (2x3 row-major) */
void gso_project_to_camera_position_jacobian(const float xyz[3], const float T_camera_world[16],
                                             const float Kmat[9], float J[6]);

/* GP3D:237-331 : dSigma'/dq (4x4 row-major) and dSigma'/ds (4x3 row-major) */
void gso_project_to_camera_covariance_jacobian(const float q_cov[4], const float log_s[3],
                                               const float T_camera_world[16], const float Kmat[9],
                                               const float xyz_camera[3],
                                               float dSigma_dq[16], float dSigma_ds[12]);

/* SH:10-32 : 16 real SH basis values of the normalised direction */
void gso_spherical_harmonics(const float d[3], float sh[16]);

/* UTIL:257-272 : conic (a,b,c) + rescale from a 2x2 covariance */
void gso_conic_and_rescale(const float cov[4], float out[4]);

/* RAST:81-103 */
void gso_bounding_box(float u, float v, float radii, int W, int H, int32_t box[4]);

/* RAST:175-193 ; arrays must be zero-initialised by the caller (RAST:954-957) */
void gso_find_tile_start_and_end(const int64_t* sorted_keys, int64_t n,
                                 int32_t* tile_start, int32_t* tile_end);

/* Whole forward, RAST:830-1023.  point_cloud_features is modified in place
 * (quaternion normalisation, RAST:264-266).  Returns NULL on bad arguments. */
gso_frame* gso_forward(const float* point_cloud, float* point_cloud_features,
                       const int8_t* point_invalid_mask, const int32_t* point_object_id,
                       int64_t N,
                       const float* q_pointcloud_camera, const float* t_pointcloud_camera,
                       int32_t n_objects,
                       const float* camera_intrinsics /*3x3*/, int32_t H, int32_t W,
                       const gso_config* cfg);

/* Whole backward, RAST:1025-1163.  All outputs caller-allocated:
 *  grad_pointcloud (N,3), grad_pointcloud_features (N,56) [band-masked and
 *  factor-scaled, RAST:1102-1125], grad_viewspace (N,2), magnitude_grad_viewspace (N),
 *  magnitude_grad_viewspace_on_image (H,W,2), num_affected_pixels (M),
 *  optional raw buffers grad_uv_cov (M,3) / grad_color (M,3) (may be NULL).
 * Cross-pixel sums (the reference's ti.atomic_add, order unspecified) are
 * accumulated in double and rounded once. */
int gso_backward(const gso_frame* f,
                 const float* point_cloud, const float* point_cloud_features,
                 const int32_t* point_object_id,
                 const float* q_pointcloud_camera, const float* t_pointcloud_camera,
                 const float* camera_intrinsics,
                 const float* grad_rasterized_image, int32_t color_max_sh_band,
                 const gso_config* cfg,
                 float* grad_pointcloud, float* grad_pointcloud_features,
                 float* grad_viewspace, float* magnitude_grad_viewspace,
                 float* magnitude_grad_viewspace_on_image, int32_t* num_affected_pixels,
                 float* grad_uv_cov_buffer, float* grad_color_buffer);

/* gso_backward plus, optionally (each may be NULL), for every element of the two returned gradients the magnitude of
 * what was summed to produce it: sum |loop-1 contribution| (RAST:674-696) carried through |loop-2 Jacobian| and the
 * grad factors -- (N,3) and (N,56).  Floor of the tests' per-element bar. */
int gso_backward_ex(const gso_frame* f,
                    const float* point_cloud, const float* point_cloud_features,
                    const int32_t* point_object_id,
                    const float* q_pointcloud_camera, const float* t_pointcloud_camera,
                    const float* camera_intrinsics,
                    const float* grad_rasterized_image, int32_t color_max_sh_band,
                    const gso_config* cfg,
                    float* grad_pointcloud, float* grad_pointcloud_features,
                    float* grad_viewspace, float* magnitude_grad_viewspace,
                    float* magnitude_grad_viewspace_on_image, int32_t* num_affected_pixels,
                    float* grad_uv_cov_buffer, float* grad_color_buffer,
                    float* summed_pointcloud, float* summed_pointcloud_features);

/* ---- the same path cut at the projected records and at the per-point sums: stand-ins, on the CPU, for the staged entry
 * points of the library under test (gs_project_shard / gs_forward_projected / gs_backward_projected / gs_backward_shard).
 * Records: 16 floats per in-camera point (layout in gs_oracle.c); sums: 12 floats per in-camera point = the reference's
 * loop-1 accumulators (post-factor, unlike the library's rows -- each half is only ever paired with its own other half). */
void gso_pack_records(const gso_frame* f, float* records_out);                 /* (M,16) from any frame */
gso_frame* gso_forward_from_projected(int64_t M, const float* records, int32_t H, int32_t W, const gso_config* cfg);
int gso_backward_sums(const gso_frame* f, const float* grad_rasterized_image, float* sums_out /* (M,12) */,
                      float* magnitude_grad_viewspace_on_image /* (H,W,2) */);
int gso_backward_points(const gso_frame* f /* the shard's own forward frame */, const float* point_cloud,
                        const float* point_cloud_features, const int32_t* point_object_id, const float* t_pointcloud_camera,
                        const float* camera_intrinsics, const float* sums /* (M,12) */, int32_t color_max_sh_band,
                        const gso_config* cfg, float* grad_pointcloud, float* grad_pointcloud_features,
                        float* grad_viewspace, float* magnitude_grad_viewspace, int32_t* num_affected_pixels);

void gso_frame_free(gso_frame* f);
int  gso_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
