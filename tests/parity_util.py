"""Shared helpers of the GPU parity tests: run the HIP operator (through the C ABI) and the
CPU oracle on the same inputs and compare every product."""
import numpy as np
import torch

from oracle import oracle
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation
from taichi_3d_gaussian_splatting_amd.synthetic import synth, view_pose

Rast = GaussianPointCloudRasterisation
DEV = "cuda:0"

INT_EXPORTS = ["point_id_in_camera_list", "num_overlap_tiles", "accumulated_num_overlap_tiles", "sort_key",
               "point_offset_with_sort_key", "tile_points_start", "tile_points_end", "point_in_camera_mask"]
FLOAT_EXPORTS = ["point_uv", "point_in_camera", "point_uv_conic_and_rescale", "point_alpha_after_activation",
                 "point_color", "point_radii"]


def make_input(scene, q, t, band=3, requires_grad=True):
    pc = torch.tensor(scene.point_cloud, device=DEV, requires_grad=requires_grad)
    feat = torch.tensor(scene.point_cloud_features, device=DEV, requires_grad=requires_grad)
    inp = Rast.GaussianPointCloudRasterisationInput(
        point_cloud=pc, point_cloud_features=feat,
        point_object_id=torch.tensor(scene.point_object_id, device=DEV),
        point_invalid_mask=torch.tensor(scene.point_invalid_mask, device=DEV),
        camera_info=CameraInfo(camera_intrinsics=torch.tensor(scene.camera_intrinsics, device=DEV),
                               camera_height=scene.height, camera_width=scene.width, camera_id=0),
        q_pointcloud_camera=torch.tensor(q, device=DEV), t_pointcloud_camera=torch.tensor(t, device=DEV),
        color_max_sh_band=band)
    return inp


def run_oracle(scene, q, t, cfg=None):
    return oracle.forward(scene.point_cloud, scene.point_cloud_features, scene.point_invalid_mask,
                          scene.point_object_id, q, t, scene.camera_intrinsics, scene.height, scene.width, cfg)


IMAGE_TOL = 2e-6      # max |a - ref| / max |ref| of the blended images (fused multiply-adds, shared weight alpha*T)


def assert_forward_parity(module, inp, outs, f, feat_after, rgb_only=False):
    """Bit-exact on every integer array and on every per-point f32 array of the forward (uv, conic, colour, radii ...,
    which decide every index); the blended images (sums of up to thousands of terms) within IMAGE_TOL, fifty times
    tighter than the 1e-4 bar; accumulated alpha (1 - T) bit-exact again because T follows the reference sequence."""
    image, depth, count = outs
    fr = module.last_frame
    assert fr.n_points_in_camera == f.M, (fr.n_points_in_camera, f.M)
    assert fr.n_keys == f.K, (fr.n_keys, f.K)
    for name in INT_EXPORTS:
        got = fr.export(name).cpu().numpy()
        assert np.array_equal(got, getattr(f, name)), f"{name} differs"
    for name in FLOAT_EXPORTS:
        got = fr.export(name).cpu().numpy()
        ref = getattr(f, name)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
            f"{name}: not bit-exact, max abs diff {np.abs(got - ref).max() if ref.size else 0}"
    # in-place quaternion normalisation, RAST:264-266
    assert np.array_equal(inp.point_cloud_features.detach().cpu().numpy().view(np.uint32), feat_after.view(np.uint32))
    e = rel_err(image.detach().cpu().numpy(), f.rasterized_image)
    assert e < IMAGE_TOL, ("image", e)
    acc = module.last_forward_outputs["pixel_accumulated_alpha"].cpu().numpy() if not rgb_only else None
    if acc is not None:
        assert np.array_equal(acc.view(np.uint32), f.pixel_accumulated_alpha.view(np.uint32)), "pixel_accumulated_alpha"
        assert np.array_equal(module.last_forward_outputs["pixel_offset_of_last_effective_point"].cpu().numpy(),
                              f.pixel_offset_of_last_effective_point), "pixel_offset_of_last_effective_point"
    if not rgb_only:
        assert np.array_equal(count.cpu().numpy(), f.pixel_valid_point_count), "pixel_valid_point_count"
        e = rel_err(depth.detach().cpu().numpy(), f.rasterized_depth)
        assert e < IMAGE_TOL, ("depth", e)


def rel_err(a, ref):
    """max |a-ref| / max|ref|  (the 1e-4 'relative fp32' bar of BASELINE.md section 4)."""
    scale = float(np.abs(ref).max())
    if scale == 0.0:
        return float(np.abs(a).max())
    return float(np.abs(a - ref).max()) / scale


GRAD_TOL = 1e-4


def assert_backward_parity(module, inp, g_image, f, band, extras=None, cfg=None):
    b = oracle.backward(f, g_image, band, cfg)
    gp = inp.point_cloud.grad.cpu().numpy()
    gf = inp.point_cloud_features.grad.cpu().numpy()
    assert rel_err(gp, b["grad_pointcloud"]) < GRAD_TOL, ("xyz", rel_err(gp, b["grad_pointcloud"]))
    for lo, hi, name in [(0, 4, "q"), (4, 7, "s"), (7, 8, "opacity"), (8, 56, "sh")]:
        e = rel_err(gf[:, lo:hi], b["grad_pointcloud_features"][:, lo:hi])
        assert e < GRAD_TOL, (name, e)
    # rows outside the frustum are exactly zero
    out = np.setdiff1d(np.arange(f.N), f.point_id_in_camera_list)
    assert not gp[out].any() and not gf[out].any()
    if extras is not None:
        assert rel_err(extras["grad_viewspace"].cpu().numpy(), b["grad_viewspace"]) < GRAD_TOL
        assert rel_err(extras["magnitude_grad_viewspace"].cpu().numpy(), b["magnitude_grad_viewspace"]) < GRAD_TOL
        assert rel_err(extras["magnitude_grad_viewspace_on_image"].cpu().numpy(), b["magnitude_grad_viewspace_on_image"]) < GRAD_TOL
        assert np.array_equal(extras["num_affected_pixels"].cpu().numpy(), b["num_affected_pixels"]), "num_affected_pixels"
    return b
