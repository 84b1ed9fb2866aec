"""Shared helpers of the GPU parity tests: run the HIP operator (through the C ABI) and the
CPU oracle on the same inputs and compare every product."""
import numpy as np
import torch

from oracle import oracle
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation
from taichi_3d_gaussian_splatting_amd.synthetic import synth, view_pose

import os

Rast = GaussianPointCloudRasterisation
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


GRAD_TOL = 1e-4          # tensor-level: max |a - ref| / max |ref| per column group
# Per-element bar: |a - ref| <= ELEM_RTOL * |ref| + ELEM_FLOOR * S.  S is the oracle's "summed" magnitude of the
# element (gso_backward_ex): the reference's own expressions for the element with every product replaced by its
# absolute value all the way down -- sum over the loop-1 contributions (RAST:674-696) of the un-cancelled size of each
# contribution (colour*T against w/(1-alpha) in d alpha, a*dx against b*dy in Sigma^-1 d, the terms of Sigma^-1 d d^T Sigma^-1),
# through |loop-2 Jacobian| and the grad factor.  That is the quantity a float evaluation can be accurate relative to
# (the componentwise forward-error bound): an element is a sum of up to ~1e4 signed terms, each itself a difference,
# and the reference adds them with unordered f32 atomics.  The floor covers, per product: hardware-exp alpha (2e-6
# relative), the T recurrence drifting by an ulp per step over a pixel's list (~6e-7 rms over 100 steps), and f32
# summation of n <= 1e4 terms (sqrt(n) * 2^-24 ~ 6e-6 worst case, measured 1.7e-6): 5e-6 in all.  Measured at
# BASELINE configs 2 and 3 (profiles/r02_parity_margins.json): no element uses more than a third of this bar.
ELEM_RTOL = 2e-5         # five times tighter than the 1e-4 of BASELINE.md
ELEM_FLOOR = 5e-6
GROUPS = [(0, 4, "q"), (4, 7, "s"), (7, 8, "opacity"), (8, 56, "sh")]


def elem_margins(a, ref, summed):
    """Per-element statistics of one column group: the worst and 99.9th-percentile use of the per-element bar
    (1.0 = at the bar), the same for the error relative to |ref| over well-conditioned elements (S <= 10 |ref|), and for
    the error relative to S over all elements."""
    a, ref, summed = a.astype(np.float64).ravel(), ref.astype(np.float64).ravel(), summed.astype(np.float64).ravel()
    err = np.abs(a - ref)
    live = summed > 0
    if not live.any():
        return dict(elements=0, bar_use_max=float(err.max()) if err.size else 0.0, bar_use_p999=0.0, rel_max=0.0, rel_p999=0.0,
                    of_summed_max=0.0, of_summed_p999=0.0, outside_live_max_abs=float(err.max()) if err.size else 0.0)
    use = err[live] / (ELEM_RTOL * np.abs(ref[live]) + ELEM_FLOOR * summed[live])
    well = live & (summed <= 10.0 * np.abs(ref))
    rel = err[well] / np.abs(ref[well]) if well.any() else np.zeros(1)
    ofs = err[live] / summed[live]
    q = lambda x: float(np.quantile(x, 0.999))
    return dict(elements=int(live.sum()), bar_use_max=float(use.max()), bar_use_p999=q(use),
                well_conditioned_elements=int(well.sum()), rel_max=float(rel.max()), rel_p999=q(rel),
                of_summed_max=float(ofs.max()), of_summed_p999=q(ofs),
                outside_live_max_abs=float(err[~live].max()) if (~live).any() else 0.0)


def backward_margins(gp, gf, b):
    """{group: elem_margins} for the two returned gradients against an oracle.backward(..., want_summed=True) result."""
    out = {"xyz": elem_margins(gp, b["grad_pointcloud"], b["summed_pointcloud"])}
    for lo, hi, name in GROUPS:
        out[name] = elem_margins(gf[:, lo:hi], b["grad_pointcloud_features"][:, lo:hi], b["summed_pointcloud_features"][:, lo:hi])
    return out


def assert_backward_parity(module, inp, g_image, f, band, extras=None, cfg=None, tensor_tol=GRAD_TOL):
    b = oracle.backward(f, g_image, band, cfg, want_summed=True)
    gp = inp.point_cloud.grad.cpu().numpy()
    gf = inp.point_cloud_features.grad.cpu().numpy()
    assert rel_err(gp, b["grad_pointcloud"]) < tensor_tol, ("xyz", rel_err(gp, b["grad_pointcloud"]))
    for lo, hi, name in GROUPS:
        e = rel_err(gf[:, lo:hi], b["grad_pointcloud_features"][:, lo:hi])
        assert e < tensor_tol, (name, e)
    # every element on its own: within ELEM_RTOL of its value plus ELEM_FLOOR of the magnitude summed to produce it;
    # where nothing was summed (rows outside the frustum, masked SH bands, pixels-free splats) exactly zero
    margins = backward_margins(gp, gf, b)
    for name, m in margins.items():
        assert m["bar_use_max"] <= 1.0, (name, m)
        assert m["outside_live_max_abs"] == 0.0, (name, m)
    # rows outside the frustum are exactly zero
    out = np.setdiff1d(np.arange(f.N), f.point_id_in_camera_list)
    assert not gp[out].any() and not gf[out].any()
    if extras is not None:
        assert rel_err(extras["grad_viewspace"].cpu().numpy(), b["grad_viewspace"]) < GRAD_TOL
        assert rel_err(extras["magnitude_grad_viewspace"].cpu().numpy(), b["magnitude_grad_viewspace"]) < GRAD_TOL
        assert rel_err(extras["magnitude_grad_viewspace_on_image"].cpu().numpy(), b["magnitude_grad_viewspace_on_image"]) < GRAD_TOL
        assert np.array_equal(extras["num_affected_pixels"].cpu().numpy(), b["num_affected_pixels"]), "num_affected_pixels"
    b["margins"] = margins
    return b


# ---- the randomised scenes of tools/parity_soak.py (also the source of the two seeds kept in the suite) ------------
def soak_case(seed):
    """Seeded scene + pose + config of the parity soak: image sizes 16..640 (half of them not multiples of 16),
    30..50 000 Gaussians, two decades of splat scale, a non-unit pose quaternion, optional invalid rows, SH band 0..3."""
    rng = np.random.default_rng(seed)
    partial = bool(rng.integers(0, 2))
    W = int(rng.integers(1, 40)) * 16 + (int(rng.integers(1, 16)) if partial else 0)
    H = int(rng.integers(1, 30)) * 16 + (int(rng.integers(1, 16)) if partial else 0)
    n = int(10 ** rng.uniform(1.5, 4.7))
    sigma0 = float(10 ** rng.uniform(-2.3, -0.2))
    band = int(rng.integers(0, 4))
    s = synth(n, W, H, sigma0, sh_deg=3, seed=seed)
    if rng.random() < 0.3:
        s.point_invalid_mask[rng.random(n) < 0.2] = 1
    ang = rng.normal(0, 0.15, 3)
    q = np.array([[ang[0], ang[1], ang[2], 1.0]], np.float32) * float(rng.uniform(0.5, 2.0))    # deliberately not unit
    t = rng.normal(0, 0.3, (1, 3)).astype(np.float32)
    return dict(scene=s, q=q, t=t, band=band, partial=partial, rng=rng, W=W, H=H, n=n, sigma0=sigma0)


def float64_autograd_gradients(scene, q, t, f, feat_after, g_image):
    """Gradients of tests/torch_ref.py (float64 torch.autograd restatement, CPU) for the oracle frame f: (xyz, features);
    all grad factors 1, all SH bands."""
    import torch_ref
    pc = torch.tensor(scene.point_cloud, dtype=torch.float64, requires_grad=True)
    ft = torch.tensor(feat_after, dtype=torch.float64, requires_grad=True)
    img, _ = torch_ref.render(pc, ft, q, t, scene.camera_intrinsics, scene.height, scene.width, f)
    img.backward(torch.as_tensor(g_image, dtype=torch.float64))
    return pc.grad.numpy(), ft.grad.numpy()
