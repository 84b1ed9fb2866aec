"""CPU: the oracle's analytic backward against float64 torch.autograd on tiny scenes
(tests/torch_ref.py).  Pins what the reference's tests leave open: loop-1 accumulation,
SH colour gradient, opacity gradient, loop-2 chain, band masks and grad factors."""
import numpy as np
import pytest
import torch

import torch_ref
from oracle import oracle
from taichi_3d_gaussian_splatting_amd.synthetic import synth


def _tiny(seed, n, sigma0, size=32, height=None):
    height = size if height is None else height
    s = synth(n, size, height, sigma0, sh_deg=3, seed=seed)
    rng = np.random.default_rng(seed + 100)
    ang = 0.05
    q = np.array([[0.02, np.sin(ang / 2), -0.01, np.cos(ang / 2)]], np.float32)   # deliberately not unit
    t = np.array([[0.03, -0.02, 0.1]], np.float32)
    target = rng.uniform(0, 1, (height, size, 3)).astype(np.float32)
    return s, q, t, target


@pytest.mark.parametrize("seed,n,sigma0,width,height", [(0, 48, 0.25, 32, 32), (1, 64, 0.6, 32, 32), (2, 24, 1.2, 32, 32),
                                                         (3, 56, 0.5, 41, 27)])
def test_backward_matches_autograd(seed, n, sigma0, width, height):
    """The last case is the partial-tile extension (41x27: neither side a multiple of 16)."""
    s, q, t, target = _tiny(seed, n, sigma0, width, height)
    partial = int(width % 16 != 0 or height % 16 != 0)
    cfg = oracle.default_config(grad_color_factor=1.0, grad_high_order_color_factor=1.0, grad_s_factor=1.0,
                                grad_q_factor=1.0, grad_alpha_factor=1.0, allow_partial_tiles=partial)
    f, feat_after = oracle.forward(s.point_cloud, s.point_cloud_features, s.point_invalid_mask,
                                   s.point_object_id, q, t, s.camera_intrinsics, s.height, s.width, cfg)
    assert f.K > 0 and f.pixel_valid_point_count.max() >= 3
    pc = torch.tensor(s.point_cloud, dtype=torch.float64, requires_grad=True)
    ft = torch.tensor(feat_after, dtype=torch.float64, requires_grad=True)
    img, aux = torch_ref.render(pc, ft, q, t, s.camera_intrinsics, s.height, s.width, f)
    # forward agreement first (f32 oracle vs f64 restatement)
    assert np.allclose(img.detach().numpy(), f.rasterized_image, atol=2e-5)
    g_img = 2.0 * (f.rasterized_image.astype(np.float64) - target)
    img.backward(torch.tensor(g_img))
    b = oracle.backward(f, g_img.astype(np.float32), color_max_sh_band=3, cfg=cfg)

    def close(a, ref, name):
        scale = np.abs(ref).max() + 1e-12
        err = np.abs(a - ref).max() / scale
        assert err < 2e-4, f"{name}: max err / max|ref| = {err:.3e}"

    close(b["grad_pointcloud"], pc.grad.numpy(), "xyz")
    gf = ft.grad.numpy()
    close(b["grad_pointcloud_features"][:, 0:4], gf[:, 0:4], "q")
    close(b["grad_pointcloud_features"][:, 4:7], gf[:, 4:7], "s")
    close(b["grad_pointcloud_features"][:, 7], gf[:, 7], "opacity")
    close(b["grad_pointcloud_features"][:, 8:], gf[:, 8:], "sh")
    ids = f.point_id_in_camera_list
    close(b["grad_viewspace"][ids], aux["uv"].grad.numpy(), "viewspace")


def test_band_mask_and_factors():
    """RAST:1102-1125, 1167-1182: masked bands are zero, the rest scaled by 5 / 1 / 0.5 / 1 / 20."""
    s, q, t, target = _tiny(3, 40, 0.5)
    unit = oracle.default_config(grad_color_factor=1.0, grad_high_order_color_factor=1.0, grad_s_factor=1.0,
                                 grad_q_factor=1.0, grad_alpha_factor=1.0)
    ref_cfg = oracle.default_config()
    f, _ = oracle.forward(s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id,
                          q, t, s.camera_intrinsics, s.height, s.width, unit)
    g = (2.0 * (f.rasterized_image - target)).astype(np.float32)
    raw = oracle.backward(f, g, 3, unit)["grad_pointcloud_features"]
    for band, keep in [(0, 1), (1, 4), (2, 9), (3, 16), (5, 16)]:
        got = oracle.backward(f, g, band, ref_cfg)["grad_pointcloud_features"]
        exp = raw.copy()
        exp[:, 0:4] *= 1.0
        exp[:, 4:7] *= 0.5
        exp[:, 7] *= 20.0
        for base in (8, 24, 40):
            exp[:, base] *= 5.0
            exp[:, base + keep: base + 16] = 0.0
        assert np.allclose(got, exp, rtol=1e-6, atol=1e-12), band


def test_hook_extras_consistency():
    """num_affected_pixels counts contributions; magnitude image is the per-pixel sum of |d uv|."""
    s, q, t, target = _tiny(4, 32, 0.5)
    f, _ = oracle.forward(s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id,
                          q, t, s.camera_intrinsics, s.height, s.width)
    g = (2.0 * (f.rasterized_image - target)).astype(np.float32)
    b = oracle.backward(f, g, 3)
    # every blended contribution is revisited in backward unless it straddles the 1/255 edge
    assert abs(int(b["num_affected_pixels"].sum()) - int(f.pixel_valid_point_count.sum())) <= 2
    assert b["magnitude_grad_viewspace_on_image"].min() >= 0
    assert b["magnitude_grad_viewspace"].min() >= 0
    # points outside the frustum get exactly zero gradient (RAST:1051-1058 zero-init)
    out = np.setdiff1d(np.arange(f.N), f.point_id_in_camera_list)
    assert np.all(b["grad_pointcloud"][out] == 0) and np.all(b["grad_pointcloud_features"][out] == 0)


def test_summed_magnitudes_bound_the_gradients():
    """gso_backward_ex's "summed" outputs (the floor of the GPU tests' per-element bar): every gradient element is bounded by
    the magnitude summed to produce it (triangle inequality, up to rounding), is exactly zero where nothing was summed,
    and masked SH bands / rows outside the frustum have nothing summed."""
    s, q, t, target = _tiny(5, 400, 0.3, 64, 48)
    f, _ = oracle.forward(s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id,
                          q, t, s.camera_intrinsics, s.height, s.width)
    g = (2.0 * (f.rasterized_image - target)).astype(np.float32)
    b = oracle.backward(f, g, 1, want_summed=True)
    plain = oracle.backward(f, g, 1)
    assert np.array_equal(plain["grad_pointcloud_features"], b["grad_pointcloud_features"])
    for gk, sk in [("grad_pointcloud", "summed_pointcloud"), ("grad_pointcloud_features", "summed_pointcloud_features")]:
        G, S = np.abs(b[gk]).astype(np.float64), b[sk].astype(np.float64)
        assert np.all(G <= S * (1 + 1e-5)), gk
        assert not G[S == 0].any(), gk
        assert (S > 0).any()
    Sf = b["summed_pointcloud_features"]
    for base in (8, 24, 40):
        assert not Sf[:, base + 4: base + 16].any()           # band 1 keeps coefficients 0..3
    out = np.setdiff1d(np.arange(f.N), f.point_id_in_camera_list)
    assert not Sf[out].any() and not b["summed_pointcloud"][out].any()
