"""Randomised parity soak (GPU): many seeded scenes of varied size, density, splat scale, pose, SH band and image
shape (including partial edge tiles) through the C ABI against the CPU oracle, with the bars of tests/test_gpu_parity.py
(integers and forward f32 bit-exact, gradients 1e-4 of the tensor maximum).  Not part of the test suite (minutes).

    python tools/parity_soak.py [n_cases] [first_seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import parity_util as P  # noqa: E402
from oracle import oracle  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import synth  # noqa: E402


def one(seed):
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
    cfg = P.Rast.GaussianPointCloudRasterisationConfig()
    cfg.allow_partial_tiles = partial
    got = {}
    module = P.Rast(cfg, backward_valid_point_hook=lambda x: got.setdefault("hook", x))
    inp = P.make_input(s, q, t, band)
    ocfg = oracle.default_config(allow_partial_tiles=int(partial))
    f, feat_after = P.run_oracle(s, q, t, ocfg)
    outs = module(inp)
    P.assert_forward_parity(module, inp, outs, f, feat_after)
    image = outs[0]
    target = torch.tensor(rng.uniform(0, 1, image.shape).astype(np.float32), device=image.device)
    g = 2.0 * (image.detach() - target)
    image.backward(g)
    note = ""
    try:
        P.assert_backward_parity(module, inp, g.cpu().numpy(), f, band, module.last_backward_extras, ocfg)
    except AssertionError as e:
        # The oracle evaluates the per-point Jacobian chain in f32 in the reference's operation order; for an ill-conditioned
        # splat (huge, strongly anisotropic) that alone can be 1e-4 of the tensor maximum away from the exact value.  A
        # float64 autograd restatement (tests/torch_ref.py, CPU, slow) arbitrates: the GPU has to be within the bar of IT.
        worst = arbitrate(s, q, t, rng_state_target=target.cpu().numpy(), partial=partial)
        if worst >= P.GRAD_TOL:
            raise AssertionError(f"{e}; against float64 autograd: {worst:.3e}")
        note = f" [oracle-limited case: oracle check said {e}, GPU vs float64 autograd {worst:.2e}]"
    return dict(seed=seed, W=W, H=H, n=n, sigma0=round(sigma0, 4), band=band, M=f.M, K=f.K, note=note)


def arbitrate(s, q, t, rng_state_target, partial):
    """max over the gradient tensors of |GPU - float64 autograd| / max|float64|, all grad factors 1, all SH bands."""
    import torch_ref
    cfg = P.Rast.GaussianPointCloudRasterisationConfig()
    cfg.allow_partial_tiles = partial
    cfg.grad_color_factor = cfg.grad_high_order_color_factor = cfg.grad_s_factor = cfg.grad_q_factor = cfg.grad_alpha_factor = 1.0
    module = P.Rast(cfg)
    inp = P.make_input(s, q, t, 3)
    ocfg = oracle.default_config(allow_partial_tiles=int(partial), grad_color_factor=1.0, grad_high_order_color_factor=1.0,
                                 grad_s_factor=1.0, grad_q_factor=1.0, grad_alpha_factor=1.0)
    f, feat_after = P.run_oracle(s, q, t, ocfg)
    image = module(inp)[0]
    g = 2.0 * (image.detach() - torch.tensor(rng_state_target, device=image.device))
    image.backward(g)
    pc = torch.tensor(s.point_cloud, dtype=torch.float64, requires_grad=True)
    ft = torch.tensor(feat_after, dtype=torch.float64, requires_grad=True)
    img, _ = torch_ref.render(pc, ft, q, t, s.camera_intrinsics, s.height, s.width, f)
    img.backward(g.cpu().double())
    worst = P.rel_err(inp.point_cloud.grad.cpu().numpy(), pc.grad.numpy())
    gf, rf = inp.point_cloud_features.grad.cpu().numpy(), ft.grad.numpy()
    for lo, hi in [(0, 4), (4, 7), (7, 8)]:
        worst = max(worst, P.rel_err(gf[:, lo:hi], rf[:, lo:hi]))
    # SH gradients: the float64 restatement evaluates the colour with the forward's ray origin, the reference's backward with
    # t_pointcloud_camera (RAST:731-732), which differ for the non-unit pose quaternions used here; they are well conditioned
    # and stay on the oracle's bar
    b = oracle.backward(f, g.cpu().numpy(), 3, ocfg)
    worst = max(worst, P.rel_err(gf[:, 8:56], b["grad_pointcloud_features"][:, 8:56]))
    return worst


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    for i in range(n_cases):
        info = one(first + i)
        note = info.pop("note")
        print(f"ok {info}{note}  [{time.time() - t0:.0f} s]", flush=True)
    print(f"{n_cases} cases passed")


if __name__ == "__main__":
    main()
