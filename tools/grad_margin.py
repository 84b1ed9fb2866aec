"""Gradient error of the HIP backward against the oracle at config 1/3 (max |diff| / max |ref| per tensor): the margin to the 1e-4 bar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_util as P
from oracle import oracle
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose

for wl in sys.argv[1:] or ["cfg1_plumbing", "cfg3_headline"]:
    s = synth(**CONFIGS[wl]); q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda x: None)
    inp = P.make_input(s, q, t, 3)
    f, _ = P.run_oracle(s, q, t)
    image = module(inp)[0]
    rng = np.random.default_rng(0)
    g = 2.0 * (image.detach() - torch.tensor(rng.uniform(0, 1, image.shape).astype(np.float32), device=image.device))
    image.backward(g)
    b = oracle.backward(f, g.cpu().numpy(), 3)
    gp, gf = inp.point_cloud.grad.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy()
    ex = module.last_backward_extras
    out = {"xyz": P.rel_err(gp, b["grad_pointcloud"])}
    for lo, hi, name in [(0, 4, "q"), (4, 7, "s"), (7, 8, "opacity"), (8, 56, "sh")]:
        out[name] = P.rel_err(gf[:, lo:hi], b["grad_pointcloud_features"][:, lo:hi])
    out["viewspace"] = P.rel_err(ex["grad_viewspace"].cpu().numpy(), b["grad_viewspace"])
    out["magnitude"] = P.rel_err(ex["magnitude_grad_viewspace"].cpu().numpy(), b["magnitude_grad_viewspace"])
    out["mag_image"] = P.rel_err(ex["magnitude_grad_viewspace_on_image"].cpu().numpy(), b["magnitude_grad_viewspace_on_image"])
    out["n_affected_equal"] = bool(np.array_equal(ex["num_affected_pixels"].cpu().numpy(), b["num_affected_pixels"]))
    print(wl, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in out.items()}, flush=True)
