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
    P.assert_backward_parity(module, inp, g.cpu().numpy(), f, band, module.last_backward_extras, ocfg)
    return dict(seed=seed, W=W, H=H, n=n, sigma0=round(sigma0, 4), band=band, M=f.M, K=f.K)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    for i in range(n_cases):
        info = one(first + i)
        print(f"ok {info}  [{time.time() - t0:.0f} s]", flush=True)
    print(f"{n_cases} cases passed")


if __name__ == "__main__":
    main()
