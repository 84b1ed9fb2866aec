"""The two forms of the blend backward side by side (GPU): default (fast: d p / d Sigma' as v v^T with fused multiply-adds) and
gs_config.bwd_reference_order = 1 (the reference's UTIL:331-348 operation order), at BASELINE configs 2 and 3 and the two
clustered workloads.  Per form: tensor-level error of every gradient group against the CPU oracle, the per-element margins of
tests/parity_util.py, the launch time of k_blend_bwd_tile; and the distribution of the difference between the two forms.

    python tools/strict_vs_fast.py [out.json]        (the judged copy: profiles/r03_strict_vs_fast.json)
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import parity_util as P  # noqa: E402
from oracle import oracle  # noqa: E402
from taichi_3d_gaussian_splatting_amd import _native  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_scene, view_pose  # noqa: E402


def kernel_ms(module, dev, fn, reps=20):
    L = _native.lib()
    names = L.gs_kernel_names().decode().split(",")
    kid = names.index("k_blend_bwd_tile")
    ctx = module._ctx_for(dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _native.check(L.gs_profile_enable(ctx, C.c_uint64(1 << kid)), "gs_profile_enable")
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (C.c_double * len(names))()
    cnt = (C.c_int64 * len(names))()
    _native.check(L.gs_profile_read(ctx, ms, cnt, len(names), 1), "gs_profile_read")
    _native.check(L.gs_profile_enable(ctx, C.c_uint64(0)), "gs_profile_enable")
    return ms[kid] / max(cnt[kid], 1)


def run(name):
    scene = make_scene(name)
    q, t = view_pose()
    f, feat_after = P.run_oracle(scene, q, t)
    res = {"workload": name, "N": int(f.N), "M": int(f.M), "K": int(f.K), "forms": {}}
    grads = {}
    g_np = None
    for form in ("fast", "reference_order"):
        cfg = P.Rast.GaussianPointCloudRasterisationConfig()
        cfg.backward_reference_order = form == "reference_order"
        module = P.Rast(cfg)
        inp = P.make_input(scene, q, t, 3)
        image = module(inp)[0]
        rng = np.random.default_rng(0)
        target = torch.tensor(rng.uniform(0, 1, image.shape).astype(np.float32), device=image.device)
        g = 2.0 * (image.detach() - target)
        image.backward(g, retain_graph=True)
        gp, gf = inp.point_cloud.grad.cpu().numpy().copy(), inp.point_cloud_features.grad.cpu().numpy().copy()
        if g_np is None:
            g_np = g.cpu().numpy()
            b = oracle.backward(f, g_np, 3, None, want_summed=True)
        m = P.backward_margins(gp, gf, b)
        row = {"xyz": {"tensor_level": P.rel_err(gp, b["grad_pointcloud"]), "bar_use_max": m["xyz"]["bar_use_max"], "of_summed_max": m["xyz"]["of_summed_max"]}}
        for lo, hi, gname in P.GROUPS:
            row[gname] = {"tensor_level": P.rel_err(gf[:, lo:hi], b["grad_pointcloud_features"][:, lo:hi]),
                          "bar_use_max": m[gname]["bar_use_max"], "of_summed_max": m[gname]["of_summed_max"]}

        def again():
            inp.point_cloud.grad = None
            inp.point_cloud_features.grad = None
            image.backward(g, retain_graph=True)
        row["k_blend_bwd_tile_ms"] = round(kernel_ms(module, image.device, again), 4)
        res["forms"][form] = row
        grads[form] = (gp, gf)
    d = {}
    for lo, hi, gname in [(None, None, "xyz")] + P.GROUPS:
        a = grads["fast"][0] if gname == "xyz" else grads["fast"][1][:, lo:hi]
        s = grads["reference_order"][0] if gname == "xyz" else grads["reference_order"][1][:, lo:hi]
        scale = float(np.abs(s).max()) or 1.0
        e = np.abs(a - s).ravel() / scale
        d[gname] = {"max": float(e.max()), "p999": float(np.quantile(e, 0.999)), "mean": float(e.mean()),
                    "elements_that_differ": int((a != s).sum()), "elements": int(e.size)}
    res["fast_minus_reference_order_over_tensor_max"] = d
    f.free()
    return res


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "strict_vs_fast.json")
    res = {"what": __doc__.split("\n\n")[0], "cases": []}
    for name in ("cfg2_truck7k", "cfg3_headline", "cfg2_clustered", "cfg3_clustered"):
        res["cases"].append(run(name))
        print(json.dumps(res["cases"][-1]), flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
