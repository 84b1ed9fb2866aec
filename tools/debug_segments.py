import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_util as P
from oracle import oracle
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 60163
c = P.soak_case(seed)
s, q, t, partial, rng = c["scene"], c["q"], c["t"], c["partial"], c["rng"]
unit = dict(grad_color_factor=1.0, grad_high_order_color_factor=1.0, grad_s_factor=1.0, grad_q_factor=1.0, grad_alpha_factor=1.0)
ocfg = oracle.default_config(allow_partial_tiles=int(partial), **unit)
f, feat_after = P.run_oracle(s, q, t, ocfg)
cfg = P.Rast.GaussianPointCloudRasterisationConfig(); cfg.allow_partial_tiles = partial
for k, v in unit.items(): setattr(cfg, k, v)
got = {}
module = P.Rast(cfg, backward_valid_point_hook=lambda x: got.setdefault("hook", x))
inp = P.make_input(s, q, t, 3)
image = module(inp)[0]
target = torch.tensor(rng.uniform(0, 1, image.shape).astype(np.float32), device=image.device)
g = 2.0 * (image.detach() - target)
image.backward(g)
fr = module.last_frame
lens = f.tile_points_end - f.tile_points_start
print("W,H", c["W"], c["H"], "T", lens.size, "lens max", lens.max(), "heavy", fr.heavy_tiles(), "items", fr.heavy_tiles(items=True), "segments env", os.environ.get("GS_BWD_SEGMENTS"))
b = oracle.backward(f, g.cpu().numpy(), 3, ocfg, want_summed=True)
gp, gf = inp.point_cloud.grad.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy()
m = P.backward_margins(gp, gf, b)
print({k: (round(v["bar_use_max"], 2), "%.1e" % P.rel_err(gp if k == "xyz" else gf[:, dict(q=(0,4),s=(4,7),opacity=(7,8),sh=(8,56))[k][0]:dict(q=(0,4),s=(4,7),opacity=(7,8),sh=(8,56))[k][1]], b["grad_pointcloud"] if k == "xyz" else b["grad_pointcloud_features"][:, dict(q=(0,4),s=(4,7),opacity=(7,8),sh=(8,56))[k][0]:dict(q=(0,4),s=(4,7),opacity=(7,8),sh=(8,56))[k][1]])) for k, v in m.items()})
mi = module.last_backward_extras["magnitude_grad_viewspace_on_image"].cpu().numpy()
print("mag image rel err", P.rel_err(mi, b["magnitude_grad_viewspace_on_image"]))
np.save(os.path.join(ROOT, "gpurun_out", f"dbg_gp_{os.environ.get('GS_BWD_SEGMENTS','1')}.npy"), gp)
# the worst element against the per-element bar: which point, which tiles hold it and how long their lists are
mx = m["xyz"]
err = np.abs(gp - b["grad_pointcloud"])
summed = b.get("summed_pointcloud")
if summed is not None:
    use = err / np.maximum(summed, 1e-30)
    flat = np.argsort(use.ravel())[::-1][:6]
    ids = np.asarray(f.point_id_in_camera_list)
    cam_of = {int(p): i for i, p in enumerate(ids)}
    box = None
    for fl in flat:
        p, comp = divmod(int(fl), 3)
        print("point", p, "component", comp, "got %.6e ref %.6e err %.2e summed %.3e err/summed %.2e" % (gp[p, comp], b["grad_pointcloud"][p, comp], err[p, comp], summed[p, comp], use[p, comp]),
              "tiles covered", int(np.asarray(f.num_overlap_tiles)[cam_of[p]]) if p in cam_of else None)
    # tiles holding the worst point: from the sorted values
    p = int(flat[0]) // 3
    m_idx = cam_of.get(p)
    vals = np.asarray(f.point_offset_with_sort_key)
    if vals is not None and m_idx is not None:
        pos = np.nonzero(vals == m_idx)[0]
        ts, te = np.asarray(f.tile_points_start), np.asarray(f.tile_points_end)
        for ps in pos[:12]:
            tl = int(np.nonzero((ts <= ps) & (te > ps))[0][0])
            print("   in tile", tl, "list", int(ts[tl]), int(te[tl]), "len", int(te[tl] - ts[tl]), "position in list", int(ps - ts[tl]))
np.save(os.path.join(ROOT, "gpurun_out", f"dbg_mag_{os.environ.get('GS_BWD_SEGMENTS','1')}.npy"), mi)
np.save(os.path.join(ROOT, "gpurun_out", "dbg_mag_ref.npy"), b["magnitude_grad_viewspace_on_image"])
