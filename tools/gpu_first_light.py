"""Diagnostic: run one BASELINE config through the operator on cuda:0, compare every product with the CPU oracle
and print per-array exactness plus rough fwd/bwd wall times.  usage: python tools/gpu_first_light.py [cfg1_plumbing|cfg3_headline|...]"""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as P
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_plumbing"
s = synth(**CONFIGS[name]); q, t = view_pose()
f, feat_after = P.run_oracle(s, q, t)
print("oracle M,K", f.M, f.K, flush=True)
module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
inp = P.make_input(s, q, t)
image, depth, count = module(inp)
torch.cuda.synchronize()
fr = module.last_frame
print("gpu M,K,bits", fr.n_points_in_camera, fr.n_keys, fr.sort_key_bits, flush=True)
for name_ in P.INT_EXPORTS + P.FLOAT_EXPORTS:
    got = fr.export(name_).cpu().numpy(); ref = getattr(f, name_)
    if got.shape != ref.shape: print(name_, "SHAPE", got.shape, ref.shape); continue
    eq = np.array_equal(got, ref)
    print(f"{name_:36s} exact={eq} maxdiff={np.abs(got.astype(np.float64)-ref).max() if ref.size else 0} nbad={(got!=ref).sum()}", flush=True)
for nm, got, ref in [("image", image, f.rasterized_image), ("depth", depth, f.rasterized_depth), ("count", count, f.pixel_valid_point_count),
                     ("last", None, None)]:
    if got is None: continue
    g = got.detach().cpu().numpy()
    print(f"{nm:36s} exact={np.array_equal(g, ref)} maxdiff={np.abs(g.astype(np.float64)-ref).max()} nbad={(g!=ref).sum()}", flush=True)
g_image = 2.0 * (image.detach() - 0.5)
image.backward(g_image)
torch.cuda.synchronize()
b = oracle.backward(f, g_image.cpu().numpy(), 3)
gp = inp.point_cloud.grad.cpu().numpy(); gf = inp.point_cloud_features.grad.cpu().numpy()
print("grad xyz rel", P.rel_err(gp, b["grad_pointcloud"]))
for lo, hi, nm in [(0, 4, "q"), (4, 7, "s"), (7, 8, "opacity"), (8, 56, "sh")]:
    print("grad", nm, P.rel_err(gf[:, lo:hi], b["grad_pointcloud_features"][:, lo:hi]))
# timing
for _ in range(3):
    inp2 = P.make_input(s, q, t)
    torch.cuda.synchronize(); t0 = time.time()
    im = module(inp2)[0]; torch.cuda.synchronize(); t1 = time.time()
    im.backward(g_image); torch.cuda.synchronize(); t2 = time.time()
    print(f"fwd {1e3*(t1-t0):.3f} ms  bwd {1e3*(t2-t1):.3f} ms", flush=True)
