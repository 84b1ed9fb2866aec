"""Start/end time of every k_blend_bwd_tile wave of one backward (stats build): wave durations and how many waves are
resident over the launch, i.e. how much of the launch is tail.  `make -C taichi_3d_gaussian_splatting_amd/csrc times` first (the timing-only diagnostic build: the counting build's
atomics make every wave tens of times slower)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GSRAST_LIB", os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd", "lib", "libgsrast_times.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast, _native  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_scene, view_pose  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_headline"
dev = torch.device("cuda", 0)
s = make_scene(wl); q, t = view_pose()
pc = torch.tensor(s.point_cloud, device=dev, requires_grad=True); feat = torch.tensor(s.point_cloud_features, device=dev, requires_grad=True)
inp = Rast.GaussianPointCloudRasterisationInput(
    point_cloud=pc, point_cloud_features=feat, point_object_id=torch.tensor(s.point_object_id, device=dev),
    point_invalid_mask=torch.tensor(s.point_invalid_mask, device=dev),
    camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
    q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3)
module = Rast(Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda p: None)
for _ in range(3):
    pc.grad = None; feat.grad = None
    image, _, _ = module(inp)
    image.backward(2.0 * (image.detach() - 0.5))
torch.cuda.synchronize()
T = ((s.height + 15) // 16) * ((s.width + 15) // 16)
G = int(os.environ.get("GS_BWD_WAVES_PER_TILE", 0)) or (4 if T < 2000 else 2 if T < 6144 else 1)     # gs_api.hip: waves_per_tile
n_heavy = module.last_frame.heavy_tiles()
n_items = module.last_frame.heavy_tiles(items=True)
# one record per wave, indexed by workgroup * 4 + wave: the heavy tiles' work items first (one workgroup each: a 512-entry segment of
# a cut list, or a whole list), then the ordinary work items four to a workgroup (k_backward.hip: k_blend_bwd_tile)
nrec = min(4 * (n_items + ((T - n_heavy) * G + 3) // 4), 65536)
buf = (C.c_ulonglong * (2 * nrec))()
_native.lib().gs_debug_wave_times_read(buf, nrec)
a = np.array(buf, dtype=np.uint64).reshape(nrec, 2).astype(np.int64)
idx = np.arange(a.shape[0])
keep = a[:, 1] > a[:, 0]
a, idx = a[keep], idx[keep]
keep = a[:, 1] > np.percentile(a[:, 0], 10)
a, idx = a[keep], idx[keep]        # (a wave that left at once wrote nothing: its slot still holds a stamp of the forward blend, which ended before this launch began)
T = a.shape[0]
print(f"{wl}: {n_heavy} heavy tiles handed out as {n_items} work items (4 cooperating waves each), {G} wave(s) per ordinary tile")
heavy_waves = 4 * n_items
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0).astype(float), (a[:, 1] - t0).astype(float)      # wall-clock ticks (only ratios are used)
dur = en - st
span = en.max()
print(f"{wl}: {G} wave(s) per tile; longest wave / launch span = {dur.max() / span:.3f}")
print(f"{T} waves, launch span {span:.0f} ticks of the 100 MHz wall clock = {span / 100:.1f} us; wave duration mean {dur.mean() / 100:.1f} us, max {dur.max() / 100:.1f}, p99 {np.percentile(dur, 99) / 100:.1f}, min {dur.min() / 100:.1f}")
print(f"sum of wave durations / (span * 5120 slots) = {dur.sum() / (span * 5120):.3f}")
for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
    tt = span * frac - 1e-6
    print(f"  at {frac:4.2f} of the span: {int(((st <= tt) & (en > tt)).sum())} waves resident")
print("first waves (heaviest tiles): durations", np.round(dur[:8], 1), " last:", np.round(dur[-8:], 1))
late = np.argsort(en)[-5:]
print("last five waves to finish: block ids", late, "start", np.round(st[late], 1), "dur", np.round(dur[late], 1))
print("by decile of dispatch order (heaviest first): mean start, mean duration, max end (fractions of the span)")
for d in range(10):
    sl = slice(d * T // 10, (d + 1) * T // 10)
    print(f"  {d}: start {st[sl].mean() / span:5.2f}  dur {dur[sl].mean() / span:5.2f}  max end {en[sl].max() / span:5.2f}")
longest = np.argsort(dur)[-8:]
print("longest waves: block ids", longest, "durations/span", np.round(dur[longest] / span, 2), "start/span", np.round(st[longest] / span, 2))

hv = idx < heavy_waves
if hv.any() and (~hv).any():
    print(f"heavy items: {hv.sum()} waves, duration mean {dur[hv].mean() / 100:.1f} us max {dur[hv].max() / 100:.1f}, last end {en[hv].max() / span:.2f} of the span; "
          f"ordinary: {(~hv).sum()} waves, mean {dur[~hv].mean() / 100:.1f} us max {dur[~hv].max() / 100:.1f}, last end {en[~hv].max() / span:.2f}")

if os.environ.get("GS_TIMELINE_DUMP"):      # raw records for a closer look: wave id, start, end (ticks from the first start), plus the tile lists' lengths
    lens = (module.last_frame.export("tile_points_end").cpu().numpy().astype(np.int64)
            - module.last_frame.export("tile_points_start").cpu().numpy().astype(np.int64))
    np.savez(os.environ["GS_TIMELINE_DUMP"], idx=idx, st=st, en=en, n_heavy=n_heavy, n_items=n_items, G=G, tile_len=lens)
