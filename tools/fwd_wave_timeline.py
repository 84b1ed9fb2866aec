"""As bwd_wave_timeline.py, for the forward blend (stats build): one record per (tile, quadrant) wave of the last forward."""
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
inp = Rast.GaussianPointCloudRasterisationInput(
    point_cloud=torch.tensor(s.point_cloud, device=dev), point_cloud_features=torch.tensor(s.point_cloud_features, device=dev),
    point_object_id=torch.tensor(s.point_object_id, device=dev), point_invalid_mask=torch.tensor(s.point_invalid_mask, device=dev),
    camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
    q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3)
module = Rast(Rast.GaussianPointCloudRasterisationConfig())
with torch.no_grad():
    for _ in range(3):
        module(inp)
torch.cuda.synchronize()
T = ((s.height + 15) // 16) * ((s.width + 15) // 16)
n = min(4 * T, 65536)
buf = (C.c_ulonglong * (2 * n))()
_native.lib().gs_debug_wave_times_read(buf, n)
a = np.array(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0).astype(float), (a[:, 1] - t0).astype(float)
dur = en - st
span = en.max()
slots = 8 * 1024
print(f"{n} waves, wave duration / span: mean {dur.mean() / span:.3f} max {dur.max() / span:.3f} p99 {np.percentile(dur, 99) / span:.3f}")
print(f"sum of wave durations / (span * {slots} slots) = {dur.sum() / (span * slots):.3f}")
for frac in (0.25, 0.5, 0.75, 0.9):
    tt = span * frac
    print(f"  at {frac:4.2f} of the span: {int(((st <= tt) & (en > tt)).sum())} waves resident")
