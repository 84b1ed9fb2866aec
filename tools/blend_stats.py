"""Event counters of the two blend kernels on one workload (diagnostic; needs `make -C taichi_3d_gaussian_splatting_amd/csrc stats`).

    GSRAST_LIB=taichi_3d_gaussian_splatting_amd/lib/libgsrast_stats.so python tools/blend_stats.py [workload]

Prints how many 64-entry batches, culled-in entries, evaluated (splat, 8x8 quadrant) pairs and contributing lanes
each kernel went through, next to the per-pixel evaluation count E of the reference algorithm.
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GSRAST_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                 "taichi_3d_gaussian_splatting_amd", "lib", "libgsrast_stats.so"))
import torch  # noqa: E402

from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast, _native  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_scene, view_pose  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_headline"
    dev = torch.device("cuda", 0)
    s = make_scene(wl)
    q, t = view_pose()
    pc = torch.tensor(s.point_cloud, device=dev, requires_grad=True)
    feat = torch.tensor(s.point_cloud_features, device=dev, requires_grad=True)
    inp = Rast.GaussianPointCloudRasterisationInput(
        point_cloud=pc, point_cloud_features=feat, point_object_id=torch.tensor(s.point_object_id, device=dev),
        point_invalid_mask=torch.tensor(s.point_invalid_mask, device=dev),
        camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
        q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3)
    module = Rast(Rast.GaussianPointCloudRasterisationConfig())
    L = _native.lib()
    buf = (C.c_ulonglong * 32)()
    image, _, _ = module(inp)
    L.gs_debug_stats_read(buf, 1)                 # discard anything earlier
    image, _, _ = module(inp)
    image.backward(2.0 * (image.detach() - 0.5))
    torch.cuda.synchronize()
    L.gs_debug_stats_read(buf, 1)
    c = list(buf)
    fr_last = module.last_forward_outputs["pixel_offset_of_last_effective_point"].to(torch.int64)
    with torch.no_grad():
        module(inp)
    fr = module.last_frame
    H, W = s.height, s.width
    tx = (W + 15) // 16
    tile_of_pixel = (torch.arange(H, device=dev) // 16)[:, None] * tx + (torch.arange(W, device=dev) // 16)[None, :]
    E = int((fr_last - fr.export("tile_points_start").to(torch.int64)[tile_of_pixel]).clamp_(min=0).sum().item())
    out = {
        "workload": wl, "sort_pairs": fr.n_keys, "tiles": fr.n_tiles, "pixel_entry_evaluations_E": E,
        "fwd": {"batches_x_waves": c[0], "entries_kept_by_cull": c[1], "quadrant_evals": c[2], "quadrant_evals_rejected_by_exponent": c[4],
                "lanes_contributing": c[3], "lanes_alive": c[5]},
        "bwd": {"batches": c[8], "splat_iterations": c[9], "quadrant_evals_entered": c[10], "quadrant_evals_past_exponent": c[11],
                "quadrant_evals_with_a_contribution": c[15], "lanes_contributing": c[12], "reductions": c[13], "exact_exp_fallbacks": c[14]},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
