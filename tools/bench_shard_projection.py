"""Owner side of the Gaussian-parallel scheme on ONE GPU: a shard of N/W Gaussians projected for W views, once with a host
wait per view (gs_project_shard) and once begun back to back and read afterwards (gs_project_shard_begin): milliseconds per
step of that stage alone, GPU otherwise idle.  `python tools/bench_shard_projection.py [W]` (GPU)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast  # noqa: E402
from taichi_3d_gaussian_splatting_amd.stages import StagedRasteriser  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    s = synth(**CONFIGS["cfg3_headline"])
    n = s.point_cloud.shape[0] // W
    inputs = []
    for v in range(W):
        q, t = view_pose(v, W)
        inputs.append(Rast.GaussianPointCloudRasterisationInput(
            point_cloud=torch.tensor(s.point_cloud[:n], device=dev), point_cloud_features=torch.tensor(s.point_cloud_features[:n], device=dev),
            point_object_id=torch.tensor(s.point_object_id[:n], device=dev), point_invalid_mask=torch.tensor(s.point_invalid_mask[:n], device=dev),
            camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
            q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3))
    st = StagedRasteriser()

    def waiting():
        return [st.project_shard(i)[0] for i in inputs]

    def begun():
        frames = [st.project_shard_begin(i) for i in inputs]
        return [st.project_shard_finish(f, want_ids=False)[0] for f in frames]

    out = {"workload": f"cfg3_headline shard of {n} Gaussians (1/{W}) projected for {W} views", "steps": 200}
    for name, fn in (("wait_per_view_ms", waiting), ("begun_back_to_back_ms", begun)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
