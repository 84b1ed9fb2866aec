"""Host-side cost of one fwd+bwd step against the GPU time of the same step (diagnostic).

The forward hands M and K to the host once per frame, so the host can never run more than the rest of a step ahead
of the GPU; if the Python + launch work of a step takes longer than that, the GPU idles at the step boundary."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_headline"
    dev = torch.device("cuda", 0)
    s = synth(**CONFIGS[wl])
    q, t = view_pose()
    pc = torch.tensor(s.point_cloud, device=dev, requires_grad=True)
    feat = torch.tensor(s.point_cloud_features, device=dev, requires_grad=True)
    inp = Rast.GaussianPointCloudRasterisationInput(
        point_cloud=pc, point_cloud_features=feat, point_object_id=torch.tensor(s.point_object_id, device=dev),
        point_invalid_mask=torch.tensor(s.point_invalid_mask, device=dev),
        camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
        q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3)
    module = Rast(Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda p: None)
    marks = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0}

    def step():
        pc.grad = None
        feat.grad = None
        a = time.perf_counter()
        image, _, _ = module(inp)
        b = time.perf_counter()
        g = 2.0 * (image.detach() - 0.5)
        c = time.perf_counter()
        image.backward(g)
        d = time.perf_counter()
        marks["fwd"] += b - a; marks["loss"] += c - b; marks["bwd"] += d - c

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    for k in marks:
        marks[k] = 0.0
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host loop {1e6 * (t1 - t0) / n:.1f} us/step, with final sync {1e6 * (t2 - t0) / n:.1f} us/step")
    print("host time inside: " + ", ".join(f"{k} {1e6 * v / n:.1f} us" for k, v in marks.items()) +
          "  (fwd includes waiting for the GPU to publish M and K)")
    # the same with the GPU idle in between (host cost alone: nothing to wait for except the prologue kernels)
    for k in marks:
        marks[k] = 0.0
    for _ in range(50):
        torch.cuda.synchronize()
        step()
    torch.cuda.synchronize()
    print("host time, GPU drained before each step: " + ", ".join(f"{k} {1e6 * v / 50:.1f} us" for k, v in marks.items()))


if __name__ == "__main__":
    main()
