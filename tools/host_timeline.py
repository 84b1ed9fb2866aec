"""Host-side cost of one fwd+bwd step against the GPU time of the same step (diagnostic).

The forward hands M and K to the host once per frame.  With predicted sizing (default) the host reads them AFTER it has queued
the whole forward, so the wait costs the GPU nothing; with GS_PREDICT_SIZES=0 it waits in the middle of the frame with the GPU
idle behind it.  Prints, per step: host time inside forward / loss / backward, the time spent waiting for the counters
(gs_ctx_counter_wait_ns) and how each frame was sized.  Run it once with and once without GS_PREDICT_SIZES=0."""
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

    from taichi_3d_gaussian_splatting_amd import _native
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    for k in marks:
        marks[k] = 0.0
    n = 200
    L = _native.lib()
    ctx = module._ctx_for(dev)
    w0 = L.gs_ctx_counter_wait_ns(ctx)
    sizing = {}
    t0 = time.perf_counter()
    for _ in range(n):
        step()
        sizing[module.last_frame.sizing] = sizing.get(module.last_frame.sizing, 0) + 1
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wait_us = (L.gs_ctx_counter_wait_ns(ctx) - w0) / 1e3 / n
    print(f"GS_PREDICT_SIZES={os.environ.get('GS_PREDICT_SIZES', '1')}: frames sized {sizing}")
    print(f"host loop {1e6 * (t1 - t0) / n:.1f} us/step, with final sync {1e6 * (t2 - t0) / n:.1f} us/step; "
          f"waiting for the frame counters {wait_us:.1f} us/step "
          f"({'after the last launch of the forward: the GPU is busy meanwhile' if os.environ.get('GS_PREDICT_SIZES', '1')[:1] != '0' else 'in the middle of the forward: the GPU has nothing queued behind the per-point kernels'})")
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
