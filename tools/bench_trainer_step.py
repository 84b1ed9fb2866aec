#!/usr/bin/env python3
"""Measure the pieces either side of the operator at BASELINE config 3 on cuda:0:
L1+SSIM loss forward+backward and the two Adam updates, libgsrast's fused kernels against the torch ops the
reference would run (conv2d-based SSIM restated from pytorch_msssim + autograd; torch.optim.Adam), and a whole
training iteration (GaussianPointTrainer.py:145-184 without data loading / logging).  Prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast  # noqa: E402
from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction  # noqa: E402
from taichi_3d_gaussian_splatting_amd.optim import FusedAdam  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose  # noqa: E402

DEV = torch.device("cuda:0")


def torch_loss(pred, gt, lam=0.2):
    coords = torch.arange(11, dtype=pred.dtype, device=pred.device) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum()).reshape(1, 1, 1, 11).repeat(3, 1, 1, 1)
    gf = lambda t: F.conv2d(F.conv2d(t, g.transpose(2, 3), groups=3), g, groups=3)
    X, Y = pred[None], gt[None]
    mu1, mu2 = gf(X), gf(Y)
    s1, s2, s12 = gf(X * X) - mu1 ** 2, gf(Y * Y) - mu2 ** 2, gf(X * Y) - mu1 * mu2
    ssim_map = ((2 * mu1 * mu2 + 1e-4) / (mu1 ** 2 + mu2 ** 2 + 1e-4)) * ((2 * s12 + 9e-4) / (s1 + s2 + 9e-4))
    return (1 - lam) * (pred - gt).abs().mean() + lam * (1 - ssim_map.mean())


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    cfg = CONFIGS["cfg3_headline"]
    s = synth(**cfg)
    H, W = s.height, s.width
    q, t = view_pose()
    gt = torch.rand(3, H, W, device=DEV)
    pred0 = torch.rand(3, H, W, device=DEV)
    lf = LossFunction(LossFunction.LossFunctionConfig(enable_regularization=False))

    def loss_fused():
        p = pred0.detach().requires_grad_(True)
        lf(p, gt)[0].backward()

    def loss_torch():
        p = pred0.detach().requires_grad_(True)
        torch_loss(p, gt).backward()

    N = cfg["N"]
    feat_a, pc_a = torch.randn(N, 56, device=DEV, requires_grad=True), torch.randn(N, 3, device=DEV, requires_grad=True)
    feat_a.grad, pc_a.grad = torch.randn_like(feat_a), torch.randn_like(pc_a)
    fa, fp = FusedAdam([feat_a], lr=1e-3), FusedAdam([pc_a], lr=1e-5)
    ta, tp = torch.optim.Adam([feat_a], lr=1e-3), torch.optim.Adam([pc_a], lr=1e-5)

    def make_iteration(fused, in_place=False):
        pc = torch.tensor(s.point_cloud, device=DEV, requires_grad=True)
        feat = torch.tensor(s.point_cloud_features, device=DEV, requires_grad=True)
        mask, obj = torch.tensor(s.point_invalid_mask, device=DEV), torch.tensor(s.point_object_id, device=DEV)
        rast = Rast(Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda x: None)
        inp = Rast.GaussianPointCloudRasterisationInput(
            point_cloud=pc, point_cloud_features=feat, point_object_id=obj, point_invalid_mask=mask,
            camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=DEV), H, W, 0),
            q_pointcloud_camera=torch.tensor(q, device=DEV), t_pointcloud_camera=torch.tensor(t, device=DEV), color_max_sh_band=3)
        if fused:
            of, op = FusedAdam([feat], lr=1e-3), FusedAdam([pc], lr=1e-5)
            loss_fn = LossFunction(LossFunction.LossFunctionConfig())
        else:
            of, op = torch.optim.Adam([feat], lr=1e-3), torch.optim.Adam([pc], lr=1e-5)
            loss_fn = None

        def it():
            of.zero_grad(); op.zero_grad()
            img, _, _ = rast(inp)
            if in_place:        # the (H,W,3) image read where it lies, clamped inside the loss kernels
                L = loss_fn(img.permute(2, 0, 1), gt, point_invalid_mask=mask, pointcloud_features=feat, clamp_predicted=True)[0]
                L.backward()
                of.step(); op.step()
                return
            img = torch.clamp(img, 0, 1).permute(2, 0, 1)               # GaussianPointTrainer.py:173-176 as written
            if fused:
                L = loss_fn(img, gt, point_invalid_mask=mask, pointcloud_features=feat)[0]
            else:
                L = torch_loss(img, gt) + 2 * torch.norm(torch.exp(feat[mask == 0, 4:7]), dim=1).mean()
            L.backward()
            of.step(); op.step()
        return it

    for flag, hwc in (("--only-loss-chw", False), ("--only-loss-hwc", True)):      # for rocprofv3: the loss kernels alone, one layout
        if flag in sys.argv:
            raw = torch.rand(H, W, 3, device=DEV) * 1.4 - 0.2

            def loss_only():
                if hwc:
                    p = raw.detach().requires_grad_(True)
                    lf(p.permute(2, 0, 1), gt, clamp_predicted=True)[0].backward()
                else:
                    loss_fused()
            print(json.dumps({flag: round(timeit(loss_only, n=100, warm=20), 4)}))
            return
    if "--only-fused-iteration" in sys.argv:          # for rocprofv3 --kernel-trace --stats: nothing but the fused iteration's launches
        print(json.dumps({"training_iteration_ms": {"fused_loss_and_adam_image_in_place": round(timeit(make_iteration(True, True), n=100, warm=20), 4)}}))
        return
    out = {
        "component": "trainer step around the rasteriser (SURVEY 8f-1), config 3, 1x MI355X",
        "loss_fwd_bwd_ms": {"fused_gs_loss_l1_ssim": round(timeit(loss_fused), 4), "torch_conv2d_autograd": round(timeit(loss_torch), 4)},
        "adam_two_tensors_ms": {"fused_gs_adam_step": round(timeit(lambda: (fa.step(), fp.step())), 4),
                                "torch_optim_adam": round(timeit(lambda: (ta.step(), tp.step())), 4)},
        "training_iteration_ms": {"fused_loss_and_adam_image_in_place": round(timeit(make_iteration(True, True), n=100, warm=20), 4),
                                  "fused_loss_and_adam": round(timeit(make_iteration(True), n=100, warm=20), 4),
                                  "torch_loss_and_adam": round(timeit(make_iteration(False), n=100, warm=20), 4)},
    }
    out["training_iterations_per_s"] = {k: round(1e3 / v, 1) for k, v in out["training_iteration_ms"].items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
