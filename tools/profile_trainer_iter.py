"""Run N fused training iterations (for use under rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_trainer_step as B
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast
from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
from taichi_3d_gaussian_splatting_amd.optim import FusedAdam
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose
s = synth(**CONFIGS["cfg3_headline"]); q, t = view_pose(); DEV = B.DEV
H, W = s.height, s.width
gt = torch.rand(3, H, W, device=DEV)
pc = torch.tensor(s.point_cloud, device=DEV, requires_grad=True); feat = torch.tensor(s.point_cloud_features, device=DEV, requires_grad=True)
mask, obj = torch.tensor(s.point_invalid_mask, device=DEV), torch.tensor(s.point_object_id, device=DEV)
rast = Rast(Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda x: None)
inp = Rast.GaussianPointCloudRasterisationInput(point_cloud=pc, point_cloud_features=feat, point_object_id=obj, point_invalid_mask=mask,
    camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=DEV), H, W, 0), q_pointcloud_camera=torch.tensor(q, device=DEV),
    t_pointcloud_camera=torch.tensor(t, device=DEV), color_max_sh_band=3)
of, op = FusedAdam([feat], lr=1e-3), FusedAdam([pc], lr=1e-5)
lf = LossFunction(LossFunction.LossFunctionConfig())
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    of.zero_grad(); op.zero_grad()
    img, _, _ = rast(inp)
    img = torch.clamp(img, 0, 1).permute(2, 0, 1)
    L = lf(img, gt, point_invalid_mask=mask, pointcloud_features=feat)[0]
    L.backward()
    of.step(); op.step()
torch.cuda.synchronize()
