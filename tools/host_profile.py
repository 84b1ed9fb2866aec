import sys, time, os
sys.path.insert(0, '/root/repo')
import torch, cProfile, pstats
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast
from taichi_3d_gaussian_splatting_amd.synthetic import make_scene, view_pose
dev = torch.device("cuda", 0)
s = make_scene("cfg3_headline"); q, t = view_pose()
pc = torch.tensor(s.point_cloud, device=dev, requires_grad=True); feat = torch.tensor(s.point_cloud_features, device=dev, requires_grad=True)
inp = Rast.GaussianPointCloudRasterisationInput(point_cloud=pc, point_cloud_features=feat, point_object_id=torch.tensor(s.point_object_id, device=dev),
    point_invalid_mask=torch.tensor(s.point_invalid_mask, device=dev), camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
    q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3)
module = Rast(Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda p: None)
minus_one = torch.full((s.height, s.width, 3), -1.0, device=dev)
def step():
    pc.grad = None; feat.grad = None
    image, _, _ = module(inp)
    g = torch.add(minus_one, image.detach(), alpha=2.0)
    image.backward(g)
for _ in range(30): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
