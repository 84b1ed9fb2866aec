"""GPU (-m gpu): the Gaussian-parallel scheme end to end with the HIP kernels, two processes sharing cuda:0, collectives
over gloo (a one-GPU box has no second card and RCCL needs one per rank): rank r owns half of the Gaussians and renders
view r through libgsrast's staged entry points (distributed.HipStageBackend + gaussian_parallel_step).  Against the
fused single-process operator on the whole scene: each rank's image bit for bit; the owners' gradients, side by side,
equal to the sum over the two views of the fused operator's gradients bit for bit as well (same kernels, same order of
the two additions).  Also the view-parallel schedules with the real operator: per-step and per-view-overlapped
reduction give the same bits."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PTS, W_IMG, H_IMG = 6000, 256, 160


def _scene():
    from taichi_3d_gaussian_splatting_amd.synthetic import synth
    return synth(N_PTS, W_IMG, H_IMG, 0.06, sh_deg=3, seed=71)


def _input(s, lo, hi, dev, q, t, requires_grad=False):
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast
    return Rast.GaussianPointCloudRasterisationInput(
        point_cloud=torch.tensor(s.point_cloud[lo:hi], device=dev, requires_grad=requires_grad),
        point_cloud_features=torch.tensor(s.point_cloud_features[lo:hi], device=dev, requires_grad=requires_grad),
        point_object_id=torch.tensor(s.point_object_id[lo:hi], device=dev),
        point_invalid_mask=torch.tensor(s.point_invalid_mask[lo:hi], device=dev),
        camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=dev), s.height, s.width, 0),
        q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev), color_max_sh_band=3)


def _grad_of_image(img):
    return 2.0 * (img - 0.4)


def _worker(rank, world, port, out_dir, backend="gloo", own_device=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if own_device else 0), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Rast
    from taichi_3d_gaussian_splatting_amd import distributed as gsd
    from taichi_3d_gaussian_splatting_amd.synthetic import view_pose
    gsd.init_from_env(backend)
    dev = torch.device("cuda", rank if own_device else 0)
    torch.cuda.set_device(dev)
    s = _scene()
    poses = [tuple(torch.tensor(x, device=dev) for x in view_pose(v, world)) for v in range(world)]
    # ---- Gaussian-parallel: own shard, render own view ----
    bounds = gsd.shard_bounds(N_PTS, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    q0, t0 = view_pose(0, world)
    backend = gsd.HipStageBackend(_input(s, lo, hi, dev, q0, t0), poses)
    image, gpc, gft, stats = gsd.gaussian_parallel_step(backend, _grad_of_image)
    np.save(os.path.join(out_dir, f"gp_img_{rank}.npy"), image.cpu().numpy())
    np.save(os.path.join(out_dir, f"gp_pc_{rank}.npy"), gpc.cpu().numpy())
    np.save(os.path.join(out_dir, f"gp_ft_{rank}.npy"), gft.cpu().numpy())
    assert stats["collectives"] == 3 and stats["bytes_sent"] > 0
    # ---- view-parallel with the fused operator: two views per rank, both reduction schedules ----
    module = Rast(Rast.GaussianPointCloudRasterisationConfig())
    results = {}
    for schedule in ("step", "view"):
        q, t = view_pose(0, 2 * world)
        inp = _input(s, 0, N_PTS, dev, q, t, requires_grad=True)
        red = gsd.OverlappedGradientReducer() if schedule == "view" else None
        for v in gsd.views_of_rank(2 * world, rank, world):
            qv, tv = view_pose(v, 2 * world)
            inp.q_pointcloud_camera, inp.t_pointcloud_camera = torch.tensor(qv, device=dev), torch.tensor(tv, device=dev)
            img = module(inp)[0]
            img.backward(_grad_of_image(img.detach()))
            if red is not None:
                red.submit(inp.point_cloud.grad, inp.point_cloud_features.grad)
                inp.point_cloud.grad = None
                inp.point_cloud_features.grad = None
        if red is not None:
            gp, gf = red.finish()
        else:
            gsd.all_reduce_point_gradients(inp.point_cloud.grad, inp.point_cloud_features.grad)
            gp, gf = inp.point_cloud.grad, inp.point_cloud_features.grad
        results[schedule] = (gp.cpu().numpy().copy(), gf.cpu().numpy().copy())
    np.save(os.path.join(out_dir, f"vp_step_{rank}.npy"), results["step"][1])
    np.save(os.path.join(out_dir, f"vp_view_{rank}.npy"), results["view"][1])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_ranks_on_one_gpu_gaussian_parallel_and_view_parallel(tmp_path):
    _two_ranks(tmp_path, "gloo", False)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the same two schemes with one card per rank over RCCL (backend nccl)")
def test_two_ranks_on_two_gpus_over_rccl(tmp_path):
    """The code paths a one-GPU box cannot reach: dist.all_to_all_single with uneven splits, the asynchronous all-reduce on RCCL's
    own stream beside the compute stream (OverlappedGradientReducer), all_gather of the count table -- same bit-exact bars."""
    _two_ranks(tmp_path, "nccl", True)


def _two_ranks(tmp_path, backend, own_device):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), backend, own_device), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Rast
    from taichi_3d_gaussian_splatting_amd.distributed import shard_bounds
    from taichi_3d_gaussian_splatting_amd.synthetic import view_pose
    dev = torch.device("cuda", 0)
    s = _scene()
    module = Rast(Rast.GaussianPointCloudRasterisationConfig())
    # The fused operator in the order the scheme imposes: ONE set of parameters, both forwards first (each normalises the
    # quaternions in place, RAST:264-266, and normalising twice is not bitwise idempotent), then both backwards.
    q, t = view_pose(0, world)
    inp = _input(s, 0, N_PTS, dev, q, t, requires_grad=True)
    images = []
    for v in range(world):
        q, t = view_pose(v, world)
        inp.q_pointcloud_camera, inp.t_pointcloud_camera = torch.tensor(q, device=dev), torch.tensor(t, device=dev)
        images.append(module(inp)[0])
        assert np.array_equal(np.load(tmp_path / f"gp_img_{v}.npy").view(np.uint32), images[-1].detach().cpu().numpy().view(np.uint32)), v
    ref_pc = ref_ft = None
    for v in range(world):
        images[v].backward(_grad_of_image(images[v].detach()))
        gp, gf = inp.point_cloud.grad.cpu().numpy().copy(), inp.point_cloud_features.grad.cpu().numpy().copy()
        inp.point_cloud.grad = None
        inp.point_cloud_features.grad = None
        ref_pc, ref_ft = (gp, gf) if ref_pc is None else (ref_pc + gp, ref_ft + gf)       # view 0 + view 1, the owners' order
    bounds = shard_bounds(N_PTS, world)
    gp_pc = np.concatenate([np.load(tmp_path / f"gp_pc_{r}.npy") for r in range(world)])
    gp_ft = np.concatenate([np.load(tmp_path / f"gp_ft_{r}.npy") for r in range(world)])
    assert gp_pc.shape[0] == N_PTS and bounds[-1] == N_PTS
    assert np.array_equal(gp_pc.view(np.uint32), ref_pc.view(np.uint32))
    assert np.array_equal(gp_ft.view(np.uint32), ref_ft.view(np.uint32))
    # view-parallel: both ranks hold the same reduced gradient; the two schedules agree to summation order
    a0, a1 = np.load(tmp_path / "vp_step_0.npy"), np.load(tmp_path / "vp_step_1.npy")
    b0 = np.load(tmp_path / "vp_view_0.npy")
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))
    assert np.abs(a0 - b0).max() <= 1e-5 * np.abs(a0).max()
