"""CPU, world_size 2 over gloo: the view-parallel path.  Each rank takes its views,
computes that view's point gradients (with the CPU oracle standing in for the GPU operator -- the
collective logic is what is under test), and the all-reduced gradient must equal the sum over
all views computed in one process."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_VIEWS = 4


def _view_grads(view, n_views):
    from oracle import oracle
    from taichi_3d_gaussian_splatting_amd.synthetic import synth, view_pose
    s = synth(1500, 96, 64, 0.1, sh_deg=3, seed=0)
    q, t = view_pose(view, n_views)
    f, _ = oracle.forward(s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id, q, t,
                          s.camera_intrinsics, s.height, s.width)
    b = oracle.backward(f, 2.0 * (f.rasterized_image - 0.5), 3)
    return b["grad_pointcloud"], b["grad_pointcloud_features"]


def _worker(rank, world, port, flat_layout, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from taichi_3d_gaussian_splatting_amd import distributed as gsd
    r, w, _ = gsd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    views = gsd.views_of_rank(N_VIEWS, rank, world)
    n = 1500
    if flat_layout:                       # the operator's layout: two views of one 59*N buffer
        flat = torch.zeros(59 * n)
        gft, gpc = flat[:56 * n].view(n, 56), flat[56 * n:].view(n, 3)
    else:
        gpc, gft = torch.zeros(n, 3), torch.zeros(n, 56)
    for v in views:
        a, b = _view_grads(v, N_VIEWS)
        gpc += torch.from_numpy(a)
        gft += torch.from_numpy(b)
    ncoll = gsd.all_reduce_point_gradients(gpc, gft)
    assert ncoll == (1 if flat_layout else 2)
    np.save(os.path.join(out_dir, f"gpc_{rank}.npy"), gpc.numpy())
    np.save(os.path.join(out_dir, f"gft_{rank}.npy"), gft.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("flat_layout", [True, False])
def test_all_reduced_gradient_equals_sum_over_views(tmp_path, flat_layout):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), flat_layout, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    ref_pc, ref_ft = np.zeros((1500, 3), np.float64), np.zeros((1500, 56), np.float64)
    for v in range(N_VIEWS):
        a, b = _view_grads(v, N_VIEWS)
        ref_pc += a
        ref_ft += b
    for rank in range(world):
        gpc = np.load(tmp_path / f"gpc_{rank}.npy")
        gft = np.load(tmp_path / f"gft_{rank}.npy")
        # within 1e-4 relative (float summation order differs, SURVEY 8e "parity check")
        assert np.abs(gpc - ref_pc).max() <= 1e-4 * np.abs(ref_pc).max()
        assert np.abs(gft - ref_ft).max() <= 1e-4 * np.abs(ref_ft).max()
    assert np.array_equal(np.load(tmp_path / "gpc_0.npy"), np.load(tmp_path / "gpc_1.npy"))   # ranks agree bit for bit


def test_round_robin_view_assignment():
    from taichi_3d_gaussian_splatting_amd.distributed import views_of_rank
    assert [views_of_rank(8, r, 8) for r in range(8)] == [[r] for r in range(8)]
    assert views_of_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((views_of_rank(13, r, 4) for r in range(4)), [])) == list(range(13))
