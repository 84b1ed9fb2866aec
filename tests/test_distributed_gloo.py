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


# ---- Gaussian-parallel data path and the overlapped view-parallel reducer --------------------
N_PTS = 1500


def _scene():
    from taichi_3d_gaussian_splatting_amd.synthetic import synth
    return synth(N_PTS, 96, 64, 0.1, sh_deg=3, seed=0)


class _OracleStageBackend:
    """The oracle's staged halves (oracle.forward on the shard -> records, forward_from_projected, backward_sums,
    backward_points) standing in for libgsrast's staged entry points on the CPU box: what is under test is the
    exchange -- counts, splits, ordering, which rows go back to whom."""

    def __init__(self, lo, hi, n_views):
        from oracle import oracle
        from taichi_3d_gaussian_splatting_amd.synthetic import view_pose
        self.o, self.s, self.lo, self.hi, self.n_views, self.view_pose = oracle, _scene(), lo, hi, n_views, view_pose

    def project(self, view):
        s, o = self.s, self.o
        q, t = self.view_pose(view, self.n_views)
        f, _ = o.forward(s.point_cloud[self.lo:self.hi], s.point_cloud_features[self.lo:self.hi], s.point_invalid_mask[self.lo:self.hi],
                         s.point_object_id[self.lo:self.hi], q, t, s.camera_intrinsics, s.height, s.width)
        return torch.from_numpy(o.pack_records(f)), f

    def render(self, records):
        f = self.o.forward_from_projected(records.numpy(), self.s.height, self.s.width)
        return torch.from_numpy(f.rasterized_image), f

    def backward_render(self, f, grad_image):
        sums, _ = self.o.backward_sums(f, grad_image.numpy())
        return torch.from_numpy(sums.copy())

    def backward_project(self, f, sums):
        b = self.o.backward_points(f, sums.numpy(), 3)
        return torch.from_numpy(b["grad_pointcloud"]), torch.from_numpy(b["grad_pointcloud_features"])


def _gp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from taichi_3d_gaussian_splatting_amd import distributed as gsd
    gsd.init_from_env("gloo")
    bounds = gsd.shard_bounds(N_PTS, world)
    backend = _OracleStageBackend(bounds[rank], bounds[rank + 1], world)
    image, gpc, gft, stats = gsd.gaussian_parallel_step(backend, lambda img: 2.0 * (img - 0.5))
    assert stats["collectives"] == 3 and stats["bytes_sent"] > 0
    np.save(os.path.join(out_dir, f"gp_img_{rank}.npy"), image.numpy())
    np.save(os.path.join(out_dir, f"gp_pc_{rank}.npy"), gpc.numpy())
    np.save(os.path.join(out_dir, f"gp_ft_{rank}.npy"), gft.numpy())
    # the view-parallel scheme on the same step, both reduction schedules, for comparison in the parent
    a, b = _view_grads(rank, world)
    flat = torch.zeros(59 * N_PTS)
    gft2, gpc2 = flat[:56 * N_PTS].view(N_PTS, 56), flat[56 * N_PTS:].view(N_PTS, 3)
    gpc2 += torch.from_numpy(a); gft2 += torch.from_numpy(b)
    gsd.all_reduce_point_gradients(gpc2, gft2)
    np.save(os.path.join(out_dir, f"vp_pc_{rank}.npy"), gpc2.numpy())
    np.save(os.path.join(out_dir, f"vp_ft_{rank}.npy"), gft2.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gaussian_parallel_exchange_equals_view_parallel_and_the_sum_over_views(tmp_path):
    """world_size 2: rank r owns half of the Gaussians and renders view r.  Records out, sums back (two all-to-alls);
    the owner-side gradients, put side by side, must equal (a) the all-reduced view-parallel gradient and (b) the sum
    over views computed in one process, within 1e-4 -- and each rank's image must be the single-process image of its
    view bit for bit (the concatenation of shards in rank order preserves the sort's tie order)."""
    world = 2
    mp.spawn(_gp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from taichi_3d_gaussian_splatting_amd.distributed import shard_bounds
    from taichi_3d_gaussian_splatting_amd.synthetic import view_pose
    s = _scene()
    ref_pc, ref_ft = np.zeros((N_PTS, 3), np.float64), np.zeros((N_PTS, 56), np.float64)
    for v in range(world):
        q, t = view_pose(v, world)
        f, _ = oracle.forward(s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id, q, t,
                              s.camera_intrinsics, s.height, s.width)
        b = oracle.backward(f, 2.0 * (f.rasterized_image - 0.5), 3)
        ref_pc += b["grad_pointcloud"]; ref_ft += b["grad_pointcloud_features"]
        assert np.array_equal(np.load(tmp_path / f"gp_img_{v}.npy"), f.rasterized_image), v
    bounds = shard_bounds(N_PTS, world)
    gp_pc = np.concatenate([np.load(tmp_path / f"gp_pc_{r}.npy") for r in range(world)])
    gp_ft = np.concatenate([np.load(tmp_path / f"gp_ft_{r}.npy") for r in range(world)])
    assert gp_pc.shape == (N_PTS, 3) and [np.load(tmp_path / f"gp_pc_{r}.npy").shape[0] for r in range(world)] == [bounds[1], N_PTS - bounds[1]]
    vp_pc, vp_ft = np.load(tmp_path / "vp_pc_0.npy"), np.load(tmp_path / "vp_ft_0.npy")
    for got_pc, got_ft in [(gp_pc, gp_ft), (vp_pc, vp_ft)]:
        assert np.abs(got_pc - ref_pc).max() <= 1e-4 * np.abs(ref_pc).max()
        assert np.abs(got_ft - ref_ft).max() <= 1e-4 * np.abs(ref_ft).max()
    assert np.abs(gp_pc - vp_pc).max() <= 1e-4 * np.abs(vp_pc).max()
    assert np.abs(gp_ft - vp_ft).max() <= 1e-4 * np.abs(vp_ft).max()


def _overlap_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from taichi_3d_gaussian_splatting_amd import distributed as gsd
    gsd.init_from_env("gloo")
    red = gsd.OverlappedGradientReducer()
    for v in gsd.views_of_rank(N_VIEWS, rank, world):                     # two views per rank
        a, b = _view_grads(v, N_VIEWS)
        flat = torch.zeros(59 * N_PTS)                                    # a fresh buffer per view, as the operator allocates
        gft, gpc = flat[:56 * N_PTS].view(N_PTS, 56), flat[56 * N_PTS:].view(N_PTS, 3)
        gpc += torch.from_numpy(a); gft += torch.from_numpy(b)
        red.submit(gpc, gft)                                              # returns at once; the next view is computed meanwhile
    gpc, gft = red.finish()
    assert red.collectives == 2
    np.save(os.path.join(out_dir, f"ov_pc_{rank}.npy"), gpc.numpy())
    np.save(os.path.join(out_dir, f"ov_ft_{rank}.npy"), gft.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_per_view_reduction_equals_sum_over_views(tmp_path):
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref_pc, ref_ft = np.zeros((N_PTS, 3), np.float64), np.zeros((N_PTS, 56), np.float64)
    for v in range(N_VIEWS):
        a, b = _view_grads(v, N_VIEWS)
        ref_pc += a; ref_ft += b
    for rank in range(world):
        assert np.abs(np.load(tmp_path / f"ov_pc_{rank}.npy") - ref_pc).max() <= 1e-4 * np.abs(ref_pc).max()
        assert np.abs(np.load(tmp_path / f"ov_ft_{rank}.npy") - ref_ft).max() <= 1e-4 * np.abs(ref_ft).max()


def test_shard_bounds_are_contiguous_and_cover():
    from taichi_3d_gaussian_splatting_amd.distributed import shard_bounds
    assert shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert shard_bounds(500000, 8)[-1] == 500000 and shard_bounds(3, 8) == [0, 1, 2, 3, 3, 3, 3, 3, 3]
