"""GPU (-m gpu): scene files through the HIP operator.  A scene is written with the reference's parquet
columns (GaussianPointCloudScene.py:132-146) and as an INRIA PLY (:148-180), loaded back, given the max_num_points_ratio
rows (:28-37) and rendered; every integer product and per-point f32 array must equal the oracle's on the same loaded
arrays bit for bit.  No real scene exists offline (the reference's parquet files are git-LFS stubs): the points are the
synthetic generator's."""
import numpy as np
import pytest
import torch

from oracle import oracle
from taichi_3d_gaussian_splatting_amd import scene_io
from taichi_3d_gaussian_splatting_amd.synthetic import synth, view_pose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import parity_util
    return parity_util


class _Scene:
    pass


def _as_scene(base, pc, ft, mask, obj):
    s = _Scene()
    s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id = pc, ft, mask, obj
    s.camera_intrinsics, s.height, s.width = base.camera_intrinsics, base.height, base.width
    return s


def test_parquet_scene_with_preallocated_rows_trains_through_the_operator(P, tmp_path):
    """save_parquet drops the invalid rows (as to_parquet does), load + preallocate(1.5) appends invalid spare rows
    (max_num_points_ratio); forward + backward on the loaded arrays against the oracle, spare rows get zero gradient."""
    base = synth(6000, 320, 192, 0.05, sh_deg=3, seed=41)
    mask = (np.random.default_rng(1).random(6000) < 0.15).astype(np.int8)
    path = str(tmp_path / "scene.parquet")
    scene_io.save_parquet(path, base.point_cloud, base.point_cloud_features, mask)
    pc, ft = scene_io.load_parquet(path)
    assert pc.shape[0] == int((mask == 0).sum())
    pc, ft, inv, obj = scene_io.preallocate(pc, ft, 1.5)
    n_valid = int((mask == 0).sum())
    assert pc.shape[0] == int(n_valid * 1.5) and inv[:n_valid].sum() == 0 and inv[n_valid:].all()
    s = _as_scene(base, pc, ft, inv, obj)
    q, t = view_pose(1, 3)
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, 3)
    f, feat_after = P.run_oracle(s, q, t)
    outs = module(inp)
    P.assert_forward_parity(module, inp, outs, f, feat_after)
    g = 2.0 * (outs[0].detach() - 0.5)
    outs[0].backward(g)
    P.assert_backward_parity(module, inp, g.cpu().numpy(), f, 3)
    assert not inp.point_cloud.grad[n_valid:].any() and not inp.point_cloud_features.grad[n_valid:].any()


def test_two_loaded_scenes_as_objects_like_the_visualiser(P, tmp_path):
    """visualizer.py:60-75, 273-284: several scene files merged into one point set, one pose row per scene (Kobj = 2),
    rendered under torch.no_grad.  One scene comes from the parquet layout, the other from an INRIA PLY."""
    a = synth(4000, 256, 160, 0.05, sh_deg=3, seed=42)
    b = synth(2500, 256, 160, 0.08, sh_deg=3, seed=43)
    pa, pb = str(tmp_path / "a.parquet"), str(tmp_path / "b.ply")
    scene_io.save_parquet(pa, a.point_cloud, a.point_cloud_features)
    scene_io.save_inria_ply(pb, b.point_cloud, b.point_cloud_features)
    la, lb = scene_io.load_parquet(pa), scene_io.load_inria_ply(pb)
    assert np.array_equal(la[1], a.point_cloud_features)                  # parquet is lossless
    pc, ft, inv, obj = scene_io.merge_scenes([la, lb])
    assert obj[:4000].max() == 0 and obj[4000:].min() == 1
    s = _as_scene(a, pc, ft, inv, obj)
    q = np.array([[0.0, 0.02, 0.0, 1.0], [0.03, -0.05, 0.01, 0.98]], np.float32)
    t = np.array([[0.0, 0.0, 0.0], [0.4, -0.2, 0.5]], np.float32)
    f, feat_after = P.run_oracle(s, q, t)
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, 3, requires_grad=False)
    with torch.no_grad():
        outs = module(inp)
    P.assert_forward_parity(module, inp, outs, f, feat_after)
    assert f.M > 3000 and (f.point_id_in_camera_list >= 4000).any()
