"""CPU: scene files.  No real scene is available offline (the reference's committed parquet files are
git-LFS stubs), so the formats are pinned by the reference's writer/reader source: column names and order of
to_parquet (GaussianPointCloudScene.py:132-146) and the PLY attribute order / rotation order / SH split of
to_ply (:148-180) and of the INRIA importer (benchmark/inference_benchmark.py:21-81)."""
import numpy as np
import pytest

from taichi_3d_gaussian_splatting_amd import scene_io
from taichi_3d_gaussian_splatting_amd.synthetic import synth


@pytest.fixture()
def scene():
    s = synth(500, 64, 64, 0.05, sh_deg=3, seed=3)
    return s.point_cloud, s.point_cloud_features


def test_parquet_round_trip_and_columns(tmp_path, scene):
    import pandas as pd
    pc, ft = scene
    mask = np.zeros(500, np.int8)
    mask[::7] = 1
    p = str(tmp_path / "scene.parquet")
    scene_io.save_parquet(p, pc, ft, mask)
    df = pd.read_parquet(p)
    assert list(df.columns) == ["x", "y", "z"] + scene_io.FEATURE_COLUMNS          # GaussianPointCloudScene.py:137-146
    assert list(df.columns[3:11]) == ["cov_q0", "cov_q1", "cov_q2", "cov_q3", "cov_s0", "cov_s1", "cov_s2", "alpha0"]
    pc2, ft2 = scene_io.load_parquet(p)
    assert np.array_equal(pc2, pc[mask == 0]) and np.array_equal(ft2, ft[mask == 0])
    pd.DataFrame(pc, columns=["x", "y", "z"]).to_parquet(str(tmp_path / "bare.parquet"))
    with pytest.raises(ValueError):
        scene_io.load_parquet(str(tmp_path / "bare.parquet"))


def test_inria_ply_layout_and_round_trip(tmp_path, scene):
    pc, ft = scene
    p = str(tmp_path / "point_cloud.ply")
    scene_io.save_inria_ply(p, pc, ft)
    raw = open(p, "rb").read()
    header, body = raw.split(b"end_header\n", 1)
    names = [l.split()[-1] for l in header.decode().splitlines() if l.startswith("property")]
    assert names == scene_io.PLY_PROPERTIES and len(names) == 62 and len(body) == 500 * 62 * 4
    rows = np.frombuffer(body, "<f4").reshape(500, 62)
    assert np.array_equal(rows[:, 0:3], pc) and not rows[:, 3:6].any()
    assert np.array_equal(rows[:, 6:9], ft[:, [8, 24, 40]])                       # f_dc = SH coefficient 0 of R,G,B
    assert np.array_equal(rows[:, 9:24], ft[:, 9:24]) and np.array_equal(rows[:, 39:54], ft[:, 41:56])   # f_rest channel-major
    assert np.array_equal(rows[:, 54], ft[:, 7]) and np.array_equal(rows[:, 55:58], ft[:, 4:7])
    assert np.array_equal(rows[:, 58:62], ft[:, [3, 0, 1, 2]])                    # rotation stored w,x,y,z
    pc2, ft2 = scene_io.load_inria_ply(p)
    assert np.array_equal(pc2, pc)
    assert np.allclose(ft2, ft, atol=1e-6)                                        # quaternions are re-normalised on import
    assert np.array_equal(ft2[:, 4:], ft[:, 4:])


def test_preallocation_rows():
    pc, ft = np.ones((10, 3), np.float32), np.ones((10, 56), np.float32)
    pc2, ft2, mask, obj = scene_io.preallocate(pc, ft, 2.5)
    assert pc2.shape == (25, 3) and ft2.shape == (25, 56) and mask.dtype == np.int8 and obj.dtype == np.int32
    assert mask[:10].sum() == 0 and mask[10:].all() and not pc2[10:].any()
    assert scene_io.preallocate(pc, ft, None)[0].shape == (10, 3)


def test_merge_scenes_assigns_object_ids_in_order(scene):
    pc, ft = scene
    mask = np.zeros(500, np.int8)
    mask[3] = 1
    mpc, mft, minv, mobj = scene_io.merge_scenes([(pc, ft), (pc[:100] + 1.0, ft[:100], mask[:100])])   # visualizer.py:292-323
    assert mpc.shape == (600, 3) and mft.shape == (600, 56)
    assert np.array_equal(mobj, np.concatenate([np.zeros(500, np.int32), np.ones(100, np.int32)]))
    assert minv.sum() == 1 and minv[503] == 1
    assert np.array_equal(mpc[500:], pc[:100] + 1.0)
