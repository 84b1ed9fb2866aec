"""Synthetic scene generator synth(N, W, H, sigma0, sh_deg, seed) of SURVEY.md 8(d).

No real scene is available offline (the reference's committed .parquet files are
git-LFS stubs), so benchmarks and parity tests use this generator.  numpy only;
draws are made in float64 in a fixed order and cast to float32 at the end, so the
same (N, W, H, sigma0, sh_deg, seed) gives the same bytes everywhere.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class SyntheticScene:
    point_cloud: np.ndarray            # (N,3) f32
    point_cloud_features: np.ndarray   # (N,56) f32  [q xyzw | log s | opacity logit | R,G,B SH x16]
    point_invalid_mask: np.ndarray     # (N,) i8
    point_object_id: np.ndarray        # (N,) i32
    camera_intrinsics: np.ndarray      # (3,3) f32
    height: int
    width: int


def synth(N, W, H, sigma0, sh_deg=3, seed=0):
    rng = np.random.default_rng(seed)
    N = int(N)
    u = rng.uniform(-0.05 * W, 1.05 * W, N)
    v = rng.uniform(-0.05 * H, 1.05 * H, N)
    z = rng.uniform(2.0, 10.0, N)
    fx = fy = 0.6 * W
    cx, cy = W / 2.0, H / 2.0
    x = (u - cx) * z / fx
    y = (v - cy) * z / fy
    log_s = rng.normal(np.log(sigma0), 0.4, (N, 3))
    q = rng.normal(0.0, 1.0, (N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opacity = rng.uniform(-2.0, 4.0, N)
    sh = np.zeros((N, 3, 16))
    sh[:, :, 0] = rng.uniform(-1.5, 1.5, (N, 3)) / 0.2820948
    if sh_deg >= 3:
        sh[:, :, 1:] = rng.normal(0.0, 0.3, (N, 3, 15))
    feat = np.concatenate([q, log_s, opacity[:, None], sh.reshape(N, 48)], axis=1)
    Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
    return SyntheticScene(
        point_cloud=np.stack([x, y, z], axis=1).astype(np.float32),
        point_cloud_features=feat.astype(np.float32),
        point_invalid_mask=np.zeros(N, np.int8),
        point_object_id=np.zeros(N, np.int32),
        camera_intrinsics=Kmat, height=int(H), width=int(W))


def view_pose(i=0, n_views=1):
    """Pose of view i of n_views: camera at the origin looking +z, rotated about y by
    (i-(V-1)/2)*2 degrees.  Returns (q_pointcloud_camera (1,4) xyzw, t_pointcloud_camera (1,3))."""
    ang = np.deg2rad((i - (n_views - 1) / 2.0) * 2.0)
    q = np.array([[0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)]], dtype=np.float32)
    t = np.zeros((1, 3), np.float32)
    return q, t


# BASELINE.json configs -> generator arguments
CONFIGS = {
    "cfg1_plumbing": dict(N=10_000, W=256, H=256, sigma0=0.05, sh_deg=0),
    "cfg2_truck7k": dict(N=230_000, W=976, H=544, sigma0=0.02, sh_deg=3),
    "cfg3_headline": dict(N=500_000, W=1920, H=1088, sigma0=0.02, sh_deg=3),
    "cfg5_infer2e6": dict(N=2_000_000, W=1920, H=1088, sigma0=0.01, sh_deg=3),
}
