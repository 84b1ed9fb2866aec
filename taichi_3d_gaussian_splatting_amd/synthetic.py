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


def synth_clustered(N, W, H, sigma0, sh_deg=3, seed=0):
    """A heavy-tailed scene in the spirit of the trained truck scene behind the reference's published numbers
    (benchmark/README.md:2-33: blend 50-66 % of the frame): most of the Gaussians sit on one object in the middle of the
    image, a sparse shell of larger ones fills the background, and a few huge faint ones (floaters, sky, ground) cover
    hundreds of tiles each.  Per-tile list lengths then span two decades (max / mean > 10) where synth()'s are within 2x
    of each other -- the case that decides whether one wave per tile still schedules well.  Same draw discipline as
    synth(): float64, fixed order, cast at the end.
      75 %  object: three overlapping blobs around (0, 0, 5) covering about a tenth of the image, scale 0.6 sigma0, mostly
            translucent (opacity logit -3.5..1.5: long walks before a pixel saturates, as in a trained scene)
      24.6 % shell: uniform over the frustum at depth 7..20, scale 2 sigma0 (similar size in pixels)
      0.4 % floaters: anywhere at depth 2..20, scale 10 sigma0, opacity logit -4..-2"""
    rng = np.random.default_rng(seed)
    N = int(N)
    n_obj, n_flt = int(0.75 * N), int(0.004 * N)
    n_shell = N - n_obj - n_flt
    fx = fy = 0.6 * W
    cx, cy = W / 2.0, H / 2.0
    # object: blobs in camera space (the identity view looks down +z)
    centres = np.array([[-0.45, 0.10, 5.0], [0.40, 0.15, 5.4], [0.0, -0.25, 4.7]])
    spread = np.array([[0.32, 0.22, 0.30], [0.28, 0.25, 0.25], [0.40, 0.16, 0.30]])
    which = rng.integers(0, 3, n_obj)
    p_obj = centres[which] + rng.normal(0.0, 1.0, (n_obj, 3)) * spread[which]
    # shell and floaters: uniform in the image plane (5 % margin) at their depth ranges
    def frustum(n, z0, z1):
        u = rng.uniform(-0.05 * W, 1.05 * W, n)
        v = rng.uniform(-0.05 * H, 1.05 * H, n)
        z = rng.uniform(z0, z1, n)
        return np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], axis=1)
    p_shell = frustum(n_shell, 7.0, 20.0)
    p_flt = frustum(n_flt, 2.0, 20.0)
    xyz = np.concatenate([p_obj, p_shell, p_flt])
    base = np.concatenate([np.full(n_obj, 0.6 * sigma0), np.full(n_shell, 2.0 * sigma0), np.full(n_flt, 10.0 * sigma0)])
    log_s = rng.normal(0.0, 0.4, (N, 3)) + np.log(base)[:, None]
    q = rng.normal(0.0, 1.0, (N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opacity = rng.uniform(-2.0, 4.0, N)
    opacity[:n_obj] = rng.uniform(-3.5, 1.5, n_obj)
    opacity[n_obj + n_shell:] = rng.uniform(-4.0, -2.0, n_flt)
    sh = np.zeros((N, 3, 16))
    sh[:, :, 0] = rng.uniform(-1.5, 1.5, (N, 3)) / 0.2820948
    if sh_deg >= 3:
        sh[:, :, 1:] = rng.normal(0.0, 0.3, (N, 3, 15))
    perm = rng.permutation(N)                    # no structure in the point order (a trained scene's rows are not grouped)
    feat = np.concatenate([q, log_s, opacity[:, None], sh.reshape(N, 48)], axis=1)[perm]
    Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
    return SyntheticScene(
        point_cloud=xyz[perm].astype(np.float32), point_cloud_features=feat.astype(np.float32),
        point_invalid_mask=np.zeros(N, np.int8), point_object_id=np.zeros(N, np.int32),
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
# heavy-tailed counterparts of configs 2 and 3 (synth_clustered); not BASELINE configs -- reported beside them
CLUSTERED = {
    "cfg2_clustered": dict(N=230_000, W=976, H=544, sigma0=0.02, sh_deg=3),
    "cfg3_clustered": dict(N=500_000, W=1920, H=1088, sigma0=0.02, sh_deg=3),
}
# a small frame for plumbing rehearsals of bench.py (tests/test_gpu_bench_rehearsal.py)
SMALL = {"tiny_rehearsal": dict(N=20_000, W=256, H=160, sigma0=0.05, sh_deg=3)}


def make_scene(name):
    """Scene of a named workload: BASELINE configs through synth(), the clustered ones through synth_clustered()."""
    if name in CLUSTERED:
        return synth_clustered(**CLUSTERED[name])
    if name in SMALL:
        return synth(**SMALL[name])
    return synth(**CONFIGS[name])


def workload_args(name):
    return CLUSTERED.get(name) or SMALL.get(name) or CONFIGS[name]
