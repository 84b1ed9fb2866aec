"""Mint golden vectors from the reference's own pure-torch helpers.

Run ONCE in the build container (needs /root/reference; never runs on the GPU
box, never imported by the tests).  Output: tests/golden/reference_torch_helpers.json and
tests/golden/reference_single_point_batch.json -- data only (inputs + expected outputs).

What is executed: the plain-PyTorch helper functions of the reference's
taichi_3d_gaussian_splatting/utils.py (torch_single_point_alpha_forward :513-558,
inverse_SE3_qt_torch :426-432, quaternion_to_rotation_matrix_torch :596-632,
rotation_matrix_to_quaternion_torch :435-483, get_spherical_harmonic_from_xyz_torch
:635-657).  `taichi` is not installable here, so the module names `taichi` and
`taichi.math` are bound to an inert placeholder whose decorators return the
function untouched; no Taichi code runs and none of the @ti.func bodies is called.
Inputs are the ones the reference's tests use (tests/GaussianPointCloudRasterisation_test.py
:354-379, tests/GaussianPoint3D_test.py:12-67) plus seeded random poses.
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


class _Inert:
    """Attribute/call sink: ti.func, ti.kernel, ti.types.matrix(...), ti.math.vec3 ..."""

    def __getattr__(self, name):
        return _Inert()

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k and not isinstance(a[0], _Inert):
            return a[0]          # used as a bare decorator
        return _Inert()

    def __getitem__(self, k):
        return _Inert()


def _load_utils():
    ti = types.ModuleType("taichi")
    ti.__getattr__ = lambda name: _Inert()
    tm = types.ModuleType("taichi.math")
    tm.__getattr__ = lambda name: _Inert()
    ti.math = tm
    sys.modules["taichi"], sys.modules["taichi.math"] = ti, tm
    pkg = types.ModuleType("taichi_3d_gaussian_splatting")
    pkg.__path__ = [os.path.join(REF, "taichi_3d_gaussian_splatting")]
    sys.modules["taichi_3d_gaussian_splatting"] = pkg
    import importlib
    return importlib.import_module("taichi_3d_gaussian_splatting.utils")


def main():
    U = _load_utils()
    out = {}

    # ---- single point alpha + autograd gradients (T_RAST:354-379) -----------------
    T_camera_pointcloud = torch.tensor([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 2.], [0., 0., 0., 1.]])
    Kmat = torch.tensor([[32., 0., 16.], [0., 32., 16.], [0., 0., 1.]])
    xyz = torch.tensor([-0.4325, -0.7224, -0.4733], dtype=torch.float32, requires_grad=True)
    feat_list = [0.0115, 0.5507, 0.6920, 0.4666, float(np.log(0.6306)), float(np.log(0.0871)), float(np.log(0.0112)), 1.7667,
                 2.2963, 0.1560, 0.8710, 0.3418, 0.3658, 0.1913, 0.8727, 0.3608,
                 0.6874, 0.7516, 0.9281, 0.5649, 0.9469, 0.9090, 0.7356, 0.5436,
                 1.7886, 0.7542, 0.9568, 0.2868, 0.3552, 0.3872, 0.0827, 0.4101,
                 0.7783, 0.6266, 0.9601, 0.8252, 0.7846, 0.0183, 0.6635, 0.4688,
                 -1.4012, 0.1584, 0.3252, 0.5403, 0.4992, 0.2780, 0.7412, 0.5056,
                 0.8236, 0.9722, 0.5467, 0.6644, 0.2583, 0.0953, 0.3986, 0.2265]
    features = torch.tensor(feat_list, dtype=torch.float32, requires_grad=True)
    pixel_uv = torch.tensor([3, 3])
    with contextlib.redirect_stdout(io.StringIO()):
        alpha = U.torch_single_point_alpha_forward(
            point_xyz=xyz, point_q=features[:4], point_s=features[4:7],
            T_camera_pointcloud=T_camera_pointcloud, camera_intrinsics=Kmat,
            point_alpha=features[7], pixel_uv=pixel_uv)
        alpha.backward()
    out["single_point"] = {
        "source": "tests/GaussianPointCloudRasterisation_test.py:354-396 via utils.py:513-558",
        "T_camera_pointcloud": T_camera_pointcloud.tolist(), "camera_intrinsics": Kmat.tolist(),
        "xyz": xyz.detach().tolist(), "features": feat_list, "pixel_uv": [3, 3],
        "alpha": float(alpha.item()),
        "grad_xyz": xyz.grad.tolist(), "grad_features_0_8": features.grad[:8].tolist(),
        "tolerance": {"alpha_atol": 1e-4, "grad_xyz_atol": 1e-4, "grad_features_atol": 1e-2},
    }

    # ---- quaternion -> rotation (T_GP3D:56-67 input) --------------------------------
    q = torch.tensor([[0.0229, 0.9774, 0.1204, 0.1725]], dtype=torch.float64)
    out["quaternion_to_rotation"] = {
        "source": "tests/GaussianPoint3D_test.py:56-67 input through utils.py:596-632",
        "q_xyzw": q[0].tolist(), "R": U.quaternion_to_rotation_matrix_torch(q)[0].tolist(), "atol": 1e-2,
        "note": "q is not exactly unit (4-digit literals); the torch helper does not renormalise the "
                "xx,yy.. products, like rotation_matrix_from_quaternion GaussianPoint3D.py:30-48"}

    # ---- pose inversion on seeded poses (T_UTIL:139-157 pattern) ------------------
    rng = np.random.default_rng(1234)
    qs = rng.random((16, 4)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    ts = rng.random((16, 3))
    qi, ti = U.inverse_SE3_qt_torch(torch.tensor(qs), torch.tensor(ts))
    out["inverse_SE3_qt"] = {"source": "utils.py:426-432", "q": qs.tolist(), "t": ts.tolist(),
                             "q_inv": qi.tolist(), "t_inv": ti.tolist()}

    # ---- rotation matrix -> quaternion (host-side helper UTIL:435-483) ------------
    Rm = U.quaternion_to_rotation_matrix_torch(torch.tensor(qs))
    out["rotation_matrix_to_quaternion"] = {"source": "utils.py:435-483", "R": Rm.tolist(),
                                            "q": U.rotation_matrix_to_quaternion_torch(Rm).tolist()}

    # ---- SH basis (UTIL:635-657 mirrors SH:10-32) -----------------------------------
    dirs = rng.normal(size=(8, 3))
    sh = [U.get_spherical_harmonic_from_xyz_torch(torch.tensor(d.copy())).tolist() for d in dirs]
    out["spherical_harmonics"] = {"source": "utils.py:635-657", "xyz": dirs.tolist(), "sh16": sh}

    # ---- covariance projection, numpy/scipy expectation of T_GP3D:12-54 -------------
    import scipy.spatial.transform as transform
    proj = np.array([[32, 0, 16], [0, 32, 16], [0, 0, 1]], dtype=np.float32)
    p = np.array([-0.1316, -0.2471, 1.0090], dtype=np.float32)
    s = np.array([np.log(0.7606), np.log(0.9650), np.log(0.1946)])
    qq = np.array([0.0229, 0.9774, 0.1204, 0.1725])
    R = transform.Rotation.from_quat(qq).as_matrix()
    S = np.diag(np.exp(s))
    fx, fy = proj[0, 0], proj[1, 1]
    x, y, z = p
    J = np.array([[fx / z, 0, -fx * x / (z * z)], [0, fy / z, -fy * y / (z * z)]])
    cov = J @ (R @ S @ S @ R.T) @ J.T
    out["project_to_camera_covariance"] = {
        "source": "tests/GaussianPoint3D_test.py:12-54 (expected value computed exactly as the test does)",
        "xyz": p.tolist(), "log_s": s.tolist(), "q_xyzw": qq.tolist(), "camera_intrinsics": proj.tolist(),
        "cov": cov.tolist(), "rtol": 1e-2}

    # ---- tile ranges, T_RAST:19-42 (literal known answer) ---------------------------
    out["find_tile_start_and_end"] = {
        "source": "tests/GaussianPointCloudRasterisation_test.py:19-42",
        "keys": [0x100000000, 0x100000001, 0x200000000, 0x200000001, 0x200000002, 0x300000000, 0x300000001],
        "n_tiles": 4, "start": [0, 0, 2, 5], "end": [0, 2, 5, 7]}

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_torch_helpers.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)
    single_point_batch(U)


N_BATCH = 96


def _run_single_point(U, dtype, T, Km, xyz, q, s, opacity_logit, pixel):
    """torch_single_point_alpha_forward (utils.py:513-558) + autograd, evaluated in `dtype`.  The helper builds J
    with torch.tensor([...]) (default dtype, detached: no gradient reaches xyz through J, like RAST:708-772), so the
    default dtype is switched for the call instead of touching the helper."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        npdt = np.float64 if dtype == torch.float64 else np.float32
        tx = torch.tensor(xyz.astype(npdt), requires_grad=True)
        f = torch.tensor(np.concatenate([q, s, [opacity_logit]]).astype(npdt), requires_grad=True)
        import warnings
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = U.torch_single_point_alpha_forward(
                point_xyz=tx, point_q=f[:4], point_s=f[4:7], T_camera_pointcloud=torch.tensor(T.astype(npdt)),
                camera_intrinsics=torch.tensor(Km.astype(npdt)), point_alpha=f[7], pixel_uv=torch.tensor(pixel))
            a.backward()
        return float(a.item()), tx.grad.double().tolist(), f.grad.double().tolist()
    finally:
        torch.set_default_dtype(old)


def single_point_batch(U):
    """N_BATCH seeded (point, pose, intrinsics, pixel) cases through the reference's single-point helper: alpha and all
    eight Jacobian groups d alpha / d(xyz, q, s, opacity logit).  Every input is an exact float32 value.  Expected values
    are the helper evaluated in float64; the helper evaluated in float32 (the precision the reference's own test
    tests/GaussianPointCloudRasterisation_test.py:353-548 runs it in) is recorded beside them as the arithmetic's own
    noise level.  Varied: pose rotation up to 26 degrees about a random axis + translation, fx != fy, principal point,
    off-axis points 1.5 to 6 units in front of the camera, random quaternions (every fourth one NOT unit: the helper and
    rotation_matrix_from_quaternion GaussianPoint3D.py:30-48 both use the raw formula), scales 0.02 to 0.8 per axis
    (anisotropy up to 40:1), opacity logits -2 to 4, pixels up to two pixels off the projected centre."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(20261004)
    cases = []
    for i in range(N_BATCH):
        ang = rng.uniform(0, 0.45)
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec(ang * ax).as_matrix()
        T[:3, 3] = rng.uniform(-1, 1, 3) * np.array([0.5, 0.5, 1.0])
        T = T.astype(np.float32)
        fx, fy = rng.uniform(20, 60, 2)
        cx, cy = rng.uniform(10, 22, 2)
        Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
        p_cam = np.array([rng.uniform(-1.2, 1.2), rng.uniform(-1.2, 1.2), rng.uniform(1.5, 6.0)])
        xyz = (np.linalg.inv(T.astype(np.float64)) @ np.append(p_cam, 1.0))[:3].astype(np.float32)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if i % 4 == 3:
            q *= rng.uniform(0.9, 1.1)
        q = q.astype(np.float32)
        s = np.log(rng.uniform(0.02, 0.8, 3)).astype(np.float32)
        opacity_logit = np.float32(rng.uniform(-2, 4))
        pc = (T.astype(np.float64) @ np.append(xyz.astype(np.float64), 1.0))[:3]
        uv = (Km.astype(np.float64) @ pc)[:2] / pc[2]
        pixel = [int(np.floor(uv[0] + rng.uniform(-2, 2))), int(np.floor(uv[1] + rng.uniform(-2, 2)))]
        a64, gx64, gf64 = _run_single_point(U, torch.float64, T, Km, xyz, q, s, opacity_logit, pixel)
        a32, gx32, gf32 = _run_single_point(U, torch.float32, T, Km, xyz, q, s, opacity_logit, pixel)
        cases.append({
            "T_camera_pointcloud": T.astype(np.float64).tolist(), "camera_intrinsics": Km.astype(np.float64).tolist(),
            "xyz": xyz.astype(np.float64).tolist(), "q_xyzw": q.astype(np.float64).tolist(),
            "log_s": s.astype(np.float64).tolist(), "opacity_logit": float(opacity_logit), "pixel_uv": pixel,
            "float64": {"alpha": a64, "grad_xyz": gx64, "grad_q_s_opacity": gf64},
            "float32": {"alpha": a32, "grad_xyz": gx32, "grad_q_s_opacity": gf32}})
    out = {"source": "utils.py:513-558 (torch_single_point_alpha_forward) + torch.autograd, inputs seeded: "
                     "numpy default_rng(20261004), see make_golden.py:single_point_batch",
           "cases": cases}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_single_point_batch.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
