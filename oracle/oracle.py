"""ctypes wrapper around the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.
Arrays go in and come out as numpy, in the reference's own layouts
(GaussianPointCloudRasterisation.py:830-1023 forward, :1025-1163 backward).
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsoracle.so")
_lib = None


class Config(C.Structure):
    _fields_ = [
        ("near_plane", C.c_float), ("far_plane", C.c_float),
        ("depth_to_sort_key_scale", C.c_float), ("rgb_only", C.c_int),
        ("grad_color_factor", C.c_float), ("grad_high_order_color_factor", C.c_float),
        ("grad_s_factor", C.c_float), ("grad_q_factor", C.c_float),
        ("grad_alpha_factor", C.c_float), ("radius_from_preblur_cov", C.c_int), ("allow_partial_tiles", C.c_int),
        ("blend_exp", C.c_int), ("bwd_strict_dpdcov", C.c_int),
    ]


EXP_POLY, EXP_LIBM, EXP_FAST2, EXP_ULP2 = 0, 1, 2, 3      # gso_config.blend_exp (gs_oracle.h)


def default_config(**kw):
    """Reference defaults, GaussianPointCloudRasterisation.py:776-786."""
    c = Config(0.8, 1000.0, 100.0, 0, 5.0, 1.0, 0.5, 1.0, 20.0, 1, 0, EXP_POLY, 1)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


_F, _I8, _I32, _I64 = C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class _Frame(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("M", C.c_int64), ("K", C.c_int64),
        ("H", C.c_int32), ("W", C.c_int32), ("tiles_x", C.c_int32), ("tiles_y", C.c_int32),
        ("T", C.c_int32), ("n_objects", C.c_int32),
        ("q_camera_pointcloud", _F), ("t_camera_pointcloud", _F),
        ("point_in_camera_mask", _I8), ("point_id_in_camera_list", _I32),
        ("point_uv", _F), ("point_in_camera", _F), ("point_uv_conic_and_rescale", _F),
        ("point_alpha_after_activation", _F), ("point_color", _F), ("point_radii", _F),
        ("num_overlap_tiles", _I32), ("accumulated_num_overlap_tiles", _I64),
        ("sort_key_unsorted", _I64), ("point_offset_unsorted", _I32),
        ("sort_key", _I64), ("point_offset_with_sort_key", _I32),
        ("tile_points_start", _I32), ("tile_points_end", _I32),
        ("rasterized_image", _F), ("rasterized_depth", _F), ("pixel_accumulated_alpha", _F),
        ("pixel_offset_of_last_effective_point", _I32), ("pixel_valid_point_count", _I32),
        ("blend_exp", C.c_int32),
    ]


def build(force=False):
    """Compile libgsoracle.so with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "gs_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "gs_oracle.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libgsoracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.gso_expf.restype = C.c_float
        L.gso_expf.argtypes = [C.c_float]
        L.gso_exp_blend.restype = C.c_float
        L.gso_exp_blend.argtypes = [C.c_float]
        L.gso_forward.restype = C.POINTER(_Frame)
        L.gso_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                  C.POINTER(Config)]
        L.gso_backward.restype = C.c_int
        L.gso_backward.argtypes = [C.POINTER(_Frame)] + [C.c_void_p] * 7 + [C.c_int32, C.POINTER(Config)] + [C.c_void_p] * 8
        L.gso_backward_ex.restype = C.c_int
        L.gso_backward_ex.argtypes = [C.POINTER(_Frame)] + [C.c_void_p] * 7 + [C.c_int32, C.POINTER(Config)] + [C.c_void_p] * 10
        L.gso_pack_records.restype = None
        L.gso_pack_records.argtypes = [C.POINTER(_Frame), C.c_void_p]
        L.gso_forward_from_projected.restype = C.POINTER(_Frame)
        L.gso_forward_from_projected.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Config)]
        L.gso_backward_sums.restype = C.c_int
        L.gso_backward_sums.argtypes = [C.POINTER(_Frame), C.c_void_p, C.c_void_p, C.c_void_p]
        L.gso_backward_points.restype = C.c_int
        L.gso_backward_points.argtypes = [C.POINTER(_Frame)] + [C.c_void_p] * 6 + [C.c_int32, C.POINTER(Config)] + [C.c_void_p] * 5
        L.gso_frame_free.argtypes = [C.POINTER(_Frame)]
        L.gso_num_threads.restype = C.c_int
        for name, nargs in [("gso_inverse_se3_qt", None), ("gso_rotation_matrix_from_quaternion", None),
                            ("gso_project_to_camera_covariance", None),
                            ("gso_project_to_camera_position_jacobian", None),
                            ("gso_project_to_camera_covariance_jacobian", None),
                            ("gso_spherical_harmonics", None), ("gso_conic_and_rescale", None),
                            ("gso_find_tile_start_and_end", None)]:
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def exp_blend(x):
    """gso_exp_blend: the exp of the two blend loops (cheaper sequence than gso_expf, same accuracy on [-10, 0])."""
    L = lib()
    x = np.asarray(x, dtype=np.float32)
    return np.array([L.gso_exp_blend(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def expf(x):
    L = lib()
    x = np.asarray(x, dtype=np.float32)
    return np.array([L.gso_expf(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def inverse_se3_qt(q, t):
    q, t = _f32(q).reshape(-1, 4), _f32(t).reshape(-1, 3)
    qi, ti = np.empty_like(q), np.empty_like(t)
    lib().gso_inverse_se3_qt(_p(q), _p(t), C.c_int(q.shape[0]), _p(qi), _p(ti))
    return qi, ti


def rotation_matrix_from_quaternion(q):
    q = _f32(q)
    R = np.empty(9, np.float32)
    lib().gso_rotation_matrix_from_quaternion(_p(q), _p(R))
    return R.reshape(3, 3)


def project_to_camera_covariance(q, log_s, T, Kmat, xyz_camera):
    out = np.empty(4, np.float32)
    a = [_f32(q), _f32(log_s), _f32(T).reshape(16), _f32(Kmat).reshape(9), _f32(xyz_camera)]
    lib().gso_project_to_camera_covariance(*[_p(x) for x in a], _p(out))
    return out.reshape(2, 2)


def project_to_camera_position_jacobian(xyz, T, Kmat):
    out = np.empty(6, np.float32)
    a = [_f32(xyz), _f32(T).reshape(16), _f32(Kmat).reshape(9)]
    lib().gso_project_to_camera_position_jacobian(*[_p(x) for x in a], _p(out))
    return out.reshape(2, 3)


def project_to_camera_covariance_jacobian(q, log_s, T, Kmat, xyz_camera):
    dq, ds = np.empty(16, np.float32), np.empty(12, np.float32)
    a = [_f32(q), _f32(log_s), _f32(T).reshape(16), _f32(Kmat).reshape(9), _f32(xyz_camera)]
    lib().gso_project_to_camera_covariance_jacobian(*[_p(x) for x in a], _p(dq), _p(ds))
    return dq.reshape(4, 4), ds.reshape(4, 3)


def spherical_harmonics(d):
    out = np.empty(16, np.float32)
    d = _f32(d)
    lib().gso_spherical_harmonics(_p(d), _p(out))
    return out


def conic_and_rescale(cov):
    out = np.empty(4, np.float32)
    cov = _f32(cov).reshape(4)
    lib().gso_conic_and_rescale(_p(cov), _p(out))
    return out


def find_tile_start_and_end(sorted_keys, n_tiles):
    keys = np.ascontiguousarray(sorted_keys, dtype=np.int64)
    ts, te = np.zeros(n_tiles, np.int32), np.zeros(n_tiles, np.int32)
    lib().gso_find_tile_start_and_end(_p(keys), C.c_int64(keys.shape[0]), _p(ts), _p(te))
    return ts, te


@dataclass
class Forward:
    """numpy copies of every forward product (reference names)."""
    N: int
    M: int
    K: int
    H: int
    W: int
    arrays: dict
    _handle: object = None
    _inputs: tuple = None

    def __getattr__(self, k):
        try:
            return self.__dict__["arrays"][k]
        except KeyError:
            raise AttributeError(k)

    def free(self):
        if self._handle is not None:
            lib().gso_frame_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _np_from(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True).reshape(shape)


def forward(point_cloud, point_cloud_features, point_invalid_mask, point_object_id,
            q_pointcloud_camera, t_pointcloud_camera, camera_intrinsics, H, W, cfg=None):
    """Returns (Forward, features_after) -- features_after has the normalised quaternions."""
    cfg = cfg or default_config()
    pc = _f32(point_cloud)
    feat = _f32(point_cloud_features).copy()
    inv = np.ascontiguousarray(point_invalid_mask, dtype=np.int8)
    obj = np.ascontiguousarray(point_object_id, dtype=np.int32)
    q = _f32(q_pointcloud_camera).reshape(-1, 4)
    t = _f32(t_pointcloud_camera).reshape(-1, 3)
    Km = _f32(camera_intrinsics).reshape(9)
    N = pc.shape[0]
    h = lib().gso_forward(_p(pc), _p(feat), _p(inv), _p(obj), N, _p(q), _p(t), q.shape[0], _p(Km),
                          int(H), int(W), C.byref(cfg))
    if not h:
        raise ValueError("gso_forward rejected the arguments (W,H must be multiples of 16)")
    f = h.contents
    M, K, T = f.M, f.K, f.T
    A = {
        "q_camera_pointcloud": _np_from(f.q_camera_pointcloud, (q.shape[0], 4), np.float32),
        "t_camera_pointcloud": _np_from(f.t_camera_pointcloud, (q.shape[0], 3), np.float32),
        "point_in_camera_mask": _np_from(f.point_in_camera_mask, (N,), np.int8),
        "point_id_in_camera_list": _np_from(f.point_id_in_camera_list, (M,), np.int32),
        "point_uv": _np_from(f.point_uv, (M, 2), np.float32),
        "point_in_camera": _np_from(f.point_in_camera, (M, 3), np.float32),
        "point_uv_conic_and_rescale": _np_from(f.point_uv_conic_and_rescale, (M, 4), np.float32),
        "point_alpha_after_activation": _np_from(f.point_alpha_after_activation, (M,), np.float32),
        "point_color": _np_from(f.point_color, (M, 3), np.float32),
        "point_radii": _np_from(f.point_radii, (M,), np.float32),
        "num_overlap_tiles": _np_from(f.num_overlap_tiles, (M,), np.int32),
        "accumulated_num_overlap_tiles": _np_from(f.accumulated_num_overlap_tiles, (M,), np.int64),
        "sort_key_unsorted": _np_from(f.sort_key_unsorted, (K,), np.int64),
        "point_offset_unsorted": _np_from(f.point_offset_unsorted, (K,), np.int32),
        "sort_key": _np_from(f.sort_key, (K,), np.int64),
        "point_offset_with_sort_key": _np_from(f.point_offset_with_sort_key, (K,), np.int32),
        "tile_points_start": _np_from(f.tile_points_start, (T,), np.int32),
        "tile_points_end": _np_from(f.tile_points_end, (T,), np.int32),
        "rasterized_image": _np_from(f.rasterized_image, (H, W, 3), np.float32),
        "rasterized_depth": _np_from(f.rasterized_depth, (H, W), np.float32),
        "pixel_accumulated_alpha": _np_from(f.pixel_accumulated_alpha, (H, W), np.float32),
        "pixel_offset_of_last_effective_point": _np_from(f.pixel_offset_of_last_effective_point, (H, W), np.int32),
        "pixel_valid_point_count": _np_from(f.pixel_valid_point_count, (H, W), np.int32),
    }
    out = Forward(N=N, M=M, K=K, H=H, W=W, arrays=A, _handle=h,
                  _inputs=(pc, feat, obj, q, t, Km))
    return out, feat


def backward(fwd, grad_rasterized_image, color_max_sh_band=2, cfg=None, want_buffers=False, want_summed=False):
    """Reference backward (RAST:1025-1163) on a Forward from forward().  want_summed adds "summed_pointcloud" (N,3) and
    "summed_pointcloud_features" (N,56): per gradient element the sum of the absolute values of everything that was
    added up to produce it (gso_backward_ex)."""
    cfg = cfg or default_config()
    pc, feat, obj, q, t, Km = fwd._inputs
    N, M, H, W = fwd.N, fwd.M, fwd.H, fwd.W
    g = _f32(grad_rasterized_image)
    assert g.shape == (H, W, 3)
    out = {
        "grad_pointcloud": np.zeros((N, 3), np.float32),
        "grad_pointcloud_features": np.zeros((N, 56), np.float32),
        "grad_viewspace": np.zeros((N, 2), np.float32),
        "magnitude_grad_viewspace": np.zeros((N,), np.float32),
        "magnitude_grad_viewspace_on_image": np.zeros((H, W, 2), np.float32),
        "num_affected_pixels": np.zeros((max(M, 1),), np.int32),
        "grad_uv_cov_buffer": np.zeros((max(M, 1), 3), np.float32),
        "grad_color_buffer": np.zeros((max(M, 1), 3), np.float32),
    }
    if want_summed:
        out["summed_pointcloud"] = np.zeros((N, 3), np.float32)
        out["summed_pointcloud_features"] = np.zeros((N, 56), np.float32)
    rc = lib().gso_backward_ex(fwd._handle, _p(pc), _p(feat), _p(obj), _p(q), _p(t), _p(Km), _p(g),
                               int(color_max_sh_band), C.byref(cfg),
                               _p(out["grad_pointcloud"]), _p(out["grad_pointcloud_features"]),
                               _p(out["grad_viewspace"]), _p(out["magnitude_grad_viewspace"]),
                               _p(out["magnitude_grad_viewspace_on_image"]), _p(out["num_affected_pixels"]),
                               _p(out["grad_uv_cov_buffer"]), _p(out["grad_color_buffer"]),
                               _p(out["summed_pointcloud"]) if want_summed else None,
                               _p(out["summed_pointcloud_features"]) if want_summed else None)
    if rc != 0:
        raise RuntimeError("gso_backward failed")
    for k in ("num_affected_pixels", "grad_uv_cov_buffer", "grad_color_buffer"):
        out[k] = out[k][:M]
    if not want_buffers:
        out.pop("grad_uv_cov_buffer"), out.pop("grad_color_buffer")
    return out


def num_threads():
    return lib().gso_num_threads()


# ---- the path cut at the projected records and the per-point sums (stand-ins for the library's staged entry points) ----
def pack_records(fwd):
    """(M,16) f32 records of a Forward: the per-point arrays of RAST:873-911, one row per in-camera point."""
    out = np.zeros((fwd.M, 16), np.float32)
    if fwd.M:
        lib().gso_pack_records(fwd._handle, _p(out))
    return out


def forward_from_projected(records, H, W, cfg=None):
    """Binning + sort + blend from records (any concatenation of shards' records).  Returns a Forward whose per-point
    arrays are the records' and which has no point ids / mask (N = M)."""
    cfg = cfg or default_config()
    rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 16)
    M = rec.shape[0]
    h = lib().gso_forward_from_projected(M, _p(rec), int(H), int(W), C.byref(cfg))
    if not h:
        raise ValueError("gso_forward_from_projected rejected the arguments")
    f = h.contents
    K, T = f.K, f.T
    A = {
        "point_uv": _np_from(f.point_uv, (M, 2), np.float32),
        "point_in_camera": _np_from(f.point_in_camera, (M, 3), np.float32),
        "num_overlap_tiles": _np_from(f.num_overlap_tiles, (M,), np.int32),
        "sort_key": _np_from(f.sort_key, (K,), np.int64),
        "point_offset_with_sort_key": _np_from(f.point_offset_with_sort_key, (K,), np.int32),
        "tile_points_start": _np_from(f.tile_points_start, (T,), np.int32),
        "tile_points_end": _np_from(f.tile_points_end, (T,), np.int32),
        "rasterized_image": _np_from(f.rasterized_image, (H, W, 3), np.float32),
        "rasterized_depth": _np_from(f.rasterized_depth, (H, W), np.float32),
        "pixel_accumulated_alpha": _np_from(f.pixel_accumulated_alpha, (H, W), np.float32),
        "pixel_offset_of_last_effective_point": _np_from(f.pixel_offset_of_last_effective_point, (H, W), np.int32),
        "pixel_valid_point_count": _np_from(f.pixel_valid_point_count, (H, W), np.int32),
    }
    return Forward(N=M, M=M, K=K, H=H, W=W, arrays=A, _handle=h, _inputs=None)


def backward_sums(fwd, grad_rasterized_image):
    """Loop 1 of the backward (RAST:531-705) for a rendered frame: (sums (M,12) f32, magnitude image (H,W,2))."""
    g = _f32(grad_rasterized_image)
    assert g.shape == (fwd.H, fwd.W, 3)
    sums = np.zeros((max(fwd.M, 1), 12), np.float32)
    mag_img = np.zeros((fwd.H, fwd.W, 2), np.float32)
    if lib().gso_backward_sums(fwd._handle, _p(g), _p(sums), _p(mag_img)) != 0:
        raise RuntimeError("gso_backward_sums failed")
    return sums[:fwd.M], mag_img


def backward_points(fwd, sums, color_max_sh_band=2, cfg=None):
    """Loop 2 of the backward (RAST:708-772 + 1102-1125) for the shard `fwd` was computed from, given that shard's rows
    of the sums.  Returns grad_pointcloud (N,3), grad_pointcloud_features (N,56) and the hook extras."""
    cfg = cfg or default_config()
    pc, feat, obj, q, t, Km = fwd._inputs
    N, M = fwd.N, fwd.M
    s = np.ascontiguousarray(sums, dtype=np.float32).reshape(-1, 12)
    assert s.shape[0] == M
    if M == 0:
        s = np.zeros((1, 12), np.float32)
    out = {"grad_pointcloud": np.zeros((N, 3), np.float32), "grad_pointcloud_features": np.zeros((N, 56), np.float32),
           "grad_viewspace": np.zeros((N, 2), np.float32), "magnitude_grad_viewspace": np.zeros((N,), np.float32),
           "num_affected_pixels": np.zeros((max(M, 1),), np.int32)}
    rc = lib().gso_backward_points(fwd._handle, _p(pc), _p(feat), _p(obj), _p(t), _p(Km), _p(s), int(color_max_sh_band),
                                   C.byref(cfg), _p(out["grad_pointcloud"]), _p(out["grad_pointcloud_features"]),
                                   _p(out["grad_viewspace"]), _p(out["magnitude_grad_viewspace"]), _p(out["num_affected_pixels"]))
    if rc != 0:
        raise RuntimeError("gso_backward_points failed")
    out["num_affected_pixels"] = out["num_affected_pixels"][:M]
    return out
