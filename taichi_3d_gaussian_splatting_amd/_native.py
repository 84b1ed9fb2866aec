"""ctypes binding of libgsrast.so (include/gs_rasterizer.h).

The library is hand-written HIP for gfx950; there is no CPU or PyTorch fallback.  If it
is missing or does not load, every entry point of this package raises -- loudly.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSRAST_LIB selects another build of the same library (e.g. the counter-instrumented `make stats` one); no other fallback
LIB_PATH = os.environ.get("GSRAST_LIB") or os.path.join(_HERE, "lib", "libgsrast.so")
ABI_VERSION = 8

_I64, _I32, _F32, _VP = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class GsConfig(C.Structure):
    _fields_ = [("near_plane", _F32), ("far_plane", _F32), ("depth_to_sort_key_scale", _F32),
                ("rgb_only", _I32), ("grad_color_factor", _F32), ("grad_high_order_color_factor", _F32),
                ("grad_s_factor", _F32), ("grad_q_factor", _F32), ("grad_alpha_factor", _F32),
                ("allow_partial_tiles", _I32), ("bwd_reference_order", _I32)]


class GsScene(C.Structure):
    _fields_ = [("point_cloud", _VP), ("point_cloud_features", _VP), ("point_invalid_mask", _VP),
                ("point_object_id", _VP), ("n_points", _I64)]


class GsCamera(C.Structure):
    _fields_ = [("q_pointcloud_camera", _VP), ("t_pointcloud_camera", _VP), ("n_objects", _I32),
                ("camera_intrinsics", _VP), ("camera_height", _I32), ("camera_width", _I32)]


class GsForwardOut(C.Structure):
    _fields_ = [("rasterized_image", _VP), ("rasterized_depth", _VP), ("pixel_accumulated_alpha", _VP),
                ("pixel_offset_of_last_effective_point", _VP), ("pixel_valid_point_count", _VP)]


class GsFrameInfo(C.Structure):
    _fields_ = [("n_points", _I64), ("n_points_in_camera", _I64), ("n_keys", _I64), ("n_tiles", _I32),
                ("camera_height", _I32), ("camera_width", _I32), ("sort_key_bits", _I32),
                ("kept_for_backward", _I32), ("stages", _I32), ("sizing", _I32)]


STAGE_PROJECT, STAGE_RASTER = 1, 2
RECORD_FLOATS, SPLAT_SUM_FLOATS = 16, 12


class GsLossImage(C.Structure):
    """a (3,H,W) f32 image by base pointer and (channel, row, column) strides in floats"""
    _fields_ = [("data", _VP), ("stride_channel", _I64), ("stride_row", _I64), ("stride_column", _I64)]

    @classmethod
    def of(cls, t):
        return cls(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


class GsControllerAccumulators(C.Structure):
    _fields_ = [("accumulated_num_in_camera", _VP), ("accumulated_num_pixels", _VP),
                ("accumulated_view_space_position_gradients", _VP),
                ("accumulated_view_space_position_gradients_avg", _VP),
                ("accumulated_position_gradients", _VP), ("accumulated_position_gradients_norm", _VP)]


class GsBackwardOut(C.Structure):
    _fields_ = [("grad_pointcloud", _VP), ("grad_pointcloud_features", _VP), ("grad_viewspace", _VP),
                ("magnitude_grad_viewspace", _VP), ("magnitude_grad_viewspace_on_image", _VP),
                ("num_affected_pixels", _VP), ("hook_grad_point_in_camera", _VP),
                ("hook_grad_pointfeatures_in_camera", _VP), ("hook_grad_viewspace", _VP),
                ("hook_magnitude_grad_viewspace", _VP), ("controller", C.POINTER(GsControllerAccumulators)),
                ("hook_point_id_in_camera_list", _VP), ("hook_num_overlap_tiles", _VP), ("hook_point_depth", _VP),
                ("hook_point_uv_in_camera", _VP)]


# gs_export ids (include/gs_rasterizer.h) -> (name, numpy/torch dtype name, trailing shape)
EXPORTS = {
    "point_id_in_camera_list": (0, "int32", ()),
    "point_uv": (1, "float32", (2,)),
    "point_in_camera": (2, "float32", (3,)),
    "point_uv_conic_and_rescale": (3, "float32", (4,)),
    "point_alpha_after_activation": (4, "float32", ()),
    "point_color": (5, "float32", (3,)),
    "point_radii": (6, "float32", ()),
    "num_overlap_tiles": (7, "int32", ()),
    "accumulated_num_overlap_tiles": (8, "int64", ()),
    "sort_key": (9, "int64", ()),
    "point_offset_with_sort_key": (10, "int32", ()),
    "tile_points_start": (11, "int32", ()),
    "tile_points_end": (12, "int32", ()),
    "point_depth": (13, "float32", ()),
    "point_in_camera_mask": (14, "int8", ()),
    "records": (15, "float32", (16,)),
}

# every symbol include/gs_rasterizer.h declares
SYMBOLS = ["gs_abi_version", "gs_last_error", "gs_create", "gs_destroy", "gs_forward", "gs_frame_get_info",
           "gs_frame_export_count", "gs_frame_export", "gs_backward", "gs_frame_release", "gs_frame_heavy_tiles",
           "gs_ctx_device_bytes", "gs_ctx_counter_wait_ns", "gs_kernel_names", "gs_profile_enable", "gs_profile_read",
           "gs_loss_l1_ssim", "gs_loss_maps_floats", "gs_loss_l1_ssim_forward", "gs_loss_l1_ssim_backward", "gs_adam_step", "gs_scale_regulariser", "gs_scale_regulariser_grad",
           "gs_project_shard", "gs_project_shard_begin", "gs_forward_projected", "gs_backward_projected", "gs_backward_shard"]

_lib = None


def source_digest():
    """sha256 (12 hex digits) over the kernel sources and the ABI header: what a profile taken with one build of the library is
    labelled with, so that counters are never quoted beside another build's timings (bench.py: roofline.traffic / .valu)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")) or f == "Makefile")
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_HERE), "include", "gs_rasterizer.h"), "rb").read())
    return h.hexdigest()[:12]


class NativeLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libgsrast.so for gfx950 with hipcc (csrc/Makefile)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=out)
    return LIB_PATH


def lib():
    """Load libgsrast.so or raise.  Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for this operator.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for s in SYMBOLS:
        if not hasattr(L, s):
            raise NativeLibraryError(f"{LIB_PATH} does not export {s}")
    L.gs_abi_version.restype = C.c_int
    L.gs_last_error.restype = C.c_char_p
    L.gs_kernel_names.restype = C.c_char_p
    L.gs_create.argtypes = [_I32, C.POINTER(_VP)]
    L.gs_destroy.argtypes = [_VP]
    L.gs_forward.argtypes = [_VP, C.POINTER(GsScene), C.POINTER(GsCamera), C.POINTER(GsConfig),
                             C.POINTER(GsForwardOut), _I32, C.POINTER(_VP), _VP]
    L.gs_frame_get_info.argtypes = [_VP, _VP, C.POINTER(GsFrameInfo)]
    L.gs_frame_export_count.argtypes = [_VP, _VP, C.c_int]
    L.gs_frame_export_count.restype = _I64
    L.gs_frame_export.argtypes = [_VP, _VP, C.c_int, _VP, _VP]
    L.gs_project_shard.argtypes = [_VP, C.POINTER(GsScene), C.POINTER(GsCamera), C.POINTER(GsConfig), _VP, _VP, _I32,
                                   C.POINTER(_VP), _VP]
    L.gs_project_shard_begin.argtypes = [_VP, C.POINTER(GsScene), C.POINTER(GsCamera), C.POINTER(GsConfig), _I32, C.POINTER(_VP), _VP]
    L.gs_forward_projected.argtypes = [_VP, _VP, _I64, C.POINTER(GsCamera), C.POINTER(GsConfig), C.POINTER(GsForwardOut), _I32,
                                       C.POINTER(_VP), _VP]
    L.gs_backward_projected.argtypes = [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]
    L.gs_backward_shard.argtypes = [_VP, _VP, C.POINTER(GsScene), C.POINTER(GsCamera), C.POINTER(GsConfig), _VP, _I32,
                                    C.POINTER(GsBackwardOut), _VP]
    L.gs_backward.argtypes = [_VP, _VP, C.POINTER(GsScene), C.POINTER(GsCamera), C.POINTER(GsConfig),
                              _VP, _VP, _VP, _I32, C.POINTER(GsBackwardOut), _VP]
    L.gs_frame_release.argtypes = [_VP, _VP]
    L.gs_frame_heavy_tiles.argtypes = [_VP, _VP, C.POINTER(_I32), _VP]
    L.gs_ctx_device_bytes.argtypes = [_VP]
    L.gs_profile_enable.argtypes = [_VP, C.c_uint64]
    L.gs_loss_l1_ssim.argtypes = [_VP, _VP, _VP, _I32, _I32, _F32, _VP, _VP, _VP]
    L.gs_loss_maps_floats.argtypes = [_I32, _I32]
    L.gs_loss_maps_floats.restype = _I64
    L.gs_loss_l1_ssim_forward.argtypes = [_VP, C.POINTER(GsLossImage), C.POINTER(GsLossImage), _I32, _I32, _I32, _F32, _VP, _VP, _VP]
    L.gs_loss_l1_ssim_backward.argtypes = [_VP, C.POINTER(GsLossImage), C.POINTER(GsLossImage), _I32, _I32, _I32, _F32, _VP, _VP,
                                           C.POINTER(GsLossImage), _VP]
    L.gs_scale_regulariser.argtypes = [_VP, _VP, _VP, _I64, _VP, _VP]
    L.gs_scale_regulariser_grad.argtypes = [_VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP]
    L.gs_adam_step.argtypes = [_VP, _VP, _VP, _VP, _VP, _I64, _F32, _F32, _F32, _F32, _I64, _VP]
    L.gs_profile_read.argtypes = [_VP, C.POINTER(C.c_double), C.POINTER(_I64), _I32, _I32]
    L.gs_ctx_device_bytes.restype = _I64
    L.gs_ctx_counter_wait_ns.argtypes = [_VP]
    L.gs_ctx_counter_wait_ns.restype = _I64
    if L.gs_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH} has ABI {L.gs_abi_version()}, this package expects {ABI_VERSION}")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().gs_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


class Context:
    """Owner of one gs_ctx.  Everything that can outlive the operator module -- above all the frame handles autograd keeps
    between forward and backward -- holds a strong reference to this object, and gs_destroy runs only from its finaliser,
    i.e. after the last frame is gone; never while the interpreter is shutting down (the HIP runtime may already be)."""

    def __init__(self, device_index: int):
        self.handle = C.c_void_p()
        self.device_index = device_index
        check(lib().gs_create(device_index, C.byref(self.handle)), "gs_create")

    def __del__(self):
        try:
            if self.handle and sys is not None and not sys.is_finalizing():
                lib().gs_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


_shared_ctx = {}


def shared_ctx(device_index: int):
    """A process-wide gs_ctx per device for the stateless helpers (loss, Adam)."""
    if device_index not in _shared_ctx:
        _shared_ctx[device_index] = Context(device_index)
    return _shared_ctx[device_index].handle
