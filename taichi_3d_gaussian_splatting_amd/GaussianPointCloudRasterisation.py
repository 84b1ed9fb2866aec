"""GaussianPointCloudRasterisation -- drop-in for the reference operator
(taichi_3d_gaussian_splatting/GaussianPointCloudRasterisation.py:775-1204) backed by
libgsrast.so (hand-written HIP for MI355X / gfx950) through its C ABI.

Same class, nested dataclass names, field order, forward() contract and backward-hook
payload as the reference.  The Python side only validates arguments, allocates the output
tensors, keeps the opaque frame handle alive between forward and backward and dispatches
the hook; all arithmetic happens in the library.  There is no fallback path.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from . import _native
from .Camera import CameraInfo
from .controller_stats import ControllerAccumulators

TILE_WIDTH = 16
TILE_HEIGHT = 16

try:  # the reference mixes in dataclass_wizard.YAMLWizard (RAST:777); optional here
    from dataclass_wizard import YAMLWizard as _ConfigBase
except Exception:  # pragma: no cover - not installed in the build image
    class _ConfigBase:
        pass

_TORCH_DTYPES = {"float32": torch.float32, "int32": torch.int32, "int64": torch.int64, "int8": torch.int8}


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def _require(t: torch.Tensor, name: str, dtype, shape_tail, device=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on a GPU (cuda/hip device), got {t.device}")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} must have shape (*, {', '.join(map(str, shape_tail))}), got {tuple(t.shape)}")


class _on_device:
    """torch.cuda.device(dev) only when dev is not already current (the context manager costs microseconds per call)."""

    def __init__(self, dev):
        self._cm = None if dev.index is None or torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self._cm is not None:
            self._cm.__enter__()

    def __exit__(self, *exc):
        if self._cm is not None:
            return self._cm.__exit__(*exc)
        return False


class _Frame:
    """Owner of a gs_frame ticket: what ctx.save_for_backward keeps in the reference (RAST:998-1021).  Holds the
    context alive (the ticket is meaningless without it) and gives the ticket back when it dies."""

    def __init__(self, context: "_native.Context", handle, device, owned=True, lazy=False):
        """lazy: the frame was only begun (gs_project_shard_begin); its counts are read -- which waits for its kernels -- the
        first time one of them is asked for."""
        self._context, self._h, self.device, self._owned = context, handle, device, owned
        self.marshalled = None              # (gs_scene, gs_camera, gs_config) of the forward that made the frame
        if not lazy:
            self._read_info()

    def _read_info(self):
        info = _native.GsFrameInfo()
        _native.check(_native.lib().gs_frame_get_info(self._context.handle, self.handle, C.byref(info)), "gs_frame_get_info")
        self.n_points, self.n_points_in_camera, self.n_keys = info.n_points, info.n_points_in_camera, info.n_keys
        self.n_tiles, self.sort_key_bits, self.stages = info.n_tiles, info.sort_key_bits, info.stages
        self.sizing = ("exact", "predicted", "redone")[info.sizing]      # gs_frame_info.sizing: how the per-pixel half was sized

    def __getattr__(self, name):            # only reached for attributes not set yet: the counts of a lazy frame
        if name in ("n_points", "n_points_in_camera", "n_keys", "n_tiles", "sort_key_bits", "stages", "sizing"):
            self._read_info()
            return self.__dict__[name]
        raise AttributeError(name)

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError("frame already released")
        return self._h

    def export(self, name: str) -> torch.Tensor:
        eid, dtype, tail = _native.EXPORTS[name]
        L = _native.lib()
        n = L.gs_frame_export_count(self._context.handle, self.handle, eid)
        if n < 0:
            raise RuntimeError(f"gs_frame_export_count({name}) failed: the frame is no longer live or does not hold that stage")
        rows = n
        for d in tail:
            rows //= d
        out = torch.empty((rows, *tail), dtype=_TORCH_DTYPES[dtype], device=self.device)
        if n > 0:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _native.check(L.gs_frame_export(self._context.handle, self.handle, eid, _ptr(out), C.c_void_p(stream)), f"gs_frame_export({name})")
        return out

    def heavy_tiles(self, items: bool = False) -> int:
        """Diagnostic: tiles the last backward blend of this frame shared among four waves, or (items=True) the work items they were
        handed out as -- one per 512-entry segment of a list the forward cut (gs_frame_heavy_tiles)."""
        n = (C.c_int32 * 2)(0, 0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _native.check(_native.lib().gs_frame_heavy_tiles(self._context.handle, self.handle, n, C.c_void_p(stream)), "gs_frame_heavy_tiles")
        return int(n[1] if items else n[0])

    def release(self):
        """Hands the ticket back.  Transient frames (forward without gradient tracking) belong to the context and are
        recycled by its next forward; their ticket then simply stops resolving."""
        if self._h is not None:
            if self._owned and self._context.handle:
                _native.lib().gs_frame_release(self._context.handle, self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class GaussianPointCloudRasterisation(torch.nn.Module):
    @dataclass
    class GaussianPointCloudRasterisationConfig(_ConfigBase):
        near_plane: float = 0.8
        far_plane: float = 1000.
        depth_to_sort_key_scale: float = 100.
        rgb_only: bool = False
        # un-annotated on purpose, as in the reference (RAST:782-786): class attributes, not fields
        grad_color_factor = 5.
        grad_high_order_color_factor = 1.
        grad_s_factor = 0.5
        grad_q_factor = 1.
        grad_alpha_factor = 20.
        # extension over the reference: True lifts the W,H % 16 == 0 requirement (e.g. a true 1920x1080 frame)
        allow_partial_tiles = False
        # extension: True runs the backward's per-contribution Gaussian gradient (utils.py:331-348) in the reference's own f32
        # operation order instead of the faster, algebraically equal form (gs_config.bwd_reference_order)
        backward_reference_order = False

    @dataclass
    class GaussianPointCloudRasterisationInput:
        point_cloud: torch.Tensor  # Nx3
        point_cloud_features: torch.Tensor  # Nx56
        point_object_id: torch.Tensor  # N, int32, index into the pose rows
        point_invalid_mask: torch.Tensor  # N, int8
        camera_info: CameraInfo
        q_pointcloud_camera: torch.Tensor  # Kx4, xyzw
        t_pointcloud_camera: torch.Tensor  # Kx3
        color_max_sh_band: int = 2

    @dataclass
    class BackwardValidPointHookInput:
        point_id_in_camera_list: torch.Tensor  # M
        grad_point_in_camera: torch.Tensor  # Mx3
        grad_pointfeatures_in_camera: torch.Tensor  # Mx56
        grad_viewspace: torch.Tensor  # Mx2
        magnitude_grad_viewspace: torch.Tensor  # M
        magnitude_grad_viewspace_on_image: torch.Tensor  # HxWx2
        num_overlap_tiles: torch.Tensor  # M
        num_affected_pixels: torch.Tensor  # M
        point_depth: torch.Tensor  # M
        point_uv_in_camera: torch.Tensor  # Mx2

    def __init__(self, config: "GaussianPointCloudRasterisation.GaussianPointCloudRasterisationConfig",
                 backward_valid_point_hook: Optional[Callable[["GaussianPointCloudRasterisation.BackwardValidPointHookInput"], None]] = None,
                 controller_accumulators: Optional[ControllerAccumulators] = None):
        """`controller_accumulators` is an extension over the reference signature (RAST:819-824): when given, every
        backward adds this view's densification statistics to them on the device (controller_stats.py)."""
        super().__init__()
        _native.lib()                       # fail now, loudly, if libgsrast.so is absent
        self.config = config
        self._hook = backward_valid_point_hook
        self.controller_accumulators = controller_accumulators
        self._ctxs = {}                     # device index -> _native.Context (owner of the gs_ctx)
        self.last_frame: Optional[_Frame] = None   # inspection aid (tests / profiling); replaced every call
        self.last_forward_outputs = {}
        module = self

        class _module_function(torch.autograd.Function):
            @staticmethod
            def forward(ctx, pointcloud, pointcloud_features, point_invalid_mask, point_object_id,
                        q_pointcloud_camera, t_pointcloud_camera, camera_info, color_max_sh_band, grad_mode):
                # ctx.needs_input_grad says whether the inputs require grad, not whether a graph is being recorded (it is True
                # under torch.no_grad() too, and grad mode is always off inside forward): the caller passes the grad mode in
                needs_grad = bool(grad_mode and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
                outs, frame = module._run_forward(pointcloud, pointcloud_features, point_invalid_mask, point_object_id,
                                                  q_pointcloud_camera, t_pointcloud_camera, camera_info, keep=needs_grad)
                image, depth, acc_alpha, last, count = outs
                ctx.frame = frame if needs_grad else None
                ctx.camera_info = camera_info
                ctx.color_max_sh_band = color_max_sh_band
                ctx.save_for_backward(pointcloud, pointcloud_features, point_invalid_mask, point_object_id,
                                      q_pointcloud_camera, t_pointcloud_camera, acc_alpha, last)
                ctx.mark_non_differentiable(count)
                # the depth gradient is ignored (RAST:1157-1163) and the count has none: do not let autograd materialise
                # two image-sized zero tensors (two fill launches) per backward for them
                ctx.set_materialize_grads(False)
                return image, depth, count

            @staticmethod
            def backward(ctx, grad_rasterized_image, grad_rasterized_depth, grad_pixel_valid_point_count):
                grad_pointcloud = grad_pointcloud_features = None
                if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:      # RAST:1028
                    if ctx.frame is None:
                        raise RuntimeError("backward through a forward that ran without gradient tracking")
                    (pointcloud, pointcloud_features, point_invalid_mask, point_object_id, q_pointcloud_camera,
                     t_pointcloud_camera, acc_alpha, last) = ctx.saved_tensors
                    if grad_rasterized_image is None:       # only the depth was used downstream: its gradient does not flow (RAST:1157-1163)
                        grad_rasterized_image = torch.zeros(ctx.camera_info.camera_height, ctx.camera_info.camera_width, 3,
                                                            dtype=torch.float32, device=pointcloud.device)
                    grad_pointcloud, grad_pointcloud_features = module._run_backward(
                        ctx.frame, pointcloud, pointcloud_features, point_invalid_mask, point_object_id,
                        q_pointcloud_camera, t_pointcloud_camera, ctx.camera_info, acc_alpha, last,
                        grad_rasterized_image.contiguous(), ctx.color_max_sh_band)
                    # the frame is NOT released here: like the reference's saved tensors it lives as long as the graph
                    # node does, so backward(retain_graph=True) followed by another backward works; it goes back to the
                    # pool when autograd drops the node (_Frame.__del__)
                return grad_pointcloud, grad_pointcloud_features, None, None, None, None, None, None, None

        self._module_function = _module_function

    # ------------------------------------------------------------------ helpers
    def _context_for(self, device: torch.device) -> "_native.Context":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._ctxs:
            self._ctxs[idx] = _native.Context(idx)
        return self._ctxs[idx]

    def _ctx_for(self, device: torch.device):
        """The raw gs_ctx* of this module on `device` (profiling / diagnostics)."""
        return self._context_for(device).handle

    def _c_config(self):
        c = self.config
        return _native.GsConfig(c.near_plane, c.far_plane, c.depth_to_sort_key_scale, 1 if c.rgb_only else 0,
                                c.grad_color_factor, c.grad_high_order_color_factor, c.grad_s_factor,
                                c.grad_q_factor, c.grad_alpha_factor, 1 if getattr(c, "allow_partial_tiles", False) else 0,
                                1 if getattr(c, "backward_reference_order", False) else 0)

    @staticmethod
    def _c_scene(pointcloud, features, mask, obj):
        return _native.GsScene(_ptr(pointcloud), _ptr(features), _ptr(mask), _ptr(obj), pointcloud.shape[0])

    @staticmethod
    def _c_camera(q, t, camera_info, Kmat):
        return _native.GsCamera(_ptr(q), _ptr(t), q.shape[0], _ptr(Kmat), camera_info.camera_height, camera_info.camera_width)

    def _marshal(self, pointcloud, features, mask, obj, q, t, camera_info):
        Kmat = self._validate(pointcloud, features, mask, obj, q, t, camera_info)
        return self._c_scene(pointcloud, features, mask, obj), self._c_camera(q, t, camera_info, Kmat), self._c_config()

    def _validate(self, pointcloud, features, mask, obj, q, t, camera_info):
        dev = pointcloud.device
        _require(pointcloud, "point_cloud", torch.float32, (3,))
        _require(features, "point_cloud_features", torch.float32, (56,), dev)
        if features.shape[0] != pointcloud.shape[0] or mask.shape[0] != pointcloud.shape[0] or obj.shape[0] != pointcloud.shape[0]:
            raise ValueError("point_cloud, point_cloud_features, point_invalid_mask and point_object_id disagree on N")
        _require(mask, "point_invalid_mask", torch.int8, (), dev)
        _require(obj, "point_object_id", torch.int32, (), dev)
        _require(q, "q_pointcloud_camera", torch.float32, (4,), dev)
        _require(t, "t_pointcloud_camera", torch.float32, (3,), dev)
        if q.shape[0] != t.shape[0] or q.shape[0] < 1:
            raise ValueError("q_pointcloud_camera and t_pointcloud_camera must have the same, non-zero number of rows")
        Kmat = camera_info.camera_intrinsics
        if tuple(Kmat.shape) != (3, 3):
            raise ValueError("camera_intrinsics must be 3x3")
        if Kmat.dtype != torch.float32 or Kmat.device != dev or not Kmat.is_contiguous():
            Kmat = Kmat.to(device=dev, dtype=torch.float32).contiguous()
        return Kmat

    def _run_forward(self, pointcloud, features, mask, obj, q, t, camera_info, keep):
        Kmat = self._validate(pointcloud, features, mask, obj, q, t, camera_info)
        dev = pointcloud.device
        H, W = camera_info.camera_height, camera_info.camera_width
        rgb_only = bool(self.config.rgb_only)
        image = torch.empty(H, W, 3, dtype=torch.float32, device=dev)                   # RAST:967-976
        depth = torch.empty(H, W, dtype=torch.float32, device=dev)
        acc_alpha = torch.empty(H, W, dtype=torch.float32, device=dev)
        last = torch.empty(H, W, dtype=torch.int32, device=dev)
        count = torch.empty(H, W, dtype=torch.int32, device=dev)
        out = _native.GsForwardOut(_ptr(image), _ptr(depth), _ptr(acc_alpha), _ptr(last), _ptr(count))
        context = self._context_for(dev)
        ctxh = context.handle
        frame_h = C.c_void_p()
        scene, cam, cfg = self._c_scene(pointcloud, features, mask, obj), self._c_camera(q, t, camera_info, Kmat), self._c_config()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with _on_device(dev):
            _native.check(_native.lib().gs_forward(ctxh, C.byref(scene), C.byref(cam), C.byref(cfg), C.byref(out),
                                                   1 if keep else 0, C.byref(frame_h), C.c_void_p(stream)), "gs_forward")
        frame = _Frame(context, frame_h, dev, owned=keep)
        if keep:
            frame.marshalled = (scene, cam, cfg)       # the backward of this frame reads the same tensors (Kmat is kept alive below)
            frame._keepalive = Kmat
        self.last_frame = frame
        self.last_forward_outputs = {"pixel_accumulated_alpha": acc_alpha, "pixel_offset_of_last_effective_point": last}
        return (image, depth, acc_alpha, last, count), frame

    def _run_backward(self, frame, pointcloud, features, mask, obj, q, t, camera_info, acc_alpha, last, grad_image, sh_band):
        dev = pointcloud.device
        N, M = pointcloud.shape[0], frame.n_points_in_camera
        H, W = camera_info.camera_height, camera_info.camera_width
        # The host part of a backward sits on the step's critical path (the GPU has about one forward blend of queued work
        # when autograd gets here), so it is kept short: the inputs were validated and marshalled by the forward of this
        # frame (same tensors: autograd's saved tensors), and everything the hook receives comes out of ONE allocation.
        ms = frame.marshalled
        if ms is None or (ms[0].point_cloud, ms[0].point_cloud_features, ms[0].point_invalid_mask, ms[0].point_object_id, ms[0].n_points,
                          ms[1].q_pointcloud_camera, ms[1].t_pointcloud_camera) != (
                pointcloud.data_ptr() or None, features.data_ptr() or None, mask.data_ptr() or None, obj.data_ptr() or None, pointcloud.shape[0],
                q.data_ptr() or None, t.data_ptr() or None):
            ms = self._marshal(pointcloud, features, mask, obj, q, t, camera_info)     # storage was swapped since the forward (p.data = ...)
        scene, cam, _ = ms
        cfg = self._c_config()          # read again: the grad factors (and bwd_reference_order) may have changed since the forward
        if grad_image.dtype != torch.float32 or tuple(grad_image.shape) != (H, W, 3):
            raise ValueError("grad of rasterized_image must be float32 (H,W,3)")
        # one allocation for both gradients so that data-parallel training all-reduces ONE buffer; the 56-float rows
        # come first: 224*N bytes keep them 16-byte aligned for any N (the kernels store them as float4)
        flat = torch.empty(N * 59, dtype=torch.float32, device=dev)
        grad_feat = flat[:N * 56].view(N, 56)
        grad_pc = flat[N * 56:].view(N, 3)
        want_hook = self._hook is not None
        grad_uv = mag = mag_img = n_aff = h_pc = h_feat = h_uv = h_mag = h_ids = h_ntiles = h_depth = h_puv = None
        if want_hook:
            # twelve arrays, one buffer and one split (the 56-float rows first: the kernel stores them as float4 and the buffer
            # is aligned); a Python-level tensor op costs 1-2 us, so fewer of them is what shortens this path
            sizes = (M * 56, M * 3, M * 2, M, M, M * 2, N * 2, N, H * W * 2, M, M, M)
            parts = torch.empty(sum(sizes), dtype=torch.float32, device=dev).split(sizes)
            h_feat, h_pc, h_uv, h_mag, h_depth, h_puv = parts[0].view(M, 56), parts[1].view(M, 3), parts[2].view(M, 2), parts[3], parts[4], parts[5].view(M, 2)
            grad_uv, mag, mag_img = parts[6].view(N, 2), parts[7], parts[8].view(H, W, 2)
            n_aff, h_ids, h_ntiles = parts[9].view(torch.int32), parts[10].view(torch.int32), parts[11].view(torch.int32)
        ctrl = None
        if self.controller_accumulators is not None:
            ca = self.controller_accumulators
            ca.validate(N, dev)
            ctrl = _native.GsControllerAccumulators(
                _ptr(ca.accumulated_num_in_camera), _ptr(ca.accumulated_num_pixels),
                _ptr(ca.accumulated_view_space_position_gradients), _ptr(ca.accumulated_view_space_position_gradients_avg),
                _ptr(ca.accumulated_position_gradients), _ptr(ca.accumulated_position_gradients_norm))
        out = _native.GsBackwardOut(_ptr(grad_pc), _ptr(grad_feat), _ptr(grad_uv), _ptr(mag), _ptr(mag_img), _ptr(n_aff),
                                    _ptr(h_pc), _ptr(h_feat), _ptr(h_uv), _ptr(h_mag),
                                    C.pointer(ctrl) if ctrl is not None and N > 0 else None,
                                    _ptr(h_ids), _ptr(h_ntiles), _ptr(h_depth), _ptr(h_puv))
        stream = torch.cuda.current_stream(dev).cuda_stream
        with _on_device(dev):
            _native.check(_native.lib().gs_backward(self._ctx_for(dev), frame.handle, C.byref(scene), C.byref(cam), C.byref(cfg),
                                                    _ptr(grad_image), _ptr(acc_alpha), _ptr(last), int(sh_band),
                                                    C.byref(out), C.c_void_p(stream)), "gs_backward")
        self.last_backward_extras = dict(grad_viewspace=grad_uv, magnitude_grad_viewspace=mag,
                                         magnitude_grad_viewspace_on_image=mag_img, num_affected_pixels=n_aff)
        if want_hook:                                                                   # RAST:1127-1142
            self._hook(GaussianPointCloudRasterisation.BackwardValidPointHookInput(
                point_id_in_camera_list=h_ids,
                grad_point_in_camera=h_pc, grad_pointfeatures_in_camera=h_feat, grad_viewspace=h_uv,
                magnitude_grad_viewspace=h_mag, magnitude_grad_viewspace_on_image=mag_img,
                num_overlap_tiles=h_ntiles, num_affected_pixels=n_aff,
                point_depth=h_depth, point_uv_in_camera=h_puv))
        return grad_pc, grad_feat

    # ------------------------------------------------------------------ nn.Module
    def forward(self, input_data: "GaussianPointCloudRasterisation.GaussianPointCloudRasterisationInput"):
        camera_info = input_data.camera_info
        if not getattr(self.config, "allow_partial_tiles", False):
            assert camera_info.camera_width % TILE_WIDTH == 0        # RAST:1193-1194
            assert camera_info.camera_height % TILE_HEIGHT == 0
        return self._module_function.apply(
            input_data.point_cloud, input_data.point_cloud_features, input_data.point_invalid_mask,
            input_data.point_object_id, input_data.q_pointcloud_camera, input_data.t_pointcloud_camera,
            camera_info, input_data.color_max_sh_band, torch.is_grad_enabled())
