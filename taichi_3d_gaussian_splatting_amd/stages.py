"""The operator's path cut at the projected records and at the per-splat sums: thin tensor wrappers over
gs_project_shard / gs_forward_projected / gs_backward_projected / gs_backward_shard (include/gs_rasterizer.h).

These are the four calls the Gaussian-parallel multi-GPU scheme is made of (distributed.GaussianParallelRasteriser,
DESIGN.md section 6): the rank that owns a shard of the Gaussians runs project_shard and backward_shard, the rank that
renders a view runs forward_projected and backward_projected; records (M,16) and sums (M,12) travel in between.
Chained on one device they reproduce GaussianPointCloudRasterisation bit for bit.  No fallback: all arithmetic is in
libgsrast.so."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _native
from .GaussianPointCloudRasterisation import GaussianPointCloudRasterisation as _Rast, _Frame, _ptr

RECORD_FLOATS = _native.RECORD_FLOATS
SPLAT_SUM_FLOATS = _native.SPLAT_SUM_FLOATS


@dataclass
class RasterOutputs:
    rasterized_image: torch.Tensor                         # (H,W,3)
    rasterized_depth: torch.Tensor                         # (H,W)
    pixel_accumulated_alpha: torch.Tensor                  # (H,W)
    pixel_offset_of_last_effective_point: torch.Tensor     # (H,W) i32
    pixel_valid_point_count: torch.Tensor                  # (H,W) i32


@dataclass
class ShardGradients:
    grad_pointcloud: torch.Tensor                          # (N,3) view of `flat`
    grad_pointcloud_features: torch.Tensor                 # (N,56) view of `flat`
    flat: torch.Tensor                                     # (59 N,) [features | positions], one buffer for collectives
    num_affected_pixels: Optional[torch.Tensor] = None     # (M) i32
    grad_viewspace: Optional[torch.Tensor] = None          # (N,2)
    magnitude_grad_viewspace: Optional[torch.Tensor] = None  # (N)


class StagedRasteriser:
    """One gs_ctx per instance per device; `module` supplies config, validation and struct marshalling."""

    def __init__(self, config: Optional["_Rast.GaussianPointCloudRasterisationConfig"] = None):
        self.module = _Rast(config or _Rast.GaussianPointCloudRasterisationConfig())

    # -- per-point half, forward ------------------------------------------------------------------------------------
    def project_shard(self, inp: "_Rast.GaussianPointCloudRasterisationInput", keep: bool = True):
        """Returns (records (M,16) f32, ids (M) i32 ascending, frame)."""
        m = self.module
        pc, ft = inp.point_cloud, inp.point_cloud_features
        Kmat = m._validate(pc, ft, inp.point_invalid_mask, inp.point_object_id, inp.q_pointcloud_camera,
                           inp.t_pointcloud_camera, inp.camera_info)
        dev = pc.device
        N = pc.shape[0]
        records = torch.empty(max(N, 1), RECORD_FLOATS, dtype=torch.float32, device=dev)
        ids = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        context = m._context_for(dev)
        frame_h = C.c_void_p()
        scene = m._c_scene(pc, ft, inp.point_invalid_mask, inp.point_object_id)
        cam = m._c_camera(inp.q_pointcloud_camera, inp.t_pointcloud_camera, inp.camera_info, Kmat)
        cfg = m._c_config()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_project_shard(
                context.handle, C.byref(scene), C.byref(cam), C.byref(cfg), _ptr(records), _ptr(ids), 1 if keep else 0,
                C.byref(frame_h), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_project_shard")
        frame = _Frame(context, frame_h, dev, owned=keep)
        M = frame.n_points_in_camera
        return records[:M], ids[:M], frame

    def project_shard_begin(self, inp: "_Rast.GaussianPointCloudRasterisationInput", keep: bool = True) -> _Frame:
        """Queues the same stage and returns at once (gs_project_shard_begin): an owner projecting its shard for several views
        begins them all, so that the GPU runs them back to back, and fetches each result with project_shard_finish()."""
        m = self.module
        pc, ft = inp.point_cloud, inp.point_cloud_features
        Kmat = m._validate(pc, ft, inp.point_invalid_mask, inp.point_object_id, inp.q_pointcloud_camera,
                           inp.t_pointcloud_camera, inp.camera_info)
        dev = pc.device
        context = m._context_for(dev)
        frame_h = C.c_void_p()
        scene = m._c_scene(pc, ft, inp.point_invalid_mask, inp.point_object_id)
        cam = m._c_camera(inp.q_pointcloud_camera, inp.t_pointcloud_camera, inp.camera_info, Kmat)
        cfg = m._c_config()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_project_shard_begin(
                context.handle, C.byref(scene), C.byref(cam), C.byref(cfg), 1 if keep else 0,
                C.byref(frame_h), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_project_shard_begin")
        frame = _Frame(context, frame_h, dev, owned=keep, lazy=True)
        frame._keepalive = (Kmat, inp.q_pointcloud_camera, inp.t_pointcloud_camera)      # read by kernels that may not have run yet
        return frame

    def project_shard_finish(self, frame: _Frame, want_ids: bool = True):
        """(records (M,16) f32, ids (M) i32 ascending or None) of a frame begun with project_shard_begin; waits for its kernels."""
        return frame.export("records"), (frame.export("point_id_in_camera_list") if want_ids else None)

    # -- per-pixel half, forward ------------------------------------------------------------------------------------
    def forward_projected(self, records: torch.Tensor, camera_info, keep: bool = True):
        """records: (M,16) f32 contiguous, any concatenation of shards' records.  Returns (RasterOutputs, frame)."""
        m = self.module
        if records.dtype != torch.float32 or not records.is_cuda or not records.is_contiguous() or records.dim() != 2 \
                or records.shape[1] != RECORD_FLOATS:
            raise TypeError("records must be a contiguous float32 (M,16) GPU tensor")
        dev = records.device
        H, W = camera_info.camera_height, camera_info.camera_width
        if not getattr(m.config, "allow_partial_tiles", False):
            assert W % 16 == 0 and H % 16 == 0                                          # RAST:1193-1194
        e = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device=dev)
        outs = RasterOutputs(e(H, W, 3), e(H, W), e(H, W), e(H, W, dtype=torch.int32), e(H, W, dtype=torch.int32))
        fo = _native.GsForwardOut(_ptr(outs.rasterized_image), _ptr(outs.rasterized_depth), _ptr(outs.pixel_accumulated_alpha),
                                  _ptr(outs.pixel_offset_of_last_effective_point), _ptr(outs.pixel_valid_point_count))
        cam = _native.GsCamera(C.c_void_p(0), C.c_void_p(0), 1, C.c_void_p(0), H, W)
        cfg = m._c_config()
        context = m._context_for(dev)
        frame_h = C.c_void_p()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_forward_projected(
                context.handle, _ptr(records), records.shape[0], C.byref(cam), C.byref(cfg), C.byref(fo), 1 if keep else 0,
                C.byref(frame_h), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_forward_projected")
        return outs, _Frame(context, frame_h, dev, owned=keep)

    # -- per-pixel half, backward -----------------------------------------------------------------------------------
    def backward_projected(self, frame: _Frame, outs: RasterOutputs, grad_rasterized_image: torch.Tensor,
                           want_magnitude_image: bool = False):
        """Returns (splat_sums (M,12) f32, magnitude_grad_viewspace_on_image (H,W,2) or None)."""
        dev = frame.device
        g = grad_rasterized_image.contiguous()
        if g.dtype != torch.float32 or tuple(g.shape) != tuple(outs.rasterized_image.shape):
            raise ValueError("grad of rasterized_image must be float32 (H,W,3)")
        M = frame.n_points_in_camera
        sums = torch.empty(max(M, 1), SPLAT_SUM_FLOATS, dtype=torch.float32, device=dev)
        mag = torch.empty(*g.shape[:2], 2, dtype=torch.float32, device=dev) if want_magnitude_image else None
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_backward_projected(
                frame._context.handle, frame.handle, _ptr(g), _ptr(outs.pixel_accumulated_alpha),
                _ptr(outs.pixel_offset_of_last_effective_point), _ptr(sums), _ptr(mag),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_backward_projected")
        return sums[:M], mag

    # -- per-point half, backward -----------------------------------------------------------------------------------
    def backward_shard(self, frame: _Frame, inp: "_Rast.GaussianPointCloudRasterisationInput", splat_sums: torch.Tensor,
                       want_extras: bool = False) -> ShardGradients:
        m = self.module
        pc, ft = inp.point_cloud, inp.point_cloud_features
        Kmat = m._validate(pc, ft, inp.point_invalid_mask, inp.point_object_id, inp.q_pointcloud_camera,
                           inp.t_pointcloud_camera, inp.camera_info)
        dev = pc.device
        N, M = pc.shape[0], frame.n_points_in_camera
        s = splat_sums.contiguous()
        if s.dtype != torch.float32 or tuple(s.shape) != (M, SPLAT_SUM_FLOATS):
            raise ValueError(f"splat_sums must be float32 ({M},{SPLAT_SUM_FLOATS})")
        flat = torch.empty(N * 59, dtype=torch.float32, device=dev)
        grad_feat, grad_pc = flat[:N * 56].view(N, 56), flat[N * 56:].view(N, 3)
        e = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device=dev)
        n_aff = e(M, dtype=torch.int32) if want_extras else None
        g_uv = e(N, 2) if want_extras else None
        mag = e(N) if want_extras else None
        out = _native.GsBackwardOut(_ptr(grad_pc), _ptr(grad_feat), _ptr(g_uv), _ptr(mag), C.c_void_p(0), _ptr(n_aff),
                                    C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), None,
                                    C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0))
        scene = m._c_scene(pc, ft, inp.point_invalid_mask, inp.point_object_id)
        cam = m._c_camera(inp.q_pointcloud_camera, inp.t_pointcloud_camera, inp.camera_info, Kmat)
        cfg = m._c_config()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_backward_shard(
                frame._context.handle, frame.handle, C.byref(scene), C.byref(cam), C.byref(cfg), _ptr(s),
                int(inp.color_max_sh_band), C.byref(out), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_backward_shard")
        return ShardGradients(grad_pc, grad_feat, flat, n_aff, g_uv, mag)
