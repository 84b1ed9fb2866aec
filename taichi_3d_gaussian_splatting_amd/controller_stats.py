"""On-device statistics of the adaptive density controller.

The reference's GaussianPointAdaptiveController.update() (GaussianPointAdaptiveController.py:130-141)
receives the backward hook payload and performs six indexed `+=` on N-row accumulators.  Those are pure
per-point updates of quantities the backward epilogue already holds in registers, so libgsrast can apply
them in place (`gs_backward_out.controller`): no hook gathers, no torch scatter-adds.  The attribute names are
the controller's own (:114-127); `reset()` is what refinement() does at densification time (:153-164), and
`all_reduce()` sums the statistics over data-parallel ranks before a densification decision.
"""
from dataclasses import dataclass, fields

import torch
import torch.distributed as dist


@dataclass
class ControllerAccumulators:
    accumulated_num_in_camera: torch.Tensor                      # (N,) int32
    accumulated_num_pixels: torch.Tensor                         # (N,) int32
    accumulated_view_space_position_gradients: torch.Tensor      # (N,) float32
    accumulated_view_space_position_gradients_avg: torch.Tensor  # (N,) float32
    accumulated_position_gradients: torch.Tensor                 # (N,3) float32
    accumulated_position_gradients_norm: torch.Tensor            # (N,) float32

    @classmethod
    def zeros(cls, n_points: int, device) -> "ControllerAccumulators":
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=device)
        return cls(z(n_points, dtype=torch.int32), z(n_points, dtype=torch.int32), z(n_points, dtype=torch.float32),
                   z(n_points, dtype=torch.float32), z(n_points, 3, dtype=torch.float32), z(n_points, dtype=torch.float32))

    def reset(self) -> None:
        for f in fields(self):
            getattr(self, f.name).zero_()

    def all_reduce(self, group=None) -> None:
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            for f in fields(self):
                dist.all_reduce(getattr(self, f.name), op=dist.ReduceOp.SUM, group=group)

    def validate(self, n_points: int, device) -> None:
        for f in fields(self):
            t = getattr(self, f.name)
            want = torch.int32 if "num_" in f.name else torch.float32
            shape = (n_points, 3) if f.name == "accumulated_position_gradients" else (n_points,)
            if t.dtype != want or tuple(t.shape) != shape or t.device != torch.device(device) or not t.is_contiguous():
                raise ValueError(f"{f.name} must be a contiguous {want} tensor of shape {shape} on {device}")
