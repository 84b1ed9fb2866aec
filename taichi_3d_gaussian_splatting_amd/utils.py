"""Host-side pose helpers of the operator boundary (torch, any device).

Same names and argument meaning as the helpers the reference's callers use around the
rasteriser (taichi_3d_gaussian_splatting/utils.py:396-492, 596-632); quaternions are
(x, y, z, w).  The rasteriser itself does not call these: the pose inversion the reference
performs with inverse_SE3_qt_torch at GaussianPointCloudRasterisation.py:845 runs inside
libgsrast.so (every k_filter block derives the pose records itself).  They are here so callers that build poses keep working.
"""
from typing import Tuple

import torch


def quaternion_conjugate_torch(q: torch.Tensor) -> torch.Tensor:
    sign = torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=q.dtype, device=q.device)
    return q * sign


def quaternion_multiply_torch(q0: torch.Tensor, q1: torch.Tensor) -> torch.Tensor:
    """Hamilton product q0 * q1."""
    v0, w0 = q0[..., :3], q0[..., 3:4]
    v1, w1 = q1[..., :3], q1[..., 3:4]
    v = w0 * v1 + w1 * v0 + torch.cross(v0, v1, dim=-1)
    w = w0 * w1 - (v0 * v1).sum(-1, keepdim=True)
    return torch.cat([v, w], dim=-1)


def quaternion_rotate_torch(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Rotate v by q; q is normalised first (reference utils.py:416-423)."""
    q = q / q.norm(dim=-1, keepdim=True)
    pure = torch.cat([v, torch.zeros_like(v[..., :1])], dim=-1)
    return quaternion_multiply_torch(quaternion_multiply_torch(q, pure), quaternion_conjugate_torch(q))[..., :3]


def inverse_SE3_qt_torch(q: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(q, t) of T  ->  (q, t) of T^-1.  The returned quaternion is the plain conjugate (not renormalised)."""
    q_inv = quaternion_conjugate_torch(q)
    return q_inv, -quaternion_rotate_torch(q_inv, t)


def quaternion_to_rotation_matrix_torch(q: torch.Tensor) -> torch.Tensor:
    x, y, z, w = q.unbind(-1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=-1).reshape(*q.shape[:-1], 3, 3)


def rotation_matrix_to_quaternion_torch(R: torch.Tensor) -> torch.Tensor:
    """(B,3,3) -> (B,4) xyzw.  Picks the largest of (w, x, y, z) as pivot, like reference utils.py:435-483."""
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    trace = m00 + m11 + m22
    case_w = trace > 0
    case_x = (~case_w) & (m00 > m11) & (m00 > m22)
    case_y = (~case_w) & (~case_x) & (m11 > m22)
    # four candidate solutions, each valid where its pivot is large
    def cand(pivot_sq):
        return 2.0 * torch.sqrt(torch.clamp(pivot_sq, min=1e-30))
    sw, sx = cand(1 + trace), cand(1 + m00 - m11 - m22)
    sy, sz = cand(1 + m11 - m00 - m22), cand(1 + m22 - m00 - m11)
    a, b, c = R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]
    d, e, f = R[..., 0, 1] + R[..., 1, 0], R[..., 0, 2] + R[..., 2, 0], R[..., 1, 2] + R[..., 2, 1]
    qw = torch.stack([a / sw, b / sw, c / sw, 0.25 * sw], -1)
    qx = torch.stack([0.25 * sx, d / sx, e / sx, a / sx], -1)
    qy = torch.stack([d / sy, 0.25 * sy, f / sy, b / sy], -1)
    qz = torch.stack([e / sz, f / sz, 0.25 * sz, c / sz], -1)
    out = torch.where(case_w[..., None], qw, torch.where(case_x[..., None], qx, torch.where(case_y[..., None], qy, qz)))
    return out


def SE3_to_quaternion_and_translation_torch(transform: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return rotation_matrix_to_quaternion_torch(transform[..., :3, :3]), transform[..., :3, 3]
