"""Float64 torch.autograd restatement of the rasteriser for TINY scenes (tests only).

Purpose: pin the oracle's hand-written backward (oracle/gs_oracle.c, restating
GaussianPointCloudRasterisation.py:488-772) against automatic differentiation,
because the reference ships no test for loop-1 accumulation, the SH gradient,
the grad factors or loop 2 as a whole (SURVEY 8c "parity unpinned" list).

The integer structure (visible ids, sorted per-tile lists, tile ranges) is taken
from the oracle; every floating-point quantity is recomputed here in float64
with autograd.  The reference's analytic backward deliberately differs from the
true derivative in a few places; the same stops are placed here with .detach():
  * rescale is a constant                          (UTIL:347 "known caveat")
  * no gradient through the 0.99 clamp test: the clamped value is used in the
    formulas but d alpha / d (g * opacity) = 1     (RAST:634-662)
  * Sigma' does not feed xyz (J is a constant)      (RAST:757-761, GP3D:237-331)
  * the SH view direction does not feed xyz         (RAST:749-756)
  * q is the already-normalised quaternion          (RAST:264-266)
  * depth / count outputs carry no gradient         (RAST:1026)
"""
import numpy as np
import torch

ALPHA_EPS = 1.0 / 255.0
F64 = torch.float64


def quat_to_R(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    xx, yy, zz, xy, xz, yz, wx, wy, wz = x * x, y * y, z * z, x * y, x * z, y * z, w * x, w * y, w * z
    R = torch.stack([
        torch.stack([1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy)], -1),
        torch.stack([2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx)], -1),
        torch.stack([2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (xx + yy)], -1)], -2)
    return R


def sh16(d):
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * x * y, -1.0925484305920792 * y * z,
        0.94617469575755997 * z * z - 0.31539156525251999,
        -1.0925484305920792 * x * z, 0.54627421529603959 * x * x - 0.54627421529603959 * y * y,
        0.59004358992664352 * y * (-3.0 * x * x + y * y), 2.8906114426405538 * x * y * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z * z), 0.3731763325901154 * z * (5.0 * z * z - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z * z), 1.4453057213202769 * z * (x * x - y * y),
        0.59004358992664352 * x * (-x * x + 3.0 * y * y)], -1)


def render(point_cloud, features, q_pc, t_pc, Kmat, H, W, fwd, object_id=None):
    """Differentiable render of the image.  `fwd` is an oracle.Forward (ints only are used).
    point_cloud (N,3) and features (N,56) are float64 leaf tensors; features[:, :4] must hold
    the normalised quaternions (i.e. the oracle's features_after)."""
    ids = torch.as_tensor(fwd.point_id_in_camera_list.astype(np.int64))
    obj = torch.zeros(point_cloud.shape[0], dtype=torch.long) if object_id is None else torch.as_tensor(object_id).long()
    q_pc = torch.as_tensor(q_pc, dtype=F64).reshape(-1, 4)
    t_pc = torch.as_tensor(t_pc, dtype=F64).reshape(-1, 3)
    Kmat = torch.as_tensor(Kmat, dtype=F64)
    # pose: inverse of (q_pc, t_pc), UTIL:426-432 (conjugate is NOT renormalised for R)
    q_cp = torch.cat([-q_pc[:, :3], q_pc[:, 3:]], -1)
    R_cp_unit = quat_to_R(q_cp / q_cp.norm(dim=-1, keepdim=True))
    t_cp = -(R_cp_unit @ t_pc[..., None])[..., 0]
    Wm = quat_to_R(q_cp)[obj[ids]]                  # (M,3,3) rotation used by the kernels
    tt = t_cp[obj[ids]]
    xyz = point_cloud[ids]
    f = features[ids]
    pcam = (Wm @ xyz[..., None])[..., 0] + tt
    uv1 = (Kmat @ pcam[..., None])[..., 0]
    uv = uv1[:, :2] / pcam[:, 2:3]
    uv.retain_grad()
    # covariance (J constant w.r.t. xyz)
    pc_d = pcam.detach()
    fx, fy = Kmat[0, 0], Kmat[1, 1]
    zero = torch.zeros_like(pc_d[:, 0])
    J = torch.stack([torch.stack([fx / pc_d[:, 2], zero, -fx * pc_d[:, 0] / pc_d[:, 2] ** 2], -1),
                     torch.stack([zero, fy / pc_d[:, 2], -fy * pc_d[:, 1] / pc_d[:, 2] ** 2], -1)], -2)
    R = quat_to_R(f[:, 0:4])
    S = torch.diag_embed(torch.exp(f[:, 4:7]))
    Sigma = R @ S @ S.transpose(-1, -2) @ R.transpose(-1, -2)
    U = J @ Wm
    cov = U @ Sigma @ U.transpose(-1, -2)
    cov_b = cov + 0.3 * torch.eye(2, dtype=F64)
    det_pre = cov[:, 0, 0] * cov[:, 1, 1] - cov[:, 0, 1] * cov[:, 1, 0]
    det = cov_b[:, 0, 0] * cov_b[:, 1, 1] - cov_b[:, 0, 1] * cov_b[:, 1, 0]
    rescale = torch.sqrt(torch.clamp(det_pre / det, min=0.0)).detach()
    conic_a, conic_b, conic_c = cov_b[:, 1, 1] / det, -cov_b[:, 0, 1] / det, cov_b[:, 0, 0] / det
    opacity = torch.sigmoid(f[:, 7])
    # colour: direction constant w.r.t. xyz; ray origin = camera centre in point-cloud frame
    Rt = quat_to_R(q_cp).transpose(-1, -2)
    origin = -(Rt @ t_cp[..., None])[..., 0]
    d = (xyz - origin[obj[ids]]).detach()
    Y = sh16(d)
    color = torch.sigmoid(torch.stack([(f[:, 8:24] * Y).sum(-1), (f[:, 24:40] * Y).sum(-1), (f[:, 40:56] * Y).sum(-1)], -1))

    image = torch.zeros(H, W, 3, dtype=F64)
    tiles_x = (W + 15) // 16                       # = W // 16 at the reference's sizes; partial edge tiles are an extension
    lst = fwd.point_offset_with_sort_key
    yy, xx = torch.meshgrid(torch.arange(16, dtype=F64), torch.arange(16, dtype=F64), indexing="ij")
    for tile in range(tiles_x * ((H + 15) // 16)):
        s, e = int(fwd.tile_points_start[tile]), int(fwd.tile_points_end[tile])
        if e <= s:
            continue
        tu, tv = tile % tiles_x, tile // tiles_x
        px = (xx + tu * 16 + 0.5).reshape(-1)
        py = (yy + tv * 16 + 0.5).reshape(-1)
        T = torch.ones(256, dtype=F64)
        C = torch.zeros(256, 3, dtype=F64)
        alive = torch.ones(256, dtype=torch.bool)
        for idx in range(s, e):
            p = int(lst[idx])
            dx, dy = px - uv[p, 0], py - uv[p, 1]
            g = torch.exp(-0.5 * (dx * dx * conic_a[p] + dy * dy * conic_c[p]) - dx * dy * conic_b[p]) * rescale[p]
            a = g * opacity[p]
            use = alive & (a.detach() >= ALPHA_EPS)
            a_c = a + (torch.clamp(a, max=0.99) - a).detach()        # clamp value, straight-through gradient
            nT = T * (1 - a_c)
            sat = use & (nT.detach() < 1e-4)
            alive = alive & ~sat
            use = use & ~sat
            w = torch.where(use, a_c * T, torch.zeros_like(T))
            C = C + w[:, None] * color[p][None, :]
            T = torch.where(use, nT, T)
        hh, ww = min(16, H - tv * 16), min(16, W - tu * 16)
        image[tv * 16:tv * 16 + hh, tu * 16:tu * 16 + ww, :] = C.reshape(16, 16, 3)[:hh, :ww]
    return image, {"uv": uv, "color": color, "opacity": opacity, "cov": cov}
