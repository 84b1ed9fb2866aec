"""GPU (-m gpu): the step either side of the operator -- fused L1+SSIM loss and Adam.
pytorch_msssim is not installed here, so SSIM parity is against a float64 torch restatement of its published
algorithm (separable 11-tap Gaussian, sigma 1.5, valid filtering, K1=0.01, K2=0.03, data_range=1): "parity
unpinned" against the package itself.  Adam is checked against torch.optim.Adam."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ssim_ref(X, Y):
    """pytorch_msssim.ssim(X, Y, data_range=1, size_average=True) restated; X, Y (1,3,H,W) float64."""
    coords = torch.arange(11, dtype=X.dtype, device=X.device) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum()).reshape(1, 1, 1, 11).repeat(3, 1, 1, 1)

    def gf(t):
        t = F.conv2d(t, g.transpose(2, 3), groups=3)       # along H first, then W
        return F.conv2d(t, g, groups=3)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = gf(X), gf(Y)
    s1, s2, s12 = gf(X * X) - mu1 ** 2, gf(Y * Y) - mu2 ** 2, gf(X * Y) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * cs
    return ssim_map.flatten(2).mean(-1).mean()


@pytest.mark.parametrize("H,W", [(48, 64), (33, 75), (1088, 1920)])
def test_fused_l1_ssim_matches_restatement(H, W):
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    torch.manual_seed(H)
    gt = torch.rand(3, H, W, device=DEV)
    pred = (gt + 0.2 * torch.randn(3, H, W, device=DEV)).clamp(0, 1).requires_grad_(True)
    lf = LossFunction(LossFunction.LossFunctionConfig(lambda_value=0.2, enable_regularization=False))
    L, L1, LD = lf(pred, gt)
    L.backward()
    p64 = pred.detach().double().requires_grad_(True)
    g64 = gt.double()
    l1_ref = (p64 - g64).abs().mean()
    ld_ref = 1 - ssim_ref(p64[None], g64[None])
    L_ref = 0.8 * l1_ref + 0.2 * ld_ref
    L_ref.backward()
    assert abs(L1.item() - l1_ref.item()) < 1e-6 and abs(LD.item() - ld_ref.item()) < 2e-6 and abs(L.item() - L_ref.item()) < 2e-6
    ref = p64.grad.float()
    err = (pred.grad - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-4, err


@pytest.mark.parametrize("H,W", [(64, 96), (45, 70)])
def test_permuted_rasteriser_layout_and_fused_clamp_equal_the_trainer_lines(H, W):
    """GaussianPointTrainer.py:173-176: image = clamp(image, 0, 1).permute(2, 0, 1) before the loss.  The fused path reads the
    (H,W,3) image in place and clamps on the fly: same value, same gradient (zero where the raw value is outside [0, 1]), and
    the gradient comes back in (H,W,3) memory order.  Also an upstream factor and a second backward through the same node."""
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    torch.manual_seed(W)
    gt = torch.rand(3, H, W, device=DEV)
    raw = (gt.permute(1, 2, 0) + 0.4 * torch.randn(H, W, 3, device=DEV)).contiguous()       # a good part outside [0, 1]
    raw[0, 0, 0], raw[1, 1, 1] = 0.0, 1.0                                                   # the closed ends pass the gradient
    lf = LossFunction(LossFunction.LossFunctionConfig(lambda_value=0.2, enable_regularization=False))
    a = raw.clone().requires_grad_(True)
    La = 3.0 * lf(torch.clamp(a, 0, 1).permute(2, 0, 1), gt)[0]
    La.backward()
    b = raw.clone().requires_grad_(True)
    Lb, L1b, LDb = lf(b.permute(2, 0, 1), gt, clamp_predicted=True)
    (3.0 * Lb).backward(retain_graph=True)
    assert torch.equal(3.0 * Lb, La)
    assert b.grad.is_contiguous() and torch.equal(b.grad, a.grad)
    outside = (raw < 0) | (raw > 1)
    assert outside.float().mean() > 0.2 and not b.grad[outside].any() and b.grad[0, 0, 0] != 0 and b.grad[1, 1, 1] != 0
    first = b.grad.clone()
    (3.0 * Lb).backward()
    assert torch.equal(b.grad, 2 * first)
    # float64 restatement of the same lines
    p64 = raw.double().requires_grad_(True)
    x64 = torch.clamp(p64, 0, 1).permute(2, 0, 1)
    ref = 0.8 * (x64 - gt.double()).abs().mean() + 0.2 * (1 - ssim_ref(x64[None], gt.double()[None]))
    ref.backward()
    assert abs(Lb.item() - ref.item()) < 2e-6
    assert (first / 3.0 - p64.grad.float()).abs().max().item() / p64.grad.abs().max().item() < 1e-4


def test_batched_images_keep_their_own_maps_and_the_one_call_entry_still_answers():
    """B > 1 (LossFunction.py:21-33): two forwards before the backwards -- each node owns its derivative maps.  And the C ABI's
    one-call form gs_loss_l1_ssim (contiguous images, upstream 1) gives the same numbers as the two-call form."""
    import ctypes as C
    from taichi_3d_gaussian_splatting_amd import _native
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    torch.manual_seed(5)
    H, W = 40, 52
    gt = torch.rand(2, 3, H, W, device=DEV)
    pred = (gt + 0.1 * torch.randn_like(gt)).clamp(0, 1)
    lf = LossFunction(LossFunction.LossFunctionConfig(lambda_value=0.2, enable_regularization=False))
    pb = pred.clone().requires_grad_(True)
    Lb = lf(pb, gt)[0]
    Lb.backward()
    singles, grads = [], []
    for i in range(2):
        pi = pred[i].clone().requires_grad_(True)
        Li = lf(pi, gt[i])[0]
        Li.backward()
        singles.append(Li); grads.append(pi.grad)
    assert torch.allclose(Lb, (singles[0] + singles[1]) / 2, rtol=1e-6)
    assert torch.allclose(pb.grad, torch.stack(grads) / 2, rtol=1e-5, atol=1e-12)
    terms = torch.empty(3, device=DEV)
    g = torch.empty(3, H, W, device=DEV)
    x, y = pred[1].contiguous(), gt[1].contiguous()
    _native.check(_native.lib().gs_loss_l1_ssim(_native.shared_ctx(0), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), H, W, 0.2,
                                                C.c_void_p(terms.data_ptr()), C.c_void_p(g.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_loss_l1_ssim")
    assert torch.equal(terms[0], singles[1].detach()) and torch.equal(g, grads[1])


def test_regulariser_and_reference_call_shapes():
    """(1,3,H,W) inputs and the scale regulariser of LossFunction.py:40-51."""
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    gt = torch.rand(1, 3, 32, 48, device=DEV)
    pred = torch.rand(1, 3, 32, 48, device=DEV, requires_grad=True)
    feat = torch.randn(100, 56, device=DEV, requires_grad=True)
    mask = torch.zeros(100, dtype=torch.int8, device=DEV)
    mask[::3] = 1
    lf = LossFunction(LossFunction.LossFunctionConfig())
    L, L1, LD = lf(pred, gt, point_invalid_mask=mask, pointcloud_features=feat)
    reg = torch.norm(torch.exp(feat[mask == 0, 4:7]), dim=1).mean()
    assert torch.allclose(L, 0.8 * L1 + 0.2 * LD + 2 * reg, rtol=1e-6)
    L.backward()
    assert pred.grad is not None and feat.grad is not None and not feat.grad[mask == 1].any()
    f64 = feat.detach().double().requires_grad_(True)
    (2 * torch.norm(torch.exp(f64[mask == 0, 4:7]), dim=1).mean()).backward()
    assert torch.allclose(feat.grad, f64.grad.float(), rtol=1e-5, atol=1e-9)
    assert not feat.grad[:, :4].any() and not feat.grad[:, 7:].any()


def test_fused_adam_matches_torch_adam():
    from taichi_3d_gaussian_splatting_amd.optim import FusedAdam
    torch.manual_seed(0)
    p0 = torch.randn(5000, 56, device=DEV)
    pa, pb = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    ref = torch.optim.Adam([pa], lr=1e-3, betas=(0.9, 0.999))
    fused = FusedAdam([pb], lr=1e-3, betas=(0.9, 0.999))
    for it in range(6):
        g = torch.randn_like(p0) * (1 + it)
        pa.grad, pb.grad = g.clone(), g.clone()
        ref.step()
        fused.step()
        if it == 3:
            ref.param_groups[0]["lr"] *= 0.97
            fused.lr *= 0.97
    assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7), (pa - pb).abs().max().item()


def test_training_step_with_fused_loss_and_adam_reduces_loss():
    """Operator + fused loss + fused Adam wired together as GaussianPointTrainer.py:160-184 does."""
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    from taichi_3d_gaussian_splatting_amd.optim import FusedAdam
    from taichi_3d_gaussian_splatting_amd.synthetic import synth, view_pose
    s = synth(2000, 128, 96, 0.1, seed=9)
    q, t = view_pose()
    pc = torch.tensor(s.point_cloud, device=DEV, requires_grad=True)
    feat = torch.tensor(s.point_cloud_features, device=DEV, requires_grad=True)
    mask, obj = torch.tensor(s.point_invalid_mask, device=DEV), torch.tensor(s.point_object_id, device=DEV)
    rast = Rast(Rast.GaussianPointCloudRasterisationConfig())
    lf = LossFunction(LossFunction.LossFunctionConfig())
    opt_f, opt_p = FusedAdam([feat], lr=1e-3), FusedAdam([pc], lr=1e-5)
    target = torch.rand(3, 96, 128, device=DEV)
    losses = []
    for it in range(40):
        opt_f.zero_grad(); opt_p.zero_grad()
        img, _, _ = rast(Rast.GaussianPointCloudRasterisationInput(
            point_cloud=pc, point_cloud_features=feat, point_object_id=obj, point_invalid_mask=mask,
            camera_info=CameraInfo(torch.tensor(s.camera_intrinsics, device=DEV), 96, 128, 0),
            q_pointcloud_camera=torch.tensor(q, device=DEV), t_pointcloud_camera=torch.tensor(t, device=DEV), color_max_sh_band=3))
        img = torch.clamp(img, 0, 1).permute(2, 0, 1)
        L, L1, LD = lf(img, target, point_invalid_mask=mask, pointcloud_features=feat)
        L.backward()
        opt_f.step(); opt_p.step()
        losses.append(L.item())
    assert losses[-1] < losses[0]
