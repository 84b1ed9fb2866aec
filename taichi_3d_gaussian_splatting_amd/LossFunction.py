"""LossFunction -- same class, config and return triple as the reference's LossFunction.py:8-51,
   L = (1 - lambda) L1 + lambda (1 - SSIM) [+ regularization_weight * mean ||exp(s)||],
with the L1 + SSIM part (value and gradient) computed by libgsrast in three HIP launches
(gs_loss_l1_ssim) instead of pytorch_msssim's conv2d chain + autograd.  SSIM follows
pytorch_msssim.ssim(data_range=1, size_average=True): 11-tap Gaussian window (sigma 1.5), valid filtering,
K1 = 0.01, K2 = 0.03.  The scale regulariser is fused as well (gs_scale_regulariser[_grad]): in torch its
boolean-mask indexing and the sort-based index_put of its backward cost more than the rasteriser's backward."""
import ctypes as C
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _native

try:
    from dataclass_wizard import YAMLWizard as _ConfigBase
except Exception:  # pragma: no cover
    class _ConfigBase:
        pass


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predicted, ground_truth, lambda_value):
        if predicted.dim() != 3 or predicted.shape[0] != 3 or predicted.shape != ground_truth.shape:
            raise ValueError("predicted_image and ground_truth_image must both be (3,H,W)")
        if predicted.dtype != torch.float32 or ground_truth.dtype != torch.float32 or not predicted.is_cuda:
            raise TypeError("images must be float32 tensors on the GPU")
        x, y = predicted.contiguous(), ground_truth.contiguous()
        dev = x.device
        terms = torch.empty(3, dtype=torch.float32, device=dev)
        need_grad = bool(ctx.needs_input_grad[0])
        grad = torch.empty_like(x) if need_grad else None
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_loss_l1_ssim(
                _native.shared_ctx(idx), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.shape[1], x.shape[2],
                float(lambda_value), C.c_void_p(terms.data_ptr()), C.c_void_p(grad.data_ptr() if need_grad else 0),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_loss_l1_ssim")
        ctx.save_for_backward(grad) if need_grad else None
        ctx.mark_non_differentiable(terms)
        return terms[0], terms

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None


class _ScaleRegulariser(torch.autograd.Function):
    """mean over valid points of ||exp(s)||_2 (LossFunction.py:40-51) without boolean-mask indexing."""

    @staticmethod
    def forward(ctx, features, invalid_mask):
        if features.dtype != torch.float32 or not features.is_cuda or not features.is_contiguous() or features.shape[1] != 56:
            raise TypeError("pointcloud_features must be a contiguous float32 (N,56) GPU tensor")
        mask = invalid_mask if invalid_mask.dtype == torch.int8 else invalid_mask.to(torch.int8)
        dev = features.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_scale_regulariser(
                _native.shared_ctx(idx), C.c_void_p(features.data_ptr()), C.c_void_p(mask.data_ptr()), features.shape[0],
                C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_scale_regulariser")
        ctx.save_for_backward(features, mask, out)
        return out[0]

    @staticmethod
    def backward(ctx, upstream):
        features, mask, out = ctx.saved_tensors
        dev = features.device
        grad = torch.empty_like(features)
        up = upstream.reshape(1).to(torch.float32).contiguous()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_scale_regulariser_grad(
                _native.shared_ctx(idx), C.c_void_p(features.data_ptr()), C.c_void_p(mask.data_ptr()), features.shape[0],
                C.c_void_p(out.data_ptr()), C.c_void_p(up.data_ptr()), C.c_void_p(grad.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_scale_regulariser_grad")
        return grad, None


class LossFunction(nn.Module):
    @dataclass
    class LossFunctionConfig(_ConfigBase):
        lambda_value: float = 0.2
        enable_regularization: bool = True
        regularization_weight: float = 2

    def __init__(self, config: "LossFunction.LossFunctionConfig"):
        super().__init__()
        self.config = config

    def forward(self, predicted_image, ground_truth_image, point_invalid_mask=None, pointcloud_features=None):
        """predicted_image / ground_truth_image: (B, C, H, W) or (C, H, W), C = 3.  Returns (L, L1, LD_SSIM).
        With B > 1 (LossFunction.py:21-33 accepts it; the reference trainer uses batch_size=None) the images are
        equally sized, so the batch means of L1 and of size_average=True SSIM are the means of the per-image values:
        the fused kernel runs once per image."""
        if predicted_image.dim() == 3:
            predicted_image = predicted_image.unsqueeze(0)
        if ground_truth_image.dim() == 3:
            ground_truth_image = ground_truth_image.unsqueeze(0)
        if predicted_image.shape != ground_truth_image.shape:
            raise ValueError("predicted_image and ground_truth_image must have the same shape")
        per_image = [_L1SSIM.apply(predicted_image[b], ground_truth_image[b], self.config.lambda_value)
                     for b in range(predicted_image.shape[0])]
        if len(per_image) == 1:
            L, terms = per_image[0]
        else:
            L = torch.stack([p[0] for p in per_image]).mean()
            terms = torch.stack([p[1] for p in per_image]).mean(dim=0)
        L1, LD_SSIM = terms[1], terms[2]
        if pointcloud_features is not None and self.config.enable_regularization:
            L = L + self.config.regularization_weight * self._regularization_loss(point_invalid_mask, pointcloud_features)
        return L, L1, LD_SSIM

    def _regularization_loss(self, point_invalid_mask, pointcloud_features):
        if pointcloud_features.is_cuda and pointcloud_features.dtype == torch.float32 and pointcloud_features.is_contiguous() \
                and pointcloud_features.dim() == 2 and pointcloud_features.shape[1] == 56:
            return _ScaleRegulariser.apply(pointcloud_features, point_invalid_mask)
        s = pointcloud_features[point_invalid_mask == 0, 4:7]           # LossFunction.py:48-50
        return torch.norm(torch.exp(s), dim=1).mean()
