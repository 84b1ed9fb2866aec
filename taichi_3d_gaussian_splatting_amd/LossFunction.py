"""LossFunction -- same class, config and return triple as the reference's LossFunction.py:8-51,
   L = (1 - lambda) L1 + lambda (1 - SSIM) [+ regularization_weight * mean ||exp(s)||],
with the L1 + SSIM part (value and gradient) computed by libgsrast in three HIP launches
(gs_loss_l1_ssim) instead of pytorch_msssim's conv2d chain + autograd.  SSIM follows
pytorch_msssim.ssim(data_range=1, size_average=True): 11-tap Gaussian window (sigma 1.5), valid filtering,
K1 = 0.01, K2 = 0.03.  The scale regulariser stays in torch (a handful of ops on an (N,3) slice)."""
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
        """predicted_image / ground_truth_image: (B=1, C, H, W) or (C, H, W).  Returns (L, L1, LD_SSIM)."""
        if predicted_image.dim() == 4:
            if predicted_image.shape[0] != 1:
                raise ValueError("the fused loss handles one image per call (the reference trains with batch_size=None)")
            predicted_image = predicted_image[0]
        if ground_truth_image.dim() == 4:
            ground_truth_image = ground_truth_image[0]
        L, terms = _L1SSIM.apply(predicted_image, ground_truth_image, self.config.lambda_value)
        L1, LD_SSIM = terms[1], terms[2]
        if pointcloud_features is not None and self.config.enable_regularization:
            L = L + self.config.regularization_weight * self._regularization_loss(point_invalid_mask, pointcloud_features)
        return L, L1, LD_SSIM

    def _regularization_loss(self, point_invalid_mask, pointcloud_features):
        s = pointcloud_features[point_invalid_mask == 0, 4:7]           # LossFunction.py:48-50
        return torch.norm(torch.exp(s), dim=1).mean()
