"""LossFunction -- same class, config and return triple as the reference's LossFunction.py:8-51,
   L = (1 - lambda) L1 + lambda (1 - SSIM) [+ regularization_weight * mean ||exp(s)||],
with the L1 + SSIM part (value and gradient) computed by libgsrast in three HIP launches
(gs_loss_l1_ssim_forward / _backward) instead of pytorch_msssim's conv2d chain + autograd.  SSIM follows
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
    """Forward: gs_loss_l1_ssim_forward (value; the derivative maps stay in a tensor this node owns).  Backward:
    gs_loss_l1_ssim_backward, scaled by the incoming gradient on the device.  The predicted image is read through its strides:
    the rasteriser's (H,W,3) output viewed as (3,H,W) by permute(2,0,1) needs no copy, and its gradient is written in the
    same layout."""

    @staticmethod
    def forward(ctx, predicted, ground_truth, lambda_value, clamp_predicted):
        if predicted.dim() != 3 or predicted.shape[0] != 3 or predicted.shape != ground_truth.shape:
            raise ValueError("predicted_image and ground_truth_image must both be (3,H,W)")
        if predicted.dtype != torch.float32 or ground_truth.dtype != torch.float32 or not predicted.is_cuda:
            raise TypeError("images must be float32 tensors on the GPU")
        dev = predicted.device
        H, W = int(predicted.shape[1]), int(predicted.shape[2])
        L = _native.lib()
        terms = torch.empty(3, dtype=torch.float32, device=dev)
        maps = torch.empty(L.gs_loss_maps_floats(H, W), dtype=torch.float32, device=dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        x, y = _native.GsLossImage.of(predicted), _native.GsLossImage.of(ground_truth)
        with torch.cuda.device(dev):
            _native.check(L.gs_loss_l1_ssim_forward(
                _native.shared_ctx(idx), C.byref(x), C.byref(y), H, W, int(bool(clamp_predicted)), float(lambda_value),
                C.c_void_p(maps.data_ptr()), C.c_void_p(terms.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_loss_l1_ssim_forward")
        ctx.save_for_backward(predicted, ground_truth, maps)
        ctx.lambda_value, ctx.clamp_predicted = float(lambda_value), int(bool(clamp_predicted))
        ctx.mark_non_differentiable(terms)
        return terms[0], terms

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        predicted, ground_truth, maps = ctx.saved_tensors
        dev = predicted.device
        H, W = int(predicted.shape[1]), int(predicted.shape[2])
        # the gradient in the memory layout of the image it belongs to (a permuted view of an (H,W,3) buffer for a permuted input)
        # (preserve_format: the input's strides when it is dense and non-overlapping, contiguous otherwise)
        grad = torch.empty_like(predicted, memory_format=torch.preserve_format)
        up = grad_loss.reshape(1).to(torch.float32).contiguous()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        x, y, g = _native.GsLossImage.of(predicted), _native.GsLossImage.of(ground_truth), _native.GsLossImage.of(grad)
        with torch.cuda.device(dev):
            _native.check(_native.lib().gs_loss_l1_ssim_backward(
                _native.shared_ctx(idx), C.byref(x), C.byref(y), H, W, ctx.clamp_predicted, ctx.lambda_value,
                C.c_void_p(maps.data_ptr()), C.c_void_p(up.data_ptr()), C.byref(g),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_loss_l1_ssim_backward")
        return grad, None, None, None


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

    def forward(self, predicted_image, ground_truth_image, point_invalid_mask=None, pointcloud_features=None, clamp_predicted=False):
        """predicted_image / ground_truth_image: (B, C, H, W) or (C, H, W), C = 3, any strides (the rasteriser's (H,W,3)
        output under permute(2,0,1) is read in place).  Returns (L, L1, LD_SSIM).
        clamp_predicted=True (not in the reference's signature) applies the trainer's torch.clamp(image, 0, 1)
        (GaussianPointTrainer.py:173) inside the kernels: loss_function(image_pred.permute(2, 0, 1), image_gt, ...,
        clamp_predicted=True) equals loss_function(torch.clamp(image_pred, 0, 1).permute(2, 0, 1), image_gt, ...) in value and
        gradient, without the clamp's kernels and copies.
        With B > 1 (LossFunction.py:21-33 accepts it; the reference trainer uses batch_size=None) the images are
        equally sized, so the batch means of L1 and of size_average=True SSIM are the means of the per-image values:
        the fused kernel runs once per image."""
        # (no unsqueeze / [b] round trip for a single image: the select's backward is a zero fill and a copy of the whole image)
        if predicted_image.dim() == 4 and predicted_image.shape[0] == 1:
            predicted_image = predicted_image.squeeze(0)
        if ground_truth_image.dim() == 4 and ground_truth_image.shape[0] == 1:
            ground_truth_image = ground_truth_image.squeeze(0)
        if predicted_image.shape != ground_truth_image.shape:
            raise ValueError("predicted_image and ground_truth_image must have the same shape")
        if predicted_image.dim() == 3:
            L, terms = _L1SSIM.apply(predicted_image, ground_truth_image, self.config.lambda_value, clamp_predicted)
        else:
            per_image = [_L1SSIM.apply(p, g, self.config.lambda_value, clamp_predicted)
                         for p, g in zip(predicted_image.unbind(0), ground_truth_image.unbind(0))]
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
