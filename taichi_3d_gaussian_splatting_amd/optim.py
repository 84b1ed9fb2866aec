"""FusedAdam -- torch.optim.Adam(params, lr, betas=(0.9, 0.999)) as the reference trainer uses it for the
feature and position tensors (GaussianPointTrainer.py:131-134, 183-184), one HIP launch per tensor
(gs_adam_step) instead of torch's multi-kernel foreach path.  `lr` is a plain attribute so an exponential
decay (GaussianPointTrainer.py:136-137,191-192) is `opt.lr *= rate`."""
import ctypes as C
from typing import Iterable

import torch

from . import _native


class FusedAdam:
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = list(params)
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise TypeError("FusedAdam handles contiguous float32 GPU tensors")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.state = [dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p in self.params]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        L = _native.lib()
        for p, st in zip(self.params, self.state):
            if p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            st["step"] += 1
            dev = p.device
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            with torch.cuda.device(dev):
                _native.check(L.gs_adam_step(_native.shared_ctx(idx), C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()),
                                             C.c_void_p(st["exp_avg"].data_ptr()), C.c_void_p(st["exp_avg_sq"].data_ptr()),
                                             p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, st["step"],
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gs_adam_step")
