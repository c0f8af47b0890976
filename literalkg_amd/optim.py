"""Fused Adam for the hot path's parameters.  Same hyper-parameters, update rule and
``state_dict`` layout ('step', 'exp_avg', 'exp_avg_sq') as ``torch.optim.Adam`` so optimizer checkpoints
interchange; each parameter is updated by ONE HIP kernel pass instead of the multi-kernel foreach path."""
from __future__ import annotations

import torch

from . import _native as N
from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.is_sparse or p.grad.is_sparse:
                    raise RuntimeError("literalkg_amd.optim.Adam handles dense parameters only "
                                       "(A_in carries no gradient)")
                ops._need_gpu(p)
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)           # host scalar like torch's default (capturable=False)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("parameters must be contiguous")
                N.call("lkg_adam_step_f32", p.numel(), N.ptr(p), N.ptr(g), N.ptr(st["exp_avg"]),
                       N.ptr(st["exp_avg_sq"]), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                       float(group["weight_decay"]), int(st["step"]), ops._stream())
                # the kernel wrote p through its raw pointer: tell autograd's version counter, as torch.optim.Adam's in-place
                # ops do -- everything keyed on a parameter's version (the inference heads' kept table, a tensor saved for a
                # backward) sees the step
                torch.autograd.graph.increment_version(p)
        return loss
