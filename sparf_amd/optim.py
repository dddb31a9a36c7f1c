"""Optimiser for the renderer's networks: per network, gradient-norm
clipping + Adam in two launches on the flat gradient buffer the HIP backward produces.

Mirrors what the reference trainer does around `Graph`
(/root/reference/source/training/nerf_trainer.py:181-185 `torch.optim.Adam(net.nerf.parameters(),
lr, betas=(0.9, 0.999))` + a second param group for `nerf_fine`; clipping by norm per network
component, /root/reference/source/training/base.py:96-97 with `nerf_gradient_clipping`).  It is a
`torch.optim.Optimizer`, so LR schedulers (`ExponentialLR`, nerf_trainer.py:196-203) work on
its param groups unchanged."""
import ctypes

import torch

from . import lib as L
from .parallel import _group_grads


class FusedAdam(torch.optim.Optimizer):
    """FusedAdam([graph.nerf, graph.nerf_fine], lr=5e-4, max_grad_norm=0.1).

    One param group per network (20 tensors W0,b0,...,W9,b9 in `NeRF.hip_params()` order; the
    scalar `progress` is never optimised -- it receives no gradient in the reference either).
    `max_grad_norm=None` disables clipping.  `last_grad_norms` holds the pre-clip norms
    (device tensors, no host sync).  `device_step=True` keeps the update count on the device
    (sparf_adam_step_dev): required when the training step is captured in a hipGraph."""

    def __init__(self, nets, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=None, device_step=False):
        self.device_step = bool(device_step)
        nets = list(nets)
        self._nets = nets
        groups = [dict(params=list(n.hip_params())) for n in nets]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm))
        self.last_grad_norms = [None] * len(groups)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.load()
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if all(p.grad is None for p in params):
                continue
            dev = params[0].device
            L.require_gpu(dev)
            st = self.state[params[0]]
            if not st:
                n = sum(p.numel() for p in params)
                st.update(step=0, exp_avg=torch.zeros(n, device=dev), exp_avg_sq=torch.zeros(n, device=dev),
                          ws=torch.empty(int(lib.sparf_adam_workspace_floats()), device=dev), norm=torch.zeros(1, device=dev),
                          step_dev=torch.zeros(1, dtype=torch.int32, device=dev))
            if "step_dev" not in st:      # a state dict saved before the device-side counter existed, or without the key
                st["step_dev"] = torch.full((1,), int(st["step"]), dtype=torch.int32, device=dev)
            st["step"] += 1
            grads = [p.grad for p in params]
            flats, loose = _group_grads(grads) if all(g is not None for g in grads) else ([], grads)
            if len(flats) == 1 and not loose and flats[0].numel() == st["exp_avg"].numel():
                flat = flats[0]                                   # the HIP backward's own buffer
            else:                                                 # gradients from elsewhere: pack them once
                flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1).float() for p, g in zip(params, grads)])
            for p in params:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.SparfError("FusedAdam needs contiguous fp32 parameters")
            arr = (ctypes.c_void_p * 20)(*[p.data_ptr() for p in params])
            b1, b2 = group["betas"]
            mg = group["max_grad_norm"]
            with L.on(dev):
                if self.device_step:      # (the host-side count above is then only informative: replays of a captured step do not pass here)
                    L.check(lib.sparf_adam_step_dev(arr, L.ptr(flat), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.ptr(st["ws"]), L.ptr(st["norm"]),
                                                    float(group["lr"]), float(b1), float(b2), float(group["eps"]), L.ptr(st["step_dev"]),
                                                    float(mg) if mg else 0.0, L.stream_ptr(dev)), "sparf_adam_step_dev")
                else:
                    L.check(lib.sparf_adam_step(arr, L.ptr(flat), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.ptr(st["ws"]), L.ptr(st["norm"]),
                                                float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]),
                                                float(mg) if mg else 0.0, L.stream_ptr(dev)), "sparf_adam_step")
            self._nets[gi].weights_changed()    # raw-pointer update: torch's version counters did not move
            self.last_grad_norms[gi] = st["norm"] if mg else None
        return loss

    def _sync_steps(self):
        """device_step mode: the authoritative update count lives on the device (hipGraph replays do not pass through step());
        bring the host-side `step` up to it before the state leaves the optimiser"""
        if not self.device_step:
            return
        for group in self.param_groups:
            st = self.state.get(group["params"][0])
            if st and "step_dev" in st:
                st["step"] = int(st["step_dev"].item())

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            st = self.state.get(group["params"][0])
            if st and "step" in st:          # one count, whichever mode saved it: the device counter restarts from the host count
                dev = group["params"][0].device
                st["step"] = int(st["step"])
                st["step_dev"] = torch.full((1,), st["step"], dtype=torch.int32, device=dev)
