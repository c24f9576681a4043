"""Drop-in for `arm.optim.lamb.Lamb` (arm/optim/lamb.py:26-127): same constructor, `step()`, and per-parameter state keys
(`step, exp_avg, exp_avg_sq, weight_norm, adam_norm, trust_ratio` - the checkpoint contract of utils.py:289), with the
whole step done by two multi-tensor HIP kernels over all parameters (`semabs_lamb_step`, csrc/optim.hip)."""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import _lib

_CHUNK = 16384


class Lamb(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0, adam=False):
        # same argument domain as the reference's constructor (arm/optim/lamb.py:47-54): a ValueError outside it
        for name, value, ok in (("lr", lr, lr >= 0.0), ("eps", eps, eps >= 0.0), ("betas[0]", betas[0], 0.0 <= betas[0] < 1.0),
                                ("betas[1]", betas[1], 0.0 <= betas[1] < 1.0)):
            if not ok:
                raise ValueError(f"Lamb: {name} = {value} is outside its domain (lr, eps >= 0; 0 <= beta < 1)")
        self.adam = adam
        self._plan = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def load_state_dict(self, state_dict):
        """`utils.py:289`'s resume path: the loaded moment tensors replace the ones the cached launch plan points at."""
        super().load_state_dict(state_dict)
        self._plan = None

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plan = None

    def _build_plan(self, ps, dev):
        for p in ps:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p.data)
                st["exp_avg_sq"] = torch.zeros_like(p.data)
            for k in ("exp_avg", "exp_avg_sq"):                 # moments restored from a CPU checkpoint / another device / non-contiguous
                m = st[k]
                if m.shape != p.shape:                           # a positional state dict from a different parameter order / network: fail loudly -
                    raise ValueError(f"Lamb: loaded state '{k}' has shape {tuple(m.shape)} for a parameter of shape {tuple(p.shape)}")   # the kernels index by raw pointers
                if m.device != p.device or m.dtype != torch.float32 or not m.is_contiguous():
                    st[k] = m.to(p.device, torch.float32).contiguous()
        # the plan holds raw device pointers: parameters, gradients AND both moment tensors are part of its identity
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel())
                    for p in ps)
        if self._plan is None or self._plan["key"] != key:
            rows = []
            for t, p in enumerate(ps):
                for off in range(0, p.numel(), _CHUNK):
                    rows.append((t, off, min(_CHUNK, p.numel() - off)))
            ptrs = [[p.data.data_ptr() for p in ps], [p.grad.data.data_ptr() for p in ps],
                    [self.state[p]["exp_avg"].data_ptr() for p in ps], [self.state[p]["exp_avg_sq"].data_ptr() for p in ps]]
            self._plan = dict(key=key, chunks=torch.tensor(rows, dtype=torch.int64, device=dev).contiguous(), n_chunks=len(rows),
                              ptrs=torch.tensor(ptrs, dtype=torch.int64, device=dev).contiguous(),
                              norms=torch.zeros(len(ps), 2, dtype=torch.float64, device=dev),
                              stats=torch.zeros(len(ps), 3, dtype=torch.float32, device=dev))
        return self._plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dev = _lib.require_gpu()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.grad.is_sparse:
                    raise RuntimeError("Lamb: sparse gradients are not supported (the step runs dense multi-tensor kernels)")
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
            plan = self._build_plan(ps, dev)
            beta1, beta2 = group["betas"]
            _lib.call("semabs_lamb_step", _lib.ptr(plan["chunks"]), plan["n_chunks"], _lib.ptr(plan["ptrs"]), len(ps), float(group["lr"]),
                      float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), int(self.adam), _lib.ptr(plan["norms"]),
                      _lib.ptr(plan["stats"]), _lib.stream())
            # The kernels wrote through raw device pointers: tell torch the parameters changed, so that everything keyed on a tensor's
            # version counter - the derived kernel operands of the nn.Modules (module.signature), autograd's saved-tensor checks - sees the step
            try:
                torch._C._increment_version(ps)                  # iterable overload (recent torch)
            except TypeError:                                    # older builds take one tensor at a time
                for p in ps:
                    torch._C._increment_version(p)
            stats = plan["stats"]
            for t, p in enumerate(ps):
                st = self.state[p]
                st["step"] += 1
                st["weight_norm"], st["adam_norm"], st["trust_ratio"] = stats[t, 0], stats[t, 1], stats[t, 2]   # device scalars, no sync
        return loss
