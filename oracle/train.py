"""ORACLE (test infrastructure, not product code) — CPU restatement of one VOOL optimisation step:
`train_vool.get_losses` (train_vool.py:118-178: forward + binary_cross_entropy_with_logits with `utils.get_bce_weight`,
utils.py:727-749) followed by `utils.loop`'s update (utils.py:404-417: backward, clip_grad_norm_, Lamb.step).

The forward is the functional restatement of oracle/vool.py; the backward is torch-CPU autograd over it (what the reference runs).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.  Pinned by tests/golden/g13_vool_train.npz.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .vool import lamb_step, vool_forward


def bce_weight(label: torch.Tensor, balance_positive_negative: bool) -> torch.Tensor:
    """utils.py:727-749.  label [B, D, M] (0/1 floats)."""
    w = torch.ones_like(label).float()
    if balance_positive_negative:
        total = w.sum()
        pos = label.bool()
        B, D, M = pos.shape
        pp = pos.float().mean(dim=2).view(-1)
        pn = 1 - pp
        w = w.view(-1, M)
        pos2 = pos.view(-1, M)
        for i in range(len(w)):
            w[i, pos2[i]] = 1.0 / (pp[i] + 1e-10)
            w[i, ~pos2[i]] = 1.0 / (pn[i] + 1e-10)
        w = w.view(label.shape)
        w = w * (total / w.sum())
    return w


def vool_loss_and_grads(sd, batch, scene_bounds, grid_shape, num_levels=6, balance_positive_negative=False):
    """-> (loss float, logits [B, D, M], {key: grad tensor} for every parameter the graph reaches)."""
    params = {k: v.detach().clone().float().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith("steps")}
    D = batch["output_label_pts"].shape[1]
    out = vool_forward(params, batch["input_xyz_pts"], batch["input_target_saliency_pts"].reshape(*batch["input_target_saliency_pts"].shape[:3], 1),
                       batch["input_reference_saliency_pts"].reshape(*batch["input_reference_saliency_pts"].shape[:3], 1),
                       batch["output_xyz_pts"], batch["spatial_relation_name"], scene_bounds, grid_shape, num_levels=num_levels)
    label = batch["output_label_pts"].float()
    loss = F.binary_cross_entropy_with_logits(out, label, weight=bce_weight(label, balance_positive_negative))
    loss.backward()
    grads = {k: p.grad for k, p in params.items() if p.grad is not None}
    return float(loss.item()), out.detach(), grads


def clip_grad_norm(grads: dict, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (norm type 2): scales in place, returns the pre-clip norm."""
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    coef = min(1.0, max_norm / (total + 1e-6))
    if coef < 1.0:
        for g in grads.values():
            g.mul_(coef)
    return total


def vool_train_step(sd, batch, scene_bounds, grid_shape, num_levels=6, lr=1e-3, weight_decay=1e-5, grad_max_norm=2.0,
                    balance_positive_negative=False):
    """First optimiser step from a fresh Lamb state -> dict(loss, logits, grads (pre-clip copies), total_norm, new_sd)."""
    loss, logits, grads = vool_loss_and_grads(sd, batch, scene_bounds, grid_shape, num_levels, balance_positive_negative)
    raw = {k: g.clone() for k, g in grads.items()}
    total = clip_grad_norm(grads, grad_max_norm)
    new_sd = {k: v.clone() for k, v in sd.items()}
    for k, g in grads.items():
        w = sd[k].float().numpy()
        nw, _, _, _ = lamb_step(w, g.numpy(), np.zeros_like(w), np.zeros_like(w), lr=lr, weight_decay=weight_decay)
        new_sd[k] = torch.from_numpy(nw)
    return dict(loss=loss, logits=logits, grads=raw, total_norm=total, new_sd=new_sd)
