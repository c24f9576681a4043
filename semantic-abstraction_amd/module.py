"""`torch.nn.Module` plumbing shared by the drop-in classes (`SemAbs3D`, `SemAbsVOOL`, `ResidualUNet3D`).

The reference builds its networks as nn.Modules and its callers rely on that surface: `net_class(**kwargs).to(device)`,
`get_n_params(net)` / `Lamb(net.parameters(), ...)`, `DistributedDataParallel(module=net)`, `net.load_state_dict(ckpt["net"])`,
`net.eval()` (utils.py:225-296, visualize.py:333,452).  The HIP kernels want their own operand layouts (fp16 hi/lo split matrices,
flattened MLP weights), so every class keeps

  * real `nn.Parameter`s / buffers under the reference's state-dict key names (nested containers built from the dotted keys), and
  * a cache of kernel operands DERIVED from them, rebuilt when a parameter's storage or version counter changes
    (`load_state_dict`, `.to()`, an optimizer step, a manual `.data` edit).
"""
from __future__ import annotations

from typing import Dict, Iterable

import torch


class _Node(torch.nn.Module):
    """Name-space container: `a.b.0.weight` -> modules a / b / 0 with parameter `weight` (what Sequential / ModuleList / ParameterDict produce
    in the reference)."""

    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError("container module")


def register_tree(root: torch.nn.Module, tensors: Dict[str, torch.Tensor], buffers: Iterable[str] = ()):
    """Register every `dotted.key -> tensor` under `root` as nn.Parameter (or buffer), creating the intermediate containers."""
    buffers = set(buffers)
    for key, t in tensors.items():
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        t = t.detach().clone()
        if key in buffers:
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], torch.nn.Parameter(t, requires_grad=torch.is_floating_point(t)))


def signature(module: torch.nn.Module):
    """Changes whenever a parameter / buffer of `module` is replaced or written in place."""
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def strip_module_prefix(sd):
    """DDP-saved checkpoints carry a `module.` prefix (utils.py:283-287 strips it by hand when not distributed)."""
    if any(k.startswith("module.") for k in sd):
        return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    return sd
