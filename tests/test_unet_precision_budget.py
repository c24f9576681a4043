"""Why the 3D-UNet keeps its three-product (hi/lo-split) MFMA arithmetic at every level.

The experiment runs on the CPU oracle: the level-0 convolutions of the UNet get either their weights or their (GroupNorm-ed) input
activations rounded to fp16 - which is exactly what dropping the `a_hi * w_lo` or the `a_lo * w_hi` MFMA product does - and the voxel logits
are compared with the unrounded run.  Measured at 64^3: 1.9e-3 (weights) / 1.5e-3 (activations) of a 0.81 logit maximum; BASELINE.json's bar
on the logits is 1e-3.  If this test starts failing because the errors became SMALLER than the bar, a two-product level 0 is worth building."""
import numpy as np
import torch
import torch.nn.functional as F

import semabs_amd  # noqa: F401
import oracle.semabs3d as O
from semabs_amd.weights import make_semabs3d_state_dict

S = 64


def _inputs():
    rng = np.random.default_rng(0)
    n = 20000
    xyz = rng.uniform(-1.0, 1.0, (1, n, 3)).astype(np.float32)
    xyz[..., 2] = (rng.integers(0, 4, (1, n)) * 0.3 - 0.5 + rng.normal(0, 0.01, (1, n))).astype(np.float32)     # a few noisy planes
    feat = (rng.standard_normal((1, 2, n, 1)) * 0.5).astype(np.float32)
    q = rng.uniform(-1.0, 1.0, (1, 2, n, 3)).astype(np.float32)
    return torch.from_numpy(xyz), torch.from_numpy(feat), torch.from_numpy(q), torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])


def test_level0_needs_both_low_order_products(monkeypatch):
    sd = {k: torch.as_tensor(np.asarray(v)).float() for k, v in make_semabs3d_state_dict(seed=0).items()}
    xyz, feat, q, bounds = _inputs()
    mode = {"what": None}
    conv3d = F.conv3d

    def rounded_conv3d(x, w, b=None, **kw):
        if x.shape[-1] == S and w.shape[-1] == 3:                   # the 3 x 3 x 3 convolutions of the full-resolution level
            if mode["what"] == "weights":
                w = w.half().float()
            elif mode["what"] == "activations":
                x = x.half().float()
        return conv3d(x, w, b, **kw)

    monkeypatch.setattr(F, "conv3d", rounded_conv3d)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    out = {}
    with torch.no_grad():
        for what in (None, "weights", "activations"):
            mode["what"] = what
            out[what] = O.semabs3d_forward(sd, xyz, feat, q, bounds, (S, S, S))
    ref = out[None]
    assert float(ref.abs().max()) > 0.3
    for what in ("weights", "activations"):
        err = float((out[what] - ref).abs().max())
        assert err > 1e-3, (what, err)                              # over the bar: that product cannot be dropped
        assert err < 1e-2, (what, err)                              # and the experiment is not broken
