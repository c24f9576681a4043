"""CPU: the training-step oracle (oracle/train.py) against the golden produced by the reference's own autograd + Lamb (G13)."""
import numpy as np
import torch

import semabs_amd  # noqa: F401
from semabs_amd.synth import SCENE_BOUNDS
from semabs_amd.weights import make_semabsvool_state_dict
from oracle import train as ot

from _train_inputs import vool_batch


import pytest


@pytest.mark.parametrize("name", ["g13_vool_train", "g20_vool_train64"])
def test_oracle_train_step_matches_reference(golden, name):
    g = golden(name)
    S, N, M, D, seed, wseed, _ = [int(v) for v in g["meta"]]
    batch = vool_batch(S, N, M, D, seed, g["label"])
    r = ot.vool_train_step(make_semabsvool_state_dict(seed=wseed), batch, SCENE_BOUNDS, (S, S, S))
    assert abs(r["loss"] - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    assert np.abs(r["logits"].numpy() - g["logits"]).max() <= 1e-5
    assert abs(r["total_norm"] - float(g["total_norm"])) <= 1e-5 * float(g["total_norm"])
    names = [str(k) for k in g["names"]]
    for k, n, has in zip(names, g["grad_norm"], g["has_grad"]):
        assert (k in r["grads"]) == bool(has), k              # visual_sampler.* stays without gradient, like p.grad = None
        if has:
            assert abs(float(r["grads"][k].double().norm()) - n) <= 1e-5 * max(n, 1e-8), k
    for k in list(g):
        if k.startswith("grad/"):
            ref = g[k]
            assert np.abs(r["grads"][k[5:]].numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-8), k
        elif k.startswith("grads/"):
            ref = g[k]
            mine = r["grads"][k[6:]].numpy().reshape(-1)[g["gradidx/" + k[6:]]]
            assert np.abs(mine - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-8), k
        elif k.startswith("new/"):
            assert np.abs(r["new_sd"][k[4:]].numpy() - g[k]).max() <= 2e-7, k


def test_oracle_bce_weight(golden):
    g = golden("g13_vool_train")
    lab = torch.from_numpy(g["label"].astype(np.float32))
    w = ot.bce_weight(lab, True)
    assert abs(float(w.double().sum()) - float(g["bce_weight_balanced_sum"])) <= 1e-3
    assert np.array_equal(w.numpy()[:, :, ::50], g["bce_weight_balanced_sub"])
    assert torch.equal(ot.bce_weight(lab, False), torch.ones_like(lab))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(torch.from_numpy(g["logits"]), lab, weight=w)
    assert abs(float(loss) - float(g["loss_balanced"])) <= 1e-6 * float(g["loss_balanced"])
