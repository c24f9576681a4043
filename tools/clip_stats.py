"""Residual-stream statistics of a CLIP state dict under the oracle's fp32 forward (CPU): what `weights.make_clip_state_dict(stats="trained")` is
calibrated with.  Per block: the spread (std over channels, massive ones excluded) of a row, the three largest channel magnitudes over it, the bulk's row DC offset (|row mean| / row spread, massive
channels excluded), sigma of the scaled attention scores, mean of the row maximum of the softmax; class-token norm over patch-token norm at the input.

    python tools/clip_stats.py [ViT-B/32|ViT-B/16] [init|trained]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semabs_amd  # noqa: F401,E402
from oracle import relevancy as orl  # noqa: E402
from semabs_amd.synth import synth_rgb  # noqa: E402
from semabs_amd.weights import make_clip_state_dict  # noqa: E402


def row_stats(x, k=3):
    a = x.abs().reshape(-1, x.shape[-1])
    ch = a.mean(0)
    top = torch.topk(ch, k).indices
    med = float(a.median())
    bulk = torch.ones(x.shape[-1], dtype=torch.bool)
    bulk[top] = False
    xb = x.reshape(-1, x.shape[-1])[:, bulk]
    dc = (xb.mean(1).abs() / xb.std(1)).mean()
    full = (x.reshape(-1, x.shape[-1]).mean(1).abs() / x.reshape(-1, x.shape[-1]).std(1)).mean()
    spread = float(xb.std(1).mean())
    return spread, [float(ch[c] / spread) for c in top], float(dc), float(full), float(a.max())


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
    stats = sys.argv[2] if len(sys.argv) > 2 else "trained"
    sd = make_clip_state_dict(arch, 0, text_tower=False, stats=stats)
    from oracle import preprocess as op
    sizes = [120, 80, 60, 30, 97]
    tiles = torch.from_numpy(np.stack([op.preprocess_tile(synth_rgb(sizes[i % 5], sizes[i % 5], seed=7 + i)) for i in range(3)]))
    with torch.no_grad():
        x = orl.vit_embed(sd, tiles)
        print(f"{arch} {stats}: |cls token| / mean |patch token| at the trunk input = {float(x[:, 0].norm(dim=-1).mean() / x[:, 1:].norm(dim=-1).mean()):.2f}")
        for i in range(12):
            pre = f"visual.transformer.resblocks.{i}."
            med, top, dc, full, amax = row_stats(x)
            h = orl._ln(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
            qkv = F.linear(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
            q, k, _ = qkv.chunk(3, -1)
            n, T, D = q.shape
            q = (q / 8).view(n, T, 12, 64).transpose(1, 2)
            k = k.view(n, T, 12, 64).transpose(1, 2)
            s = q @ k.transpose(-1, -2)
            p = s.softmax(-1)
            print(f"  block {i:2d}: bulk spread {med:6.3f}  top channels / spread {top[0]:6.1f} {top[1]:6.1f} {top[2]:6.1f}  max |x| {amax:7.1f}  bulk |mean| / spread {dc:5.2f} "
                  f"(all channels {full:4.2f})  score sigma {float(s.std(-1).mean()):5.2f}  max |q|,|k| {float(q.abs().max()):6.1f} {float(k.abs().max()):6.1f}  mean row-max prob {float(p.max(-1).values.mean()):.3f}")
            x = orl._block(sd, pre, x, 12)
        med, top, dc, full, amax = row_stats(x)
        print(f"  output  : bulk spread {med:6.3f}  top channels / spread {top[0]:6.1f} {top[1]:6.1f} {top[2]:6.1f}  max |x| {amax:7.1f}  bulk |mean| / spread {dc:5.2f}")


if __name__ == "__main__":
    main()
