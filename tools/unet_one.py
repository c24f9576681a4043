"""Run the exact-mode UNet forward a few times (profiling aid): python tools/unet_one.py [volumes] [S] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd.unet3d import ResidualUNet3D
from semabs_amd.weights import make_semabs3d_state_dict
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=6, precision="exact")
u.load_state_dict(make_semabs3d_state_dict(seed=3), prefix="vol_feature_extractor.")
x = torch.zeros(P, S, S, S, 16, device="cuda")
x[:, ::3, ::2, ::5] = torch.randn(P, (S + 2) // 3, (S + 1) // 2, (S + 4) // 5, 16, device="cuda")
for _ in range(reps):
    y = u.forward_cl(x)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
