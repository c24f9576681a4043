"""Config 3 of BASELINE.json: ResidualUNet3D forward on [16, 16, 128^3] (16 label volumes), 1 GPU: volumes/s, algorithmic TFLOP/s and the
compulsory-traffic GB/s of SURVEY.md 8(d) (0.838 G elements per volume), for the fp16 and the exact (fp32 activations, split-fp16 MFMA) modes.
    python tools/unet_bench.py [volumes] [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd.unet3d import ResidualUNet3D
from semabs_amd.weights import make_semabs3d_state_dict
if os.environ.get("SEMABS_TUNE_LIB") == "1" and os.environ.get("UNET_TUNE"):      # A/B switches of the tuning build: UNET_TUNE="key:value,..." -> semabs_conv_tune
    from semabs_amd import _lib
    for kv in os.environ["UNET_TUNE"].split(","):
        _lib.call("semabs_conv_tune", int(kv.split(":")[0]), int(kv.split(":")[1]))
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
S = 128
for prec in ("fp16", "exact"):
    u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=6, precision=prec)
    u.load_state_dict(make_semabs3d_state_dict(seed=3), prefix="vol_feature_extractor.")
    x = torch.zeros(P, S, S, S, 16, device="cuda", dtype=u.act_dtype)
    x[:, ::3, ::2, ::5] = torch.randn(P, (S + 2) // 3, (S + 1) // 2, (S + 4) // 5, 16, device="cuda").to(u.act_dtype)
    for _ in range(2):
        y = u.forward_cl(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        y = u.forward_cl(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    bpe = 2 if prec == "fp16" else 4
    print(json.dumps({"mode": prec, "volumes": P, "ms_per_batch": dt * 1e3, "volumes_per_s": P / dt, "algorithmic_TFLOP_per_s": P * 0.34081 / dt,
                      "compulsory_GB_per_s": P * 0.838 * bpe / dt, "mfma_TFLOP_per_s_issued": P * 0.34081 * (1 if prec == "fp16" else 3) / dt}))
