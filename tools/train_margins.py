import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import semabs_amd
import test_gpu_train as T
import conftest
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g13_vool_train.npz"), allow_pickle=True))
names = [str(k) for k in g["names"]]
for rep in range(12):
    tr, batch = T._g13_trainer(g)
    out = tr.forward_backward(batch)
    torch.cuda.synchronize()
    worst_n = 0; wk = None
    for k, n, has in zip(names, g["grad_norm"], g["has_grad"]):
        if has:
            mine = float(tr.grads[k].double().norm()); d = abs(mine - n) / max(n, 1e-12)
            if d > worst_n: worst_n, wk = d, k
    worst_l2 = worst_med = 0
    for k in list(g):
        if k.startswith("grad/"):
            l2, med = T._robust(tr.grads[k[5:]].cpu().numpy(), g[k]); worst_l2 = max(worst_l2, l2)
        elif k.startswith("grads/"):
            mine = tr.grads[k[6:]].cpu().numpy().reshape(-1)[g["gradidx/" + k[6:]]]
            l2, med = T._robust(mine, g[k]); worst_l2 = max(worst_l2, l2); worst_med = max(worst_med, med)
    total = float(tr.optimizer_step())
    print(f"run {rep}: loss err {abs(float(out['loss'])-float(g['loss']))/float(g['loss']):.1e}  worst norm dev {worst_n:.4f} ({wk})  worst l2 {worst_l2:.3f}  worst med {worst_med:.4f}  total dev {abs(total-float(g['total_norm']))/float(g['total_norm']):.4f}", flush=True)
