import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd
from semabs_amd import _lib
import runpy
real = _lib.call
log = []
WATCH = set(sys.argv[1:]) or {"semabs_wgrad"}
sys.argv = sys.argv[:1]
def wrapped(name, *a):
    if name in WATCH:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        real(name, *a)
        torch.cuda.synchronize(); log.append((1e6 * (time.perf_counter() - t0), name, [x for x in a if isinstance(x, (int, float))]))
    else:
        real(name, *a)
_lib.call = wrapped
sys.argv = ["train_bench.py", "--steps", "1", "--warmup", "1"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_bench.py"), run_name="__main__")
n = len(log) // 2
for us, nm, shp in log[-n:]:
    print(f"{us:8.0f} us  {nm} {shp}")
