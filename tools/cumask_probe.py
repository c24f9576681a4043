"""Where do the workgroups of a CU-masked stream land?  python tools/cumask_probe.py"""
import os, sys, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd
from semabs_amd import _lib

def stream_for(bits):
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    _lib.call("semabs_stream_create_cumask", C.byref(s), arr, 8)
    return s

def probe(s, n=2048):
    out = torch.zeros(n, 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.call("semabs_probe_placement", out.data_ptr(), n, s)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    xcc = o[:, 0]; hw = o[:, 1]
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    places = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = collections.Counter()
    for (x, _, _, _), _ in places.items():
        per_xcc[x] += 1
    return len(places), dict(sorted(per_xcc.items())), [int(x) for x in xcc[:16]]

for name, bits in [("all 256", range(256)), ("bits 0-127", range(128)), ("bits 128-255", range(128, 256)), ("even bits", range(0, 256, 2)),
                   ("bits 0-7", range(8)), ("bits 0,8,16,..", range(0, 256, 8)), ("bits with (b//8)%2==0", [b for b in range(256) if (b // 8) % 2 == 0]),
                   ("bits with (b//16)%2==0", [b for b in range(256) if (b // 16) % 2 == 0])]:
    s = stream_for(list(bits))
    n, per, first = probe(s)
    print(f"{name:26s}: {n:3d} distinct CUs; CUs per XCC {per}; XCC of blocks 0..15 {first}", flush=True)
    _lib.call("semabs_stream_destroy", s)
