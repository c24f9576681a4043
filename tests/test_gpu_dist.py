"""GPU: the process-group paths with the `nccl` backend (= RCCL on ROCm) and bench.py's own rank launcher.

The test box has ONE GPU, where RCCL only admits a one-rank group: that still initialises the backend and pushes the product's collectives
(`allgather_tile_relevance`, `gather_results`, `allreduce_flat_gradients`) through RCCL device buffers.  When >= 2 devices are visible the same
worker runs with two ranks and `bench.py --gpus 2` is run end to end; otherwise those cases skip.  (Two-rank equivalence itself is covered on
one GPU over gloo: tests/test_gpu_sharded_scene.py, tests/test_gpu_train_dp.py.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SEMABS_ROOT"])
import semabs_amd
from semabs_amd import dist as sd
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
assert dist.get_backend() == "nccl"
L, N, g = 3, 11, 2
full = torch.arange(L * N * g * g, dtype=torch.float32, device="cuda").view(L, N, g, g)
lo, hi = sd.shard_range(N, rank, world)
rel = sd.allgather_tile_relevance([full[:, lo:hi].contiguous(), 2 * full[:, lo:hi].contiguous()], N)
assert torch.equal(rel[0], full) and torch.equal(rel[1], 2 * full)
gathered = sd.gather_results(torch.full((2, 4), float(rank), device="cuda"))
assert gathered.shape == (world, 2, 4) and all(float(gathered[r, 0, 0]) == r for r in range(world))
flat = torch.cat([torch.arange(6, dtype=torch.float32) * (rank + 1), torch.tensor([1.0, 0.0, 0.0] if rank == 0 else [0.0, 0.0, 1.0])]).cuda()
scale, used = sd.allreduce_flat_gradients(flat, 3)
tot = sum(r + 1 for r in range(world))
assert scale == 1.0 / world and torch.equal(flat[:6].cpu(), torch.arange(6, dtype=torch.float32) * tot)
assert used.tolist() == ([True, False, True] if world > 1 else [True, False, False])
# the bucketed, asynchronous gradient exchange of the training step (RCCL works on the process group's stream): bit-identical to the single call
gen = torch.Generator().manual_seed(7 + rank)
base = torch.randn(1 << 20, generator=gen).cuda()
want = base.clone(); sd.allreduce_flat_gradients(want, 0)
for announce in ((), (0, 2), (0, 1, 2, 3)):
    buf = base.clone()
    br = sd.BucketedAllReduce(buf, [(600000, 1 << 20), (250000, 600000), (100000, 250000), (0, 100000)])
    br.begin_step()
    for i in announce:
        br.ready(i)
        buf2 = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")      # compute queued behind the announcement
    assert br.finish() == 1.0 / world
    torch.cuda.synchronize()
    if world <= 2:                                          # two addends: the order cannot matter - bit-identical to the single call (ADVICE r5)
        assert torch.equal(buf, want), announce
    else:                                                   # more ranks: RCCL may pick another reduction order per message size
        assert float((buf - want).abs().max()) <= 1e-5 * float(want.abs().max()), announce
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK", rank, world, flush=True)
'''


def _run_ranks(n):
    env = dict(os.environ, SEMABS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                           "--master-port", str(29700 + os.getpid() % 200 + n), _worker_file()], env=env, capture_output=True, text=True, timeout=600)


def _worker_file():
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"semabs_rccl_worker_{os.getpid()}.py")
    with open(path, "w") as f:
        f.write(WORKER)
    return path


def test_rccl_backend_single_rank_collectives():
    r = _run_ranks(1)
    assert r.returncode == 0 and "RCCL_OK 0 1" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_rccl_backend_two_ranks_collectives():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices (RCCL admits one rank per device)")
    r = _run_ranks(2)
    assert r.returncode == 0 and "RCCL_OK 0 2" in r.stdout and "RCCL_OK 1 2" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_gpus_more_than_devices_fails_loudly():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "HIP device(s) visible" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())            # never a JSON line claiming fewer GPUs than asked


def test_bench_world_size_mismatch_fails_loudly():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.parametrize("extra,key", [(["--mode", "latency"], "identical_across_ranks"), (["--workload", "train"], "parameters_identical_across_ranks")])
def test_bench_two_ranks_over_rccl_agree(extra, key):
    """With two devices the RCCL paths MUST work: one scene tile- / label-sharded over two ranks ends with identical maps and labels on both, and the
    data-parallel training step (bucketed all-reduce overlapped with the backward pass) with identical parameters.  Fails - does not skip - on any
    box that has the devices; a one-GPU box cannot form a two-rank RCCL group (one rank per device) and runs the same code over
    gloo instead (tests/test_bench_contract.py)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("ONE HIP device: RCCL admits one rank per device (the same modes run over gloo in tests/test_bench_contract.py)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity",
                        "--no-stages"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["backend"] == "nccl" and d["collectives"][key] is True
    assert len(d["collectives"]["per_rank"]) == 2 and all(v for v in d["collectives"]["per_rank"])


def test_bench_spawns_its_own_ranks():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
