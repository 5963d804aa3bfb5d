"""RCCL on the device (-m gpu): the one collective of the path, `parallel.gather_episode_block`, over the `nccl` backend (= RCCL on ROCm).

A one-GPU box cannot run two ranks on two devices, so this is world size 1 -- but it is a real process group on `cuda:0`: librccl is loaded, a communicator is
created, and `all_gather_into_tensor` / `all_reduce` run on device float64 tensors of the real shapes ([E/R, 6] rows + counts = one [4096, 7] block).
The ranks > 1 logic (rank-major order = env order, shard ranges, sums payload) is covered by the gloo world-size-2 tests in tests/test_host_cpu.py.
The group lives in a fresh child process so that the test process itself never initialises a communicator."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from benchpush_amd.parallel import gather_episode_block, gather_episode_sums, summarize_episode_block, allgather_episode_metrics
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
E = 4096
g = torch.Generator(device=dev); g.manual_seed(5)
rows = torch.rand((E, 6), generator=g, device=dev, dtype=torch.float64) * 100 - 50
cnt = torch.randint(0, 9, (E,), generator=g, device=dev, dtype=torch.int32)
dist.barrier()
torch.cuda.synchronize()
allr, allc = gather_episode_block(rows, cnt, dist)          # one all_gather_into_tensor of the [4096, 7] float64 block through RCCL
torch.cuda.synchronize()
assert allr.device.type == "cuda" and allr.shape == (E, 6) and allc.shape == (E,)
assert torch.equal(allr, rows) and torch.equal(allc, cnt.to(torch.int64))
s = summarize_episode_block(allr, allc)
assert s["episodes"] == int(cnt.sum().item())
sums, c2 = gather_episode_sums(rows * 3, cnt, dist)
assert torch.equal(sums, rows * 3) and torch.equal(c2, cnt.to(torch.int64))
# the other two collectives bench.py issues: MAX over ranks of the wall time, the [1, 2] episode counters
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
m = allgather_episode_metrics(torch.tensor([[3.0, 1.0]], dtype=torch.float64, device=dev), dist)
assert m.shape == (1, 2) and m[0, 0].item() == 3.0
dist.barrier()
dist.destroy_process_group()
print("rccl-ok")
'''


def test_episode_block_allgather_through_rccl_on_the_device(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "rccl-ok" in out, out[-3000:]
