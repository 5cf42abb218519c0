"""mh_volume_reduce (RCCL through the C ABI).  On a one-GPU box only the single-rank communicator can
be exercised (RCCL refuses two ranks on one device): that covers the run-time binding of librccl, mh_comm_unique_id /
mh_comm_init / mh_comm_destroy and both modes of the call.  With two or more GPUs the real exchange runs: two processes,
one GPU each, slab gather and dense reduce against the volume a single process builds."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_and_both_modes():
    import torch

    from monohair_amd import dist as mdist

    dev = torch.device("cuda:0")
    ctx, comm = mdist.rccl_comm(dev)
    assert comm.value
    g = torch.Generator().manual_seed(0)
    vol = torch.rand((16, 12, 10, 4), generator=g).to(dev)
    want = vol.clone()
    for mode in (0, 1):
        mdist.volume_reduce(vol, dev, mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(vol, want), mode            # one rank owns everything: the volume is unchanged


WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from monohair_amd import dist as mdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
X, Y, Z, C = 64, 48, 40, 4
full = torch.rand((X, Y, Z, C), generator=torch.Generator().manual_seed(3))
b = mdist.slab_bounds(X, world)
for mode in (0, 1):
    vol = torch.zeros((X, Y, Z, C), device=dev)
    vol[b[rank]:b[rank + 1]] = full[b[rank]:b[rank + 1]].to(dev)      # every rank fills only the slab it owns
    mdist.volume_reduce(vol, dev, mode=mode)
    torch.cuda.synchronize()
    if rank == 0:
        assert torch.equal(vol.cpu(), full), mode
dist.barrier()
if rank == 0:
    print("VOLUME_REDUCE_OK")
dist.destroy_process_group()
"""


def test_two_gpus_slab_gather_equals_single_process_volume(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "VOLUME_REDUCE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
