"""world_size-2 gloo test (CPU) of the N > 1 path: contiguous sharding + the per-step gather to rank 0
reproduce the single-process ordering bit for bit (no reductions involved)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package  # noqa: E402


def _fake_results(lo, hi, with_gains):
    """Deterministic stand-in for the per-rollout solver outputs, a function of the GLOBAL rollout index."""
    idx = torch.arange(lo, hi, dtype=torch.float64)
    u0 = torch.sin(idx[:, None] * 0.1 + torch.arange(19, dtype=torch.float64)[None, :])
    cost = 100.0 + idx
    K0 = torch.cos(idx[:, None, None] * 0.01 + torch.arange(19 * 51, dtype=torch.float64).reshape(1, 19, 51)) if with_gains else None
    return u0, cost, K0


def _worker(rank, world, port, with_gains, global_batch, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    from mpc_ilqr_mujoco_amd import sharding as sh
    lo, hi = sh.shard_range(global_batch, rank, world)
    u0, cost, K0 = _fake_results(lo, hi, with_gains)
    payload = torch.zeros(hi - lo, sh.payload_width(with_gains), dtype=torch.float64)
    sh.pack_payload(payload, u0, cost, K0)
    g = sh.gather_first_knot(payload, dst=0)
    if rank == 0:
        torch.save(g, out)
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("with_gains", [False, True])
def test_shard_and_gather_matches_single_process(tmp_path, with_gains):
    pkg = load_package()
    from mpc_ilqr_mujoco_amd import sharding as sh
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "g.pt")
    G = 16
    mp.spawn(_worker, args=(2, port, with_gains, G, out), nprocs=2, join=True)
    g = torch.load(out)
    u0, cost, K0 = sh.unpack_payload(g, with_gains)
    ru0, rcost, rK0 = _fake_results(0, G, with_gains)
    assert torch.equal(u0, ru0) and torch.equal(cost, rcost)
    if with_gains:
        assert torch.equal(K0, rK0)


def test_shard_range_rules():
    pkg = load_package()
    from mpc_ilqr_mujoco_amd import sharding as sh
    assert sh.shard_range(32768, 3, 8) == (3 * 4096, 4 * 4096)
    with pytest.raises(ValueError):
        sh.shard_range(10, 0, 4)
    # single process: gather is the identity
    p = torch.arange(40, dtype=torch.float64).reshape(2, 20)
    assert sh.gather_first_knot(p) is p
