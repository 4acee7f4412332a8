"""Batch sharding across the GPUs of one node and the ONE collective of an MPC step.

The reference is single-process; rollouts are independent, so the batch is split
into contiguous ranges (rollout b of the global batch lives on rank b // per_rank) and the only exchange
is a gather of the first-knot results {u0[19], cost, optionally K0[19x51]} to rank 0 once per MPC step
(RCCL over xGMI when the backend is "nccl"; every peer uses its own link to rank 0).  No reduction, no
per-iteration exchange.
"""
import numpy as np
import torch
import torch.distributed as dist

NU, NX = 19, 51


def shard_range(global_batch, rank, world):
    """Contiguous rollout range [lo, hi) owned by `rank`; requires world | global_batch (weak scaling)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def payload_width(with_gains):
    return NU + 1 + (NU * NX if with_gains else 0)


def pack_payload(payload, u0, cost, K0=None):
    """payload[B, width] <- [u0 | cost | K0.flatten] (all tensors on the same device)."""
    payload[:, :NU] = u0
    payload[:, NU] = cost
    if K0 is not None:
        payload[:, NU + 1:] = K0.reshape(K0.shape[0], -1)
    return payload


def gather_first_knot(payload, dst=0):
    """Gather every rank's payload on `dst`; returns the [world * B, width] tensor there, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return payload
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    return torch.cat(bufs, dim=0) if rank == dst else None


def unpack_payload(gathered, with_gains):
    g = gathered
    u0 = g[:, :NU]
    cost = g[:, NU]
    K0 = g[:, NU + 1:].reshape(-1, NU, NX) if with_gains else None
    return u0, cost, K0
