"""Env sharding across the GPUs of one node: contiguous global env-id ranges per rank,
no data-path collective for stepping, one all-gather of observations per batched step."""
from __future__ import annotations


def shard_range(rank: int, world: int, envs_per_rank: int):
    """Global env ids [lo, hi) owned by `rank` (weak scaling: every rank owns envs_per_rank envs)."""
    if not (0 <= rank < world) or envs_per_rank <= 0:
        raise ValueError("bad shard arguments")
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def gather_obs(dist, obs_local, out=None):
    """All-gather obs[B_local, D] -> obs[world*B_local, D] in global env-id order (RCCL on GPUs, gloo on CPU)."""
    import torch
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * obs_local.shape[0],) + tuple(obs_local.shape[1:]), dtype=obs_local.dtype, device=obs_local.device)
    dist.all_gather_into_tensor(out, obs_local.contiguous())
    return out


def u01(seed: int, a: int, b: int) -> float:
    """Host twin of the device counter RNG (csrc/myo_hip.hip u01): splitmix64 of (seed, a, b) -> [0, 1)."""
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (a + 1) + 0xBF58476D1CE4E5B9 * (b + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    return (z >> 40) / 16777216.0
