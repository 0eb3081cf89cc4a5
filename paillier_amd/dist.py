"""One-process-per-GPU plumbing (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The Paillier hot path shards embarrassingly: ciphertexts are independent, the key is replicated, and no data-path
collective is needed for Encrypt / Decrypt / Add / ConstMult / proofs.  The only exchange step in the reference's
domain is threshold decryption when each GPU plays one decryption server: the partial decryptions c_i of a ciphertext
live on different ranks and must meet on the rank that combines them -> one all-gather of fixed-stride byte buffers
(big-integer modular products are not an RCCL reduction op, so "reduce" = gather + local combine kernel).
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_slice(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of `total` units for `rank`; sizes differ by at most one; covers [0, total)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (the bench's elapsed time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_bytes(local, world: int):
    """All-gather of equally shaped uint8 tensors (partial decryptions): returns a tensor [world, *local.shape]."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.unsqueeze(0)
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)  # concatenation along dim 0 (the layout both RCCL and gloo accept)
    return out.view((world,) + tuple(local.shape))
