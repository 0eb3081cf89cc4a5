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
    if dist.get_backend() == "gloo":
        device = None                      # rehearsals over gloo reduce on the host
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(flag: int, device=None) -> int:
    """MIN all-reduce of a host integer (parity flags: 1 = this rank's results are right)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(flag)
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([int(flag)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def all_gather_bytes(local, world: int):
    """All-gather of equally shaped uint8 tensors (partial decryptions): returns a tensor [world, *local.shape]."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.unsqueeze(0)
    local = local.contiguous()
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()                # rehearsal of the exchange with several ranks on ONE GPU: through host memory
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)  # concatenation along dim 0 (the layout both RCCL and gloo accept)
    return out.view((world,) + tuple(local.shape)).to(dev)


def _sync(t):
    if getattr(t, "is_cuda", False):
        import torch
        torch.cuda.synchronize(t.device)


def threshold_decrypt_sharded(c, n_servers: int, rank: int, world: int, partial_fn, combine_fn, pad_rows: bool = True,
                              units_fn=None, servers_fn=None, range_fn=None, timings=None):
    """Threshold decryption of B ciphertexts with the work sharded over `world` ranks and ONE exchange step
    (thresholdkey.go:149-201: every server's PartialDecrypt, then CombinePartialDecryptions).

    Work units are (server, ciphertext) pairs, server-major: unit u = s * B + i.  Rank r computes the partial decryptions
    of its contiguous unit range (so it touches at most two servers' shares -- shares are not replicated everywhere), the
    fixed-stride partials are all-gathered (RCCL over xGMI with one GPU per rank; big-integer modular products are not a
    collective reduction op: gather + local combine kernel), then every rank combines its own ciphertext slice.

      c            uint8 tensor [B, cipher_bytes] (same on every rank: ciphertexts are public), on the device the
                   collective backend moves (cuda for nccl, cpu for gloo)
      partial_fn   (server_index, c_rows) -> uint8 tensor [len(c_rows), cipher_bytes]      PartialDecrypt of one server
      combine_fn   ([rows of server 0, rows of server 1, ...]) -> uint8 tensor [rows, plain_bytes]   Combine
      units_fn     optional: (server_index int32 numpy [units], c_rows [units, cipher_bytes]) -> uint8 tensor [units,
                   cipher_bytes]: this rank's units in ONE call (pgpu_partial_decrypt_indexed) instead of one partial_fn
                   call per server -- a shard of a few thousand units fills the GPU only when its servers share a launch
      servers_fn   optional, used when ONE rank holds every share (world == 1): (c) -> [rows of server 0, rows of server 1, ...]
                   for the whole batch in one call (pgpu_partial_decrypt_multi: two servers' ladders per launch)
      range_fn     optional, preferred when given: (c, unit_begin, unit_end) -> uint8 tensor [unit_end - unit_begin,
                   cipher_bytes]: this rank's units straight from the whole ciphertext batch (pgpu_partial_decrypt_units:
                   ciphertexts the range wants under several shares walk one chain of squarings -- a rank that holds one
                   server whole and half of the next, N = 2, computes 1.5 ladders' worth instead of 3 half-batch ladders)
      timings      optional dict: receives what the exchange of this step cost -- "exchange_s" (seconds of the all-gather on this
                   rank, between device synchronisations; added to what is there), "exchange_bytes" (payload: the partials of
                   every unit), "exchange_padded_bytes" (what the collective moved into this rank), "exchange_world" (the world
                   size the BACKEND reports) and "exchange_backend"
    Returns (plaintext rows of this rank's ciphertext slice, (begin, end) of that slice).
    """
    import time
    import torch
    B, cbytes = int(c.shape[0]), int(c.shape[1])
    units = n_servers * B
    ub, ue = shard_slice(units, rank, world)
    per = -(-units // world) if pad_rows else (ue - ub)      # all-gather needs equal shapes
    local = torch.zeros((max(per, 1), cbytes), dtype=torch.uint8, device=c.device)
    if range_fn is not None and ue > ub:
        local[:ue - ub] = range_fn(c, ub, ue)
    elif servers_fn is not None and world == 1:
        for s_, rows in enumerate(servers_fn(c)):
            local[s_ * B:(s_ + 1) * B] = rows
    elif units_fn is not None and ue > ub:
        import numpy as np
        us = np.arange(ub, ue, dtype=np.int64)
        rows = c[torch.from_numpy(us % B).to(c.device)]
        local[:ue - ub] = units_fn((us // B).astype(np.int32), rows)
    else:
        u = ub
        while u < ue:
            s, i0 = divmod(u, B)
            cnt = min(ue - u, B - i0)
            local[u - ub:u - ub + cnt] = partial_fn(s, c[i0:i0 + cnt])
            u += cnt
    if timings is not None:
        _sync(local)
        t0 = time.perf_counter()
    g = all_gather_bytes(local, world).reshape(world * max(per, 1), cbytes)
    if timings is not None:
        import torch.distributed as dist
        _sync(g)
        live = world > 1 and dist.is_available() and dist.is_initialized()
        timings["exchange_s"] = timings.get("exchange_s", 0.0) + (time.perf_counter() - t0)
        timings["exchange_bytes"] = units * cbytes if world > 1 else 0
        timings["exchange_padded_bytes"] = world * max(per, 1) * cbytes if world > 1 else 0
        timings["exchange_world"] = dist.get_world_size() if live else 1
        timings["exchange_backend"] = dist.get_backend() if live else None
    parts = torch.empty((units, cbytes), dtype=torch.uint8, device=c.device)
    for q in range(world):          # un-pad: rank q's units sit at rows [q*per, q*per + len_q)
        qb, qe = shard_slice(units, q, world)
        parts[qb:qe] = g[q * max(per, 1):q * max(per, 1) + (qe - qb)]
    parts = parts.view(n_servers, B, cbytes)
    cb, ce = shard_slice(B, rank, world)
    if ce == cb:
        return None, (cb, ce)
    return combine_fn([parts[s, cb:ce].contiguous() for s in range(n_servers)]), (cb, ce)


THRESHOLD_SHARDS = ("units", "ciphertext")


def threshold_bench_entries(world: int):
    """(entry name, shard) pairs bench.py reports for BASELINE config 4.  `threshold_2048` is ALWAYS the flow north_star names: the
    servers' shares on different ranks, every rank computes its (server, ciphertext) unit range, the partials are all-gathered over
    RCCL / xGMI, every rank combines its own ciphertext slice.  With more than one rank the bench adds the no-exchange shard of a
    holder of every share beside it (`threshold_2048_replicated`); on one rank the two are the same call."""
    return [("threshold_2048", "units")] + ([("threshold_2048_replicated", "ciphertext")] if world > 1 else [])


def threshold_step(shard: str, c, n_servers: int, rank: int, world: int, *, partial_fn, combine_fn, units_fn=None, servers_fn=None,
                   range_fn=None, timings=None):
    """One threshold decryption of the batch `c` in the named shard -- the function bench.py times and the gloo tests call.
      "units"       threshold_decrypt_sharded: unit ranges, ONE all-gather of the partials, local combine
      "ciphertext"  threshold_decrypt_ciphertext_major: ciphertext slices under every share, no exchange (needs range_fn)
    Returns (plaintext rows of this rank's ciphertext slice or None, (begin, end))."""
    if shard == "units":
        return threshold_decrypt_sharded(c, n_servers, rank, world, partial_fn, combine_fn, units_fn=units_fn, servers_fn=servers_fn,
                                         range_fn=range_fn, timings=timings)
    if shard == "ciphertext":
        if range_fn is None:
            raise ValueError("the ciphertext-major shard needs range_fn")
        if timings is not None:
            timings.update(exchange_bytes=0, exchange_padded_bytes=0, exchange_world=world, exchange_backend=None)
            timings.setdefault("exchange_s", 0.0)
        return threshold_decrypt_ciphertext_major(c, n_servers, rank, world, range_fn, combine_fn)
    raise ValueError(f"unknown threshold shard {shard!r}: one of {THRESHOLD_SHARDS}")


def threshold_shard_mode(n_ciphertexts: int, world: int) -> str:
    """Which shard of the threshold flow a holder of EVERY share should take (a caller's choice; bench.py reports both).  Measured per rank on one MI355X (t = 3, 2048-bit key,
    16 384 ciphertexts, tools/threshold_shard_probe.py): unit ranges 81 / 70 / 59 / 36 ms at N = 1 / 2 / 4 / 8 plus the exchange of the
    partials; ciphertext slices 81 / 49 / 43 / 36 ms and no exchange (the t ladders of a ciphertext share a chain of squarings while
    the chip is full, and split into chains of their own on the eight-lane kernel as it empties: plan::shared_chain_groups).  So:
    "ciphertext" whenever there is more than one rank; a single rank is the same call either way."""
    return "ciphertext" if world > 1 else "units"


def threshold_decrypt_ciphertext_major(c, n_servers: int, rank: int, world: int, range_fn, combine_fn):
    """Threshold decryption of B ciphertexts by ranks that ALL hold every one of the t shares (shares replicated -- a benchmark or a
    single trust domain with several GPUs; with one share per machine use threshold_decrypt_sharded): rank r takes the ciphertexts
    shard_slice(B, r, world), computes their partial decryptions under all t shares (thresholdkey.go:192-201; one call,
    pgpu_partial_decrypt_units over the slice's whole unit range: the t ladders of a ciphertext share one chain of squarings) and
    combines them locally (thresholdkey.go:149-190).  No exchange step at all.

      range_fn     (c_rows, unit_begin, unit_end) -> uint8 tensor [unit_end - unit_begin, cipher_bytes], units server-major over c_rows
      combine_fn   ([rows of server 0, rows of server 1, ...]) -> uint8 tensor [rows, plain_bytes]
    Returns (plaintext rows of this rank's ciphertext slice, (begin, end) of that slice)."""
    B, cbytes = int(c.shape[0]), int(c.shape[1])
    cb, ce = shard_slice(B, rank, world)
    if ce == cb:
        return None, (cb, ce)
    cnt = ce - cb
    parts = range_fn(c[cb:ce], 0, n_servers * cnt)
    if tuple(parts.shape) != (n_servers * cnt, cbytes):
        raise ValueError("range_fn must return one row per (server, ciphertext) unit of the slice")
    parts = parts.view(n_servers, cnt, cbytes)
    return combine_fn([parts[s_] for s_ in range(n_servers)]), (cb, ce)


def ddleq_prove_verify_sharded(n_statements: int, rank: int, world: int, prove_fn, verify_fn, device=None):
    """ProveDDLEQ + VerifyDDLEQProof (ddleq.go:27-53) for `n_statements` statements with the statements sharded over the ranks
    (BASELINE config 5).  Proofs of different statements are independent: rank r proves and verifies the contiguous slice
    shard_slice(n_statements, r, world) with the key replicated, and the ONLY cross-rank step is the MIN-reduction of the
    verdict flag -- no data-path collective (scaling: strong, the whole job is fixed).

      prove_fn   (begin, end) -> proofs of statements [begin, end)      (any object verify_fn understands; [] for an empty slice)
      verify_fn  (begin, end, proofs) -> sequence of bool, one verdict per statement of the slice
    Returns ((begin, end), proofs of this rank's slice, verdicts of this rank's slice, every statement of every rank accepted)."""
    b, e = shard_slice(n_statements, rank, world)
    proofs = prove_fn(b, e) if e > b else []
    verdicts = list(verify_fn(b, e, proofs)) if e > b else []
    if len(verdicts) != e - b:
        raise ValueError("verify_fn must return one verdict per statement of the slice")
    return (b, e), proofs, verdicts, bool(min_over_ranks(1 if all(verdicts) else 0, device))
