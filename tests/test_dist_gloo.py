"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: batch sharding covers the batch exactly once, the
MAX-over-ranks timing reduction and the all-gather used by the threshold share-combine exchange work across processes."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paillier_amd import dist as pd
    b, e = pd.shard_slice(total, rank, world)
    # every rank "processes" its slice: here a checksum of the unit ids
    local = torch.zeros(4, dtype=torch.int64)
    local[0], local[1], local[2] = b, e, sum(range(b, e))
    gathered = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, local)
    t = pd.max_over_ranks(1.0 + rank)
    # the exchange used by threshold combine: fixed-stride byte buffers, one per server rank
    part = torch.full((3, 8), rank + 1, dtype=torch.uint8)
    allp = pd.all_gather_bytes(part, world)
    q.put((rank, [g.tolist() for g in gathered], t, allp.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [65536, 1001])
def test_two_rank_sharding_and_exchange(total):
    world, port = 2, _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, t, allp in res:
        assert t == 2.0  # MAX over ranks of (1 + rank)
        assert gathered[0][0] == 0 and gathered[-1][1] == total
        assert all(gathered[i][1] == gathered[i + 1][0] for i in range(world - 1))  # contiguous, no overlap
        assert sum(g[2] for g in gathered) == total * (total - 1) // 2             # every unit exactly once
        assert allp == [[[1] * 8] * 3, [[2] * 8] * 3]


def test_shard_slice_properties():
    from paillier_amd.dist import shard_slice
    for total in (0, 1, 7, 16384, 65537):
        for world in (1, 2, 3, 8):
            sl = [shard_slice(total, r, world) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == total
            assert all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 1
