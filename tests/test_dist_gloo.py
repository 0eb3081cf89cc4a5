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


def test_threshold_bench_entries():
    """BASELINE config 4 in the N-rank bench: `threshold_2048` is the exchange flow on EVERY N; the no-exchange
    shard of a holder of every share is a second entry beside it from two ranks up."""
    from paillier_amd import dist as pd
    assert pd.threshold_bench_entries(1) == [("threshold_2048", "units")]
    for world in (2, 3, 4, 8):
        assert pd.threshold_bench_entries(world) == [("threshold_2048", "units"), ("threshold_2048_replicated", "ciphertext")]
    with pytest.raises(ValueError):
        pd.threshold_step("rows", None, 3, 0, 1, partial_fn=None, combine_fn=None)
    # bench.py goes through threshold_step for both and reports the exchange of the first
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "pdist.threshold_step(shard_mode" in src and "pdist.threshold_bench_entries(world)" in src
    assert "exchange_bytes_per_step" in src and "exchange_ms_per_step" in src and "exchange_world_size" in src


def test_shard_slice_properties():
    from paillier_amd.dist import shard_slice
    for total in (0, 1, 7, 16384, 65537):
        for world in (1, 2, 3, 8):
            sl = [shard_slice(total, r, world) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == total
            assert all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 1


# ---- threshold decryption across ranks: PartialDecrypt -> exchange -> Combine (thresholdkey.go:149-201) -----------------

def _threshold_inputs(count):
    import json
    import random
    from oracle import paillier_oracle as po
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))["threshold"]["512"]
    n, shares = int(k["n"], 16), [int(s, 16) for s in k["shares"]]
    rng = random.Random(77)
    pk = po.PublicKey(N=n, G=n + 1)
    ms = [0, n - 1] + [rng.randrange(n) for _ in range(count - 2)]
    cts = [po.encrypt_with_r(pk, m, po.rand_unit(n, rng)).C for m in ms]
    return n, shares, ms, cts


def _threshold_worker(rank, world, port, ids, count, use_gpu, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from paillier_amd import dist as pd
    from oracle import paillier_oracle as po
    n, shares, ms, cts = _threshold_inputs(count)
    cb, pb = 128, 64
    to_rows = lambda vals, st: torch.from_numpy(np.frombuffer(b"".join(int(v).to_bytes(st, "big") for v in vals),
                                                              dtype=np.uint8).reshape(len(vals), st).copy())
    to_ints = lambda t: [int.from_bytes(bytes(r.tolist()), "big") for r in t]
    c = to_rows(cts, cb)
    if use_gpu:   # the product path through the C ABI (host buffers; both ranks share GPU 0 in this rehearsal)
        import paillier_amd as pa
        ctx = pa.Context(0)
        tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3)

        def partial_fn(s, rows):
            rows = rows.contiguous().numpy()
            out = np.zeros((rows.shape[0], cb), dtype=np.uint8)
            tk.partial_decrypt_raw(shares[ids[s] - 1], rows.shape[0], rows, cb, out, cb)
            return torch.from_numpy(out)

        def combine_fn(parts):
            arrs = [x.contiguous().numpy() for x in parts]
            out = np.zeros((arrs[0].shape[0], pb), dtype=np.uint8)
            tk.combine_raw(ids, arrs[0].shape[0], [a.ctypes.data for a in arrs], cb, out, pb)
            return torch.from_numpy(out)
    else:         # CPU rehearsal of the exchange logic: the oracle plays the kernels
        tsk = {i: po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=5, Threshold=3, ID=i, Share=shares[i - 1])
               for i in ids}

        def partial_fn(s, rows):
            return to_rows([po.partial_decrypt(tsk[ids[s]], x).Decryption for x in to_ints(rows)], cb)

        def combine_fn(parts):
            cols = [to_ints(x) for x in parts]
            return to_rows([po.combine_partial_decryptions(tsk[ids[0]], [po.PartialDecryption(i, col[j]) for i, col in zip(ids, cols)])
                            for j in range(len(cols[0]))], pb)
    units_fn = None
    if use_gpu:   # the one-launch form of a rank's units (pgpu_partial_decrypt_indexed)
        def units_fn(server_index, rows):
            rows = rows.contiguous().numpy()
            out = np.zeros((rows.shape[0], cb), dtype=np.uint8)
            tk.partial_decrypt_indexed_raw([shares[i - 1] for i in ids], server_index, rows.shape[0], rows, cb, out, cb)
            return torch.from_numpy(out)
    range_fn = None
    if use_gpu:   # the rank's unit range straight from the ciphertext batch (pgpu_partial_decrypt_units); checked against units_fn
        def range_fn(rows, ub, ue):
            rows = rows.contiguous().numpy()
            out = np.zeros((ue - ub, cb), dtype=np.uint8)
            tk.partial_decrypt_units_raw([shares[i - 1] for i in ids], rows.shape[0], rows, cb, ub, ue, out, cb)
            return torch.from_numpy(out)
        out2, _ = pd.threshold_decrypt_sharded(c, len(ids), rank, world, partial_fn, combine_fn, units_fn=units_fn)
    # bench.py's tstep: pd.threshold_step in the shard of BASELINE config 4 -- unit ranges, ONE all-gather, local combine -- with
    # the exchange accounted for
    tm = {}
    out, (b, e) = pd.threshold_step("units", c, len(ids), rank, world, partial_fn=partial_fn, combine_fn=combine_fn, units_fn=units_fn,
                                    range_fn=range_fn, timings=tm)
    assert tm["exchange_world"] == world == dist.get_world_size() and tm["exchange_backend"] == "gloo"
    assert tm["exchange_bytes"] == len(ids) * count * cb and tm["exchange_padded_bytes"] >= tm["exchange_bytes"]
    assert tm["exchange_padded_bytes"] == world * -(-len(ids) * count // world) * cb and tm["exchange_s"] > 0
    if use_gpu and out is not None:
        assert torch.equal(out, out2)
    # the ciphertext-major shard (every rank holds every share: no exchange) gives the same plaintexts for the same slice
    if range_fn is None:
        def range_fn(rows, ub, ue):          # CPU rehearsal: server-major units of the slice, one oracle call each
            vals = to_ints(rows)
            return to_rows([po.partial_decrypt(tsk[ids[u // len(vals)]], vals[u % len(vals)]).Decryption for u in range(ub, ue)], cb)
    tm3 = {}
    out3, (b3, e3) = pd.threshold_step("ciphertext", c, len(ids), rank, world, partial_fn=partial_fn, combine_fn=combine_fn,
                                       range_fn=range_fn, timings=tm3)
    assert (b3, e3) == (b, e) and tm3["exchange_bytes"] == 0 and tm3["exchange_s"] == 0.0
    if out is not None:
        assert torch.equal(out, out3)
    q.put((rank, b, e, to_ints(out) if out is not None else []))
    dist.barrier()
    dist.destroy_process_group()


def _run_threshold(world, ids, count, use_gpu):
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_threshold_worker, args=(r, world, port, ids, count, use_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n, shares, ms, cts = _threshold_inputs(count)
    got = []
    for rank, b, e, vals in res:
        assert b == len(got)          # contiguous ciphertext slices in rank order
        got += vals
    assert got == ms


@pytest.mark.parametrize("ids,count", [([1, 2, 3], 7), ([1, 3, 5], 10)])
def test_threshold_partial_exchange_combine_two_ranks_cpu(ids, count):
    """world_size 2, gloo: each rank computes the partial decryptions of its (server, ciphertext) units, the partials are
    all-gathered, each rank combines its ciphertext slice; the plaintexts come back.  7 ciphertexts x 3 servers = 21 units
    over 2 ranks: a rank's range straddles two servers and the last rank is padded."""
    _run_threshold(2, ids, count, use_gpu=False)


@pytest.mark.gpu
def test_threshold_partial_exchange_combine_two_ranks_gpu():
    """The same flow with the HIP path doing PartialDecrypt and Combine (two gloo ranks sharing GPU 0)."""
    _run_threshold(2, [1, 3, 5], 9, use_gpu=True)


def _ddleq_inputs(n_statements):
    """statements (ct1, ct2 = NestedRandomize(ct1; a, b), a, b) on a toy key -- the oracle plays the kernels on CPU ranks"""
    import random
    from oracle import paillier_oracle as po
    sk, p, q = po.keygen_seeded(256, 31)
    n = sk.N
    rng = random.Random(91)
    L2 = po.ENC_LEVEL_TWO
    st = []
    for _ in range(n_statements):
        inner = po.encrypt_with_r(sk, rng.randrange(n), po.rand_unit(n, rng)).C
        ct1 = po.encrypt_with_r_at_level(sk, inner, po.rand_unit(n, rng), L2).C
        a, b = po.rand_unit(n, rng), po.rand_unit(n, rng)
        st.append((ct1, po.nested_randomize_with_ab(sk, po.Ciphertext(ct1, L2), a, b).C, a, b))
    return sk, st


def _ddleq_worker(rank, world, port, n_statements, secpar, tamper, q, use_gpu=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    from paillier_amd import dist as pd
    from oracle import paillier_oracle as po
    sk, st = _ddleq_inputs(n_statements)
    L2 = po.ENC_LEVEL_TWO
    rng = random.Random(1000 + rank)
    if use_gpu:   # the product path: pgpu_ddleq_prove_secpar / pgpu_ddleq_verify on this rank's slice (both ranks share GPU 0)
        import paillier_amd as pa
        from paillier_amd import protocols as pr
        ctx = pa.Context(0)
        gpk = pa.PublicKey(ctx, sk.N, sk.N + 1)
        gsk = pa.SecretKey(ctx, gpk, sk.Lambda)

        def prove_fn(b, e):
            out = pr.prove_ddleq_batch(gsk, secpar, [t[0] for t in st[b:e]], [t[1] for t in st[b:e]], [t[2] for t in st[b:e]],
                                       [t[3] for t in st[b:e]])
            # every instance must also satisfy the ORACLE's verifier (ddleq.go:129-153 restated)
            for (c1, c2, _, _), prf in zip(st[b:e], out):
                for i in prf:
                    assert po.verify_ddleq_proof_instance(sk, po.Ciphertext(c1, L2), po.Ciphertext(c2, L2),
                                                          po.DDLEQProofInstance(i.X, i.Y, i.Alpha, i.E, i.F))
            if tamper is not None and b <= tamper < e:
                i = out[tamper - b][0]
                out[tamper - b][0] = pr.DDLEQProofInstance(i.X, i.Y, i.Alpha, i.E, i.F ^ 1)
            return out

        def verify_fn(b, e, proofs):
            return pr.verify_ddleq_proof_batch(gpk, [t[0] for t in st[b:e]], [t[1] for t in st[b:e]], proofs)

        (b, e), proofs, verdicts, all_ok = pd.ddleq_prove_verify_sharded(n_statements, rank, world, prove_fn, verify_fn)
        q.put((rank, b, e, len(proofs), verdicts, all_ok))
        dist.barrier()
        dist.destroy_process_group()
        return

    def prove_fn(b, e):
        out = []
        for (c1, c2, a, bb) in st[b:e]:
            out.append([po.prove_ddleq_instance_xy(sk, po.Ciphertext(c1, L2), po.Ciphertext(c2, L2), a, bb, po.rand_unit(sk.N, rng),
                                                   po.rand_unit(sk.N, rng)) for _ in range(secpar)])
        if tamper is not None and b <= tamper < e:
            out[tamper - b][0].F ^= 1
        return out

    def verify_fn(b, e, proofs):
        return [all(po.verify_ddleq_proof_instance(sk, po.Ciphertext(c1, L2), po.Ciphertext(c2, L2), pf) for pf in pr)
                for (c1, c2, _, _), pr in zip(st[b:e], proofs)]

    (b, e), proofs, verdicts, all_ok = pd.ddleq_prove_verify_sharded(n_statements, rank, world, prove_fn, verify_fn)
    q.put((rank, b, e, len(proofs), verdicts, all_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_statements,secpar,tamper", [(5, 3, None), (4, 2, 3), (1, 2, None)])
def test_ddleq_statements_shard_over_two_ranks_cpu(n_statements, secpar, tamper):
    """BASELINE config 5 on two gloo ranks with the oracle as the kernels: the statements split contiguously (5 over 2 ranks: 3 +
    2; 1 over 2: the second rank's slice is empty), every rank proves and verifies its own slice, and the MIN-reduced flag tells
    EVERY rank when some rank's proof failed (a tampered F on the last statement)."""
    world, port = 2, _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_ddleq_worker, args=(r, world, port, n_statements, secpar, tamper, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    covered = 0
    for rank, b, e, nproofs, verdicts, all_ok in res:
        assert b == covered and nproofs == e - b == len(verdicts)
        covered = e
        assert all_ok == (tamper is None)                      # the same flag on every rank
        for j, v in enumerate(verdicts):
            assert v == (tamper is None or b + j != tamper)
    assert covered == n_statements


@pytest.mark.gpu
def test_ddleq_statements_shard_over_two_ranks_gpu():
    """The same flow with the HIP path proving (pgpu_ddleq_prove_secpar, library-drawn x, y) and verifying each rank's slice --
    two gloo ranks sharing GPU 0; every instance is also checked by the oracle's verifier; a tampered F on the last statement
    is seen by every rank."""
    world, port = 2, _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_ddleq_worker, args=(r, world, port, 5, 3, 4, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    covered = 0
    for rank, b, e, nproofs, verdicts, all_ok in res:
        assert b == covered and nproofs == e - b == len(verdicts)
        covered = e
        assert all_ok is False
        assert verdicts == [b + j != 4 for j in range(e - b)]
    assert covered == 5
