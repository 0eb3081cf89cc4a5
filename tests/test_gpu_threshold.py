"""GPU parity for threshold decryption (thresholdkey.go): PartialDecrypt and CombinePartialDecryptions, against the
reference's own toy KATs (run THROUGH the GPU path) and against the oracle on the committed safe-prime keys."""
import itertools
import json
import os
import random

import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


def test_reference_kats_on_gpu(ctx):
    import paillier_amd as pa
    # thresholdkey_test.go:58-74 TestDecrypt: N = 101*103, share 862, l = 10, c = 56 -> 40644522
    tk = pa.ThresholdPublicKey(ctx, 101 * 103, total=10, threshold=1)
    assert tk.PartialDecryptBatch(9, 862, [56]) == (9, [40644522])
    # thresholdkey_test.go:267-281 TestDecryption: shares (1, 384111638639), (2, 235243761043), N = 637753, l = 2 -> 100
    tk = pa.ThresholdPublicKey(ctx, 637753, total=2, threshold=2)
    assert tk.CombinePartialDecryptionsBatch([(1, [384111638639]), (2, [235243761043])]) == [100]
    # thresholdkey_test.go:151-166: threshold not met / duplicate ids
    with pytest.raises(pa.PaillierHipError) as ei:
        tk.CombinePartialDecryptionsBatch([(1, [384111638639])])
    assert ei.value.code == -6
    with pytest.raises(pa.PaillierHipError) as ei:
        tk.CombinePartialDecryptionsBatch([(1, [384111638639]), (1, [235243761043])])
    assert ei.value.code == -6


@pytest.mark.parametrize("bits", ["512", "2048"])
def test_partial_decrypt_and_combine_all_subsets(ctx, bits):
    """BASELINE config 4 shape: l = 5, t = 3.  Every 3-subset has at least one negative Lagrange coefficient."""
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"][bits]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    v, vks = int(k["v"], 16), [int(x, 16) for x in k["vks"]]
    tsks = [po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, VerificationKey=v,
                                  VerificationKeys=vks, ID=i + 1, Share=shares[i]) for i in range(total)]
    rng = random.Random(int(bits))
    ms = [0, 1, n - 1] + [rng.randrange(n) for _ in range(9)]
    cts = [po.encrypt_with_r(tsks[0], m, po.rand_unit(n, rng)).C for m in ms]
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    parts = {}
    for t in tsks:
        ID, dec = tk.PartialDecryptBatch(t.ID, t.Share, cts)
        assert dec == [po.partial_decrypt(t, c).Decryption for c in cts]
        parts[ID] = dec
    subsets = list(itertools.combinations(range(1, total + 1), thr)) + [(5, 3, 1), (2, 4, 1, 5), (1, 2, 3, 4, 5)]
    for ids in subsets:
        got = tk.CombinePartialDecryptionsBatch([(i, parts[i]) for i in ids])
        assert got == ms, f"subset {ids}"
        want = [po.combine_partial_decryptions(tsks[0], [po.PartialDecryption(i, parts[i][j]) for i in ids])
                for j in range(3)]
        assert got[:3] == want
    # tampered share: still must equal what the reference computes on the same (wrong) inputs
    bad = [(1, parts[1]), (3, [x ^ 5 for x in parts[3]]), (4, parts[4])]
    got = tk.CombinePartialDecryptionsBatch(bad)
    want = [po.combine_partial_decryptions(tsks[0], [po.PartialDecryption(i, d[j]) for i, d in bad]) for j in range(len(ms))]
    assert got == want


def test_create_verification_keys_kat_and_large(ctx):
    """thresholdkey_generator_test.go:314-324 through the GPU (v = 54, n^2 = 101^2, l = 10, shares (12, 90, 103) -> (6162, 304,
    2728)), and a 100-server key as in thresholdkey_test.go:329-355."""
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    tk = pa.ThresholdPublicKey(ctx, 101, total=10, threshold=1)
    assert pr.create_verification_keys(tk, 54, [12, 90, 103]) == [6162, 304, 2728]
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"]["512"]
    n = int(k["n"], 16)
    tk = pa.ThresholdPublicKey(ctx, n, total=100, threshold=75)
    rng = random.Random(100)
    shares = [rng.randrange(n * n) for _ in range(100)]
    v = int(k["v"], 16)
    assert pr.create_verification_keys(tk, v, shares) == po.tkg_create_verification_keys(v, n * n, 100, shares)


@pytest.mark.parametrize("bits", ["512", "2048"])
def test_partial_decrypt_indexed_units(ctx, bits):
    """pgpu_partial_decrypt_indexed: (share, ciphertext) units of several servers in ONE launch (per-unit exponents) must
    equal the per-server PartialDecrypt, i.e. the committed partials / the oracle."""
    import numpy as np
    import paillier_amd as pa
    from paillier_amd.api import be_to_ints, ints_to_be
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"][bits]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    rng = random.Random(int(bits) + 3)
    pkk = po.PublicKey(N=n, G=n + 1)
    cts = [po.encrypt_with_r(pkk, rng.randrange(n), po.rand_unit(n, rng)).C for _ in range(7)]
    units = [(s, i) for s in (0, 2, 4, 1) for i in range(len(cts))][3:]          # ragged: starts in the middle of a server
    cb = tk.cipher_bytes()
    rows = ints_to_be([cts[i] for _, i in units], cb)
    out = np.zeros((len(units), cb), dtype=np.uint8)
    tk.partial_decrypt_indexed_raw(shares, np.array([s for s, _ in units], dtype=np.int32), len(units), rows, cb, out, cb)
    want = {s: tk.PartialDecryptBatch(s + 1, shares[s], cts)[1] for s in (0, 1, 2, 4)}
    assert be_to_ints(out) == [want[s][i] for s, i in units]
    tsk = po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, ID=3, Share=shares[2])
    assert want[2][:2] == [po.partial_decrypt(tsk, c).Decryption for c in cts[:2]]
    if bits == "2048":
        th = json.load(open(os.path.join(G, "proofs.json")))["threshold"]
        cs = [int(x, 16) for x in th["c"]]
        rows = ints_to_be(cs * 5, cb)
        out = np.zeros((len(cs) * 5, cb), dtype=np.uint8)
        tk.partial_decrypt_indexed_raw(shares, np.repeat(np.arange(5, dtype=np.int32), len(cs)), len(cs) * 5, rows, cb, out, cb)
        assert be_to_ints(out) == [int(x, 16) for row in th["partials"] for x in row]


@pytest.mark.parametrize("bits", ["512", "2048"])
def test_partial_decrypt_multi_servers(ctx, bits):
    """pgpu_partial_decrypt_multi: one ciphertext batch under all five shares (pairs of servers share a launch, the odd one
    runs alone) must equal the per-server PartialDecrypt / the committed partials; also with the pair kernels switched off."""
    import numpy as np
    import paillier_amd as pa
    from paillier_amd.api import be_to_ints, ints_to_be
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"][bits]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    cb = tk.cipher_bytes()
    if bits == "2048":
        th = json.load(open(os.path.join(G, "proofs.json")))["threshold"]
        cts = [int(x, 16) for x in th["c"]]
        want = [[int(x, 16) for x in row] for row in th["partials"]]
    else:
        rng = random.Random(12)
        pkk = po.PublicKey(N=n, G=n + 1)
        cts = [po.encrypt_with_r(pkk, rng.randrange(n), po.rand_unit(n, rng)).C for _ in range(9)]
        want = [tk.PartialDecryptBatch(i + 1, shares[i], cts)[1] for i in range(total)]
    rows = ints_to_be(cts, cb)
    for flag in (1, 0):
        ctx.set_flag("pair", flag)
        try:
            outs = [np.zeros((len(cts), cb), dtype=np.uint8) for _ in shares]
            tk.partial_decrypt_multi_raw(shares, len(cts), rows, cb, outs, cb)
            assert [be_to_ints(o) for o in outs] == want
        finally:
            ctx.set_flag("pair", 1)


@pytest.mark.parametrize("lanes_wanted,nshares", [(1, 3), (4096, 5), (1, 2)])
def test_partial_decrypt_multi_shares_one_chain_of_squarings(ctx, lanes_wanted, nshares):
    """A batch that fills the chip takes pgpu_partial_decrypt_multi's shared chain: right-to-left sliding windows into Yao
    buckets, ONE chain of squarings for all the shares (vm_emit.cpp emit_multi_exp_shared_base).  Forced here for a small batch
    (lanes_wanted = 1: the two-lane pair kernel; 4096: the four-lane one) and compared with c^(2 Delta s_i) mod n^2
    (thresholdkey.go:192-201) from Python and with the separate ladders (flag off)."""
    import numpy as np
    import paillier_amd as pa
    from paillier_amd.api import be_to_ints, ints_to_be
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"]["2048"]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]][:nshares]
    if nshares == 2:
        shares[1] = (1 << 300) + 12345          # a short exponent next to a long one (still on the pair path)
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    cb = tk.cipher_bytes()
    rng = random.Random(77 + nshares)
    n2 = n * n
    cts = [rng.randrange(n2) for _ in range(290)] + [0, 1, n2 - 1, n, 3 * n]
    rows = ints_to_be(cts, cb)
    got = {}
    try:
        ctx.set_flag("lanes_wanted", lanes_wanted)
        for flag in (1, 0):
            ctx.set_flag("shared_chain", flag)
            outs = [np.zeros((len(cts), cb), dtype=np.uint8) for _ in shares]
            tk.partial_decrypt_multi_raw(shares, len(cts), rows, cb, outs, cb)
            got[flag] = [be_to_ints(o) for o in outs]
    finally:
        ctx.set_flag("shared_chain", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert got[1] == got[0]
    delta2 = 2 * 120
    for s, col in zip(shares, got[1]):
        assert col[:24] + col[-5:] == [pow(c, delta2 * s, n2) for c in cts[:24] + cts[-5:]]


@pytest.mark.parametrize("lens,lanes_wanted", [((300, 41), 0), ((7, 250, 130), 1), ((513,), 0)])
def test_partial_decrypt_indexed_runs_of_one_share(ctx, lens, lanes_wanted):
    """A rank's shard of the threshold flow (paillier_amd/dist.py) is one to three RUNS of units under the same share:
    pgpu_partial_decrypt_indexed turns those into shared-exponent ladders, one program segment per run, in ONE launch.
    Ragged run lengths, both pair kernels; equal to the per-unit ladders (flag off) and to c^(2 Delta s_i) mod n^2."""
    import numpy as np
    import paillier_amd as pa
    from paillier_amd.api import be_to_ints, ints_to_be
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"]["2048"]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    cb = tk.cipher_bytes()
    rng = random.Random(91 + len(lens))
    n2 = n * n
    which = [4, 1, 3][:len(lens)]                       # share numbers of the runs
    idx = np.concatenate([np.full(m, w, dtype=np.int32) for m, w in zip(lens, which)])
    cts = [rng.randrange(n2) for _ in range(len(idx))]
    cts[0], cts[-1] = 0, n
    rows = ints_to_be(cts, cb)
    got = {}
    try:
        ctx.set_flag("lanes_wanted", lanes_wanted)
        for flag in (1, 0):
            ctx.set_flag("shared_chain", flag)
            out = np.zeros((len(cts), cb), dtype=np.uint8)
            tk.partial_decrypt_indexed_raw(shares, idx, len(cts), rows, cb, out, cb)
            got[flag] = be_to_ints(out)
    finally:
        ctx.set_flag("shared_chain", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert got[1] == got[0]
    pick = list(range(0, len(cts), 37)) + [len(cts) - 1] + [sum(lens[:i + 1]) - 1 for i in range(len(lens))] + [sum(lens[:i]) for i in range(len(lens))]
    assert [got[1][i] for i in pick] == [pow(cts[i], 2 * 120 * shares[int(idx[i])], n2) for i in pick]


@pytest.mark.parametrize("B,ub,ue,lanes_wanted", [(300, 0, 450, 0),        # N = 2, rank 0: server 0 whole + half of server 1
                                                   (300, 450, 900, 1),      # N = 2, rank 1: half of server 1 + server 2 whole
                                                   (300, 225, 450, 0),      # N = 4, rank 1: disjoint ciphertext ranges of two servers
                                                   (257, 100, 700, 0),      # three servers touched: tail, whole, head
                                                   (300, 0, 900, 0),        # every unit of a small batch: a chain per share (three groups)
                                                   (300, 0, 900, 8192),     # ... room for two groups: {s0}, {s1, s2}
                                                   (300, 0, 900, 4096),     # ... for one: one chain for the three shares
                                                   (300, 310, 590, 1)])     # inside one server
def test_partial_decrypt_units_of_a_rank(ctx, B, ub, ue, lanes_wanted):
    """pgpu_partial_decrypt_units: the contiguous, server-major unit range of ONE rank of the sharded threshold flow over the
    whole ciphertext batch.  Ciphertexts the range wants under several shares walk ONE chain of squarings (N = 2: a rank holds
    one server whole and half of the next).  Equal to the per-server ladders (shared_chain off / pgpu_partial_decrypt) and to
    c^(2 Delta s_i) mod n^2; ragged sizes, c = 0 and c = n included, both pair kernels."""
    import numpy as np
    import paillier_amd as pa
    from paillier_amd.api import be_to_ints, ints_to_be
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"]["2048"]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    sh = [shares[0], shares[2], shares[4]]               # servers 1, 3, 5 hold unit blocks 0, 1, 2
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    cb = tk.cipher_bytes()
    rng = random.Random(B + ub)
    n2 = n * n
    cts = [rng.randrange(n2) for _ in range(B)]
    cts[0], cts[-1] = 0, n
    rows = ints_to_be(cts, cb)
    got = {}
    kernels = {}
    try:
        ctx.set_flag("lanes_wanted", lanes_wanted)
        # (shared_chain, lanes8): a shard this small takes the eight-lane pair kernel (vm_asm_19_96: 76-limb digits, the radix
        # changed on the way in and out) unless lanes8 is switched off -- then the four-lane one
        # (the third flag: bucket products as VM_MULS -- the current power never leaves the registers -- or as LOAD / MUL / STORE)
        for flag in ((1, 1, 1), (1, 0, 1), (0, 1, 1), (1, 1, 0), (1, 0, 0)):
            ctx.set_flag("shared_chain", flag[0])
            ctx.set_flag("lanes8", flag[1])
            ctx.set_flag("muls", flag[2])
            out = np.zeros((ue - ub, cb), dtype=np.uint8)
            tk.partial_decrypt_units_raw(sh, B, rows, cb, ub, ue, out, cb)
            got[flag] = be_to_ints(out)
            kernels[flag] = ctx.last_profile()["kernel"]
            if flag[0]:
                assert ctx.last_vm_asm() > 0
    finally:
        ctx.set_flag("shared_chain", 1)
        ctx.set_flag("lanes8", 1)
        ctx.set_flag("muls", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert got[(1, 1, 1)] == got[(1, 0, 1)] == got[(0, 1, 1)] == got[(1, 1, 0)] == got[(1, 0, 0)]
    if lanes_wanted == 0:
        assert kernels[(1, 1, 1)] == "vm_asm_19_96" and kernels[(1, 0, 1)] == "vm_asm_37_64", kernels
    got = {1: got[(1, 1, 1)]}
    # against the one-server entry point on the same units, and against pow on a sample
    u = ub
    while u < ue:
        s, i0 = divmod(u, B)
        cnt = min(ue - u, B - i0)
        one = np.zeros((cnt, cb), dtype=np.uint8)
        tk.partial_decrypt_raw(sh[s], cnt, rows[i0:i0 + cnt], cb, one, cb)
        assert be_to_ints(one) == got[1][u - ub:u - ub + cnt], (s, i0, cnt)
        u += cnt
    pick = sorted(set(list(range(ub, ue, 53)) + [ub, ue - 1]))
    assert [got[1][u - ub] for u in pick] == [pow(cts[u % B], 2 * 120 * sh[u // B], n2) for u in pick]
    with pytest.raises(pa.PaillierHipError):
        tk.partial_decrypt_units_raw(sh, B, rows, cb, 5, 3 * B + 1, np.zeros((3 * B, cb), dtype=np.uint8), cb)
