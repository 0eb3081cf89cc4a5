"""Batch forms of the reference's proof protocols, orchestrated on the host over the GPU batch primitives.

Every modular exponentiation, product and inverse below runs on the GPU through the C ABI (`Modulus.exp_batch`,
`mul_batch`, `inv_batch`, `PublicKey.*Batch`, `SecretKey.DecryptBatch`); the host does what the reference does outside
gmp: SHA-256 over `Bytes()` of the transcript, the unreduced integers that only feed the hash, and control flow.

Reference functions mirrored (file:line in /root/reference):
  ExtractRandonness                       operations.go:75-91
  NestedRandomize (randomness supplied)   operations.go:96-118
  PartialDecryptionWithZKP / VerifyProof  thresholdkey.go:225-311
  ProveDDLEQ / VerifyDDLEQProof           ddleq.go:27-153
  RandomOracleDigest / RandomOracleBit    random_oracle.go:10-32
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Optional, Sequence

from .api import ENC_LEVEL_ONE, ENC_LEVEL_TWO, Context, Modulus, PublicKey, SecretKey, ThresholdPublicKey


def _bytes(x: int) -> bytes:
    """gmp.Int.Bytes(): minimal big-endian magnitude, empty for zero."""
    x = abs(int(x))
    return x.to_bytes((x.bit_length() + 7) // 8, "big")


def random_oracle_digest(*values: int) -> bytes:
    """random_oracle.go:20-32 -- the FIRST argument is skipped (:24-26)."""
    return hashlib.sha256(b"".join(_bytes(v) for v in values[1:])).digest()


def random_oracle_bit(*values: int) -> bool:
    """random_oracle.go:10-16."""
    return int.from_bytes(random_oracle_digest(*values), "big") % 2 == 1


def _factorial(n: int) -> int:
    r = 1
    for i in range(1, n + 1):
        r *= i
    return r


class _Mods:
    """Moduli n, n^2, n^3 of a key on one context (cached on the key object)."""

    def __init__(self, ctx: Context, n: int):
        self.n, self.n2, self.n3 = n, n * n, n ** 3
        self.m1, self.m2, self.m3 = Modulus(ctx, n), Modulus(ctx, n * n), Modulus(ctx, n ** 3)


def _mods(key) -> _Mods:
    m = getattr(key, "_mods", None)
    if m is None:
        m = _Mods(key.ctx, key.N)
        key._mods = m
    return m


# ---------------------------------------------------------------------------------------------------------------
# operations.go
# ---------------------------------------------------------------------------------------------------------------

def extract_randomness_batch(sk: SecretKey, cts: Sequence[int], level: int = ENC_LEVEL_ONE) -> List[int]:
    """operations.go:75-91 for each ciphertext."""
    pk, mods = sk.pk, _mods(sk.pk)
    n = pk.N
    ns, ns1, mod1 = (n, mods.n2, mods.m2) if level == ENC_LEVEL_ONE else (mods.n2, mods.n3, mods.m3)
    ns_inv = pow(ns, -1, sk.Lambda)                       # ModInverse(ns, Lambda): one scalar per key
    if pk.G == n + 1:
        # z = G^(-v) c is only used modulo n (:88), and (1 + n)^v = 1 (mod n) whatever v = Decrypt(c) is: z = c (mod n).
        # The same r without the decryption, G^v and the inversion (what the DDLEQ prover on the device does as well).
        z = [int(c) % n for c in cts]
        return mods.m1.exp_batch(z, ns_inv)
    v = sk.DecryptBatch(cts, level=level)
    gv = mod1.exp_batch([pk.G] * len(cts), v)             # G^v mod n^(s+1), per-ciphertext exponent
    gv_inv = mod1.inv_batch(gv)
    z = mod1.mul_batch(gv_inv, list(cts))
    if level == ENC_LEVEL_TWO:                            # z < n^3: bring it below n^2 before the exponentiation mod n
        z = mods.m2.exp_batch(z, 1, base_bytes=2 * mods.m2.nbytes)
    return mods.m1.exp_batch(z, ns_inv, base_bytes=2 * mods.m1.nbytes)


def nested_randomize_with_ab_batch(pk: PublicKey, cts: Sequence[int], a_s: Sequence[int], b_s: Sequence[int]) -> List[int]:
    """operations.go:96-118 with the draws (a, b) supplied: ct^(a^n mod n^2) * b^(n^2) mod n^3 (one interleaved ladder per
    ciphertext on the device: pgpu_nested_randomize_with_ab)."""
    return pk.NestedRandomizeWithABBatch(list(cts), list(a_s), list(b_s))


# ---------------------------------------------------------------------------------------------------------------
# thresholdkey.go: share-decryption proofs
# ---------------------------------------------------------------------------------------------------------------

@dataclass
class PartialDecryptionZKP:
    ID: int
    Decryption: int
    E: int
    Z: int
    C: int


def partial_decryption_with_zkp_batch(tk: ThresholdPublicKey, ID: int, share: int, verification_key: int,
                                      cts: Sequence[int], rs: Sequence[int]) -> List[PartialDecryptionZKP]:
    """thresholdkey.go:225-257 for each ciphertext, with the random r (< n^2, :233) supplied."""
    mods = _mods(tk)
    delta = _factorial(tk.TotalNumberOfDecryptionServers)
    _, dec = tk.PartialDecryptBatch(ID, share, cts)
    c4m = tk.ConstMultBatch(list(cts), 4)                              # Exp(c^4, r, n^2) = (c^4 mod n^2)^r
    a = mods.m2.exp_batch(c4m, list(rs))
    b = mods.m2.exp_batch([verification_key] * len(cts), list(rs))
    out = []
    for c, d, ai, bi, r in zip(cts, dec, a, b, rs):
        c4, ci2 = c ** 4, d ** 2                                       # unreduced: only the hash sees them (:241,248)
        e = int.from_bytes(hashlib.sha256(_bytes(ai) + _bytes(bi) + _bytes(c4) + _bytes(ci2)).digest(), "big")
        out.append(PartialDecryptionZKP(ID, d, e, r + e * delta * share, c))   # computeZ :313-317
    return out


def verify_proof_batch(tk: ThresholdPublicKey, verification_key: int, verification_keys: Sequence[int],
                       proofs: Sequence[PartialDecryptionZKP]) -> List[bool]:
    """thresholdkey.go:278-311 for each proof."""
    mods = _mods(tk)
    cs, ds = [p.C for p in proofs], [p.Decryption for p in proofs]
    zs, es = [p.Z for p in proofs], [p.E for p in proofs]
    c4m = tk.ConstMultBatch(cs, 4)
    a1 = mods.m2.exp_batch(c4m, zs)
    a2 = mods.m2.inv_batch(mods.m2.exp_batch(mods.m2.mul_batch(ds, ds), es))
    a = mods.m2.mul_batch(a1, a2)
    b1 = mods.m2.exp_batch([verification_key] * len(proofs), zs)
    b2 = mods.m2.inv_batch(mods.m2.exp_batch([verification_keys[p.ID - 1] for p in proofs], es))
    b = mods.m2.mul_batch(b1, b2)
    ok = []
    for p, ai, bi in zip(proofs, a, b):
        h = hashlib.sha256(_bytes(ai) + _bytes(bi) + _bytes(p.C ** 4) + _bytes(p.Decryption ** 2)).digest()
        ok.append(p.E == int.from_bytes(h, "big"))
    return ok


# ---------------------------------------------------------------------------------------------------------------
# ddleq.go
# ---------------------------------------------------------------------------------------------------------------

@dataclass
class DDLEQProofInstance:
    X: int
    Y: int
    Alpha: int
    E: int
    F: int


class DDLEQPanic(Exception):
    """ddleq.go:68 panics when the statement is false."""


def prove_ddleq_instances(sk: SecretKey, ct1s: Sequence[int], ct2s: Sequence[int], a_s: Sequence[int], b_s: Sequence[int],
                          xs: Sequence[int], ys: Sequence[int]) -> List[DDLEQProofInstance]:
    """ddleq.go:55-127 for a batch of (statement, instance) pairs with the draws (x, y) supplied."""
    pk, mods = sk.pk, _mods(sk.pk)
    m2, m3, n, n2 = mods.m2, mods.m3, mods.n, mods.n2
    B = len(ct1s)
    an = m2.exp_batch(list(a_s), n)
    sanity = m3.mul_batch(m3.exp_batch(list(ct1s), an), m3.exp_batch(list(b_s), n2))
    if sanity != list(ct2s):
        raise DDLEQPanic("cannot prove re-encryption because inputs are wrong")
    xn = m2.exp_batch(list(xs), n)
    yn2 = m3.exp_batch(list(ys), n2)
    alpha = m3.mul_batch(m3.exp_batch(list(ct1s), xn), yn2)
    chal = [random_oracle_bit(c1, c2, x, y, al) for c1, c2, x, y, al in zip(ct1s, ct2s, xs, ys, alpha)]
    e, f = list(xs), list(ys)
    idx = [i for i in range(B) if chal[i]]
    if idx:
        sel = lambda v: [v[i] for i in idx]
        ainv = m2.inv_batch(sel(a_s))
        e_sel = m2.mul_batch(sel(xs), ainv)
        s = extract_randomness_batch(sk, sel(ct1s), level=ENC_LEVEL_TWO)
        en = m2.exp_batch(e_sel, n)
        c = m3.exp_batch(s, sel(an))
        c = m3.mul_batch(c, sel(b_s))
        c = m3.inv_batch(m3.exp_batch(c, en))
        c = m3.mul_batch(c, m3.exp_batch(s, sel(xn)))
        f_sel = m3.mul_batch(sel(ys), c)
        for j, i in enumerate(idx):
            e[i], f[i] = e_sel[j], f_sel[j]
    return [DDLEQProofInstance(xs[i], ys[i], alpha[i], e[i], f[i]) for i in range(B)]


def verify_ddleq_instances(pk: PublicKey, ct1s: Sequence[int], ct2s: Sequence[int],
                           proofs: Sequence[DDLEQProofInstance]) -> List[bool]:
    """ddleq.go:129-153 for a batch of (statement, instance) pairs."""
    mods = _mods(pk)
    chal = [random_oracle_bit(c1, c2, p.X, p.Y, p.Alpha) for c1, c2, p in zip(ct1s, ct2s, proofs)]
    check = [c2 if ch else c1 for c1, c2, ch in zip(ct1s, ct2s, chal)]
    en = mods.m2.exp_batch([p.E for p in proofs], mods.n)
    fn2 = mods.m3.exp_batch([p.F for p in proofs], mods.n2)
    got = mods.m3.mul_batch(mods.m3.exp_batch(check, en), fn2)
    return [g == p.Alpha for g, p in zip(got, proofs)]


# ---------------------------------------------------------------------------------------------------------------
# whole-protocol forms with the reference's signatures (randomness drawn on the host, as crypto/rand does there)
# ---------------------------------------------------------------------------------------------------------------

def prove_ddleq(sk: SecretKey, secpar: int, ct1: int, ct2: int, a: int, b: int) -> List[DDLEQProofInstance]:
    """ddleq.go:27-40 ProveDDLEQ: `secpar` independent instances for ONE statement, proved as one device batch; what depends
    on the statement only is computed once (pgpu_ddleq_prove_secpar)."""
    return prove_ddleq_batch(sk, secpar, [ct1], [ct2], [a], [b])[0]


def prove_ddleq_batch(sk: SecretKey, secpar: int, ct1s: Sequence[int], ct2s: Sequence[int], a_s: Sequence[int],
                      b_s: Sequence[int]) -> List[List[DDLEQProofInstance]]:
    """ProveDDLEQ (ddleq.go:27-40) for a batch of statements: one proof of `secpar` instances per statement, randomness drawn as
    crypto/rand does there (utils.go:26-49)."""
    S = len(ct1s)
    flat_x, flat_y = sk.pk.random_units(S * secpar), sk.pk.random_units(S * secpar)
    xs = [flat_x[j * secpar:(j + 1) * secpar] for j in range(S)]
    ys = [flat_y[j * secpar:(j + 1) * secpar] for j in range(S)]
    al, es, fs = sk.ProveDDLEQBatch(secpar, ct1s, ct2s, a_s, b_s, xs, ys)
    return [[DDLEQProofInstance(xs[j][k], ys[j][k], al[j][k], es[j][k], fs[j][k]) for k in range(secpar)] for j in range(S)]


def verify_ddleq_proof_batch(pk: PublicKey, ct1s: Sequence[int], ct2s: Sequence[int],
                             proofs: Sequence[Sequence[DDLEQProofInstance]]) -> List[bool]:
    """VerifyDDLEQProof (ddleq.go:44-53) for a batch of statements: one verdict per statement (every instance must verify), all
    instances of all statements in one device batch."""
    if not (len(ct1s) == len(ct2s) == len(proofs)):      # (a statement without a proof must not pass for want of a verdict)
        raise ValueError("verify_ddleq_proof_batch: one ct2 and one proof per statement")
    flat = [(c1, c2, p) for c1, c2, pr in zip(ct1s, ct2s, proofs) for p in pr]
    if not flat:
        return [True] * len(proofs)
    ok = pk.VerifyDDLEQInstancesBatch([t[0] for t in flat], [t[1] for t in flat], [t[2].X for t in flat], [t[2].Y for t in flat],
                                      [t[2].Alpha for t in flat], [t[2].E for t in flat], [t[2].F for t in flat])
    out, k = [], 0
    for pr in proofs:
        out.append(all(ok[k:k + len(pr)]))
        k += len(pr)
    return out


def verify_ddleq_proof(pk: PublicKey, ct1: int, ct2: int, proof: Sequence[DDLEQProofInstance]) -> bool:
    """ddleq.go:44-53 VerifyDDLEQProof: every instance must verify."""
    k = len(proof)
    if k == 0:
        return True
    return all(pk.VerifyDDLEQInstancesBatch([ct1] * k, [ct2] * k, [p.X for p in proof], [p.Y for p in proof],
                                            [p.Alpha for p in proof], [p.E for p in proof], [p.F for p in proof]))


def combine_partial_decryptions_zkp(tk: ThresholdPublicKey, verification_key: int, verification_keys: Sequence[int],
                                    shares: Sequence[Sequence[PartialDecryptionZKP]]) -> List[int]:
    """thresholdkey.go:164-172 CombinePartialDecryptionsZKP for a batch of ciphertexts.  shares[k] = the proofs of server k
    (one per ciphertext, same ciphertext order).  A server whose proof fails for a ciphertext is dropped FOR THAT ciphertext,
    as the reference drops it; ciphertexts are regrouped by their surviving server set and combined per group."""
    B = len(shares[0])
    valid = []
    for srv in shares:
        sid = srv[0].ID
        valid.append(tk.VerifyProofBatch(verification_key, verification_keys[sid - 1], [p.C for p in srv],
                                         [p.Decryption for p in srv], [p.E for p in srv], [p.Z for p in srv]))
    groups = {}
    for i in range(B):
        key = tuple(k for k in range(len(shares)) if valid[k][i])
        groups.setdefault(key, []).append(i)
    out: List[Optional[int]] = [None] * B
    for key, idxs in groups.items():
        sub = [(shares[k][0].ID, [shares[k][i].Decryption for i in idxs]) for k in key]
        res = tk.CombinePartialDecryptionsBatch(sub)      # raises PaillierHipError(-6) when fewer than Threshold survive
        for i, v in zip(idxs, res):
            out[i] = v
    return out  # type: ignore[return-value]


def verify_decryption(tk: ThresholdPublicKey, verification_key: int, verification_keys: Sequence[int], encrypted: Sequence[int],
                      decrypted: Sequence[int], shares: Sequence[Sequence[PartialDecryptionZKP]]) -> None:
    """thresholdkey.go:175-189 VerifyDecryption for a batch; raises ValueError with the reference's messages."""
    for srv in shares:
        for p, c in zip(srv, encrypted):
            if p.C != c:
                raise ValueError("The encrypted message is not the same than the one in the shares")
    res = combine_partial_decryptions_zkp(tk, verification_key, verification_keys, shares)
    if list(res) != list(decrypted):
        raise ValueError("The decrypted message is not the same than the one in the shares")


def verify_partial_decryption(tk: ThresholdPublicKey, ID: int, share: int, verification_key: int,
                              verification_keys: Sequence[int], trials: int = 1) -> None:
    """thresholdkey.go:258-275 VerifyPartialDecryption: a server tests its own share -- encrypt a random m < n, make the
    partial decryption with its proof, verify the proof; raises ValueError("Invalid share") as the reference returns that
    error.  `trials` independent tests in one batch (the reference does one)."""
    import secrets
    n = tk.N
    ms = [secrets.randbelow(n) for _ in range(trials)]
    cts = tk.EncryptBatch(ms)
    rs = [secrets.randbelow(n * n) for _ in range(trials)]             # thresholdkey.go:233
    proofs = partial_decryption_with_zkp_batch(tk, ID, share, verification_key, cts, rs)
    if not all(verify_proof_batch(tk, verification_key, verification_keys, proofs)):
        raise ValueError("Invalid share")


def create_verification_keys(tk: ThresholdPublicKey, v: int, shares: Sequence[int]) -> List[int]:
    """thresholdkey_generator.go:246-254 createVerificationKeys: v_i = v^(l! * s_i) mod n^2 for every server, as one batch
    with per-server exponents (the modexp-heavy part of threshold key generation; l = 100 in the
    reference's tests)."""
    delta = _factorial(tk.TotalNumberOfDecryptionServers)
    return _mods(tk).m2.exp_batch([v] * len(shares), [s * delta for s in shares])
