"""CPU ORACLE (test infrastructure only) -- Python-int restatement of sachaservan/paillier's hot path.

This file is NOT product code.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker.  The product
path (`paillier_amd/`) never imports anything under `oracle/`.

What it restates: every arithmetic formula on the hot path of the reference
(`/root/reference`, a Go package on top of github.com/ncw/gmp -> libgmp), one function
per reference function, each citing the reference file:line it follows.  Python's
arbitrary-precision `int` replaces `gmp.Int`; that is an implementation of integer
arithmetic independent from libgmp, which `oracle/gmp_oracle.c` (same formulas, on the
very libgmp the reference bottoms out in) is cross-checked against.

Third-party arithmetic: github.com/ncw/gmp, version UNPINNED by the reference (no
go.mod / go.sum / vendor dir).  Semantics restated from its math/big-compatible API:
  Exp(x, y, m):  y <= 0 -> 1 ; m nil/0 -> plain power ; else mpz_powm
  Div / Mod:     Euclidean (remainder always non-negative)
  ModInverse:    mpz_invert, the unique value in [0, m)
  Bytes():       minimal big-endian magnitude, empty for 0 ; SetBytes its inverse

Parity pin: the reference cannot be built or run here (no Go toolchain, dependency not
vendored).  The oracle is pinned by every known-answer test the reference's own test
files hold for this path (tests/test_oracle_kats.py lists them with file:line); those
are toy-sized.  At cryptographic sizes (>= 1024 bits) the reference holds NO golden
vector, so parity there is pinned by (1) uniqueness of canonical residues, (2) agreement
of two independent oracles (this file and gmp_oracle.c), (3) round-trip properties the
reference's tests check.  That limit is restated in DESIGN.md.
"""
from __future__ import annotations

import hashlib
import random as _random
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

# ----------------------------------------------------------------------------------
# gmp.Int semantics
# ----------------------------------------------------------------------------------

ENC_LEVEL_ONE = 0  # paillier.go:17-23
ENC_LEVEL_TWO = 1

REGULAR_ENCRYPTION = 0  # paillier.go:29-39
ALTERNATIVE_ENCRYPTION = 1
MIXED_ENCRYPTION = 2

DEFAULT_ENCRYPTION_LEVEL = ENC_LEVEL_ONE  # paillier.go:42


def gmp_exp(x: int, y: int, m: Optional[int]) -> int:
    """gmp.Int.Exp: y<=0 -> 1; m nil or 0 -> x**y; else x**y mod |m|."""
    if y <= 0:
        return 1
    if m is None or m == 0:
        return x ** y
    return pow(x, y, abs(m))


def gmp_div(x: int, y: int) -> int:
    """gmp.Int.Div: Euclidean division (pinned by thresholdkey_test.go:168-177)."""
    q, r = divmod(x, y)  # floor division; remainder has sign of y
    if r < 0:  # only when y < 0
        q += 1
    return q


def gmp_mod(x: int, y: int) -> int:
    """gmp.Int.Mod: Euclidean modulus (non-negative)."""
    return x % abs(y)


def gmp_mod_inverse(x: int, m: int) -> int:
    """gmp.Int.ModInverse = mpz_invert: value in [0, m).  Non-invertible input is
    undefined in the reference (no check at any call site); the oracle raises."""
    return pow(x, -1, m)


def gmp_bytes(x: int) -> bytes:
    """gmp.Int.Bytes: minimal big-endian magnitude; empty for zero."""
    x = abs(x)
    return x.to_bytes((x.bit_length() + 7) // 8, "big")


def factorial(n: int) -> int:
    """utils.go:17-23."""
    ret = 1
    for i in range(1, n + 1):
        ret = ret * i
    return ret


# ----------------------------------------------------------------------------------
# Keys and ciphertexts  (paillier.go:46-69)
# ----------------------------------------------------------------------------------


@dataclass
class PublicKey:
    N: int
    G: int
    H: int = 0
    K: int = 0

    def get_n2(self) -> int:  # paillier.go:72-79
        return self.N * self.N

    def get_n3(self) -> int:  # paillier.go:82-90
        return self.N * self.N * self.N

    def get_moduli_for_level(self, level: int) -> Tuple[int, int, int]:
        """paillier.go:403-414 -> (s, n^s, n^(s+1))."""
        if level == ENC_LEVEL_TWO:
            return 2, self.get_n2(), self.get_n3()
        return 1, self.N, self.get_n2()

    def get_generator_of_qr_for_level(self, level: int) -> int:
        """paillier.go:416-434."""
        if level == ENC_LEVEL_ONE:
            return gmp_exp(self.N - self.H, self.N, self.get_n2())
        return gmp_exp(self.get_n2() - self.H, self.get_n2(), self.get_n3())


@dataclass
class SecretKey(PublicKey):
    Lambda: int = 0


@dataclass
class Ciphertext:
    C: int
    Level: int = ENC_LEVEL_ONE
    EncMethod: int = REGULAR_ENCRYPTION


def L(u: int, n: int) -> int:
    """paillier.go:436-440: Euclidean (floor for n>0) division of u-1 by n."""
    return gmp_div(u - 1, n)


def encrypt_with_r_at_level(pk: PublicKey, m: int, r: int, level: int) -> Ciphertext:
    """paillier.go:206-218: c = G^m * r^(n^s) mod n^(s+1)."""
    _, ns, ns1 = pk.get_moduli_for_level(level)
    gm = gmp_exp(pk.G, m, ns1)
    rn = gmp_exp(r, ns, ns1)
    c = gmp_mod(gm * rn, ns1)
    return Ciphertext(c, level, REGULAR_ENCRYPTION)


def encrypt_with_r(pk: PublicKey, m: int, r: int) -> Ciphertext:
    """paillier.go:185-187."""
    return encrypt_with_r_at_level(pk, m, r, DEFAULT_ENCRYPTION_LEVEL)


def alt_encrypt_with_r_at_level(pk: PublicKey, m: int, r: int, level: int) -> Tuple[Ciphertext, int]:
    """paillier.go:221-238: c = G^m * h_s^(r mod K) mod n^(s+1).  The reference mutates
    the caller's r (paillier.go:228); the reduced r is returned alongside."""
    _, _, ns1 = pk.get_moduli_for_level(level)
    h = pk.get_generator_of_qr_for_level(level)
    r = gmp_mod(r, pk.K)
    gm = gmp_exp(pk.G, m, ns1)
    hr = gmp_exp(h, r, ns1)
    c = gmp_mod(gm * hr, ns1)
    return Ciphertext(c, level, ALTERNATIVE_ENCRYPTION), r


def recovery_algorithm(sk: SecretKey, a: int, s: int) -> int:
    """paillier.go:308-340, statement for statement (including the aliasing of `i` with
    `t1` and its in-place decrement at :323)."""
    i = 0
    for j in range(1, s + 1):
        nj = gmp_exp(sk.N, j, None)
        nj1 = gmp_exp(sk.N, j + 1, None)
        amod = gmp_mod(a, nj1)
        t1 = L(amod, sk.N)
        t2 = int.from_bytes(gmp_bytes(i), "big")
        for k in range(2, j + 1):
            nk = gmp_exp(sk.N, k - 1, None)
            i = i - 1
            t2 = gmp_mod(t2 * i, nj)
            t2 = t2 * nk
            k_fac = gmp_mod_inverse(factorial(k), nj)
            t2 = t2 * k_fac
            t2 = t1 - t2
            t1 = gmp_mod(t2, nj)
        i = t1
    return i


def decrypt(sk: SecretKey, ct: Ciphertext) -> int:
    """paillier.go:292-303."""
    s, ns, ns1 = sk.get_moduli_for_level(ct.Level)
    tmp = gmp_exp(ct.C, sk.Lambda, ns1)
    ml = recovery_algorithm(sk, tmp, s)
    mu = gmp_mod_inverse(sk.Lambda, ns)
    return gmp_mod(ml * mu, ns)


def decrypt_nested_ciphertext_layer(sk: SecretKey, ct: Ciphertext) -> Ciphertext:
    """paillier.go:357-372."""
    if ct.Level == ENC_LEVEL_ONE:
        raise RuntimeError("no nested ciphertexts to recover")  # panic
    value = decrypt(sk, ct)
    if ct.Level == ENC_LEVEL_TWO:
        return Ciphertext(value, ENC_LEVEL_ONE, MIXED_ENCRYPTION)
    raise RuntimeError("not implemented")


def nested_decrypt(sk: SecretKey, ct: Ciphertext) -> int:
    """paillier.go:344-355."""
    ct1 = decrypt_nested_ciphertext_layer(sk, ct)
    if ct1.C == 0:
        return 0
    return decrypt(sk, ct1)


# ----------------------------------------------------------------------------------
# Homomorphic operations (operations.go)
# ----------------------------------------------------------------------------------


def add(pk: PublicKey, *cts: Ciphertext) -> Ciphertext:
    """operations.go:11-29."""
    acc = 1
    level = cts[0].Level
    _, _, ns1 = pk.get_moduli_for_level(level)
    for c in cts:
        acc = gmp_mod(acc * c.C, ns1)
    return Ciphertext(acc, level, MIXED_ENCRYPTION)


def sub(pk: PublicKey, *cts: Ciphertext) -> Ciphertext:
    """operations.go:32-55.  Note: the first operand is returned unreduced when it is
    the only operand (accumulator := cts[0].C)."""
    acc = cts[0].C
    level = cts[0].Level
    _, _, ns1 = pk.get_moduli_for_level(level)
    for i, c in enumerate(cts):
        if i == 0:
            continue
        neg = gmp_mod_inverse(c.C, ns1)
        acc = gmp_mod(acc * neg, ns1)
    return Ciphertext(acc, level, MIXED_ENCRYPTION)


def const_mult(pk: PublicKey, ct: Ciphertext, k: int) -> Ciphertext:
    """operations.go:58-64."""
    _, _, ns1 = pk.get_moduli_for_level(ct.Level)
    return Ciphertext(gmp_exp(ct.C, k, ns1), ct.Level, ct.EncMethod)


def extract_randomness(sk: SecretKey, ct: Ciphertext) -> int:
    """operations.go:75-91 (ExtractRandonness)."""
    _, ns, ns1 = sk.get_moduli_for_level(ct.Level)
    ns_inv = gmp_mod_inverse(ns, sk.Lambda)
    v = decrypt(sk, ct)
    gv = gmp_exp(sk.G, v, ns1)
    gv_inv = gmp_mod_inverse(gv, ns1)
    z = gmp_mod(gv_inv * ct.C, ns1)
    return gmp_exp(z, ns_inv, sk.N)


def nested_randomize_with_ab(pk: PublicKey, ct: Ciphertext, a: int, b: int) -> Ciphertext:
    """operations.go:96-118 with the two random draws (a, b) supplied."""
    if ct.Level != ENC_LEVEL_TWO:
        raise RuntimeError("can only homomorphically randomize doubly encrypted values")
    n, n2, n3 = pk.N, pk.get_n2(), pk.get_n3()
    an = gmp_exp(a, n, n2)
    bn2 = gmp_exp(b, n2, n3)
    r = gmp_exp(ct.C, an, n3)
    r = gmp_mod(r * bn2, n3)
    return Ciphertext(r, ct.Level, REGULAR_ENCRYPTION)


def nested_add(pk: PublicKey, ct1: Ciphertext, ct2: Ciphertext) -> Ciphertext:
    """operations.go:121-127."""
    if ct1.Level != ENC_LEVEL_TWO or ct2.Level != ENC_LEVEL_ONE:
        raise RuntimeError("can only homomorphically add an encrypted value to a doubly encrypted value")
    return const_mult(pk, ct1, ct2.C)


def nested_sub(pk: PublicKey, ct1: Ciphertext, ct2: Ciphertext) -> Ciphertext:
    """operations.go:130-140."""
    if ct1.Level != ENC_LEVEL_TWO or ct2.Level != ENC_LEVEL_ONE:
        raise RuntimeError("can only homomorphically add an encrypted value to a doubly encrypted value")
    _, _, ns1 = pk.get_moduli_for_level(ct2.Level)
    neg = gmp_mod_inverse(ct2.C, ns1)
    return const_mult(pk, ct1, neg)


# ----------------------------------------------------------------------------------
# Random oracle (random_oracle.go)
# ----------------------------------------------------------------------------------


def random_oracle_digest(*values: int) -> bytes:
    """random_oracle.go:20-32: SHA-256 over the minimal big-endian bytes of every
    argument EXCEPT the first (index 0 is skipped at :24-26)."""
    data = b""
    for i, b in enumerate(values):
        if i == 0:
            continue
        data += gmp_bytes(b)
    return hashlib.sha256(data).digest()


def random_oracle_bit(*values: int) -> bool:
    """random_oracle.go:10-16."""
    res = random_oracle_digest(*values)
    return int.from_bytes(res, "big") % 2 == 1


# ----------------------------------------------------------------------------------
# Threshold decryption (thresholdkey.go)
# ----------------------------------------------------------------------------------


@dataclass
class ThresholdPublicKey(PublicKey):
    TotalNumberOfDecryptionServers: int = 0
    Threshold: int = 0
    VerificationKey: int = 0
    VerificationKeys: List[int] = field(default_factory=list)

    def delta(self) -> int:  # thresholdkey.go:70-72
        return factorial(self.TotalNumberOfDecryptionServers)

    def combine_shares_constant(self) -> int:  # thresholdkey.go:63-66
        tmp = 4 * (self.delta() * self.delta())
        return gmp_mod_inverse(tmp, self.N)


@dataclass
class ThresholdSecretKey(ThresholdPublicKey):
    ID: int = 0
    Share: int = 0


@dataclass
class PartialDecryption:
    ID: int
    Decryption: int


@dataclass
class PartialDecryptionZKP(PartialDecryption):
    Key: Optional[ThresholdPublicKey] = None
    E: int = 0
    Z: int = 0
    C: int = 0


class ThresholdError(Exception):
    pass


def verify_partial_decryptions(tk: ThresholdPublicKey, shares: Sequence[PartialDecryption]) -> None:
    """thresholdkey.go:77-89."""
    if len(shares) < tk.Threshold:
        raise ThresholdError("Threshold not meet")
    if len({s.ID for s in shares}) != len(shares):
        raise ThresholdError("two shares has been created by the same server")


def update_lambda(share1_id: int, share2_id: int, lam: int) -> int:
    """thresholdkey.go:91-95 (Euclidean Div)."""
    num = lam * (-share2_id)
    denom = share1_id - share2_id
    return gmp_div(num, denom)


def compute_lambda(tk: ThresholdPublicKey, share: PartialDecryption, shares: Sequence[PartialDecryption]) -> int:
    """thresholdkey.go:99-107."""
    lam = tk.delta()
    for share2 in shares:
        if share2.ID != share.ID:
            lam = update_lambda(share.ID, share2.ID, lam)
    return lam


def tk_exp(a: int, b: int, c: int) -> int:
    """thresholdkey.go:132-138."""
    if b < 0:
        ret = gmp_exp(a, -b, c)
        return gmp_mod_inverse(ret, c)
    return gmp_exp(a, b, c)


def update_cprime(tk: ThresholdPublicKey, cprime: int, lam: int, share: PartialDecryption) -> int:
    """thresholdkey.go:119-124."""
    two_lambda = 2 * lam
    ret = tk_exp(share.Decryption, two_lambda, tk.get_n2())
    return gmp_mod(cprime * ret, tk.get_n2())


def compute_decryption(tk: ThresholdPublicKey, cprime: int) -> int:
    """thresholdkey.go:143-146."""
    l = L(cprime, tk.N)
    return gmp_mod(tk.combine_shares_constant() * l, tk.N)


def combine_partial_decryptions(tk: ThresholdPublicKey, shares: Sequence[PartialDecryption]) -> int:
    """thresholdkey.go:149-161."""
    verify_partial_decryptions(tk, shares)
    cprime = 1
    for share in shares:
        lam = compute_lambda(tk, share, shares)
        cprime = update_cprime(tk, cprime, lam, share)
    return compute_decryption(tk, cprime)


def partial_decrypt(tsk: ThresholdSecretKey, c: int) -> PartialDecryption:
    """thresholdkey.go:192-201 (operands pass through Bytes(): magnitudes)."""
    exp = tsk.Share * (2 * tsk.delta())
    g_exp = int.from_bytes(gmp_bytes(exp), "big")
    g_c = int.from_bytes(gmp_bytes(c), "big")
    g_n2 = int.from_bytes(gmp_bytes(tsk.get_n2()), "big")
    return PartialDecryption(tsk.ID, gmp_exp(g_c, g_exp, g_n2))


def compute_hash(a: int, b: int, c4: int, ci2: int) -> int:
    """thresholdkey.go:319-326."""
    h = hashlib.sha256()
    h.update(gmp_bytes(a))
    h.update(gmp_bytes(b))
    h.update(gmp_bytes(c4))
    h.update(gmp_bytes(ci2))
    return int.from_bytes(h.digest(), "big")


def compute_z(tsk: ThresholdSecretKey, r: int, e: int) -> int:
    """thresholdkey.go:313-317 (plain integers, no reduction)."""
    return r + e * tsk.delta() * tsk.Share


def threshold_public_key_of(tsk: ThresholdSecretKey) -> ThresholdPublicKey:
    """thresholdkey.go:213-222: note only N is copied into the embedded PublicKey."""
    return ThresholdPublicKey(
        N=tsk.N, G=0, H=0, K=0,
        TotalNumberOfDecryptionServers=tsk.TotalNumberOfDecryptionServers,
        Threshold=tsk.Threshold, VerificationKey=tsk.VerificationKey,
        VerificationKeys=list(tsk.VerificationKeys))


def partial_decryption_with_zkp_r(tsk: ThresholdSecretKey, c: int, r: int) -> PartialDecryptionZKP:
    """thresholdkey.go:225-257 with the random r (< n^2, :233) supplied."""
    pd = PartialDecryptionZKP(ID=tsk.ID, Decryption=partial_decrypt(tsk, c).Decryption,
                              Key=threshold_public_key_of(tsk), C=c)
    c4 = gmp_exp(c, 4, None)
    a = gmp_exp(c4, r, tsk.get_n2())
    b = gmp_exp(tsk.VerificationKey, r, tsk.get_n2())
    ci2 = gmp_exp(pd.Decryption, 2, None)
    pd.E = compute_hash(a, b, c4, ci2)
    pd.Z = compute_z(tsk, r, pd.E)
    return pd


def verify_part1(pd: PartialDecryptionZKP) -> int:
    """thresholdkey.go:294-303."""
    n2 = pd.Key.get_n2()
    c4 = gmp_exp(pd.C, 4, None)
    decryption2 = gmp_exp(pd.Decryption, 2, None)
    a1 = gmp_exp(c4, pd.Z, n2)
    a2 = gmp_exp(decryption2, pd.E, n2)
    a2 = gmp_mod_inverse(a2, n2)
    return gmp_mod(a1 * a2, n2)


def verify_part2(pd: PartialDecryptionZKP) -> int:
    """thresholdkey.go:305-311."""
    n2 = pd.Key.get_n2()
    vi = pd.Key.VerificationKeys[pd.ID - 1]
    b1 = gmp_exp(pd.Key.VerificationKey, pd.Z, n2)
    b2 = gmp_exp(vi, pd.E, n2)
    b2 = gmp_mod_inverse(b2, n2)
    return gmp_mod(b1 * b2, n2)


def verify_proof(pd: PartialDecryptionZKP) -> bool:
    """thresholdkey.go:278-292."""
    a = verify_part1(pd)
    b = verify_part2(pd)
    c4 = gmp_exp(pd.C, 4, None)
    ci2 = gmp_exp(pd.Decryption, 2, None)
    return pd.E == compute_hash(a, b, c4, ci2)


def combine_partial_decryptions_zkp(tk: ThresholdPublicKey, shares: Sequence[PartialDecryptionZKP]) -> int:
    """thresholdkey.go:164-172: shares with invalid proofs are silently dropped."""
    ret = [PartialDecryption(s.ID, s.Decryption) for s in shares if verify_proof(s)]
    return combine_partial_decryptions(tk, ret)


# ----------------------------------------------------------------------------------
# DDLEQ proof (ddleq.go)
# ----------------------------------------------------------------------------------


@dataclass
class DDLEQProofInstance:
    X: int
    Y: int
    Alpha: int
    E: int
    F: int


class DDLEQPanic(Exception):
    pass


def prove_ddleq_instance_xy(sk: SecretKey, ct1: Ciphertext, ct2: Ciphertext, a: int, b: int,
                            x: int, y: int) -> DDLEQProofInstance:
    """ddleq.go:55-127 with the two random draws (x, y; :71-79) supplied."""
    n, n2, n3 = sk.N, sk.get_n2(), sk.get_n3()
    sanity = gmp_exp(ct1.C, gmp_exp(a, n, n2), n3)
    sanity = sanity * gmp_exp(b, n2, n3)
    sanity = gmp_mod(sanity, n3)
    if sanity != ct2.C:
        raise DDLEQPanic("cannot prove re-encryption because inputs are wrong")

    xn = gmp_exp(x, n, n2)
    yn2 = gmp_exp(y, n2, n3)
    alpha = gmp_exp(ct1.C, xn, n3)
    alpha = gmp_mod(alpha * yn2, n3)

    chal_bit = random_oracle_bit(ct1.C, ct2.C, x, y, alpha)

    e = x
    if chal_bit:
        ainv = gmp_mod_inverse(a, n2)
        e = gmp_mod(e * ainv, n2)

    f = y
    if chal_bit:
        s = extract_randomness(sk, ct1)
        an = gmp_exp(a, n, n2)
        en = gmp_exp(e, n, n2)
        c = gmp_exp(s, an, n3)
        c = c * b                       # ddleq.go:108 -- not reduced
        c = gmp_exp(c, en, n3)
        c = gmp_mod_inverse(c, n3)
        c = c * gmp_exp(s, xn, n3)      # ddleq.go:112 -- not reduced
        f = f * c
        f = gmp_mod(f, n3)
    return DDLEQProofInstance(x, y, alpha, e, f)


def verify_ddleq_proof_instance(pk: PublicKey, ct1: Ciphertext, ct2: Ciphertext,
                                proof: DDLEQProofInstance) -> bool:
    """ddleq.go:129-153."""
    n, n2, n3 = pk.N, pk.get_n2(), pk.get_n3()
    chal_bit = random_oracle_bit(ct1.C, ct2.C, proof.X, proof.Y, proof.Alpha)
    check = ct2.C if chal_bit else ct1.C
    en = gmp_exp(proof.E, n, n2)
    fn2 = gmp_exp(proof.F, n2, n3)
    check = gmp_exp(check, en, n3)
    check = gmp_mod(check * fn2, n3)
    return proof.Alpha == check


def prove_ddleq_xy(sk: SecretKey, ct1: Ciphertext, ct2: Ciphertext, a: int, b: int,
                   xs: Sequence[int], ys: Sequence[int]) -> List[DDLEQProofInstance]:
    """ddleq.go:27-40 with secpar = len(xs) and the draws supplied."""
    return [prove_ddleq_instance_xy(sk, ct1, ct2, a, b, x, y) for x, y in zip(xs, ys)]


def verify_ddleq_proof(pk: PublicKey, ct1: Ciphertext, ct2: Ciphertext,
                       instances: Sequence[DDLEQProofInstance]) -> bool:
    """ddleq.go:44-53."""
    for inst in instances:
        if not verify_ddleq_proof_instance(pk, ct1, ct2, inst):
            return False
    return True


# ----------------------------------------------------------------------------------
# Threshold key generation pieces that the reference's KATs pin
# (thresholdkey_generator.go) -- setup-time code, restated for fixtures only.
# ----------------------------------------------------------------------------------


def tkg_init_shortcuts(p: int, p1: int, q: int, q1: int) -> Tuple[int, int, int, int]:
    """thresholdkey_generator.go:118-123 -> (n, m, n2, nm)."""
    n = p * q
    m = p1 * q1
    return n, m, n * n, n * m


def tkg_init_d(m: int, n: int) -> int:
    """thresholdkey_generator.go:171-174."""
    return gmp_mod_inverse(m, n) * m


def tkg_compute_share(coeffs: Sequence[int], index: int, nm: int) -> int:
    """thresholdkey_generator.go:213-223 (index is 0-based; authority = index+1)."""
    share = 0
    for i, a in enumerate(coeffs):
        b = gmp_exp(index + 1, i, None)
        share = share + a * b
    return gmp_mod(share, nm)


def tkg_create_verification_keys(v: int, n2: int, total: int, shares: Sequence[int]) -> List[int]:
    """thresholdkey_generator.go:246-254."""
    delta = factorial(total)
    return [gmp_exp(v, s * delta, n2) for s in shares]


# ----------------------------------------------------------------------------------
# Deterministic synthetic keys / inputs for fixtures and benches (NOT in the
# reference: its KeyGen draws from crypto/rand).  Structure follows
# paillier.go:106-179 and thresholdkey_generator.go:47-278.
# ----------------------------------------------------------------------------------

_SMALL_PRIMES = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def _is_probable_prime(n: int, rng: _random.Random, rounds: int = 24) -> bool:
    if n < 2:
        return False
    for p in [2] + _SMALL_PRIMES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(rounds):
        a = rng.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_prime_3mod4(bits: int, rng: _random.Random) -> int:
    while True:
        p = rng.getrandbits(bits) | (1 << (bits - 1)) | (1 << (bits - 2)) | 3
        if _is_probable_prime(p, rng):
            return p


def gen_safe_prime(bits: int, rng: _random.Random) -> Tuple[int, int]:
    """Returns (p, p1) with p = 2*p1+1, both prime, p of `bits` bits."""
    while True:
        p1 = rng.getrandbits(bits - 1) | (1 << (bits - 2)) | (1 << (bits - 3)) | 1
        p = 2 * p1 + 1
        if any(p1 % s == 0 or p % s == 0 for s in _SMALL_PRIMES):
            continue
        if pow(2, p1 - 1, p1) != 1 or pow(2, p - 1, p) != 1:
            continue
        if _is_probable_prime(p1, rng, 8) and _is_probable_prime(p, rng, 8):
            return p, p1


def key_from_primes(p: int, q: int, h_seed_r: int) -> SecretKey:
    """paillier.go:146-178 given the primes and the r whose square gives H."""
    n = p * q
    g = n + 1
    k = 1 << (n.bit_length() // 2)  # K = 2^(secparam/2)
    lam = (p - 1) * (q - 1)
    h = (h_seed_r * h_seed_r) % n  # utils.go:53-59
    return SecretKey(N=n, G=g, H=h, K=k, Lambda=lam)


def keygen_seeded(bits: int, seed: int) -> Tuple[SecretKey, int, int]:
    rng = _random.Random(seed)
    while True:
        p = gen_prime_3mod4(bits // 2, rng)
        q = gen_prime_3mod4(bits // 2, rng)
        if p != q and (p * q).bit_length() == bits:
            break
    r = rand_unit(p * q, rng)
    return key_from_primes(p, q, r), p, q


def rand_unit(n: int, rng: _random.Random) -> int:
    """utils.go:36-49 with a seeded PRNG in place of crypto/rand."""
    import math
    while True:
        r = rng.randrange(0, n)
        if r != 0 and math.gcd(r, n) == 1:
            return r


def threshold_keys_from_primes(p: int, p1: int, q: int, q1: int, total: int, threshold: int,
                               rng: _random.Random) -> List[ThresholdSecretKey]:
    """thresholdkey_generator.go:47-56,118-278 given the safe primes."""
    n, m, n2, nm = tkg_init_shortcuts(p, p1, q, q1)
    d = tkg_init_d(m, n)
    r = rand_unit(n2, rng)
    v = (r * r) % n2  # thresholdkey_generator.go:145-149
    coeffs = [d] + [rng.randrange(0, nm) for _ in range(1, threshold)]
    shares = [tkg_compute_share(coeffs, i, nm) for i in range(total)]
    vks = tkg_create_verification_keys(v, n2, total, shares)
    return [ThresholdSecretKey(N=n, G=n + 1, H=0, K=0, TotalNumberOfDecryptionServers=total,
                               Threshold=threshold, VerificationKey=v, VerificationKeys=list(vks),
                               ID=i + 1, Share=shares[i]) for i in range(total)]
