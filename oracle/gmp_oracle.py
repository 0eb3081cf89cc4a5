"""ctypes loader for oracle/_build/libgmp_oracle.so (CPU ORACLE -- test infrastructure / CPU baseline only).
See oracle/gmp_oracle.c for what it restates.  Never imported by paillier_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libgmp_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_PATH)
        _lib.oracle_gmp_version.restype = C.c_char_p
        for name in ("oracle_decrypt_batch", "oracle_encrypt_batch", "oracle_modexp_batch", "oracle_decrypt_crt_batch",
                     "oracle_ddleq_verify_batch", "oracle_ddleq_prove_batch", "oracle_encrypt_l2_batch", "oracle_nested_randomize_batch",
                     "oracle_threshold_decrypt_batch", "oracle_add_sub_batch", "oracle_const_mult_batch", "oracle_alt_encrypt_batch",
                     "oracle_share_zkp_prove_batch", "oracle_share_zkp_verify_batch", "oracle_decrypt_l2_batch"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _be(v):
    return int(v).to_bytes(max(1, (int(v).bit_length() + 7) // 8), "big")


def _p(a):
    return C.c_void_p(a.ctypes.data)


def decrypt_batch_raw(n, lam, c_buf: np.ndarray, m_stride: int, threads: int = 1):
    """c_buf: uint8[batch, stride] big-endian.  Returns (uint8[batch, m_stride], threads_used)."""
    lib = load()
    nb, lb = _be(n), _be(lam)
    out = np.zeros((c_buf.shape[0], m_stride), dtype=np.uint8)
    used = lib.oracle_decrypt_batch(nb, C.c_size_t(len(nb)), lb, C.c_size_t(len(lb)), C.c_size_t(c_buf.shape[0]),
                                    _p(c_buf), C.c_size_t(c_buf.shape[1]), _p(out), C.c_size_t(m_stride), threads)
    return out, used


def encrypt_batch_raw(n, g, m_buf: np.ndarray, r_buf: np.ndarray, c_stride: int, threads: int = 1):
    lib = load()
    nb, gb = _be(n), _be(g)
    out = np.zeros((m_buf.shape[0], c_stride), dtype=np.uint8)
    used = lib.oracle_encrypt_batch(nb, C.c_size_t(len(nb)), gb, C.c_size_t(len(gb)), C.c_size_t(m_buf.shape[0]),
                                    _p(m_buf), C.c_size_t(m_buf.shape[1]), _p(r_buf), C.c_size_t(r_buf.shape[1]),
                                    _p(out), C.c_size_t(c_stride), threads)
    return out, used


def modexp_batch_raw(mod, e, b_buf: np.ndarray, o_stride: int, threads: int = 1):
    lib = load()
    mb, eb = _be(mod), _be(e)
    out = np.zeros((b_buf.shape[0], o_stride), dtype=np.uint8)
    used = lib.oracle_modexp_batch(mb, C.c_size_t(len(mb)), eb, C.c_size_t(len(eb)), C.c_size_t(b_buf.shape[0]),
                                   _p(b_buf), C.c_size_t(b_buf.shape[1]), _p(out), C.c_size_t(o_stride), threads)
    return out, used


def encrypt_l2_batch_raw(n, g, m_buf: np.ndarray, r_buf: np.ndarray, c_stride: int, threads: int = 1):
    """paillier.go:206-218 at level two: c = G^m r^(n^2) mod n^3 (m_buf: residues modulo n^2)."""
    lib = load()
    nb, gb = _be(n), _be(g)
    out = np.zeros((m_buf.shape[0], c_stride), dtype=np.uint8)
    used = lib.oracle_encrypt_l2_batch(nb, C.c_size_t(len(nb)), gb, C.c_size_t(len(gb)), C.c_size_t(m_buf.shape[0]),
                                       _p(m_buf), C.c_size_t(m_buf.shape[1]), _p(r_buf), C.c_size_t(r_buf.shape[1]),
                                       _p(out), C.c_size_t(c_stride), threads)
    return out, used


def nested_randomize_batch_raw(n, ct_buf: np.ndarray, a_buf: np.ndarray, b_buf: np.ndarray, threads: int = 1):
    """operations.go:96-118 with the draws supplied: ct^(a^n mod n^2) * b^(n^2) mod n^3."""
    lib = load()
    nb = _be(n)
    assert a_buf.shape == b_buf.shape and a_buf.shape[0] == ct_buf.shape[0]
    out = np.zeros_like(ct_buf)
    used = lib.oracle_nested_randomize_batch(nb, C.c_size_t(len(nb)), C.c_size_t(ct_buf.shape[0]), _p(ct_buf),
                                             C.c_size_t(ct_buf.shape[1]), _p(a_buf), _p(b_buf), C.c_size_t(a_buf.shape[1]), _p(out),
                                             C.c_size_t(out.shape[1]), threads)
    return out, used


def threshold_decrypt_batch_raw(n, total_servers, ids, shares, c_buf: np.ndarray, m_stride: int, threads: int = 1,
                                want_partials: bool = False):
    """thresholdkey.go:192-201 + :149-161 per ciphertext: PartialDecrypt under every share of `ids` (1-based server IDs, shares[k]
    the share of ids[k]), then CombinePartialDecryptions.  Returns (plaintexts, threads_used[, partials uint8[batch, t, stride]])."""
    lib = load()
    nb = _be(n)
    t = len(ids)
    ss = max(len(_be(s)) for s in shares)
    sh = np.frombuffer(b"".join(int(s).to_bytes(ss, "big") for s in shares), dtype=np.uint8).copy()
    idarr = (C.c_int * t)(*[int(i) for i in ids])
    out = np.zeros((c_buf.shape[0], m_stride), dtype=np.uint8)
    parts = np.zeros((c_buf.shape[0], t, c_buf.shape[1]), dtype=np.uint8) if want_partials else None
    used = lib.oracle_threshold_decrypt_batch(nb, C.c_size_t(len(nb)), int(total_servers), t, idarr, _p(sh), C.c_size_t(ss),
                                              C.c_size_t(c_buf.shape[0]), _p(c_buf), C.c_size_t(c_buf.shape[1]), _p(out),
                                              C.c_size_t(m_stride), _p(parts) if want_partials else None, threads)
    return (out, used, parts) if want_partials else (out, used)


def _ints_to_be(vals, stride):
    return np.frombuffer(b"".join(int(v).to_bytes(stride, "big") for v in vals), dtype=np.uint8).reshape(len(vals), stride).copy()


def _be_to_ints(arr):
    raw, s = arr.tobytes(), arr.shape[1]
    return [int.from_bytes(raw[i * s:(i + 1) * s], "big") for i in range(arr.shape[0])]


def decrypt_batch(n, lam, cts, threads=1):
    nb = (n.bit_length() + 7) // 8
    out, _ = decrypt_batch_raw(n, lam, _ints_to_be(cts, 2 * nb), nb, threads)
    return _be_to_ints(out)


def encrypt_batch(n, g, ms, rs, threads=1):
    nb = (n.bit_length() + 7) // 8
    out, _ = encrypt_batch_raw(n, g, _ints_to_be(ms, nb), _ints_to_be(rs, nb), 2 * nb, threads)
    return _be_to_ints(out)


def modexp_batch(mod, e, bases, threads=1):
    nb = (mod.bit_length() + 7) // 8
    out, _ = modexp_batch_raw(mod, e, _ints_to_be(bases, nb), nb, threads)
    return _be_to_ints(out)


def decrypt_crt_batch_raw(p, q, c_buf: np.ndarray, m_stride: int, threads: int = 1):
    """Textbook CRT decryption on libgmp (second CPU-baseline line; not the reference's algorithm)."""
    lib = load()
    pb, qb = _be(p), _be(q)
    out = np.zeros((c_buf.shape[0], m_stride), dtype=np.uint8)
    used = lib.oracle_decrypt_crt_batch(pb, C.c_size_t(len(pb)), qb, C.c_size_t(len(qb)), C.c_size_t(c_buf.shape[0]),
                                        _p(c_buf), C.c_size_t(c_buf.shape[1]), _p(out), C.c_size_t(m_stride), threads)
    return out, used


def ddleq_verify_batch(n, ct1s, ct2s, xs, ys, alphas, es, fs, threads=1):
    """ddleq.go:129-153 on libgmp (+ SHA-256 in C) for a batch of (statement, instance) pairs -> list of bool."""
    lib = load()
    s1 = (n.bit_length() + 7) // 8
    s2, s3 = ((n * n).bit_length() + 7) // 8, ((n ** 3).bit_length() + 7) // 8
    B = len(ct1s)
    bufs = [_ints_to_be(ct1s, s3), _ints_to_be(ct2s, s3), _ints_to_be(xs, s1), _ints_to_be(ys, s1), _ints_to_be(alphas, s3),
            _ints_to_be(es, s2), _ints_to_be(fs, s3)]
    ok = np.zeros(B, dtype=np.int32)
    nb = _be(n)
    lib.oracle_ddleq_verify_batch(nb, C.c_size_t(len(nb)), C.c_size_t(B), _p(bufs[0]), _p(bufs[1]), C.c_size_t(s3), _p(bufs[2]),
                                  _p(bufs[3]), C.c_size_t(s1), _p(bufs[4]), _p(bufs[5]), C.c_size_t(s2), _p(bufs[6]), _p(ok), threads)
    return [bool(v) for v in ok]


def ddleq_prove_batch(n, lam, ct1s, ct2s, a_s, b_s, xs, ys, threads=1):
    """ddleq.go:55-127 with (x, y) supplied, on libgmp -> (alphas, es, fs, challenge bits); raises on a false statement."""
    lib = load()
    s1 = (n.bit_length() + 7) // 8
    s2, s3 = ((n * n).bit_length() + 7) // 8, ((n ** 3).bit_length() + 7) // 8
    B = len(ct1s)
    bufs = [_ints_to_be(ct1s, s3), _ints_to_be(ct2s, s3), _ints_to_be(a_s, s1), _ints_to_be(b_s, s1), _ints_to_be(xs, s1),
            _ints_to_be(ys, s1)]
    al, eo, fo = np.zeros((B, s3), np.uint8), np.zeros((B, s2), np.uint8), np.zeros((B, s3), np.uint8)
    bits = np.zeros(B, dtype=np.int32)
    nb, lb = _be(n), _be(lam)
    rc = lib.oracle_ddleq_prove_batch(nb, C.c_size_t(len(nb)), lb, C.c_size_t(len(lb)), C.c_size_t(B), _p(bufs[0]), _p(bufs[1]),
                                      C.c_size_t(s3), _p(bufs[2]), _p(bufs[3]), _p(bufs[4]), _p(bufs[5]), C.c_size_t(s1), _p(al),
                                      _p(eo), C.c_size_t(s2), _p(fo), _p(bits), threads)
    if rc != 0:
        raise RuntimeError("cannot prove re-encryption because inputs are wrong")
    return _be_to_ints(al), _be_to_ints(eo), _be_to_ints(fo), [int(v) for v in bits]


# ---- round 5: Add / Sub / ConstMult, AltEncrypt, the share ZKP (rows of SURVEY 8(a) that had no CPU figure) ----------------------

def add_sub_batch_raw(mod, sub: bool, a_buf: np.ndarray, b_buf: np.ndarray, o_stride: int, threads: int = 1):
    """operations.go:11-29 Add(a, b) / :32-55 Sub(a, b) modulo `mod` = n^(s+1).  Returns (out, threads used, ok int32[batch])."""
    lib = load()
    mb = _be(mod)
    out = np.zeros((a_buf.shape[0], o_stride), dtype=np.uint8)
    ok = np.zeros(a_buf.shape[0], dtype=np.int32)
    used = lib.oracle_add_sub_batch(mb, C.c_size_t(len(mb)), int(bool(sub)), C.c_size_t(a_buf.shape[0]), _p(a_buf),
                                    C.c_size_t(a_buf.shape[1]), _p(b_buf), C.c_size_t(b_buf.shape[1]), _p(out), C.c_size_t(o_stride),
                                    _p(ok), threads)
    return out, used, ok


def const_mult_batch_raw(mod, c_buf: np.ndarray, k, o_stride: int, threads: int = 1):
    """operations.go:58-64 ConstMult: c^k mod `mod`; k an int (shared) or a uint8[batch, k_len] array (one per ciphertext)."""
    lib = load()
    mb = _be(mod)
    out = np.zeros((c_buf.shape[0], o_stride), dtype=np.uint8)
    if isinstance(k, int):
        kb = np.frombuffer(_be(k), dtype=np.uint8).copy()
        k_len, k_stride = kb.size, 0
    else:
        kb, k_len, k_stride = k, k.shape[1], k.shape[1]
    used = lib.oracle_const_mult_batch(mb, C.c_size_t(len(mb)), C.c_size_t(c_buf.shape[0]), _p(c_buf), C.c_size_t(c_buf.shape[1]),
                                       _p(kb), C.c_size_t(k_len), C.c_size_t(k_stride), _p(out), C.c_size_t(o_stride), threads)
    return out, used


def alt_encrypt_batch_raw(n, g, h, k, m_buf: np.ndarray, r_buf: np.ndarray, c_stride: int, threads: int = 1):
    """paillier.go:221-238 AltEncryptWithRAtLevel, level one.  Returns (ciphertexts, threads used, r mod K rows)."""
    lib = load()
    nb, gb, hb, kb = _be(n), _be(g), _be(h), _be(k)
    out = np.zeros((m_buf.shape[0], c_stride), dtype=np.uint8)
    rred = np.zeros_like(r_buf)
    used = lib.oracle_alt_encrypt_batch(nb, C.c_size_t(len(nb)), gb, C.c_size_t(len(gb)), hb, C.c_size_t(len(hb)), kb, C.c_size_t(len(kb)),
                                        C.c_size_t(m_buf.shape[0]), _p(m_buf), C.c_size_t(m_buf.shape[1]), _p(r_buf),
                                        C.c_size_t(r_buf.shape[1]), _p(out), C.c_size_t(c_stride), _p(rred), threads)
    return out, used, rred


def share_zkp_prove_batch_raw(n, total_servers, share, vkey, c_buf: np.ndarray, r_buf: np.ndarray, z_stride: int, threads: int = 1):
    """thresholdkey.go:225-257 with r supplied.  Returns (decryptions, E rows uint8[batch, 32], Z rows, threads used)."""
    lib = load()
    nb, sb, vb = _be(n), _be(share), _be(vkey)
    B, cs = c_buf.shape
    dec, eo, zo = np.zeros((B, cs), np.uint8), np.zeros((B, 32), np.uint8), np.zeros((B, z_stride), np.uint8)
    used = lib.oracle_share_zkp_prove_batch(nb, C.c_size_t(len(nb)), int(total_servers), sb, C.c_size_t(len(sb)), vb, C.c_size_t(len(vb)),
                                            C.c_size_t(B), _p(c_buf), C.c_size_t(cs), _p(r_buf), C.c_size_t(r_buf.shape[1]), _p(dec),
                                            C.c_size_t(cs), _p(eo), _p(zo), C.c_size_t(z_stride), threads)
    return dec, eo, zo, used


def share_zkp_verify_batch_raw(n, vkey, vi, c_buf: np.ndarray, dec_buf: np.ndarray, e_buf: np.ndarray, z_buf: np.ndarray, threads: int = 1,
                               want_ab: bool = False):
    """thresholdkey.go:278-311 VerifyProof for proofs of one server.  Returns (ok int32[batch], threads used[, a | b rows])."""
    lib = load()
    nb, vb, ib = _be(n), _be(vkey), _be(vi)
    B = c_buf.shape[0]
    assert e_buf.shape == (B, 32)
    ok = np.zeros(B, dtype=np.int32)
    ab = np.zeros((B, 2, c_buf.shape[1]), dtype=np.uint8) if want_ab else None
    used = lib.oracle_share_zkp_verify_batch(nb, C.c_size_t(len(nb)), vb, C.c_size_t(len(vb)), ib, C.c_size_t(len(ib)), C.c_size_t(B),
                                             _p(c_buf), C.c_size_t(c_buf.shape[1]), _p(dec_buf), C.c_size_t(dec_buf.shape[1]), _p(e_buf),
                                             _p(z_buf), C.c_size_t(z_buf.shape[1]), _p(ok), _p(ab) if want_ab else None, threads)
    return (ok, used, ab) if want_ab else (ok, used)


def decrypt_l2_batch_raw(n, lam, c_buf: np.ndarray, m_stride: int, threads: int = 1):
    """paillier.go:292-340 at level two (the recovery algorithm for s = 2): c < n^3 -> m < n^2."""
    lib = load()
    nb, lb = _be(n), _be(lam)
    out = np.zeros((c_buf.shape[0], m_stride), dtype=np.uint8)
    used = lib.oracle_decrypt_l2_batch(nb, C.c_size_t(len(nb)), lb, C.c_size_t(len(lb)), C.c_size_t(c_buf.shape[0]), _p(c_buf),
                                       C.c_size_t(c_buf.shape[1]), _p(out), C.c_size_t(m_stride), threads)
    return out, used
