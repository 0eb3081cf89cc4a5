"""ctypes binding of include/paillier_hip.h, mirroring the reference's Go API for the hot path.

Reference methods mirrored (file:line in /root/reference):
  PublicKey.EncryptWithR / EncryptWithRAtLevel   paillier.go:185,206
  SecretKey.Decrypt                              paillier.go:292
  PublicKey.Add / ConstMult                      operations.go:11,58
  gmp.Int.Exp / Mul+Mod                          (ncw/gmp seam)
Batch variants take and return Python ints (tests) or raw fixed-stride big-endian buffers (bench).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

ENC_LEVEL_ONE = 0
ENC_LEVEL_TWO = 1
MEM_HOST = 0
MEM_DEVICE = 1
DECRYPT_DEFAULT = 0
DECRYPT_NO_CRT = 1
LANE_NONUNIT = 1
LANE_NOT_INVERTIBLE = 2


class PaillierHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def library_path() -> str:
    if os.environ.get("PGPU_LIBRARY"):            # an alternative build of the same sources (A/B measurements: tools/)
        return os.environ["PGPU_LIBRARY"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpaillier_hip.so")


_lib = None

_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p
_sz = C.c_size_t
_int = C.c_int

# name -> (restype, argtypes); every symbol include/paillier_hip.h declares
SIGNATURES = {
    "pgpu_last_error": (C.c_char_p, []),
    "pgpu_version": (C.c_char_p, []),
    "pgpu_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "pgpu_ctx_destroy": (None, [_vp]),
    "pgpu_ctx_last_profile": (_int, [_vp, C.POINTER(C.c_double), C.POINTER(_int), C.POINTER(C.c_double)]),
    "pgpu_ctx_set_flag": (_int, [_vp, C.c_char_p, _int]),
    "pgpu_ctx_last_vm_asm": (_int, [_vp]),
    "pgpu_ctx_last_vm_launches": (_int, [_vp]),
    "pgpu_ctx_last_kernel": (C.c_char_p, [_vp]),
    "pgpu_pubkey_create": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, C.POINTER(_vp)]),
    "pgpu_pubkey_destroy": (None, [_vp]),
    "pgpu_pubkey_plain_bytes": (_sz, [_vp, _int]),
    "pgpu_pubkey_cipher_bytes": (_sz, [_vp, _int]),
    "pgpu_seckey_create": (_int, [_vp, _vp, _vp, _sz, C.POINTER(_vp)]),
    "pgpu_seckey_destroy": (None, [_vp]),
    "pgpu_seckey_has_crt": (_int, [_vp]),
    "pgpu_encrypt_with_r": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_encrypt_with_r_sk": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_encrypt": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_random_units": (_int, [_vp, _sz, _vp, _sz, _int]),
    "pgpu_alt_encrypt_with_r": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _int]),
    "pgpu_decrypt": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _int, _int, _vp]),
    "pgpu_add": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_sub": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _int, _vp]),
    "pgpu_add_many": (_int, [_vp, _int, _int, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_sub_many": (_int, [_vp, _int, _int, _sz, _vp, _sz, _vp, _sz, _int, _vp]),
    "pgpu_partial_decrypt": (_int, [_vp, _int, _vp, _sz, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_partial_decrypt_multi": (_int, [_vp, _int, _int, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _int]),
    "pgpu_partial_decrypt_indexed": (_int, [_vp, _int, _int, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _int]),
    "pgpu_partial_decrypt_units": (_int, [_vp, _int, _int, _vp, _vp, _sz, _vp, _sz, _sz, _sz, _vp, _sz, _int]),
    "pgpu_combine_partial_decryptions": (_int, [_vp, _int, _int, _int, _vp, _sz, _vp, _sz, _vp, _sz, _int, _vp]),
    "pgpu_random_oracle_digest": (_int, [_vp, _int, _vp, _vp, _sz, _vp, _int]),
    "pgpu_nested_randomize_with_ab": (_int, [_vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int]),
    "pgpu_ddleq_verify": (_int, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _int]),
    "pgpu_share_zkp_prove": (_int, [_vp, _int, _vp, _sz, _vp, _sz, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _int]),
    "pgpu_share_zkp_verify": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp, _int]),
    "pgpu_ddleq_prove": (_int, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _int]),
    "pgpu_ddleq_prove_secpar": (_int, [_vp, _sz, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _int]),
    "pgpu_const_mult": (_int, [_vp, _int, _sz, _vp, _sz, _vp, _sz, _sz, _vp, _sz, _int]),
    "pgpu_modulus_create": (_int, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "pgpu_modulus_destroy": (None, [_vp]),
    "pgpu_modulus_bytes": (_sz, [_vp]),
    "pgpu_modexp": (_int, [_vp, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _int]),
    "pgpu_modinv": (_int, [_vp, _sz, _vp, _sz, _sz, _vp, _sz, _int, _vp]),
    "pgpu_gob_max_bytes": (_sz, [_sz]),
    "pgpu_gob_unpack": (_int, [_vp, _sz, _vp, _vp, _vp, _sz, _int, _vp, _vp]),
    "pgpu_gob_pack": (_int, [_vp, _sz, _vp, _sz, _int, _int, _int, _vp, _sz, _vp]),
    "pgpu_modmul": (_int, [_vp, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _int]),
}

# test hooks (include/paillier_hip_debug.h: not part of the drop-in boundary)
DEBUG_SIGNATURES = {
    "pgpu_vm_debug_run": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _int, C.POINTER(_int)]),
    "pgpu_pair_debug_run": (_int, [_vp, _vp, _sz, _int, _vp, _sz, _vp, _sz, _sz, _vp, C.POINTER(_int)]),
    "pgpu_plan_query": (_int, [C.c_char_p, _vp, _int, _vp, _int]),
}


def hip_runtimes_mapped() -> List[str]:
    """Paths of the libamdhip64 copies mapped into this process (more than one = two HIP runtimes: the second one to
    initialise finds no device)."""
    seen = []
    try:
        for line in open("/proc/self/maps"):
            f = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64" in f and f not in seen:
                seen.append(f)
    except OSError:
        pass
    return seen


def _preload_hip_runtime():
    """ONE HIP runtime per process.  libpaillier_hip.so needs `libamdhip64.so.7` (by SONAME: the system ROCm's, /opt/rocm/lib);
    PyTorch ships its own copy of the same runtime and asks for it by FILE name (`libamdhip64.so`, found through its RPATH),
    which the loader does not match against an already mapped SONAME.  So `import torch` AFTER this library maps a second HIP
    runtime, and the runtime that initialises second reports "No HIP GPUs are available" (the first holds the device).  The
    other order works by itself: torch's copy carries the SONAME libamdhip64.so.7 and satisfies this library's dependency.
    Fix: when PyTorch is installed but not imported yet, map ITS runtime first (RTLD_GLOBAL), exactly what `import torch`
    would have done; both then share it in either order.  PGPU_HIP_RUNTIME=system keeps the system runtime (a process that
    never imports torch does not care)."""
    if os.environ.get("PGPU_HIP_RUNTIME", "").lower() == "system" or hip_runtimes_mapped():
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                       # fall back to the system runtime: the library still works on its own


def load_library():
    """Loads libpaillier_hip.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise PaillierHipError(-100, f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    if len(hip_runtimes_mapped()) > 1:
        raise PaillierHipError(-100, "two HIP runtimes are mapped into this process (" + ", ".join(hip_runtimes_mapped()) +
                               "): the second one to initialise will find no GPU; import paillier_amd (or torch) before whatever "
                               "loaded the other copy, or link everything against one libamdhip64")
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def version() -> str:
    """pgpu_version()"""
    return load_library().pgpu_version().decode()


def plan_query(what: str, *args: int) -> List[int]:
    """The planning predicates of csrc/plan.hpp by name (pgpu_plan_query; no GPU, no context): tests/test_plan_cpu.py."""
    lib = load_library()
    a = (C.c_uint64 * max(1, len(args)))(*[int(v) for v in args])
    out = (C.c_int64 * 8)()
    n = lib.pgpu_plan_query(what.encode(), a, len(args), out, 8)
    if n < 0:
        raise PaillierHipError(n, lib.pgpu_last_error().decode())
    return [int(out[i]) for i in range(n)]


def gob_unpack_raw(ctx, blobs: Sequence[bytes], out, out_stride: int, mem: int = MEM_HOST):
    """pgpu_gob_unpack: NewCiphertextFromBytes (paillier.go:376-391) for a batch of gob blobs.  `out`: numpy uint8[batch, stride]
    (MEM_HOST) or a device pointer (MEM_DEVICE); ctx may be None for MEM_HOST.  Returns (levels, methods)."""
    lib = load_library()
    n = len(blobs)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in blobs])
    cat = np.frombuffer(b"".join(blobs) or b"\0", dtype=np.uint8)
    levels, methods = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    _check(lib.pgpu_gob_unpack(ctx.h if ctx is not None else None, n, _ptr(cat), _ptr(offs), _ptr(out), out_stride, mem,
                               _ptr(levels), _ptr(methods)))
    return levels, methods


def gob_pack_raw(ctx, batch: int, buf, stride: int, level: int = 0, enc_method: int = 0, mem: int = MEM_HOST) -> List[bytes]:
    """pgpu_gob_pack: Ciphertext.Bytes() (paillier.go:393-401) for every row of a flat big-endian buffer (numpy array for MEM_HOST,
    device pointer for MEM_DEVICE).  Returns one gob blob per ciphertext."""
    lib = load_library()
    cap = batch * lib.pgpu_gob_max_bytes(stride)
    blobs = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(batch + 1, dtype=np.uint64)
    _check(lib.pgpu_gob_pack(ctx.h if ctx is not None else None, batch, _ptr(buf), stride, mem, level, enc_method, _ptr(blobs), cap,
                             _ptr(offs)))
    raw = blobs.tobytes()
    return [raw[int(offs[i]):int(offs[i + 1])] for i in range(batch)]


def _check(rc: int):
    if rc != 0:
        raise PaillierHipError(rc, load_library().pgpu_last_error().decode())


def _be(v: int, n: Optional[int] = None) -> bytes:
    if n is None:
        n = max(1, (v.bit_length() + 7) // 8)
    return int(v).to_bytes(n, "big")


def ints_to_be(vals: Sequence[int], stride: int) -> np.ndarray:
    """Python ints -> uint8[batch, stride], big-endian, left-padded (what Go's Bytes() + padding gives)."""
    buf = b"".join(int(v).to_bytes(stride, "big") for v in vals)
    return np.frombuffer(buf, dtype=np.uint8).reshape(len(vals), stride).copy()


def be_to_ints(arr: np.ndarray) -> List[int]:
    raw = arr.tobytes()
    stride = arr.shape[1]
    return [int.from_bytes(raw[i * stride:(i + 1) * stride], "big") for i in range(arr.shape[0])]


def _ptr(a) -> int:
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)  # raw device pointer (e.g. torch.Tensor.data_ptr())


class Context:
    """One device + stream.  `stream` is a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, own_stream: bool = False):
        """own_stream: the context creates a non-blocking stream of its own (PGPU_STREAM_NEW) -- for overlapping small
        batches issued from several host threads (paillier_amd.concurrent)."""
        self.lib = load_library()
        h = _vp()
        _check(self.lib.pgpu_ctx_create(device, _vp(-1 if own_stream else (stream or 0)), C.byref(h)))
        self.h = h
        self.device = device

    def last_profile(self):
        ms, n, mads = C.c_double(), C.c_int(), C.c_double()
        _check(self.lib.pgpu_ctx_last_profile(self.h, C.byref(ms), C.byref(n), C.byref(mads)))
        return {"vm_ms": ms.value, "vm_launches": n.value, "vm_mads": mads.value,
                "kernel": self.lib.pgpu_ctx_last_kernel(self.h).decode()}

    def set_flag(self, name: str, value: int):
        _check(self.lib.pgpu_ctx_set_flag(self.h, name.encode(), value))

    def pair_debug_run(self, prime: int, prog_words: Sequence[int], mem: np.ndarray, nslots: int, nb: int, lanes: int = 1):
        """Test hook (pgpu_pair_debug_run).  mem: uint32[nslots, 2H, nb].  Returns (memory after the run, constants p|Cadj, H)."""
        pw = np.asarray(prog_words, dtype=np.uint32)
        m = np.ascontiguousarray(mem, dtype=np.uint32).copy()
        pb = np.frombuffer(_be(prime), dtype=np.uint8).copy()
        consts = np.zeros(m.shape[1], dtype=np.uint32)
        h = C.c_int()
        _check(self.lib.pgpu_pair_debug_run(self.h, _ptr(pb), pb.size, lanes, _ptr(pw), pw.size, _ptr(m), nslots, nb, _ptr(consts),
                                            C.byref(h)))
        return m, consts, h.value

    def last_vm_asm(self) -> int:
        return self.lib.pgpu_ctx_last_vm_asm(self.h)

    def last_vm_launches(self) -> int:
        """VM launches of the last call, assembly or compiler-generated (== last_vm_asm() when nothing fell back)"""
        return self.lib.pgpu_ctx_last_vm_launches(self.h)

    def random_oracle_digest_batch(self, columns: Sequence[Sequence[int]]) -> List[bytes]:
        """SHA-256(Bytes(col0[i]) || Bytes(col1[i]) || ...) on the device, one digest per row i."""
        n = len(columns)
        batch = len(columns[0])
        strides = [max(1, max((int(v).bit_length() + 7) // 8 for v in col)) for col in columns]
        bufs = [ints_to_be(col, st) for col, st in zip(columns, strides)]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        sts = (C.c_size_t * n)(*strides)
        out = np.zeros((batch, 32), dtype=np.uint8)
        _check(self.lib.pgpu_random_oracle_digest(self.h, n, ptrs, sts, batch, _ptr(out), MEM_HOST))
        return [out[i].tobytes() for i in range(batch)]

    def close(self):
        if self.h:
            self.lib.pgpu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Modulus:
    """The gmp.Int seam: batched Exp(x, e, N) and Mod(Mul(a, b), N)."""

    def __init__(self, ctx: Context, n: int):
        self.ctx, self.n = ctx, n
        h = _vp()
        b = _be(n)
        _check(ctx.lib.pgpu_modulus_create(ctx.h, b, len(b), C.byref(h)))
        self.h = h
        self.nbytes = ctx.lib.pgpu_modulus_bytes(h)

    def exp_batch(self, bases: Sequence[int], e, base_bytes: Optional[int] = None) -> List[int]:
        """e: one int (shared exponent) or a sequence (one per base)."""
        lib = self.ctx.lib
        bl = base_bytes or self.nbytes
        bb = ints_to_be(bases, bl)
        out = np.zeros((len(bases), self.nbytes), dtype=np.uint8)
        if isinstance(e, int):
            eb = np.frombuffer(_be(e), dtype=np.uint8).copy()
            el, es = eb.size, 0
        else:
            el = max(1, max((int(v).bit_length() + 7) // 8 for v in e))
            eb = ints_to_be(e, el)
            es = el
        _check(lib.pgpu_modexp(self.h, len(bases), _ptr(bb), bl, bl, _ptr(eb), el, es, _ptr(out), self.nbytes, MEM_HOST))
        return be_to_ints(out)

    def exp_raw(self, batch, base, base_stride, e, e_len, e_stride, out, out_stride, mem=MEM_HOST):
        """pgpu_modexp on raw big-endian buffers (numpy arrays or device pointers); e_stride = 0: one shared exponent
        (e is then a bytes-like of e_len bytes on the HOST)."""
        _check(self.ctx.lib.pgpu_modexp(self.h, batch, _ptr(base), base_stride, base_stride, e if isinstance(e, bytes) else _ptr(e),
                                        e_len, e_stride, _ptr(out), out_stride, mem))

    def mul_raw(self, batch, a, a_stride, b, b_stride, out, out_stride, mem=MEM_HOST):
        _check(self.ctx.lib.pgpu_modmul(self.h, batch, _ptr(a), a_stride, a_stride, _ptr(b), b_stride, b_stride, _ptr(out),
                                        out_stride, mem))

    def vm_debug_run(self, prog_words: Sequence[int], mem: np.ndarray, nslots: int, nb: int, use_asm: bool) -> np.ndarray:
        """Test hook (include/paillier_hip_debug.h pgpu_vm_debug_run).  mem: uint32[nslots, WT, nb]; returns the memory after the run."""
        pw = np.asarray(prog_words, dtype=np.uint32)
        m = np.ascontiguousarray(mem, dtype=np.uint32).copy()
        wt = C.c_int()
        _check(self.ctx.lib.pgpu_vm_debug_run(self.h, _ptr(pw), pw.size, _ptr(m), nslots, nb, int(use_asm), C.byref(wt)))
        return m

    def inv_batch(self, xs: Sequence[int], return_status: bool = False):
        """gmp.Int.ModInverse for each x.  Non-units raise PaillierHipError(-5) unless return_status is set, in which case
        they come back as 0 with status LANE_NOT_INVERTIBLE (the other lanes are computed either way)."""
        xb = ints_to_be(xs, self.nbytes)
        out = np.zeros((len(xs), self.nbytes), dtype=np.uint8)
        status = np.zeros(len(xs), dtype=np.int32) if return_status else None
        _check(self.ctx.lib.pgpu_modinv(self.h, len(xs), _ptr(xb), self.nbytes, self.nbytes, _ptr(out), self.nbytes, MEM_HOST,
                                        _ptr(status) if return_status else None))
        return (be_to_ints(out), status) if return_status else be_to_ints(out)

    def mul_batch(self, a: Sequence[int], b: Sequence[int]) -> List[int]:
        lib = self.ctx.lib
        ab, bb = ints_to_be(a, self.nbytes), ints_to_be(b, self.nbytes)
        out = np.zeros((len(a), self.nbytes), dtype=np.uint8)
        _check(lib.pgpu_modmul(self.h, len(a), _ptr(ab), self.nbytes, self.nbytes, _ptr(bb), self.nbytes, self.nbytes,
                               _ptr(out), self.nbytes, MEM_HOST))
        return be_to_ints(out)

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.pgpu_modulus_destroy(self.h)
                self.h = None
        except Exception:
            pass


class PublicKey:
    """paillier.go:46-57 PublicKey{N, G, H, K}; batch variants of the exported methods."""

    def __init__(self, ctx: Context, N: int, G: Optional[int] = None, H: Optional[int] = None, K: Optional[int] = None):
        self.ctx = ctx
        self.N, self.G, self.H, self.K = N, (N + 1 if G is None else G), H, K
        nb, gb = _be(N), _be(self.G)
        hb = _be(H) if H else None
        kb = _be(K) if K else None
        h = _vp()
        _check(ctx.lib.pgpu_pubkey_create(ctx.h, nb, len(nb), gb, len(gb), hb, len(hb) if hb else 0,
                                          kb, len(kb) if kb else 0, C.byref(h)))
        self.h = h

    def plain_bytes(self, level: int = ENC_LEVEL_ONE) -> int:
        return self.ctx.lib.pgpu_pubkey_plain_bytes(self.h, level)

    def cipher_bytes(self, level: int = ENC_LEVEL_ONE) -> int:
        return self.ctx.lib.pgpu_pubkey_cipher_bytes(self.h, level)

    # -- raw-buffer forms (host numpy arrays or device pointers) -----------------------------------
    def encrypt_with_r_raw(self, batch, m, m_stride, r, r_stride, c, c_stride, mem=MEM_HOST, level=ENC_LEVEL_ONE):
        _check(self.ctx.lib.pgpu_encrypt_with_r(self.h, level, batch, _ptr(m), m_stride, _ptr(r), r_stride, _ptr(c),
                                                c_stride, mem))

    def add_raw(self, batch, a, a_stride, b, b_stride, out, out_stride, mem=MEM_HOST, level=ENC_LEVEL_ONE):
        """pgpu_add: out[i] = a[i] * b[i] mod n^(s+1) (operations.go:11-29 with two operands)."""
        _check(self.ctx.lib.pgpu_add(self.h, level, batch, _ptr(a), a_stride, _ptr(b), b_stride, _ptr(out), out_stride, mem))

    def sub_raw(self, batch, a, a_stride, b, b_stride, out, out_stride, mem=MEM_HOST, level=ENC_LEVEL_ONE,
                status: Optional[np.ndarray] = None):
        """pgpu_sub: out[i] = a[i] * b[i]^-1 mod n^(s+1) (operations.go:32-55 with two operands)."""
        _check(self.ctx.lib.pgpu_sub(self.h, level, batch, _ptr(a), a_stride, _ptr(b), b_stride, _ptr(out), out_stride, mem,
                                     _ptr(status) if status is not None else None))

    def const_mult_raw(self, batch, c, c_stride, k, k_len, k_stride, out, out_stride, mem=MEM_HOST, level=ENC_LEVEL_ONE):
        """pgpu_const_mult: out[i] = c[i]^k mod n^(s+1) (operations.go:58-64); k_stride 0 = one shared k (a HOST buffer of k_len
        bytes), else k[i] at k + i * k_stride in `mem`."""
        _check(self.ctx.lib.pgpu_const_mult(self.h, level, batch, _ptr(c), c_stride, _ptr(k), k_len, k_stride, _ptr(out), out_stride, mem))

    def alt_encrypt_with_r_raw(self, batch, m, m_stride, r, r_stride, c, c_stride, r_reduced=None, mem=MEM_HOST, level=ENC_LEVEL_ONE):
        """pgpu_alt_encrypt_with_r (paillier.go:221-238); r_reduced (optional, r_stride bytes per row) receives r mod K."""
        _check(self.ctx.lib.pgpu_alt_encrypt_with_r(self.h, level, batch, _ptr(m), m_stride, _ptr(r), r_stride, _ptr(c), c_stride,
                                                    _ptr(r_reduced) if r_reduced is not None else None, mem))

    # -- int forms ----------------------------------------------------------------------------------
    def EncryptWithRBatch(self, ms: Sequence[int], rs: Sequence[int], level: int = ENC_LEVEL_ONE) -> List[int]:
        """paillier.go:206-218 for each (m, r)."""
        pb, cb = self.plain_bytes(level), self.cipher_bytes(level)
        mb, rb = ints_to_be(ms, pb), ints_to_be(rs, pb)
        out = np.zeros((len(ms), cb), dtype=np.uint8)
        self.encrypt_with_r_raw(len(ms), mb, pb, rb, pb, out, cb, MEM_HOST, level)
        return be_to_ints(out)

    # -- forms that draw their own randomness on the host, as the reference does with crypto/rand -----------------
    def random_units(self, count: int) -> List[int]:
        """utils.go:36-49 GetRandomNumberInMultiplicativeGroup for a batch (pgpu_random_units): uniform r in [0, n) with
        r != 0 and gcd(r, n) = 1; bytes from getrandom(2), rejection as crypto/rand.Int, gcd test on the device."""
        pb = self.plain_bytes()
        out = np.zeros((count, pb), dtype=np.uint8)
        _check(self.ctx.lib.pgpu_random_units(self.h, count, _ptr(out), pb, MEM_HOST))
        return be_to_ints(out)

    def encrypt_raw(self, batch, m, m_stride, c, c_stride, r_out=None, r_stride=0, mem=MEM_HOST, level=ENC_LEVEL_ONE):
        """pgpu_encrypt: the library draws r (see random_units); r_out optionally receives it."""
        _check(self.ctx.lib.pgpu_encrypt(self.h, level, batch, _ptr(m), m_stride, _ptr(c), c_stride,
                                         _ptr(r_out) if r_out is not None else None, r_stride, mem))

    def EncryptBatch(self, ms: Sequence[int], level: int = ENC_LEVEL_ONE, return_r: bool = False):
        """paillier.go:192-194,258-269 EncryptAtLevel for each m (fresh r per ciphertext from the OS CSPRNG)."""
        pb, cb, rb = self.plain_bytes(level), self.cipher_bytes(level), self.plain_bytes()
        mb = ints_to_be(ms, pb)
        out = np.zeros((len(ms), cb), dtype=np.uint8)
        rr = np.zeros((len(ms), rb), dtype=np.uint8) if return_r else None
        self.encrypt_raw(len(ms), mb, pb, out, cb, rr, rb, MEM_HOST, level)
        return (be_to_ints(out), be_to_ints(rr)) if return_r else be_to_ints(out)

    def NestedEncryptBatch(self, ms: Sequence[int]) -> List[int]:
        """paillier.go:199-203: level-one encryption, then level-two encryption of the ciphertext value."""
        return self.EncryptBatch(self.EncryptBatch(ms, ENC_LEVEL_ONE), ENC_LEVEL_TWO)

    def RandomizeBatch(self, cts: Sequence[int], level: int = ENC_LEVEL_ONE) -> List[int]:
        """operations.go:67-69: Add(ct, Encrypt(0))."""
        return self.AddBatch(cts, self.EncryptBatch([0] * len(cts), level), level=level)

    def AltEncryptWithRBatch(self, ms: Sequence[int], rs: Sequence[int], level: int = ENC_LEVEL_ONE):
        """paillier.go:221-238 for each (m, r); returns (ciphertexts, r mod K) -- the reference overwrites r in place."""
        pb, cb = self.plain_bytes(level), self.cipher_bytes(level)
        rb_len = max(1, max((int(r).bit_length() + 7) // 8 for r in rs))
        mb, rb = ints_to_be(ms, pb), ints_to_be(rs, rb_len)
        out = np.zeros((len(ms), cb), dtype=np.uint8)
        rred = np.zeros((len(ms), rb_len), dtype=np.uint8)
        _check(self.ctx.lib.pgpu_alt_encrypt_with_r(self.h, level, len(ms), _ptr(mb), pb, _ptr(rb), rb_len, _ptr(out), cb,
                                                    _ptr(rred), MEM_HOST))
        return be_to_ints(out), be_to_ints(rred)

    def AltEncryptBatch(self, ms: Sequence[int], level: int = ENC_LEVEL_ONE) -> List[int]:
        """paillier.go:244-255 AltEncryptAtLevel: r drawn from Z_n^* as GetRandomNumberInMultiplicativeGroup does (on the
        device-checked path of random_units), then AltEncryptWithRAtLevel."""
        return self.AltEncryptWithRBatch(ms, self.random_units(len(ms)), level)[0]

    def EncryptZeroBatch(self, count: int, level: int = ENC_LEVEL_ONE) -> List[int]:
        """paillier.go:272-274,282-284 EncryptZero / EncryptZeroAtLevel, `count` fresh ones."""
        return self.EncryptBatch([0] * count, level)

    def EncryptOneBatch(self, count: int, level: int = ENC_LEVEL_ONE) -> List[int]:
        """paillier.go:277-279,287-289 EncryptOne / EncryptOneAtLevel, `count` fresh ones."""
        return self.EncryptBatch([1] * count, level)

    def AddBatch(self, *cts: Sequence[int], level: int = ENC_LEVEL_ONE) -> List[int]:
        """operations.go:11-29 Add(cts...), element-wise over the batch: every positional argument is one operand vector."""
        cb = self.cipher_bytes(level)
        st = max(cb, max((int(v).bit_length() + 7) // 8 for op in cts for v in op))
        bufs = [ints_to_be(op, st) for op in cts]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        B = len(cts[0])
        if any(len(op) != B for op in cts):
            raise ValueError("Add: operand vectors differ in length")
        out = np.zeros((B, cb), dtype=np.uint8)
        if len(bufs) == 2:      # the two-operand form takes flat buffers (what the Go shim does too)
            _check(self.ctx.lib.pgpu_add(self.h, level, B, _ptr(bufs[0]), st, _ptr(bufs[1]), st, _ptr(out), cb, MEM_HOST))
        else:
            _check(self.ctx.lib.pgpu_add_many(self.h, level, len(bufs), B, ptrs, st, _ptr(out), cb, MEM_HOST))
        return be_to_ints(out)

    def SubBatch(self, *cts: Sequence[int], level: int = ENC_LEVEL_ONE, return_status: bool = False):
        """operations.go:32-55 Sub(cts...), element-wise: cts[0][i] * prod_{k>=1} cts[k][i]^-1 mod n^(s+1).  A single operand
        is returned unreduced, as the reference does.  A subtrahend that is not a unit raises PaillierHipError(-5) unless
        return_status is set (then that lane is 0 with LANE_NOT_INVERTIBLE and the others are computed)."""
        cb = self.cipher_bytes(level)
        st = max(cb, max((int(v).bit_length() + 7) // 8 for op in cts for v in op))
        bufs = [ints_to_be(op, st) for op in cts]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        B = len(cts[0])
        out = np.zeros((B, st if len(cts) == 1 else cb), dtype=np.uint8)
        if any(len(op) != B for op in cts):
            raise ValueError("Sub: operand vectors differ in length")
        status = np.zeros(B, dtype=np.int32) if return_status else None
        if len(bufs) == 2:
            _check(self.ctx.lib.pgpu_sub(self.h, level, B, _ptr(bufs[0]), st, _ptr(bufs[1]), st, _ptr(out), out.shape[1], MEM_HOST,
                                         _ptr(status) if return_status else None))
        else:
            _check(self.ctx.lib.pgpu_sub_many(self.h, level, len(bufs), B, ptrs, st, _ptr(out), out.shape[1], MEM_HOST,
                                              _ptr(status) if return_status else None))
        return (be_to_ints(out), status) if return_status else be_to_ints(out)

    def ConstMultBatch(self, cts: Sequence[int], k, level: int = ENC_LEVEL_ONE) -> List[int]:
        """operations.go:58-64; k is one int (shared) or one per ciphertext."""
        cb = self.cipher_bytes(level)
        cbuf = ints_to_be(cts, cb)
        out = np.zeros((len(cts), cb), dtype=np.uint8)
        if isinstance(k, int):
            kb = np.frombuffer(_be(k), dtype=np.uint8).copy()
            kl, ks = kb.size, 0
        else:
            kl = max(1, max((int(v).bit_length() + 7) // 8 for v in k))
            kb = ints_to_be(k, kl)
            ks = kl
        _check(self.ctx.lib.pgpu_const_mult(self.h, level, len(cts), _ptr(cbuf), cb, _ptr(kb), kl, ks, _ptr(out), cb,
                                            MEM_HOST))
        return be_to_ints(out)

    def VerifyDDLEQInstancesBatch(self, ct1s, ct2s, xs, ys, alphas, es, fs) -> List[bool]:
        """ddleq.go:129-153 for a batch of (statement, instance) pairs, on the device (hash included)."""
        cb3, pb1, pb2 = self.cipher_bytes(ENC_LEVEL_TWO), self.plain_bytes(ENC_LEVEL_ONE), self.plain_bytes(ENC_LEVEL_TWO)
        B = len(ct1s)
        bufs = [ints_to_be(ct1s, cb3), ints_to_be(ct2s, cb3), ints_to_be(xs, pb1), ints_to_be(ys, pb1), ints_to_be(alphas, cb3),
                ints_to_be(es, pb2), ints_to_be(fs, cb3)]
        ok = np.zeros(B, dtype=np.int32)
        _check(self.ctx.lib.pgpu_ddleq_verify(self.h, B, _ptr(bufs[0]), _ptr(bufs[1]), cb3, _ptr(bufs[2]), _ptr(bufs[3]), pb1,
                                              _ptr(bufs[4]), cb3, _ptr(bufs[5]), pb2, _ptr(bufs[6]), cb3, _ptr(ok), MEM_HOST))
        return [bool(v) for v in ok]

    def ddleq_verify_raw(self, batch, ct1, ct2, x, y, alpha, e, f, ok: np.ndarray, mem=MEM_HOST):
        """pgpu_ddleq_verify on raw buffers with the natural strides (ct/alpha/f: bytes of n^3; x, y: bytes of n; e: bytes of
        n^2); ok: host int32[batch]."""
        cb3, pb1, pb2 = self.cipher_bytes(ENC_LEVEL_TWO), self.plain_bytes(ENC_LEVEL_ONE), self.plain_bytes(ENC_LEVEL_TWO)
        _check(self.ctx.lib.pgpu_ddleq_verify(self.h, batch, _ptr(ct1), _ptr(ct2), cb3, _ptr(x), _ptr(y), pb1, _ptr(alpha), cb3,
                                              _ptr(e), pb2, _ptr(f), cb3, _ptr(ok), mem))

    def nested_randomize_with_ab_raw(self, batch, ct, a, b, out, mem=MEM_HOST):
        """pgpu_nested_randomize_with_ab with the natural strides (ct / out: bytes of n^3; a, b: bytes of n)."""
        cb3, pb1 = self.cipher_bytes(ENC_LEVEL_TWO), self.plain_bytes(ENC_LEVEL_ONE)
        _check(self.ctx.lib.pgpu_nested_randomize_with_ab(self.h, batch, _ptr(ct), cb3, _ptr(a), _ptr(b), pb1, _ptr(out), cb3, mem))

    def NestedRandomizeWithABBatch(self, cts: Sequence[int], a_s: Sequence[int], b_s: Sequence[int]) -> List[int]:
        """operations.go:96-118 for each ciphertext with the draws (a, b) supplied, on the device."""
        cb3, pb1 = self.cipher_bytes(ENC_LEVEL_TWO), self.plain_bytes(ENC_LEVEL_ONE)
        bufs = [ints_to_be(cts, cb3), ints_to_be(a_s, pb1), ints_to_be(b_s, pb1)]
        out = np.zeros((len(cts), cb3), dtype=np.uint8)
        self.nested_randomize_with_ab_raw(len(cts), bufs[0], bufs[1], bufs[2], out)
        return be_to_ints(out)

    def NestedRandomizeBatch(self, cts: Sequence[int]):
        """operations.go:96-118 NestedRandomize: draws a, b in Z_n^* (library CSPRNG) and returns (ciphertexts, a, b)."""
        a_s, b_s = self.random_units(len(cts)), self.random_units(len(cts))
        return self.NestedRandomizeWithABBatch(cts, a_s, b_s), a_s, b_s

    def NestedAddBatch(self, ct1s: Sequence[int], ct2s: Sequence[int]) -> List[int]:
        """operations.go:121-127: level-two ciphertext ^ (level-one ciphertext value)."""
        return self.ConstMultBatch(ct1s, list(ct2s), level=ENC_LEVEL_TWO)

    def NestedSubBatch(self, ct1s: Sequence[int], ct2s: Sequence[int]) -> List[int]:
        """operations.go:130-140: ConstMult(ct1, ModInverse(ct2.C, n^2)) at level two."""
        m2 = getattr(self, "_mod_n2", None)
        if m2 is None:
            m2 = self._mod_n2 = Modulus(self.ctx, self.N * self.N)
        return self.ConstMultBatch(ct1s, m2.inv_batch(list(ct2s)), level=ENC_LEVEL_TWO)

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.pgpu_pubkey_destroy(self.h)
                self.h = None
        except Exception:
            pass


class ThresholdPublicKey(PublicKey):
    """thresholdkey.go:26-32 ThresholdPublicKey{PublicKey; TotalNumberOfDecryptionServers, Threshold, ...}."""

    def __init__(self, ctx: Context, N: int, total: int, threshold: int, G: Optional[int] = None):
        super().__init__(ctx, N, G)
        self.TotalNumberOfDecryptionServers, self.Threshold = total, threshold

    def partial_decrypt_raw(self, share: int, batch, c, c_stride, out, out_stride, mem=MEM_HOST):
        sb = _be(share)
        _check(self.ctx.lib.pgpu_partial_decrypt(self.h, self.TotalNumberOfDecryptionServers, sb, len(sb), batch, _ptr(c),
                                                 c_stride, _ptr(out), out_stride, mem))

    def partial_decrypt_multi_raw(self, shares: Sequence[int], batch, c, c_stride, outs: Sequence, out_stride, mem=MEM_HOST):
        """pgpu_partial_decrypt_multi: the same ciphertext batch under several shares; outs[k] receives server k's partials."""
        bs = [_be(s_) for s_ in shares]
        arr = (C.c_char_p * len(bs))(*bs)
        lens = (C.c_size_t * len(bs))(*[len(b) for b in bs])
        op = (C.c_void_p * len(bs))(*[_ptr(o) for o in outs])
        _check(self.ctx.lib.pgpu_partial_decrypt_multi(self.h, self.TotalNumberOfDecryptionServers, len(bs), arr, lens, batch,
                                                       _ptr(c), c_stride, op, out_stride, mem))

    def partial_decrypt_indexed_raw(self, shares: Sequence[int], share_index: np.ndarray, batch, c, c_stride, out, out_stride,
                                    mem=MEM_HOST):
        """pgpu_partial_decrypt_indexed: unit i uses shares[share_index[i]] (share_index: host int32[batch])."""
        bs = [_be(s_) for s_ in shares]
        arr = (C.c_char_p * len(bs))(*bs)
        lens = (C.c_size_t * len(bs))(*[len(b) for b in bs])
        si = np.ascontiguousarray(share_index, dtype=np.int32)
        _check(self.ctx.lib.pgpu_partial_decrypt_indexed(self.h, self.TotalNumberOfDecryptionServers, len(bs), arr, lens, batch,
                                                         _ptr(c), c_stride, _ptr(si), _ptr(out), out_stride, mem))

    def partial_decrypt_units_raw(self, shares: Sequence[int], batch, c, c_stride, unit_begin: int, unit_end: int, out, out_stride,
                                  mem=MEM_HOST):
        """pgpu_partial_decrypt_units: the (server, ciphertext) units u = s * batch + i of [unit_begin, unit_end) over ONE
        ciphertext batch (server s uses shares[s]; only the shares the range touches are read -- pass 0 for the others);
        ciphertexts wanted under several shares share one chain of squarings."""
        bs = [_be(s_) for s_ in shares]
        arr = (C.c_char_p * len(bs))(*bs)
        lens = (C.c_size_t * len(bs))(*[len(b) for b in bs])
        _check(self.ctx.lib.pgpu_partial_decrypt_units(self.h, self.TotalNumberOfDecryptionServers, len(bs), arr, lens, batch,
                                                       _ptr(c), c_stride, unit_begin, unit_end, _ptr(out), out_stride, mem))

    def combine_raw(self, ids: Sequence[int], batch, partial_ptrs: Sequence[int], stride, m, m_stride, mem=MEM_HOST,
                    status: Optional[np.ndarray] = None):
        n = len(ids)
        ida = (C.c_int * n)(*ids)
        pa_ = (C.c_void_p * n)(*partial_ptrs)
        _check(self.ctx.lib.pgpu_combine_partial_decryptions(self.h, self.TotalNumberOfDecryptionServers, self.Threshold, n,
                                                             ida, batch, pa_, stride, _ptr(m), m_stride, mem,
                                                             _ptr(status) if status is not None else None))

    def share_zkp_prove_raw(self, share: int, verification_key: int, batch, c, c_stride, r, r_stride, dec, dec_stride, e_out, z_out,
                            z_stride, mem=MEM_HOST):
        """pgpu_share_zkp_prove (thresholdkey.go:225-257, r supplied); e_out: 32 bytes per proof."""
        sb, vb = _be(share), _be(verification_key)
        _check(self.ctx.lib.pgpu_share_zkp_prove(self.h, self.TotalNumberOfDecryptionServers, sb, len(sb), vb, len(vb), batch,
                                                 _ptr(c), c_stride, _ptr(r), r_stride, _ptr(dec), dec_stride, _ptr(e_out), _ptr(z_out),
                                                 z_stride, mem))

    def share_zkp_verify_raw(self, verification_key: int, vi: int, batch, c, c_stride, dec, dec_stride, e, z, z_stride, ok: np.ndarray,
                             mem=MEM_HOST):
        """pgpu_share_zkp_verify (thresholdkey.go:278-311) for proofs of one server; ok: host int32[batch]."""
        vb, ib = _be(verification_key), _be(vi)
        _check(self.ctx.lib.pgpu_share_zkp_verify(self.h, vb, len(vb), ib, len(ib), batch, _ptr(c), c_stride, _ptr(dec), dec_stride,
                                                  _ptr(e), _ptr(z), z_stride, _ptr(ok), mem))

    def PartialDecryptionWithZKPBatch(self, ID: int, share: int, verification_key: int, cts: Sequence[int], rs: Sequence[int]):
        """thresholdkey.go:225-257 with r supplied, entirely on the device.  Returns (decryptions, Es, Zs)."""
        cb = self.cipher_bytes()
        zb = cb + 48
        B = len(cts)
        cbuf, rbuf = ints_to_be(cts, cb), ints_to_be(rs, cb)
        dec = np.zeros((B, cb), np.uint8)
        eo = np.zeros((B, 32), np.uint8)
        zo = np.zeros((B, zb), np.uint8)
        sb, vb = _be(share), _be(verification_key)
        _check(self.ctx.lib.pgpu_share_zkp_prove(self.h, self.TotalNumberOfDecryptionServers, sb, len(sb), vb, len(vb), B,
                                                 _ptr(cbuf), cb, _ptr(rbuf), cb, _ptr(dec), cb, _ptr(eo), _ptr(zo), zb, MEM_HOST))
        return be_to_ints(dec), be_to_ints(eo), be_to_ints(zo)

    def VerifyProofBatch(self, verification_key: int, vi: int, cts, decs, es, zs) -> List[bool]:
        """thresholdkey.go:278-311 for a batch of proofs of one server (vi = VerificationKeys[ID-1]), on the device."""
        cb = self.cipher_bytes()
        zb = max(cb + 48, max((int(z).bit_length() + 7) // 8 for z in zs))
        B = len(cts)
        bufs = [ints_to_be(cts, cb), ints_to_be(decs, cb), ints_to_be(es, 32), ints_to_be(zs, zb)]
        ok = np.zeros(B, np.int32)
        vb, ib = _be(verification_key), _be(vi)
        _check(self.ctx.lib.pgpu_share_zkp_verify(self.h, vb, len(vb), ib, len(ib), B, _ptr(bufs[0]), cb, _ptr(bufs[1]), cb,
                                                  _ptr(bufs[2]), _ptr(bufs[3]), zb, _ptr(ok), MEM_HOST))
        return [bool(v) for v in ok]

    def PartialDecryptBatch(self, ID: int, share: int, cts: Sequence[int]):
        """thresholdkey.go:192-201 for each ciphertext; returns (ID, [decryptions])."""
        cb = self.cipher_bytes()
        cbuf = ints_to_be(cts, cb)
        out = np.zeros((len(cts), cb), dtype=np.uint8)
        self.partial_decrypt_raw(share, len(cts), cbuf, cb, out, cb)
        return ID, be_to_ints(out)

    def CombinePartialDecryptionsBatch(self, shares, return_status: bool = False):
        """thresholdkey.go:149-161; shares = [(ID, [decryption per ciphertext]), ...]."""
        cb, pb = self.cipher_bytes(), self.plain_bytes()
        bufs = [ints_to_be(d, cb) for _, d in shares]
        batch = bufs[0].shape[0] if bufs else 0
        out = np.zeros((max(batch, 1), pb), dtype=np.uint8)
        status = np.zeros(max(batch, 1), dtype=np.int32) if return_status else None
        self.combine_raw([i for i, _ in shares], batch, [b.ctypes.data for b in bufs], cb, out, pb, status=status)
        res = be_to_ints(out[:batch])
        return (res, status[:batch]) if return_status else res


class SecretKey:
    """paillier.go:60-63 SecretKey{PublicKey; Lambda}."""

    def __init__(self, ctx: Context, pk: PublicKey, Lambda: int):
        self.ctx, self.pk, self.Lambda = ctx, pk, Lambda
        lb = _be(Lambda)
        h = _vp()
        _check(ctx.lib.pgpu_seckey_create(ctx.h, pk.h, lb, len(lb), C.byref(h)))
        self.h = h

    @property
    def has_crt(self) -> bool:
        return bool(self.ctx.lib.pgpu_seckey_has_crt(self.h))

    def decrypt_raw(self, batch, c, c_stride, m, m_stride, mem=MEM_HOST, level=ENC_LEVEL_ONE, flags=DECRYPT_DEFAULT,
                    status: Optional[np.ndarray] = None):
        _check(self.ctx.lib.pgpu_decrypt(self.h, level, batch, _ptr(c), c_stride, _ptr(m), m_stride, mem, flags,
                                         _ptr(status) if status is not None else None))

    def ddleq_prove_raw(self, batch, ct1, ct2, a, b, x, y, alpha, e, f, mem=MEM_HOST):
        """pgpu_ddleq_prove on raw buffers with the natural strides (see PublicKey.ddleq_verify_raw)."""
        pk = self.pk
        cb3, pb1, pb2 = pk.cipher_bytes(ENC_LEVEL_TWO), pk.plain_bytes(ENC_LEVEL_ONE), pk.plain_bytes(ENC_LEVEL_TWO)
        _check(self.ctx.lib.pgpu_ddleq_prove(self.h, batch, _ptr(ct1), _ptr(ct2), cb3, _ptr(a), _ptr(b), _ptr(x), _ptr(y), pb1,
                                             _ptr(alpha), _ptr(e), pb2, _ptr(f), mem))

    def ddleq_prove_secpar_raw(self, n_statements, secpar, ct1, ct2, a, b, x, y, alpha, e, f, mem=MEM_HOST):
        """pgpu_ddleq_prove_secpar on raw buffers with the natural strides: ct1, ct2, a, b one row per statement; x, y, alpha, e,
        f one row per (statement, instance), statement-major."""
        pk = self.pk
        cb3, pb1, pb2 = pk.cipher_bytes(ENC_LEVEL_TWO), pk.plain_bytes(ENC_LEVEL_ONE), pk.plain_bytes(ENC_LEVEL_TWO)
        _check(self.ctx.lib.pgpu_ddleq_prove_secpar(self.h, n_statements, secpar, _ptr(ct1), _ptr(ct2), cb3, _ptr(a), _ptr(b),
                                                    _ptr(x), _ptr(y), pb1, _ptr(alpha), _ptr(e), pb2, _ptr(f), mem))

    def ProveDDLEQBatch(self, secpar: int, ct1s, ct2s, a_s, b_s, xs, ys):
        """ddleq.go:27-40 ProveDDLEQ for a batch of statements with the draws supplied: xs[j][k], ys[j][k] = the draws of
        instance k of statement j.  The per-statement work (sanity check, a^n, a^-1, ExtractRandonness(ct1)) is done once per
        statement on the device.  Returns (alphas, es, fs), each [statement][instance]."""
        pk = self.pk
        cb3, pb1, pb2 = pk.cipher_bytes(ENC_LEVEL_TWO), pk.plain_bytes(ENC_LEVEL_ONE), pk.plain_bytes(ENC_LEVEL_TWO)
        S = len(ct1s)
        flat = lambda v: [t for row in v for t in row]
        # the C side reads S rows of every per-statement buffer and S * secpar rows of x and y: refuse ragged input here
        if not (len(ct2s) == len(a_s) == len(b_s) == len(xs) == len(ys) == S):
            raise ValueError("ProveDDLEQBatch: ct1s, ct2s, a_s, b_s, xs, ys must have one entry per statement")
        if secpar < 1 or not (all(len(r) == secpar for r in xs) and all(len(r) == secpar for r in ys)):
            raise ValueError("ProveDDLEQBatch: every statement needs exactly secpar draws x and y")
        bufs = [ints_to_be(ct1s, cb3), ints_to_be(ct2s, cb3), ints_to_be(a_s, pb1), ints_to_be(b_s, pb1), ints_to_be(flat(xs), pb1),
                ints_to_be(flat(ys), pb1)]
        B = S * secpar
        al, eo, fo = np.zeros((B, cb3), np.uint8), np.zeros((B, pb2), np.uint8), np.zeros((B, cb3), np.uint8)
        self.ddleq_prove_secpar_raw(S, secpar, *bufs, al, eo, fo, MEM_HOST)
        rows = lambda v: [v[j * secpar:(j + 1) * secpar] for j in range(S)]
        return rows(be_to_ints(al)), rows(be_to_ints(eo)), rows(be_to_ints(fo))

    def EncryptWithRBatch(self, ms: Sequence[int], rs: Sequence[int], level: int = ENC_LEVEL_ONE) -> List[int]:
        """sk.EncryptWithR (SecretKey embeds PublicKey, paillier.go:59-62,185): the key holder's r^n goes through p^2 and q^2
        (pgpu_encrypt_with_r_sk); the ciphertexts are the public path's."""
        pk = self.pk
        pb, cb, rb = pk.plain_bytes(level), pk.cipher_bytes(level), pk.plain_bytes()
        rlen = max(rb, max((int(r).bit_length() + 7) // 8 for r in rs))
        mb, rbuf = ints_to_be(ms, pb), ints_to_be(rs, rlen)
        out = np.zeros((len(ms), cb), dtype=np.uint8)
        _check(self.ctx.lib.pgpu_encrypt_with_r_sk(self.h, level, len(ms), _ptr(mb), pb, _ptr(rbuf), rlen, _ptr(out), cb, MEM_HOST))
        return be_to_ints(out)

    def encrypt_with_r_raw(self, batch, m, m_stride, r, r_stride, c, c_stride, mem=MEM_HOST, level=ENC_LEVEL_ONE):
        _check(self.ctx.lib.pgpu_encrypt_with_r_sk(self.h, level, batch, _ptr(m), m_stride, _ptr(r), r_stride, _ptr(c), c_stride, mem))

    def NestedDecryptBatch(self, cts: Sequence[int]) -> List[int]:
        """paillier.go:344-355: peel the level-two layer, then decrypt at level one (0 stays 0: the reference's edge case)."""
        layer = self.DecryptBatch(cts, level=ENC_LEVEL_TWO)
        nz = [i for i, v in enumerate(layer) if v != 0]
        out = [0] * len(layer)
        if nz:
            for i, v in zip(nz, self.DecryptBatch([layer[i] for i in nz], level=ENC_LEVEL_ONE)):
                out[i] = v
        return out

    def ProveDDLEQInstancesBatch(self, ct1s, ct2s, a_s, b_s, xs, ys):
        """ddleq.go:55-127 for a batch with the draws supplied, on the device.  Returns (alphas, es, fs)."""
        pk = self.pk
        cb3, pb1, pb2 = pk.cipher_bytes(ENC_LEVEL_TWO), pk.plain_bytes(ENC_LEVEL_ONE), pk.plain_bytes(ENC_LEVEL_TWO)
        B = len(ct1s)
        bufs = [ints_to_be(ct1s, cb3), ints_to_be(ct2s, cb3), ints_to_be(a_s, pb1), ints_to_be(b_s, pb1), ints_to_be(xs, pb1),
                ints_to_be(ys, pb1)]
        al, eo, fo = np.zeros((B, cb3), np.uint8), np.zeros((B, pb2), np.uint8), np.zeros((B, cb3), np.uint8)
        _check(self.ctx.lib.pgpu_ddleq_prove(self.h, B, _ptr(bufs[0]), _ptr(bufs[1]), cb3, _ptr(bufs[2]), _ptr(bufs[3]),
                                             _ptr(bufs[4]), _ptr(bufs[5]), pb1, _ptr(al), _ptr(eo), pb2, _ptr(fo), MEM_HOST))
        return be_to_ints(al), be_to_ints(eo), be_to_ints(fo)

    def DecryptBatch(self, cts: Sequence[int], level: int = ENC_LEVEL_ONE, flags: int = DECRYPT_DEFAULT,
                     return_status: bool = False):
        """paillier.go:292-303 for each ciphertext."""
        pb, cb = self.pk.plain_bytes(level), self.pk.cipher_bytes(level)
        cbuf = ints_to_be(cts, cb)
        out = np.zeros((len(cts), pb), dtype=np.uint8)
        status = np.zeros(len(cts), dtype=np.int32)
        self.decrypt_raw(len(cts), cbuf, cb, out, pb, MEM_HOST, level, flags, status)
        res = be_to_ints(out)
        return (res, status) if return_status else res

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.pgpu_seckey_destroy(self.h)
                self.h = None
        except Exception:
            pass
