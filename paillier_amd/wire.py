"""The reference's wire format for ciphertexts: `Ciphertext.Bytes()` / `PublicKey.NewCiphertextFromBytes`
(paillier.go:374-401) = encoding/gob of

    type Ciphertext struct { C *gmp.Int; Level EncryptionLevel; EncMethod EncryptionMethod }

written by a FRESH gob.Encoder per ciphertext, i.e. every blob carries its type definitions:

    message 1   type -65: struct "Ciphertext" { C: type 66, Level: int, EncMethod: int }
    message 2   type -66: GobEncoder type "Int"             (gmp.Int implements gob.GobEncoder, as math/big.Int does)
    message 3   value of type 65: field C = GobEncode() bytes = [version 1 << 1 | sign] ++ big-endian magnitude;
                zero-valued fields (Level = EncLevelOne, EncMethod = RegularEncryption, a nil C) are omitted, as gob does

Type ids: gob numbers user types from 65 in order of first use in the PROCESS; the bytes below are those of a process whose
first gob type is Ciphertext (the reference's own TestToFromBytes).  The decoder accepts any ids, as Go's does.

Status: written from the gob specification (pkg.go.dev/encoding/gob, "Encoding Details") -- its documented example vector
is reproduced byte for byte by the primitives here (tests/test_wire.py) -- and from math/big's GobEncode layout, which
ncw/gmp mirrors.  NOT cross-checked against a Go toolchain (none in the build image; ncw/gmp is unpinned and un-vendored).

`pack_gob_batch` / `unpack_gob_batch` move whole batches between this format and the C ABI's flat big-endian buffers.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

FIRST_USER_ID = 65
T_INT = 2          # gob's built-in type id of int


def enc_uint(v: int) -> bytes:
    if v < 128:
        return bytes([v])
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def enc_int(v: int) -> bytes:
    return enc_uint(((~v) << 1) | 1 if v < 0 else v << 1)


def enc_string(s: bytes) -> bytes:
    return enc_uint(len(s)) + s


def _common(name: str, tid: int) -> bytes:
    return b"\x01" + enc_string(name.encode()) + b"\x01" + enc_int(tid) + b"\x00"      # CommonType{Name, Id}


def struct_typedef(tid: int, name: str, fields: Sequence[Tuple[str, int]]) -> bytes:
    """message body of a struct type definition: (-id, wireType{StructT: ...})"""
    body = enc_int(-tid) + b"\x03" + b"\x01" + _common(name, tid) + b"\x01" + enc_uint(len(fields))
    for fname, ftid in fields:
        body += b"\x01" + enc_string(fname.encode()) + b"\x01" + enc_int(ftid) + b"\x00"
    return body + b"\x00\x00"


def gobencoder_typedef(tid: int, name: str) -> bytes:
    """message body of a GobEncoder type definition: (-id, wireType{GobEncoderT: ...}) -- field 4 of wireType"""
    return enc_int(-tid) + b"\x05" + b"\x01" + _common(name, tid) + b"\x00\x00"


def message(body: bytes) -> bytes:
    return enc_uint(len(body)) + body


def gmp_int_gob(v: int) -> bytes:
    """gmp.Int.GobEncode (= math/big.Int.GobEncode): version 1, sign bit, big-endian magnitude."""
    a = abs(v)
    return bytes([(1 << 1) | (1 if v < 0 else 0)]) + a.to_bytes((a.bit_length() + 7) // 8, "big")


def ciphertext_to_gob(c: int, level: int = 0, enc_method: int = 0) -> bytes:
    """Ciphertext{C, Level, EncMethod}.Bytes() (paillier.go:393-401)."""
    out = message(struct_typedef(FIRST_USER_ID, "Ciphertext", [("C", FIRST_USER_ID + 1), ("Level", T_INT), ("EncMethod", T_INT)]))
    out += message(gobencoder_typedef(FIRST_USER_ID + 1, "Int"))
    val = enc_int(FIRST_USER_ID) + b"\x01" + enc_string(gmp_int_gob(c))
    delta = 1
    for fv in (level, enc_method):
        if fv:
            val += enc_uint(delta) + enc_int(fv)
            delta = 1
        else:
            delta += 1
    return out + message(val + b"\x00")


class GobError(ValueError):
    pass


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.i = data, 0

    def byte(self) -> int:
        if self.i >= len(self.d):
            raise GobError("unexpected end of gob data")
        self.i += 1
        return self.d[self.i - 1]

    def take(self, n: int) -> bytes:
        if self.i + n > len(self.d):
            raise GobError("unexpected end of gob data")
        self.i += n
        return self.d[self.i - n:self.i]

    def uint(self) -> int:
        b = self.byte()
        if b < 128:
            return b
        n = 256 - b
        if n > 8:
            raise GobError("bad unsigned integer")
        return int.from_bytes(self.take(n), "big")

    def int(self) -> int:
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def done(self) -> bool:
        return self.i >= len(self.d)


def _read_common(r: _Reader):
    name, tid, f = None, None, -1
    while True:
        d = r.uint()
        if d == 0:
            return name, tid
        f += d
        if f == 0:
            name = r.take(r.uint()).decode()
        elif f == 1:
            tid = r.int()
        else:
            raise GobError("unknown CommonType field")


def ciphertext_from_gob(data: bytes) -> Tuple[int, int, int]:
    """NewCiphertextFromBytes (paillier.go:376-391): returns (C, Level, EncMethod).  Accepts any type ids and field order, as
    gob does (fields are matched by NAME).  A wire field the struct does not have is skipped when its extent is known without
    its definition (gob's basic types, GobEncoder values), as Go's decoder skips it; a value without C is an error (Go would
    hand back a Ciphertext with a nil C)."""
    if len(data) == 0:
        raise GobError("no data provided")                 # paillier.go:377
    r = _Reader(data)
    structs, gobenc = {}, set()
    while not r.done():
        body = _Reader(r.take(r.uint()))
        tid = body.int()
        if tid < 0:                                        # type definition
            f = body.uint() - 1                            # which wireType field
            if f == 2:                                     # StructT
                fields, g = [], -1
                name = None
                while True:
                    d = body.uint()
                    if d == 0:
                        break
                    g += d
                    if g > 1:
                        raise GobError("structType field index out of range")
                    if g == 0:
                        name, _ = _read_common(body)
                    else:
                        for _ in range(body.uint()):
                            fname, ftid, h = None, None, -1
                            while True:
                                d2 = body.uint()
                                if d2 == 0:
                                    break
                                h += d2
                                if h > 1:
                                    raise GobError("fieldType field index out of range")
                                if h == 0:
                                    fname = body.take(body.uint()).decode()
                                else:
                                    ftid = body.int()
                            fields.append((fname, ftid))
                structs[-tid] = (name, fields)
            elif f == 4:                                   # GobEncoderT
                body.uint()
                _read_common(body)
                gobenc.add(-tid)
            else:
                raise GobError("unsupported gob wire type")
            continue
        if tid not in structs:
            raise GobError("value of an undefined type")
        _, fields = structs[tid]
        c, level, method, f = None, 0, 0, -1
        while True:
            d = body.uint()
            if d == 0:
                break
            f += d
            if f >= len(fields):
                raise GobError("field index out of range")
            fname, ftid = fields[f]
            if fname == "C":
                if ftid not in gobenc:
                    raise GobError("field C is not a GobEncoder type")
                raw = body.take(body.uint())
                if not raw or raw[0] >> 1 != 1:
                    raise GobError("Int.GobDecode: encoding version not supported")
                c = int.from_bytes(raw[1:], "big")
                if raw[0] & 1:
                    c = -c
            elif fname == "Level":
                level = body.int()
            elif fname == "EncMethod":
                method = body.int()
            elif ftid in gobenc or ftid in (5, 6):         # a field Ciphertext lacks: skipped, as Go's decoder does
                body.take(body.uint())
            elif ftid in (1, 2, 3, 4):
                body.uint()
            elif ftid == 7:
                body.uint(), body.uint()
            else:
                raise GobError(f"cannot skip field {fname} (not in Ciphertext, not a basic type)")
        if c is None:
            raise GobError("the value has no field C")
        return c, level, method
    raise GobError("no value in gob data")


def pack_gob_batch(blobs: Sequence[bytes], stride: int) -> Tuple[np.ndarray, List[int], List[int]]:
    """gob blobs -> (uint8[batch, stride] big-endian buffer for the C ABI, levels, methods)"""
    vals = [ciphertext_from_gob(b) for b in blobs]
    buf = np.frombuffer(b"".join(int(v[0]).to_bytes(stride, "big") for v in vals), dtype=np.uint8).reshape(len(vals), stride).copy()
    return buf, [v[1] for v in vals], [v[2] for v in vals]


def unpack_gob_batch(buf: np.ndarray, level: int = 0, enc_method: int = 0) -> List[bytes]:
    """C-ABI result buffer uint8[batch, stride] -> one gob blob per ciphertext"""
    raw, s = buf.tobytes(), buf.shape[1]
    return [ciphertext_to_gob(int.from_bytes(raw[i * s:(i + 1) * s], "big"), level, enc_method) for i in range(buf.shape[0])]
