"""paillier_amd/wire.py: the reference's gob wire format of a Ciphertext (paillier.go:374-401).  No Go toolchain here, so the
pins are (1) the byte-exact example of the gob specification itself (pkg.go.dev/encoding/gob, "Encoding Details": the
stream for `type Point struct {X, Y int}` with value {22, 33}), which the encoder primitives must reproduce, (2) math/big's
documented GobEncode layout, (3) round trips incl. the reference's TestToFromBytes shape."""
import random

import numpy as np
import pytest

from paillier_amd import wire


def test_primitives_reproduce_the_gob_specification_example():
    spec = bytes.fromhex("1f ff 81 03 01 01 05 50 6f 69 6e 74 01 ff 82 00 01 02 01 01 58 01 04 00 01 01 59 01 04 00 00 00"
                         "07 ff 82 01 2c 01 42 00".replace(" ", ""))
    typedef = wire.message(wire.struct_typedef(65, "Point", [("X", wire.T_INT), ("Y", wire.T_INT)]))
    value = wire.message(wire.enc_int(65) + b"\x01" + wire.enc_int(22) + b"\x01" + wire.enc_int(33) + b"\x00")
    assert typedef + value == spec
    # the specification's integer examples: 7 -> 07, 256 -> FE 01 00, -129 -> FE 01 01, -1 -> 01
    assert wire.enc_uint(7) == b"\x07" and wire.enc_uint(256) == bytes.fromhex("fe0100")
    assert wire.enc_int(-129) == bytes.fromhex("fe0101") and wire.enc_int(-1) == b"\x01" and wire.enc_int(65) == bytes.fromhex("ff82")


def test_ciphertext_blob_layout():
    blob = wire.ciphertext_to_gob(0x1234, level=0, enc_method=0)
    t1 = wire.message(bytes.fromhex("ff81") + b"\x03\x01\x01\x0aCiphertext\x01\xff\x82\x00\x01\x03"
                      b"\x01\x01C\x01\xff\x84\x00" b"\x01\x05Level\x01\x04\x00" b"\x01\x09EncMethod\x01\x04\x00" b"\x00\x00")
    t2 = wire.message(bytes.fromhex("ff83") + b"\x05\x01\x01\x03Int\x01\xff\x84\x00\x00\x00")
    val = wire.message(bytes.fromhex("ff82") + b"\x01\x03\x02\x12\x34\x00")          # C = GobEncode = 02 12 34; zero fields omitted
    assert blob == t1 + t2 + val
    b2 = wire.ciphertext_to_gob(5, level=1, enc_method=2)
    assert b2.endswith(bytes.fromhex("ff82") + b"\x01\x02\x02\x05" b"\x01\x02" b"\x01\x04" b"\x00")
    b3 = wire.ciphertext_to_gob(5, level=0, enc_method=2)                             # Level omitted: delta 2 to EncMethod
    assert b3.endswith(bytes.fromhex("ff82") + b"\x01\x02\x02\x05" b"\x02\x04" b"\x00")
    assert wire.gmp_int_gob(0) == b"\x02" and wire.gmp_int_gob(-255) == b"\x03\xff"  # math/big: version 1, sign bit, magnitude


def test_round_trips():
    rng = random.Random(3)
    for _ in range(200):                                     # paillier_test.go:140-156 TestToFromBytes
        c = rng.getrandbits(rng.choice([1, 64, 128, 1023, 4096, 6144]))
        lvl, meth = rng.randrange(2), rng.randrange(3)
        assert wire.ciphertext_from_gob(wire.ciphertext_to_gob(c, lvl, meth)) == (c, lvl, meth)
    # other type ids (a process that used gob for other types first) and swapped field order decode the same
    c = rng.getrandbits(4000)
    alt = wire.message(wire.struct_typedef(70, "Ciphertext", [("Level", wire.T_INT), ("C", 71), ("EncMethod", wire.T_INT)]))
    alt += wire.message(wire.gobencoder_typedef(71, "Int"))
    alt += wire.message(wire.enc_int(70) + b"\x01" + wire.enc_int(1) + b"\x01" + wire.enc_string(wire.gmp_int_gob(c)) + b"\x00")
    assert wire.ciphertext_from_gob(alt) == (c, 1, 0)
    with pytest.raises(wire.GobError):
        wire.ciphertext_from_gob(b"")                         # paillier.go:377 "no data provided"
    with pytest.raises(wire.GobError):
        wire.ciphertext_from_gob(wire.ciphertext_to_gob(7)[:-3])


def test_batch_bridge_to_the_c_abi_format():
    rng = random.Random(4)
    vals = [rng.getrandbits(4096) for _ in range(10)]
    blobs = [wire.ciphertext_to_gob(v, 0, 2) for v in vals]
    buf, levels, methods = wire.pack_gob_batch(blobs, 512)
    assert buf.shape == (10, 512) and levels == [0] * 10 and methods == [2] * 10
    assert [int.from_bytes(buf[i].tobytes(), "big") for i in range(10)] == vals
    assert wire.unpack_gob_batch(buf, 0, 2) == blobs
