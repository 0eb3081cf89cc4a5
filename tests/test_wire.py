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


# ---- the same format at the C ABI (pgpu_gob_pack / pgpu_gob_unpack, paillier_amd/csrc/wire.cpp), host-memory mode: nothing
# touches the device and no context is needed, so the byte-for-byte comparison with the restatement above runs in the CPU suite
@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as ge
    ge.build()
    from paillier_amd import api
    api.load_library()
    return api


def _rows(vals, stride):
    return np.frombuffer(b"".join(int(v).to_bytes(stride, "big") for v in vals), dtype=np.uint8).reshape(len(vals), stride).copy()


@pytest.mark.parametrize("level,method", [(0, 0), (1, 0), (0, 2), (1, 1)])
def test_c_abi_pack_is_the_python_restatement_byte_for_byte(api, level, method):
    rng = random.Random(10 * level + method)
    vals = [0, 1, 127, 128, 255, 256, (1 << 1016) - 1, 1 << 1016, (1 << 4096) - 1] + [rng.getrandbits(rng.choice([7, 900, 2048, 4095, 4096]))
                                                                                          for _ in range(9000)]
    blobs = api.gob_pack_raw(None, len(vals), _rows(vals, 512), 512, level, method)
    assert blobs == [wire.ciphertext_to_gob(v, level, method) for v in vals]
    assert blobs == wire.unpack_gob_batch(_rows(vals, 512), level, method)
    assert max(len(b) for b in blobs) <= api.load_library().pgpu_gob_max_bytes(512)


def test_c_abi_unpack_accepts_what_the_python_decoder_accepts(api):
    rng = random.Random(5)
    vals = [0, 1, (1 << 4096) - 1] + [rng.getrandbits(rng.choice([1, 64, 1023, 4096])) for _ in range(9000)]
    lv = [rng.randrange(2) for _ in vals]
    me = [rng.randrange(3) for _ in vals]
    blobs = [wire.ciphertext_to_gob(v, a, b) for v, a, b in zip(vals, lv, me)]
    # other type ids (a process that used gob for other types first), swapped field order, a non-minimal magnitude
    c = rng.getrandbits(4000)
    alt = wire.message(wire.struct_typedef(70, "Ciphertext", [("Level", wire.T_INT), ("C", 71), ("EncMethod", wire.T_INT)]))
    alt += wire.message(wire.gobencoder_typedef(71, "Int"))
    alt += wire.message(wire.enc_int(70) + b"\x01" + wire.enc_int(1) + b"\x01" + wire.enc_string(b"\x02\x00\x00" + c.to_bytes(500, "big")) + b"\x00")
    blobs.append(alt)
    vals.append(c); lv.append(1); me.append(0)
    out = np.full((len(blobs), 512), 0xAA, dtype=np.uint8)
    levels, methods = api.gob_unpack_raw(None, blobs, out, 512)
    assert [int.from_bytes(out[i].tobytes(), "big") for i in range(len(vals))] == vals
    assert list(levels) == lv and list(methods) == me
    buf, l2, m2 = wire.pack_gob_batch(blobs, 512)
    assert (buf == out).all() and l2 == lv and m2 == me


def test_c_abi_unpack_errors(api):
    good = wire.ciphertext_to_gob(12345)
    out = np.zeros((2, 8), dtype=np.uint8)
    for bad in (b"", good[:-3], good[:10], b"\x05abc", wire.ciphertext_to_gob(-5)):
        with pytest.raises(api.PaillierHipError) as ei:
            api.gob_unpack_raw(None, [good, bad], out, 8)
        assert ei.value.code == -1
    with pytest.raises(api.PaillierHipError, match="no data provided"):          # paillier.go:377
        api.gob_unpack_raw(None, [b""], out, 8)
    with pytest.raises(api.PaillierHipError, match="wider"):
        api.gob_unpack_raw(None, [wire.ciphertext_to_gob(1 << 64)], out, 8)
    with pytest.raises(api.PaillierHipError):                                      # a device buffer needs a context
        api.gob_unpack_raw(None, [good], 0x1000, 8, mem=api.MEM_DEVICE)
    # an undefined type, an unknown field
    undefined = wire.message(wire.enc_int(65) + b"\x01\x02\x02\x05\x00")
    with pytest.raises(api.PaillierHipError):
        api.gob_unpack_raw(None, [undefined], out, 8)
    other = wire.message(wire.struct_typedef(65, "Ciphertext", [("Q", wire.T_INT)])) + wire.message(wire.enc_int(65) + b"\x01\x02\x00")
    with pytest.raises(api.PaillierHipError, match="no field C"):
        api.gob_unpack_raw(None, [other], out, 8)
    with pytest.raises(wire.GobError, match="no field C"):
        wire.ciphertext_from_gob(other)
    # a too small blob buffer is refused, never overrun
    lib = api.load_library()
    blobs = np.zeros(16, dtype=np.uint8)
    offs = np.zeros(3, dtype=np.uint64)
    rows = _rows([1, 2], 8)
    assert lib.pgpu_gob_pack(None, 2, rows.ctypes.data, 8, api.MEM_HOST, 0, 0, blobs.ctypes.data, 16, offs.ctypes.data) == -1


def _typedefs(extra=()):
    fields = [("C", 66), ("Level", wire.T_INT), ("EncMethod", wire.T_INT)] + list(extra)
    return wire.message(wire.struct_typedef(65, "Ciphertext", fields)) + wire.message(wire.gobencoder_typedef(66, "Int"))


def test_c_abi_unpack_survives_malformed_blobs(api):
    """Blobs come from the network (wire.cpp:193): a field delta is a 64-bit integer off the wire and must be
    bounded before the field table is indexed.  Each of these once read out of bounds or desynchronised the walk."""
    out = np.zeros((2, 8), dtype=np.uint8)
    good = wire.ciphertext_to_gob(12345)
    huge = b"\xf8" + b"\xff" * 8                                     # the gob uint 2^64 - 1
    wrap = b"\xf8" + b"\xff" * 7 + b"\xfe"                            # 2^64 - 2: -1 + delta wraps to -3
    c_val = wire.enc_string(b"\x02\x07")
    bad = {
        "huge value delta": _typedefs() + wire.message(wire.enc_int(65) + huge + b"\x00"),
        "wrapping value delta": _typedefs() + wire.message(wire.enc_int(65) + wrap + b"\x00"),
        "delta back to -1": _typedefs() + wire.message(wire.enc_int(65) + b"\x01" + c_val + huge + b"\x00"),
        "delta past the last field": _typedefs() + wire.message(wire.enc_int(65) + b"\x04\x02\x00"),
        "truncated varint": _typedefs() + wire.message(wire.enc_int(65) + b"\x01" + b"\xfc\x01"),
        "varint wider than 8 bytes": _typedefs() + wire.message(wire.enc_int(65) + b"\xf7" + b"\x01" * 9),
        "type id INT64_MIN": wire.message(huge + b"\x03\x00"),
        "huge structType delta": wire.message(wire.enc_int(-65) + b"\x03" + huge + b"\x00\x00") + good,
        "huge CommonType delta": wire.message(wire.enc_int(-65) + b"\x03\x01" + huge + b"\x00\x00\x00") + good,
        "huge fieldType delta": wire.message(wire.enc_int(-65) + b"\x03\x02\x01" + huge + b"\x00\x00\x00") + good,
        "nf larger than the body": wire.message(wire.enc_int(-65) + b"\x03\x02" + huge + b"\x00\x00") + good,
        "C longer than the blob": _typedefs() + wire.message(wire.enc_int(65) + b"\x01" + huge + b"\x02\x07\x00"),
        "message longer than the blob": _typedefs() + huge + wire.enc_int(65),
    }
    for name, blob in bad.items():
        with pytest.raises(api.PaillierHipError) as ei:
            api.gob_unpack_raw(None, [good, blob], out, 8)
        assert ei.value.code == -1, name
        with pytest.raises(wire.GobError):
            wire.ciphertext_from_gob(blob)
    # the control: the same prefix with a well-formed value
    levels, methods = api.gob_unpack_raw(None, [_typedefs() + wire.message(wire.enc_int(65) + b"\x01" + c_val + b"\x00")], out[:1], 8)
    assert int.from_bytes(out[0].tobytes(), "big") == 7


def test_fields_ciphertext_lacks_are_skipped_as_gob_does(api):
    """Go's decoder ignores a wire field the receiving struct does not have (wire.cpp:211)."""
    out = np.zeros((1, 8), dtype=np.uint8)
    extra = [("Note", 6), ("Tag", wire.T_INT), ("Aux", 66), ("W", 4), ("Z", 7)]
    value = (wire.enc_int(65) + b"\x01" + wire.enc_string(b"\x02\x01\x00") + b"\x01" + wire.enc_int(1)      # C = 256, Level = 1
             + b"\x02" + wire.enc_string(b"hello") + b"\x01" + wire.enc_int(-9) + b"\x01" + wire.enc_string(b"\x02\x63")
             + b"\x01" + b"\xfe\xf0\x3f" + b"\x01" + b"\x01\x02" + b"\x00")
    blob = _typedefs(extra) + wire.message(value)
    assert wire.ciphertext_from_gob(blob) == (256, 1, 0)
    levels, methods = api.gob_unpack_raw(None, [blob], out, 8)
    assert int.from_bytes(out[0].tobytes(), "big") == 256 and list(levels) == [1] and list(methods) == [0]
    # a nested struct as an extra field cannot be walked without its definition: an error, not a desynchronised walk
    nested = _typedefs([("S", 70)]) + wire.message(wire.enc_int(65) + b"\x04\x01\x02\x00\x00")
    with pytest.raises(api.PaillierHipError, match="cannot skip"):
        api.gob_unpack_raw(None, [nested], out, 8)
    with pytest.raises(wire.GobError, match="cannot skip"):
        wire.ciphertext_from_gob(nested)
