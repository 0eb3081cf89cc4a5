"""pgpu_gob_unpack / pgpu_gob_pack with the flat buffer in HBM: the payload bytes are moved by kernels (k_bytes_gather_be,
k_be_lengths, k_gob_emit) and the unpacked batch feeds Decrypt without returning to the host.  Byte-for-byte against the Python
restatement paillier_amd/wire.py of Ciphertext.Bytes() / NewCiphertextFromBytes (paillier.go:374-401).  Parity with Go's own gob
output is unpinned (no Go toolchain; the reference holds no gob bytes: paillier_test.go:140-156 is a round trip)."""
import json
import os
import random

import numpy as np
import pytest

from paillier_amd import wire

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


@pytest.mark.parametrize("batch", [1, 300, 20000])
def test_device_pack_and_unpack_are_the_python_restatement(ctx, batch):
    import torch
    from paillier_amd import api
    rng = random.Random(batch)
    vals = ([0, 1, 255, 256, (1 << 4096) - 1] + [rng.getrandbits(rng.choice([9, 2048, 4090, 4096])) for _ in range(batch)])[:batch]
    rows = np.frombuffer(b"".join(v.to_bytes(512, "big") for v in vals), dtype=np.uint8).reshape(batch, 512).copy()
    dev = torch.from_numpy(rows).cuda()
    blobs = api.gob_pack_raw(ctx, batch, dev.data_ptr(), 512, 1, 2, mem=api.MEM_DEVICE)
    assert blobs == [wire.ciphertext_to_gob(v, 1, 2) for v in vals]
    assert blobs == api.gob_pack_raw(ctx, batch, rows, 512, 1, 2)                      # host mode: the same bytes
    back = torch.full((batch, 640), 0x55, dtype=torch.uint8, device="cuda")            # a wider stride: zero-padded on the left
    levels, methods = api.gob_unpack_raw(ctx, blobs, back.data_ptr(), 640, mem=api.MEM_DEVICE)
    got = back.cpu().numpy()
    assert (got[:, :128] == 0).all() and (got[:, 128:] == rows).all()
    assert set(levels) == {1} and set(methods) == {2}


def test_gob_blobs_in_plaintexts_out_through_hbm(ctx):
    """TestToFromBytes (paillier_test.go:140-156) as a pipeline: Encrypt -> Bytes() -> NewCiphertextFromBytes -> Decrypt, the
    ciphertexts staying in HBM between unpack and Decrypt."""
    import torch
    import paillier_amd as pa
    from paillier_amd import api
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n, lam = int(k["n"], 16), int(k["lambda"], 16)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(77)
    B = 1000
    ms = [rng.randrange(n) for _ in range(B)]
    rs = [rng.randrange(1, n) | 1 for _ in range(B)]
    cts = pk.EncryptWithRBatch(ms, rs)
    blobs = [wire.ciphertext_to_gob(c) for c in cts]                                   # what a Go peer would send
    cdev = torch.zeros((B, 512), dtype=torch.uint8, device="cuda")
    api.gob_unpack_raw(ctx, blobs, cdev.data_ptr(), 512, mem=api.MEM_DEVICE)
    mdev = torch.zeros((B, 256), dtype=torch.uint8, device="cuda")
    sk.decrypt_raw(B, cdev.data_ptr(), 512, mdev.data_ptr(), 256, api.MEM_DEVICE)
    assert api.be_to_ints(mdev.cpu().numpy()) == ms
    # ... and out again in the reference's format
    assert api.gob_pack_raw(ctx, B, cdev.data_ptr(), 512, mem=api.MEM_DEVICE) == blobs
