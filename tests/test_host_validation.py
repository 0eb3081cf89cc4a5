"""Host-side argument checks of the Python mirror that guard native reads (no GPU needed: every call below must fail BEFORE it
reaches the C ABI -- a shorter list would be an out-of-bounds host read in native code)."""
import pytest

import paillier_amd as pa
from paillier_amd import api, protocols


class _StubKey:
    """stands in for a PublicKey: byte widths only (a C call through it would fail loudly: there is no handle)"""
    h = None

    def cipher_bytes(self, level=0):
        return 768 if level else 512

    def plain_bytes(self, level=0):
        return 512 if level else 256


def _secret_key_without_a_device():
    sk = object.__new__(pa.SecretKey)
    sk.pk, sk.ctx, sk.h = _StubKey(), None, None
    return sk


@pytest.mark.parametrize("drop", ["ct2s", "a_s", "b_s", "xs", "ys"])
def test_prove_ddleq_batch_refuses_ragged_statement_lists(drop):
    sk = _secret_key_without_a_device()
    S, secpar = 3, 2
    args = dict(ct1s=[5] * S, ct2s=[6] * S, a_s=[7] * S, b_s=[8] * S, xs=[[1] * secpar] * S, ys=[[2] * secpar] * S)
    args[drop] = args[drop][:-1]
    with pytest.raises(ValueError):
        sk.ProveDDLEQBatch(secpar, **args)


def test_prove_ddleq_batch_refuses_wrong_draw_counts():
    sk = _secret_key_without_a_device()
    with pytest.raises(ValueError):
        sk.ProveDDLEQBatch(2, [5, 5], [6, 6], [7, 7], [8, 8], [[1, 1], [1]], [[2, 2], [2, 2]])
    with pytest.raises(ValueError):
        sk.ProveDDLEQBatch(0, [5], [6], [7], [8], [[]], [[]])


def test_verify_ddleq_proof_batch_needs_a_proof_per_statement():
    """fewer proofs than statements must not come back as fewer verdicts: all(...) over them would accept unproven statements"""
    inst = api.DDLEQProofInstance(1, 2, 3, 4, 5) if hasattr(api, "DDLEQProofInstance") else protocols.DDLEQProofInstance(1, 2, 3, 4, 5)
    with pytest.raises(ValueError):
        protocols.verify_ddleq_proof_batch(_StubKey(), [1, 2, 3], [4, 5, 6], [[inst], [inst]])
    with pytest.raises(ValueError):
        protocols.verify_ddleq_proof_batch(_StubKey(), [1, 2], [4], [[inst], [inst]])
    assert protocols.verify_ddleq_proof_batch(_StubKey(), [1, 2], [3, 4], [[], []]) == [True, True]      # ddleq.go:44-53: no instances


def test_add_and_sub_refuse_operand_vectors_of_different_length():
    pk = object.__new__(pa.PublicKey)
    pk.cipher_bytes = lambda level=0: 512
    with pytest.raises(ValueError):
        pk.AddBatch([1, 2, 3], [4, 5])
    with pytest.raises(ValueError):
        pk.SubBatch([1, 2, 3], [4, 5])
