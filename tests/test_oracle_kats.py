"""Pins the CPU oracle (oracle/paillier_oracle.py) to every known-answer test the reference's
own test files hold for the hot path.  Each case cites the reference test it restates.
These are the only numeric pins the reference offers (all toy-sized)."""
import pytest

from oracle import paillier_oracle as po


def test_delta():  # thresholdkey_test.go:24-30
    tk = po.ThresholdPublicKey(N=0, G=0, TotalNumberOfDecryptionServers=6)
    assert tk.delta() == 720


def test_exp():  # thresholdkey_test.go:32-46
    assert po.tk_exp(720, 10, 49) == 43
    assert po.tk_exp(720, 0, 49) == 1
    assert po.tk_exp(720, -10, 49) == 8


def test_combine_shares_constant():  # thresholdkey_test.go:48-56
    tk = po.ThresholdPublicKey(N=101 * 103, G=0, TotalNumberOfDecryptionServers=6)
    assert tk.combine_shares_constant() == 4558


def test_partial_decrypt():  # thresholdkey_test.go:58-74 (TestDecrypt)
    key = po.ThresholdSecretKey(N=101 * 103, G=0, TotalNumberOfDecryptionServers=10, Share=862, ID=9)
    partial = po.partial_decrypt(key, 56)
    assert partial.ID == 9
    assert partial.Decryption == 40644522


def test_verify_part1():  # thresholdkey_test.go:109-121
    pd = po.PartialDecryptionZKP(ID=0, Decryption=101, Key=po.ThresholdPublicKey(N=131, G=0), C=99, E=112, Z=88)
    assert po.verify_part1(pd) == 11986


def test_verify_part2():  # thresholdkey_test.go:123-135
    key = po.ThresholdPublicKey(N=131, G=0, VerificationKey=101, VerificationKeys=[77, 67])
    pd = po.PartialDecryptionZKP(ID=1, Decryption=0, Key=key, E=112, Z=88)
    assert po.verify_part2(pd) == 14602


def test_verify_partial_decryptions():  # thresholdkey_test.go:151-166
    tk = po.ThresholdPublicKey(N=0, G=0, Threshold=2)
    with pytest.raises(po.ThresholdError):
        po.verify_partial_decryptions(tk, [])
    prms = [po.PartialDecryption(0, 0), po.PartialDecryption(1, 0)]
    po.verify_partial_decryptions(tk, prms)
    prms[1].ID = 0
    with pytest.raises(po.ThresholdError):
        po.verify_partial_decryptions(tk, prms)


def test_update_lambda():  # thresholdkey_test.go:168-177 -- Euclidean division: -77 / -4 = 20
    assert po.update_lambda(3, 7, 11) == 20


def test_update_cprime():  # thresholdkey_test.go:179-190
    tk = po.ThresholdPublicKey(N=99, G=0)
    assert po.update_cprime(tk, 77, 52, po.PartialDecryption(3, 5)) == 8558


def test_decryption():  # thresholdkey_test.go:267-281
    tk = po.ThresholdPublicKey(N=637753, G=0, Threshold=2, TotalNumberOfDecryptionServers=2,
                               VerificationKey=70661107826)
    shares = [po.PartialDecryption(1, 384111638639), po.PartialDecryption(2, 235243761043)]
    assert po.combine_partial_decryptions(tk, shares) == 100


def test_L():  # paillier_test.go:20-27
    assert po.L(21, 3) == 6


def test_factorial():  # utils_test.go:58-62
    assert po.factorial(6) == 720


def test_init_shortcuts():  # thresholdkey_generator_test.go:213-230
    n, m, n2, nm = po.tkg_init_shortcuts(839, 419, 887, 443)
    assert (n, m, nm, n2) == (744193, 185617, 744193 * 185617, 744193 * 744193)


def test_init_d():  # thresholdkey_generator_test.go:232-243
    n, m, _, _ = po.tkg_init_shortcuts(863, 431, 839, 419)
    d = po.tkg_init_d(m, n)
    assert d % m == 0 and d % n == 1


def test_compute_share():  # thresholdkey_generator_test.go:282-294
    assert po.tkg_compute_share([29, 88, 51], 2, 103) == 31


def test_create_verification_keys():  # thresholdkey_generator_test.go:314-324
    assert po.tkg_create_verification_keys(54, 101 * 101, 10, [12, 90, 103]) == [6162, 304, 2728]


def test_gmp_semantics():
    # Exp with y <= 0 -> 1 (pinned by thresholdkey_test.go:39), nil modulus -> plain power
    assert po.gmp_exp(7, 0, 13) == 1 and po.gmp_exp(7, -3, 13) == 1 and po.gmp_exp(3, 4, None) == 81
    # Euclidean Div/Mod
    assert po.gmp_div(-77, -4) == 20 and po.gmp_div(-77, 4) == -20 and po.gmp_div(77, -4) == -19
    assert po.gmp_mod(-7, 3) == 2 and po.gmp_mod(-7, -3) == 2
    # Bytes: minimal big-endian magnitude, empty for zero
    assert po.gmp_bytes(0) == b"" and po.gmp_bytes(256) == b"\x01\x00" and po.gmp_bytes(-5) == b"\x05"
