"""CPU: the algebra behind the pair kernels (gen_vm_asm.py GenP / GenQ), in plain Python integers.

A residue y modulo N = n^2 (n any odd modulus: a prime p for the CRT halves of Decrypt, the public n for Encrypt) is kept as
y R mod n^2 = a0 + a1 n.  The product of two such pairs is two Montgomery steps modulo n; Cadj is the multiple of n whose
limbs can all be taken from [2^28, 2^29), so that Cadj - m never borrows limb-wise.  These tests pin the identities the
kernels rely on, independent of any GPU."""
import random

LB = 28


def cadj_limbs(n, H):
    D = sum(1 << (LB * j + LB) for j in range(H))
    E = (D // n + 1) * n - D
    assert 0 < E <= n < 1 << (LB * H)
    return [((E >> (LB * j)) & ((1 << LB) - 1)) + (1 << LB) for j in range(H)]


def make(n, H):
    R = 1 << (LB * H)
    nneg = (-pow(n, -1, R)) % R
    cadj = sum(c << (LB * j) for j, c in enumerate(cadj_limbs(n, H)))

    def mont(u):
        m = (u * nneg) % R
        assert (u + m * n) % R == 0
        return (u + m * n) // R, m

    def pmul(a, b):
        t, m = mont(a[0] * b[0])
        c1, _ = mont(a[0] * b[1] + a[1] * b[0] + cadj - m)
        return (t, c1)
    return R, cadj, mont, pmul


def test_cadj_is_a_multiple_with_big_limbs():
    rng = random.Random(1)
    for bits, H in ((1024, 37), (1536, 55), (2048, 74)):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        limbs = cadj_limbs(n, H)
        assert all((1 << LB) <= c < (2 << LB) for c in limbs)
        assert sum(c << (LB * j) for j, c in enumerate(limbs)) % n == 0


def test_pair_product_is_the_product_modulo_n_squared():
    rng = random.Random(2)
    for bits, H in ((1024, 37), (2048, 74)):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        R, cadj, mont, pmul = make(n, H)
        n2 = n * n
        rinv = pow(R, -1, n2)
        for _ in range(50):
            a = (rng.randrange(2 * n), rng.randrange(4 * n))     # lazy digits, as the kernels keep them
            b = (rng.randrange(2 * n), rng.randrange(4 * n))
            c = pmul(a, b)
            assert (c[0] + c[1] * n) % n2 == (a[0] + a[1] * n) * (b[0] + b[1] * n) * rinv % n2
            assert c[0] < 2 * n and c[1] < 2 * n                  # outputs stay lazy-bounded without a final subtraction


def test_pair_ladder_gives_the_fermat_quotient_path_of_decrypt():
    """x^(p-1) mod p^2 through pairs equals pow(); its first digit is 1 mod p (what makes L_p exact)."""
    rng = random.Random(3)
    p = 2 ** 521 - 1                                             # a prime that needs no search
    H = (521 + 3 + LB - 1) // LB
    R, cadj, mont, pmul = make(p, H)
    p2 = p * p
    rinv = pow(R, -1, p2)
    c = rng.randrange(p2)
    ct = c * R % p2
    x = (ct % p, ct // p)
    acc = x
    for bit in bin(p - 1)[3:]:
        acc = pmul(acc, acc)
        if bit == "1":
            acc = pmul(acc, x)
    F = (acc[0] + acc[1] * p) * rinv % p2
    assert F == pow(c, p - 1, p2) and F % p == 1
