"""Bit-level Python model of paillier_amd/csrc/bigint28.h's montmul (K lanes per number, radix 2^28,
lazy carries, accumulator flush).  Test infrastructure: it checks the ALGORITHM and its overflow bounds
on the CPU (every 64-bit accumulator write asserts < 2^64), so the HIP code is only debugged for
transcription errors on the GPU."""
LB = 28
MASK = (1 << LB) - 1
U64 = 1 << 64


def limbs_for_bits(bits):
    return (bits + 3 + LB - 1) // LB


def to_limbs(v, n):
    out = []
    for _ in range(n):
        out.append(v & MASK)
        v >>= LB
    assert v == 0
    return out


def from_limbs(l):
    return sum(int(x) << (LB * i) for i, x in enumerate(l))


def montmul_model(x, a, n, n0inv, WL, K, flush_every=48):
    """x, a: lazy limb lists of WT = WL*K limbs.  Returns lazy limbs of a*x*R^-1 mod N (< 2N)."""
    WT = WL * K
    need_flush = (2 * WT + 1) > 255
    t = [[0] * WL for _ in range(K)]  # t[k][j]
    xs = [x[k * WL:(k + 1) * WL] for k in range(K)]
    ns = [n[k * WL:(k + 1) * WL] for k in range(K)]

    def chk(v):
        assert 0 <= v < U64, "accumulator overflow"
        return v

    for i in range(WT):
        ai = a[i]
        for k in range(K):
            for j in range(WL):
                t[k][j] = chk(t[k][j] + ai * xs[k][j])
        m = ((t[0][0] & 0xFFFFFFFF) * n0inv) & MASK
        y0 = [chk(t[k][0] + m * ns[k][0]) for k in range(K)]
        assert y0[0] & MASK == 0
        newt = [[0] * WL for _ in range(K)]
        for k in range(K):
            c = (y0[k] >> LB) if k == 0 else 0
            newt[k][0] = chk(t[k][1] + m * ns[k][1] + c)
            for j in range(2, WL):
                newt[k][j - 1] = chk(t[k][j] + m * ns[k][j])
            newt[k][WL - 1] = y0[k + 1] if k < K - 1 else 0
        t = newt
        if need_flush and i % flush_every == flush_every - 1:
            top_c = [0] * K
            for k in range(K):
                if k < K - 1:
                    top_c[k] = t[k][WL - 1] >> LB
                    t[k][WL - 1] &= MASK
                for j in range(WL - 2, -1, -1):
                    t[k][j + 1] = chk(t[k][j + 1] + (t[k][j] >> LB))
                    t[k][j] &= MASK
            for k in range(1, K):
                t[k][0] = chk(t[k][0] + top_c[k - 1])
    flat = [v for k in range(K) for v in t[k]]
    # lazy normalisation
    s = []
    for j in range(WT):
        lo = flat[j] & MASK
        mid = (flat[j - 1] >> LB) & MASK if j >= 1 else 0
        hi = flat[j - 2] >> (2 * LB) if j >= 2 else 0
        s.append(lo + mid + hi)
    assert flat[WT - 1] >> LB == 0 and (flat[WT - 2] >> (2 * LB)) == 0
    assert s[WT - 1] >> LB == 0
    out = [(s[j] & MASK) + ((s[j - 1] >> LB) if j >= 1 else 0) for j in range(WT)]
    assert all(v <= MASK + 2 for v in out)
    return out


def mont_consts(N, WT):
    R = 1 << (LB * WT)
    assert R > 4 * N and N % 2 == 1
    n0inv = (-pow(N, -1, 1 << LB)) % (1 << LB)
    return {"n": to_limbs(N, WT), "n0inv": n0inv, "R": R, "r2": to_limbs(R * R % N, WT),
            "one": to_limbs(R % N, WT), "plain_one": to_limbs(1, WT)}
