#!/usr/bin/env python3
"""Soak of the lane-sliced kernels on edge values and many random ones: x^e modulo n^2 and n^3 (shared and per-number exponents)
at batch sizes that select each kernel (two / four / eight lanes per number modulo n^2; one / two lanes per digit modulo n^3),
with the switches lanes8 / nm4 on and off.  Every result of the edge lanes and a sample of the random ones is compared with pow();
the runs with different switches must agree on ALL lanes."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import paillier_amd as pa

K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]
ctx = pa.Context(0)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 2024)
t0 = time.time()
for bits in ("1024", "2048", "3072"):
    k = K[bits]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n = p * q
    for power, N in ((2, n * n), (3, n ** 3)):
        pk = pa.PublicKey(ctx, n, n + 1)
        level = pa.ENC_LEVEL_ONE if power == 2 else pa.ENC_LEVEL_TWO
        # ConstMult(c, k) = c^k modulo n^(s+1) (operations.go:58-64): the public-key ladders on the pair / digit kernels
        class M:
            @staticmethod
            def exp_batch(xs, e):
                return pk.ConstMultBatch(xs, e, level=level)
        edge = [0, 1, 2, N - 1, N - 2, n, n - 1, n + 1, p, q, p * p, (n * n) % N, (1 << (N.bit_length() - 1)) - 1, (1 << (N.bit_length() - 1)),
                N // 2, N // 3, (N - 1) // 2 + 1, p * q * q % N, (n - 1) * n % N, (1 << 28) - 1, 1 << 28, (1 << 56) - 1]
        for B in (96, 4096, 12288, 20480):
            xs = edge + [rng.randrange(N) for _ in range(B - len(edge))]
            e_shared = rng.getrandbits(n.bit_length()) | 1
            es = [0, 1, 2, n, n - 1] + [rng.randrange(n) for _ in range(B - 5)]
            ref = None
            for lanes8 in (1, 0):
                for nm4 in (1, 0):
                    ctx.set_flag("lanes8", lanes8); ctx.set_flag("nm4", nm4)
                    r_sh = M.exp_batch(xs, e_shared)
                    kern = ctx.last_profile()["kernel"]
                    r_pn = M.exp_batch(xs, es) if B <= 4096 else None
                    if ref is None:
                        ref = (r_sh, r_pn)
                        idx = list(range(len(edge) + 8)) + [rng.randrange(B) for _ in range(24)]
                        for i in idx:
                            assert r_sh[i] == pow(xs[i], e_shared, N), (bits, power, B, kern, i)
                            if r_pn is not None:
                                assert r_pn[i] == pow(xs[i], es[i], N), (bits, power, B, "per-number", i)
                    else:
                        assert r_sh == ref[0] and r_pn == ref[1], (bits, power, B, kern, lanes8, nm4)
            ctx.set_flag("lanes8", 1); ctx.set_flag("nm4", 1)
            print(f"{bits}-bit n^{power} batch {B}: ok ({kern}, {time.time() - t0:.0f} s)", flush=True)
print("soak ok")
