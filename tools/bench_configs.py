#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on ONE GPU (bench.py stays the headline Decrypt-2048 line).
Prints one JSON line per config.  Every config ends with a parity check against the oracle on a sample."""
import json, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE, ints_to_be, be_to_ints
from paillier_amd import protocols as pr
from oracle import paillier_oracle as po

K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
which = sys.argv[1:] or ["encrypt2048", "altencrypt2048", "decrypt3072", "threshold2048", "zkp2048", "ddleq2048"]


def rand_below(n, count, nbytes, rng):
    raw = rng.integers(0, 256, size=(count, nbytes), dtype=np.uint8)
    top = n >> (8 * (nbytes - 1))
    raw[:, 0] %= np.uint8(top) if top < 256 else np.uint8(255)
    return raw


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def key(bits):
    k = K["paillier"][str(bits)]
    p, q = int(k["p"], 16), int(k["q"], 16)
    return p * q, (p - 1) * (q - 1)


if "encrypt2048" in which:   # BASELINE config 2
    n, lam = key(2048)
    pk = pa.PublicKey(ctx, n); B = 65536; rng = np.random.default_rng(2)
    m = torch.from_numpy(rand_below(n, B, 256, rng)).to(dev); r_h = rand_below(n, B, 256, rng); r_h[:, -1] |= 1
    r = torch.from_numpy(r_h).to(dev); c = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    dt = timed(lambda: pk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE))
    prof = ctx.last_profile()
    mi, ri, ci = be_to_ints(m[:3].cpu().numpy()), be_to_ints(r[:3].cpu().numpy()), be_to_ints(c[:3].cpu().numpy())
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    assert ci == [po.encrypt_with_r(sk_o, a, b).C for a, b in zip(mi, ri)]
    print(json.dumps({"config": "Batch 65536 Encrypt, 2048-bit n (r^n * g^m mod n^2)", "value": B / dt, "unit": "encryptions/s",
                      "ms_per_batch": dt * 1e3, "vm_ms": prof["vm_ms"], "executed_Tmad_per_s": prof["vm_mads"] / prof["vm_ms"] / 1e9,
                      "parity": "3 lanes vs oracle"}), flush=True)

if "altencrypt2048" in which:   # AltEncryptWithR (paillier.go:221-238): fixed-base comb, no squarings
    k = K["paillier"]["2048"]; n = int(k["n"], 16); lam = int(k["lambda"], 16)
    pk = pa.PublicKey(ctx, n, n + 1, H=int(k["h"], 16), K=int(k["k"], 16)); sk = pa.SecretKey(ctx, pk, lam)
    B = 65536; rng = np.random.default_rng(6)
    m_h = rand_below(n, B, 256, rng); r_h = rng.integers(0, 256, size=(B, 128), dtype=np.uint8)
    m = torch.from_numpy(m_h).to(dev); r = torch.from_numpy(r_h).to(dev)
    c = torch.zeros((B, 512), dtype=torch.uint8, device=dev); out = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
    import ctypes as C
    def run():
        rc = ctx.lib.pgpu_alt_encrypt_with_r(pk.h, 0, B, m.data_ptr(), 256, r.data_ptr(), 128, c.data_ptr(), 512, None, MEM_DEVICE)
        assert rc == 0, ctx.lib.pgpu_last_error()
    dt = timed(run); prof = ctx.last_profile()
    sk.decrypt_raw(B, c.data_ptr(), 512, out.data_ptr(), 256, MEM_DEVICE)
    assert torch.equal(out, m), "alt-encrypt round trip failed"
    sk_o = po.SecretKey(N=n, G=n + 1, H=int(k["h"], 16), K=int(k["k"], 16), Lambda=lam)
    mi, ri, ci = be_to_ints(m_h[:2]), be_to_ints(r_h[:2]), be_to_ints(c[:2].cpu().numpy())
    assert ci == [po.alt_encrypt_with_r_at_level(sk_o, a, b, 0)[0].C for a, b in zip(mi, ri)]
    print(json.dumps({"config": "Batch 65536 AltEncrypt, 2048-bit n (g^m * h^(r mod K), fixed-base comb)", "value": B / dt,
                      "unit": "encryptions/s", "ms_per_batch": dt * 1e3, "vm_ms": prof["vm_ms"],
                      "parity": "65536-lane decrypt round trip + 2 lanes vs oracle"}), flush=True)

if "decrypt3072" in which:   # BASELINE config 3
    n, lam = key(3072)
    pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, lam); B = 65536; rng = np.random.default_rng(3)
    m_h = rand_below(n, B, 384, rng); r_h = rand_below(n, B, 384, rng); r_h[:, -1] |= 1
    m = torch.from_numpy(m_h).to(dev); r = torch.from_numpy(r_h).to(dev)
    c = torch.zeros((B, 768), dtype=torch.uint8, device=dev); out = torch.zeros((B, 384), dtype=torch.uint8, device=dev)
    pk.encrypt_with_r_raw(B, m.data_ptr(), 384, r.data_ptr(), 384, c.data_ptr(), 768, MEM_DEVICE)
    enc = ctx.last_profile()
    dt = timed(lambda: sk.decrypt_raw(B, c.data_ptr(), 768, out.data_ptr(), 384, MEM_DEVICE))
    prof = ctx.last_profile()
    assert torch.equal(out, m), "round trip failed"
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    assert be_to_ints(out[:2].cpu().numpy()) == [po.decrypt(sk_o, po.Ciphertext(x)) for x in be_to_ints(c[:2].cpu().numpy())]
    print(json.dumps({"config": "Batch 65536 Decrypt, 3072-bit n (CRT over p^2,q^2)", "value": B / dt, "unit": "decryptions/s",
                      "ms_per_batch": dt * 1e3, "vm_ms": prof["vm_ms"], "executed_Tmad_per_s": prof["vm_mads"] / prof["vm_ms"] / 1e9,
                      "encrypt3072_per_s": B / (enc["vm_ms"] * 1e-3), "parity": "65536-lane round trip + 2 lanes vs oracle"}), flush=True)

if "decrypt2048_l2" in which:   # level two (Damgard-Jurik s = 2, paillier.go:292-340): CRT over p^3, q^3 vs the reference formula
    n, lam = key(2048)
    pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, lam); B = 32768; rng = np.random.default_rng(7)
    m_h = rand_below(n * n, B, 512, rng); r_h = rand_below(n, B, 256, rng); r_h[:, -1] |= 1
    m = torch.from_numpy(m_h).to(dev); r = torch.from_numpy(r_h).to(dev)
    c = torch.zeros((B, 768), dtype=torch.uint8, device=dev); out = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    pk.encrypt_with_r_raw(B, m.data_ptr(), 512, r.data_ptr(), 256, c.data_ptr(), 768, MEM_DEVICE, level=1)
    enc = ctx.last_profile()
    dt = timed(lambda: sk.decrypt_raw(B, c.data_ptr(), 768, out.data_ptr(), 512, MEM_DEVICE, level=1))
    prof = ctx.last_profile()
    assert torch.equal(out, m), "level-two round trip failed"
    dt_ref = timed(lambda: sk.decrypt_raw(B, c.data_ptr(), 768, out.data_ptr(), 512, MEM_DEVICE, level=1, flags=pa.DECRYPT_NO_CRT), reps=1)
    assert torch.equal(out, m), "level-two round trip (reference formula) failed"
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    assert be_to_ints(out[:2].cpu().numpy()) == [po.decrypt(sk_o, po.Ciphertext(x, po.ENC_LEVEL_TWO)) for x in be_to_ints(c[:2].cpu().numpy())]
    print(json.dumps({"config": "Batch 32768 level-two Decrypt, 2048-bit n (CRT over p^3,q^3)", "value": B / dt, "unit": "decryptions/s",
                      "ms_per_batch": dt * 1e3, "vm_ms": prof["vm_ms"], "executed_Tmad_per_s": prof["vm_mads"] / prof["vm_ms"] / 1e9,
                      "reference_formula_per_s": B / dt_ref, "encrypt_l2_per_s": B / (enc["vm_ms"] * 1e-3),
                      "parity": "32768-lane round trip (both paths) + 2 lanes vs oracle"}), flush=True)

if "threshold2048" in which:  # BASELINE config 4, single-GPU part: 3 x PartialDecrypt + Combine for 16384 ciphertexts
    k = K["threshold"]["2048"]; n = int(k["n"], 16); shares = [int(s, 16) for s in k["shares"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3); B = 16384; rng = np.random.default_rng(4)
    m_h = rand_below(n, B, 256, rng); r_h = rand_below(n, B, 256, rng); r_h[:, -1] |= 1
    m = torch.from_numpy(m_h).to(dev); r = torch.from_numpy(r_h).to(dev)
    c = torch.zeros((B, 512), dtype=torch.uint8, device=dev); out = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
    tk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
    ids = [1, 3, 5]
    parts = [torch.zeros((B, 512), dtype=torch.uint8, device=dev) for _ in ids]
    def run():
        for i, pbuf in zip(ids, parts):
            tk.partial_decrypt_raw(shares[i - 1], B, c.data_ptr(), 512, pbuf.data_ptr(), 512, MEM_DEVICE)
        tk.combine_raw(ids, B, [pbuf.data_ptr() for pbuf in parts], 512, out.data_ptr(), 256, MEM_DEVICE)
    dt = timed(run, reps=2)
    assert torch.equal(out, m), "threshold round trip failed"
    t0 = time.perf_counter(); tk.partial_decrypt_raw(shares[0], B, c.data_ptr(), 512, parts[0].data_ptr(), 512, MEM_DEVICE); torch.cuda.synchronize()
    t_pd = time.perf_counter() - t0
    t0 = time.perf_counter(); tk.combine_raw(ids, B, [pbuf.data_ptr() for pbuf in parts], 512, out.data_ptr(), 256, MEM_DEVICE); torch.cuda.synchronize()
    t_cb = time.perf_counter() - t0
    print(json.dumps({"config": "Threshold (t=3,l=5) 3 x PartialDecrypt + Combine, 2048-bit, 16384 ciphertexts, servers {1,3,5}",
                      "value": B / dt, "unit": "threshold decryptions/s", "ms_per_batch": dt * 1e3,
                      "partial_decrypt_per_s": B / t_pd, "combine_per_s": B / t_cb, "parity": "16384-lane round trip"}), flush=True)

if "zkp2048" in which:   # PartialDecryptionWithZKP (r supplied) and VerifyProof (thresholdkey.go:225-311), device-resident, raw buffers
    import ctypes as C
    k = K["threshold"]["2048"]; n = int(k["n"], 16); shares = [int(s, 16) for s in k["shares"]]
    v, vks = int(k["v"], 16), [int(x, 16) for x in k["vks"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3); B = 16384; rng = np.random.default_rng(8); sid = 2
    m_h = rand_below(n, B, 256, rng); r_h = rand_below(n, B, 256, rng); r_h[:, -1] |= 1
    c_d = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    tk.encrypt_with_r_raw(B, torch.from_numpy(m_h).to(dev).data_ptr(), 256, torch.from_numpy(r_h).to(dev).data_ptr(), 256, c_d.data_ptr(), 512, MEM_DEVICE)
    c_h = c_d.cpu().numpy(); rz = rand_below(n * n, B, 512, rng)
    dec = np.zeros((B, 512), np.uint8); eo = np.zeros((B, 32), np.uint8); zb = 512 + 48; zo = np.zeros((B, zb), np.uint8)
    from paillier_amd.api import _be, _ptr, MEM_HOST
    sb, vb, ib = _be(shares[sid - 1]), _be(v), _be(vks[sid - 1])
    def prove():
        rc = ctx.lib.pgpu_share_zkp_prove(tk.h, 5, sb, len(sb), vb, len(vb), B, _ptr(c_h), 512, _ptr(rz), 512, _ptr(dec), 512, _ptr(eo), _ptr(zo), zb, MEM_HOST)
        assert rc == 0, ctx.lib.pgpu_last_error()
    dtp = timed(prove, reps=2)
    ok = np.zeros(B, np.int32)
    def verify():
        rc = ctx.lib.pgpu_share_zkp_verify(tk.h, vb, len(vb), ib, len(ib), B, _ptr(c_h), 512, _ptr(dec), 512, _ptr(eo), _ptr(zo), zb, _ptr(ok), MEM_HOST)
        assert rc == 0, ctx.lib.pgpu_last_error()
    dtv = timed(verify, reps=2)
    assert ok.all(), "a valid share proof was rejected"
    tsk = po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=5, Threshold=3, VerificationKey=v, VerificationKeys=vks, ID=sid, Share=shares[sid - 1])
    ref = po.partial_decryption_with_zkp_r(tsk, be_to_ints(c_h[:1])[0], be_to_ints(rz[:1])[0])
    assert (be_to_ints(dec[:1])[0], be_to_ints(eo[:1])[0], be_to_ints(zo[:1])[0]) == (ref.Decryption, ref.E, ref.Z)
    print(json.dumps({"config": "Share-decryption ZKP, 2048-bit, 16384 ciphertexts, one server, host buffers", "prove_per_s": B / dtp,
                      "verify_per_s": B / dtv, "parity": "all 16384 proofs verify; proof 0 vs oracle"}), flush=True)

if "ddleq2048" in which:  # BASELINE config 5 (per-instance throughput, secpar = 1)
    n, lam = key(2048)
    pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, lam); B = int(os.environ.get('DDLEQ_B', '2048')); rng = random.Random(5)
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    ms = [rng.randrange(n) for _ in range(B)]
    inner = pk.EncryptWithRBatch(ms, [rng.randrange(1, n) | 1 for _ in ms])
    ct1 = pk.EncryptWithRBatch(inner, [rng.randrange(1, n) | 1 for _ in ms], level=pa.ENC_LEVEL_TWO)
    a_s = [rng.randrange(1, n) | 1 for _ in ms]; b_s = [rng.randrange(1, n) | 1 for _ in ms]
    ct2 = pr.nested_randomize_with_ab_batch(pk, ct1, a_s, b_s)
    xs = [rng.randrange(1, n) | 1 for _ in ms]; ys = [rng.randrange(1, n) | 1 for _ in ms]
    t0 = time.perf_counter(); proofs = pr.prove_ddleq_instances(sk, ct1, ct2, a_s, b_s, xs, ys); t_p = time.perf_counter() - t0
    t0 = time.perf_counter(); ok = pr.verify_ddleq_instances(pk, ct1, ct2, proofs); t_v = time.perf_counter() - t0
    assert all(ok)
    # device-resident verification through the C ABI (hash + 3 modexps per instance on the GPU), raw buffers in HBM
    cb3, pb1, pb2 = pk.cipher_bytes(1), pk.plain_bytes(0), pk.plain_bytes(1)
    tb = lambda vals, st: torch.from_numpy(ints_to_be(vals, st)).to(dev)
    d = [tb(ct1, cb3), tb(ct2, cb3), tb(xs, pb1), tb(ys, pb1), tb([p_.Alpha for p_ in proofs], cb3), tb([p_.E for p_ in proofs], pb2),
         tb([p_.F for p_ in proofs], cb3)]
    okh = np.zeros(B, dtype=np.int32)
    def vrun():
        rc = ctx.lib.pgpu_ddleq_verify(pk.h, B, d[0].data_ptr(), d[1].data_ptr(), cb3, d[2].data_ptr(), d[3].data_ptr(), pb1,
                                       d[4].data_ptr(), cb3, d[5].data_ptr(), pb2, d[6].data_ptr(), cb3, okh.ctypes.data, MEM_DEVICE)
        assert rc == 0, ctx.lib.pgpu_last_error()
    t_vd = timed(vrun, reps=2)
    assert okh.all()
    # device-resident prover through the C ABI
    da, db = tb(a_s, pb1), tb(b_s, pb1)
    oa = torch.zeros((B, cb3), dtype=torch.uint8, device=dev); oe = torch.zeros((B, pb2), dtype=torch.uint8, device=dev)
    of = torch.zeros((B, cb3), dtype=torch.uint8, device=dev)
    def prun():
        rc = ctx.lib.pgpu_ddleq_prove(sk.h, B, d[0].data_ptr(), d[1].data_ptr(), cb3, da.data_ptr(), db.data_ptr(), d[2].data_ptr(),
                                      d[3].data_ptr(), pb1, oa.data_ptr(), oe.data_ptr(), pb2, of.data_ptr(), MEM_DEVICE)
        assert rc == 0, ctx.lib.pgpu_last_error()
    t_pd = timed(prun, reps=1)
    assert torch.equal(oa, d[4]) and torch.equal(oe, d[5]) and torch.equal(of, d[6]), "device prover differs from the host-orchestrated one"
    ref = po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(ct1[0], 1), po.Ciphertext(ct2[0], 1), a_s[0], b_s[0], xs[0], ys[0])
    assert (proofs[0].Alpha, proofs[0].E, proofs[0].F) == (ref.Alpha, ref.E, ref.F)
    print(json.dumps({"config": f"DDLEQ 2048-bit, {B} instances (secpar=1 each), int-list API incl. host packing",
                      "prove_instances_per_s": B / t_p, "verify_instances_per_s_host_orchestrated": B / t_v,
                      "verify_instances_per_s_device_resident": B / t_vd, "prove_instances_per_s_device_resident": B / t_pd,
                      "parity": "all verify; instance 0 vs oracle"}), flush=True)

if "l2_3072" in which:   # level two at 3072 bits: n^3 is 9 216 bits -- vm_asm_42_8, eight lanes per number (round 3; rounds 1-2: hipcc (83,4))
    n, lam = key(3072)
    pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, lam)
    B = 8192; rng = np.random.default_rng(33)
    m_h = rand_below(n * n, B, 768, rng); r_h = rand_below(n, B, 384, rng); r_h[:, -1] |= 1
    m = torch.from_numpy(m_h).to(dev); r = torch.from_numpy(r_h).to(dev)
    c = torch.zeros((B, 1152), dtype=torch.uint8, device=dev); o = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    dt = timed(lambda: pk.encrypt_with_r_raw(B, m.data_ptr(), 768, r.data_ptr(), 384, c.data_ptr(), 1152, MEM_DEVICE, level=1), reps=2)
    prof = ctx.last_profile()
    all_asm = ctx.last_vm_asm() == ctx.last_vm_launches()
    dtd = timed(lambda: sk.decrypt_raw(B, c.data_ptr(), 1152, o.data_ptr(), 768, MEM_DEVICE, level=1), reps=2)
    profd = ctx.last_profile()
    assert torch.equal(o, m), "level-two round trip at 3072 bits"
    mi, ri, ci = be_to_ints(m_h[:2]), be_to_ints(r_h[:2]), be_to_ints(c[:2].cpu().numpy())
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    assert ci == [po.encrypt_with_r_at_level(sk_o, a, b, po.ENC_LEVEL_TWO).C for a, b in zip(mi, ri)]
    print(json.dumps({"config": "Batch 8192 level-two EncryptWithR / Decrypt, 3072-bit n (n^3 = 9216 bits)", "encrypt_l2_3072_per_s": B / dt,
                      "encrypt_kernel": prof["kernel"], "encrypt_vm_ms": prof["vm_ms"],
                      "encrypt_frac_of_issue_peak": prof["vm_mads"] / (prof["vm_ms"] * 1e-3) / 39.3216e12,
                      "every_vm_launch_in_assembly": bool(all_asm), "decrypt_l2_3072_per_s": B / dtd, "decrypt_kernel": profd["kernel"],
                      "parity": "8192-lane round trip + 2 lanes vs oracle"}), flush=True)
