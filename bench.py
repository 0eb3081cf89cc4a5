#!/usr/bin/env python3
"""bench.py -- headline benchmark: 2048-bit Paillier decryptions/s on N MI355X (BASELINE.json `metric`).

One "step" = one pass of the hot path (SecretKey.Decrypt, paillier.go:292-303, for a whole batch) over one
batch of synthetic ciphertexts that are already resident in HBM (big-endian, element-major: the C-ABI operand
format).  The timed region contains everything a caller pays per batch: unpack -> CRT modexp over p^2 and
q^2 (the VM kernel) -> L / CRT recombination -> pack.  Results are checked bit-exactly after the timed
region (decrypt(encrypt(m, r)) == m for the full batch, plus an oracle spot check on rank 0).

N > 1: launched by torch.distributed.run, one rank per GPU.  Ciphertext batches shard with no data-path
collective (scaling = weak: every rank decrypts its own `--batch`); RCCL is used for the barrier and the
MAX-over-ranks time only.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# measured on MI355X: profiles/r01_valu_rates.txt (v_mad_u64_u32, 4 waves/SIMD): 36.0 T lane-ops/s.
# Theoretical issue limit: 256 CU x 4 SIMD x 64 lanes / 4 cycles x 2.4 GHz = 39.3 T/s.  The roofline uses the
# theoretical figure (the stricter denominator).
PEAK_MAD_PER_S = 256 * 4 * 64 / 4 * 2.4e9
HBM_PEAK_GBPS = 8000.0
TRAFFIC_DEFAULT = 8.0e9   # HBM bytes per launch of the dominant kernel for the default workload (PMC passes, profiles/)


def alg_mul32_per_modexp(mod_bits: int, exp_bits: int, w: int = 5) -> float:
    """SURVEY.md §8(d) unit: CIOS on W 32-bit words = 2W^2+W multiply-adds; fixed window w:
    e + ceil(e/w) + 2^w - 2 + 2 Montgomery products."""
    W = mod_bits // 32
    return (exp_bits + -(-exp_bits // w) + (1 << w)) * (2 * W * W + W)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="ciphertexts per GPU per step")
    ap.add_argument("--bits", type=int, default=2048, choices=[1024, 2048, 3072])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (wall seconds)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import paillier_amd as pa
    from paillier_amd.api import MEM_DEVICE

    with open(os.path.join(ROOT, "tests", "golden", "keys.json")) as f:
        k = json.load(f)["paillier"][str(args.bits)]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)

    ctx = pa.Context(local_rank, torch.cuda.current_stream().cuda_stream)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    assert sk.has_crt
    B = args.batch
    pb, cb = pk.plain_bytes(), pk.cipher_bytes()

    # synthetic inputs: uniform m in [0, n) and r in Z_n^* from a seeded PRNG (seed differs per rank);
    # ciphertexts are produced by the engine's own Encrypt (setup, untimed).
    rng = np.random.default_rng(1234 + rank)
    def rand_below_n(count):
        raw = rng.integers(0, 256, size=(count, pb), dtype=np.uint8)
        raw[:, 0] %= np.uint8(n >> (8 * (pb - 1)))  # top byte strictly below n's top byte => value < n
        return raw
    m_host = rand_below_n(B)
    r_host = rand_below_n(B)
    r_host[:, -1] |= 1
    m_dev = torch.from_numpy(m_host).to(dev)
    r_dev = torch.from_numpy(r_host).to(dev)
    c_dev = torch.zeros((B, cb), dtype=torch.uint8, device=dev)
    out_dev = torch.zeros((B, pb), dtype=torch.uint8, device=dev)
    pk.encrypt_with_r_raw(B, m_dev.data_ptr(), pb, r_dev.data_ptr(), pb, c_dev.data_ptr(), cb, MEM_DEVICE)
    enc_prof = ctx.last_profile()

    def step():
        sk.decrypt_raw(B, c_dev.data_ptr(), cb, out_dev.data_ptr(), pb, MEM_DEVICE)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    vm_ms = []
    for _ in range(args.steps):
        step()
        vm_ms.append(ctx.last_profile()["vm_ms"])  # HIP events on the launch stream; already complete (blocking API)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # parity at full size: round trip is the identity on every lane, on every rank
    ok = bool(torch.equal(out_dev, m_dev))
    if not ok:
        bad = int((out_dev != m_dev).any(dim=1).sum().item())
        raise SystemExit(f"[bench] rank {rank}: PARITY FAILURE: {bad} of {B} decryptions differ from the plaintexts")

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    prof = ctx.last_profile()
    vm_ms_avg = sum(vm_ms) / len(vm_ms)
    value = world * B * args.steps / elapsed
    half = args.bits // 2
    # SURVEY.md §8(d) unit (schoolbook CIOS on 32-bit words, fixed 5-bit window): kept for reference.  The engine needs
    # fewer multiplies than that (squaring symmetry, sliding windows and -- for 2048-bit keys -- residues modulo p^2 as two
    # base-p digits: 3.5 instead of 6 half-width products per squaring), so a rate in SURVEY units can exceed the issue
    # peak.  The roofline therefore prices the kernel in the multiplies its own algorithm needs: 28-bit limb products,
    # counted per VM opcode by the library (squarings H(H-1)/2 + H + 3H^2, products 5H^2 on the pair kernel).
    alg_mul32 = 2 * alg_mul32_per_modexp(args.bits, half)
    alg_mul32_noncrt = alg_mul32_per_modexp(2 * args.bits, args.bits)
    alg_mads = prof["vm_mads"] / B                     # 28-bit multiply-adds the ladder programs need per decryption
    achieved = prof["vm_mads"] / (vm_ms_avg * 1e-3)
    alg_bytes = cb + pb  # read c (n^2 bytes) + write m (n bytes)
    kernel = {2048: "vm_asm_37_16 (pair kernel: x^(p-1) mod p^2 and x^(q-1) mod q^2 ladders, both halves in one launch)",
              1024: "vm_asm_37_1 (CRT modexp over p^2 and q^2)",
              3072: "vm_asm_55_32 (two-lane pair kernel: ladders modulo p^2 and q^2, both halves in one launch)"}[args.bits]
    roofline = {
        "bound": "valu",  # integer multiply issue (v_mad_u64_u32); neither HBM nor MFMA binds
        "kernel": kernel,
        "achieved": achieved / 1e12,
        "peak": PEAK_MAD_PER_S / 1e12,
        "unit": "Tmad28/s",
        "frac": achieved / PEAK_MAD_PER_S,
        "kernel_ms_per_launch": vm_ms_avg,
        "alg_mad28_per_decrypt": alg_mads,
        "survey_unit": {"alg_mul32_per_decrypt_crt": alg_mul32, "alg_mul32_per_decrypt_noncrt": alg_mul32_noncrt,
                        "rate_Tmul32_per_s": alg_mul32 * B / (vm_ms_avg * 1e-3) / 1e12,
                        "frac_of_issue_peak": alg_mul32 * B / (vm_ms_avg * 1e-3) / PEAK_MAD_PER_S,
                        "note": "above 1: the engine's algorithm needs fewer multiplies than the schoolbook count of SURVEY 8(d)"},
        # HBM bytes per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # runs, 2 x FETCH_SIZE gfx950 correction): profiles/r01_bench_pmc_summary.txt.  It is the per-lane window table
        # (written once, one entry read per window product), not re-reads of the inputs.  Only known for the default workload.
        "traffic": TRAFFIC_DEFAULT if (args.bits == 2048 and B == 65536) else None,
        "hbm": {"achieved": alg_bytes * B / (vm_ms_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": alg_bytes * B / (vm_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, "alg_bytes_per_decrypt": alg_bytes},
    }

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import gmp_oracle as go  # CPU baseline leg: the libgmp restatement of paillier.go:292-303
        threads = min(16, os.cpu_count() or 1)
        c_host = c_dev[:4096].cpu().numpy()
        t = time.perf_counter()
        out1, _ = go.decrypt_batch_raw(n, lam, c_host[:32], pb, threads=1)
        per_op = (time.perf_counter() - t) / 32
        sample = int(max(64, min(4096, args.cpu_seconds * threads / per_op)))
        t = time.perf_counter()
        outc, used = go.decrypt_batch_raw(n, lam, c_host[:sample], pb, threads=threads)
        dt = time.perf_counter() - t
        assert (outc == m_host[:sample]).all(), "CPU baseline disagrees with the plaintexts"
        assert (out_dev[:sample].cpu().numpy() == outc).all(), "GPU result differs from the libgmp oracle"
        cpu_baseline = {
            "value": sample / dt, "unit": "decryptions/s", "cores": int(used), "kind": "port",
            "single_thread_value": 1.0 / per_op,
            "sample": f"{sample} of the same {args.bits}-bit ciphertexts; libgmp {go.load().oracle_gmp_version().decode()} "
                      f"mpz_powm call sequence of paillier.go:292-303 (no CRT, lambda^-1 per call), {used} OpenMP threads",
        }

    line = {
        "metric": f"paillier_{args.bits}bit_decryptions_per_s",
        "value": value,
        "unit": "decryptions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (28-bit limbs, 64-bit accumulators)",
        "data": "synthetic",
        "bit_exact": ok,
        "config": {"workload": f"Batch {B} Decrypt per GPU, {args.bits}-bit n, level 1, CRT over p^2,q^2; inputs resident "
                               f"in HBM as big-endian element-major bytes",
                   "batch_per_gpu": B, "key_bits": args.bits, "parallelism": f"batch-sharded x{world}, no data-path collective"},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "encrypt_setup": {"vm_ms": enc_prof["vm_ms"], "encryptions_per_s": B / (enc_prof["vm_ms"] * 1e-3)},
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
