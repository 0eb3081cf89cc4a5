#!/usr/bin/env python3
"""BASELINE config 4: threshold (t=3, l=5) PartialDecrypt + Combine, 2048-bit, 16 384 ciphertexts over N GPUs, with
the share-combine exchange over RCCL.

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
             tools/bench_threshold_multi.py [--backend nccl|gloo] [--batch 16384]

Work units are (server, ciphertext) pairs, server-major: unit u = s * B + i.  Rank r computes the partial decryptions
of its contiguous unit range (so it needs at most two servers' shares -- shares are not replicated everywhere), the
fixed-stride partials are all-gathered (RCCL over xGMI; 3 x B x 512 B = 25 MB at B = 16 384), then every rank combines
its own ciphertext slice locally (big-integer modular products are not a collective reduction op: gather + local kernel).
--backend gloo moves the exchange through host memory so the logic can be rehearsed with several ranks on ONE GPU.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
from paillier_amd.dist import shard_slice, max_over_ranks, env_rank_world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--one-gpu", action="store_true", help="all ranks share GPU 0 (rehearsal with --backend gloo)")
    args = ap.parse_args()
    rank, world, local_rank = env_rank_world()
    devi = 0 if args.one_gpu else local_rank
    torch.cuda.set_device(devi)
    dev = torch.device("cuda", devi)
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    k = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["threshold"]["2048"]
    n, shares = int(k["n"], 16), [int(s, 16) for s in k["shares"]]
    ids = [1, 3, 5]                      # every 3-subset of 5 has a negative Lagrange coefficient
    T, B = len(ids), args.batch
    ctx = pa.Context(devi, torch.cuda.current_stream().cuda_stream)
    tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3)
    rng = np.random.default_rng(4)       # same ciphertexts on every rank (inputs are public)
    def below_n(cnt):
        raw = rng.integers(0, 256, size=(cnt, 256), dtype=np.uint8); raw[:, 0] %= np.uint8(n >> 2040); return raw
    m_h, r_h = below_n(B), below_n(B); r_h[:, -1] |= 1
    m = torch.from_numpy(m_h).to(dev); r = torch.from_numpy(r_h).to(dev)
    c = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    tk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
    units = T * B
    ub, ue = shard_slice(units, rank, world)
    per = -(-units // world)             # padded per-rank unit count (all-gather needs equal shapes)
    cb, ce = shard_slice(B, rank, world)
    out = torch.zeros((max(ce - cb, 1), 256), dtype=torch.uint8, device=dev)

    def step():
        local = torch.zeros((per, 512), dtype=torch.uint8, device=dev)
        u = ub
        while u < ue:                    # at most two servers per rank
            s, i0 = divmod(u, B)
            cnt = min(ue - u, B - i0)
            tk.partial_decrypt_raw(shares[ids[s] - 1], cnt, c[i0:i0 + cnt].data_ptr(), 512, local[u - ub:].data_ptr(), 512, MEM_DEVICE)
            u += cnt
        if world > 1:
            if args.backend == "gloo":
                g = torch.empty((world * per, 512), dtype=torch.uint8)
                dist.all_gather_into_tensor(g, local.cpu())
                g = g.to(dev)
            else:
                g = torch.empty((world * per, 512), dtype=torch.uint8, device=dev)
                dist.all_gather_into_tensor(g, local)
        else:
            g = local
        # un-pad: rank q's units sit at rows [q*per, q*per + len_q); rebuild the server-major [T, B, 512] view
        parts = torch.empty((T * B, 512), dtype=torch.uint8, device=dev)
        for q in range(world):
            qb, qe = shard_slice(units, q, world)
            parts[qb:qe] = g[q * per:q * per + (qe - qb)]
        parts = parts.view(T, B, 512)
        if ce > cb:
            ptrs = [parts[s, cb:ce].contiguous() for s in range(T)]
            tk.combine_raw(ids, ce - cb, [p.data_ptr() for p in ptrs], 512, out.data_ptr(), 256, MEM_DEVICE)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step(); barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    el = max_over_ranks(time.perf_counter() - t0, dev if args.backend == "nccl" else None)
    ok = bool(torch.equal(out[:ce - cb], m[cb:ce]))
    okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "threshold_2048bit_decryptions_per_s", "value": B * args.steps / el, "unit": "decryptions/s",
                          "n_gpus": world, "steps": args.steps, "ms_per_step": el / args.steps * 1e3, "scaling": "strong",
                          "bit_exact": bool(okt.item()), "exchange_MB": T * B * 512 / 1e6, "backend": args.backend,
                          "config": {"workload": f"t=3,l=5 servers {ids}, {B} ciphertexts, (server,ciphertext) units sharded over {world} ranks, all-gather + local combine"}}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    if not okt.item():
        sys.exit(1)


if __name__ == "__main__":
    main()
