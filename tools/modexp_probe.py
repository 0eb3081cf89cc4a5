#!/usr/bin/env python3
"""Shared- and per-number-exponent ladders on the GENERIC kernels by modulus width and batch size (pgpu_modexp): executed fraction of
the issue peak from the library's own launch profile.  usage: modexp_probe.py [bits ...]   (default 1024 2048: vm_asm_37_1 / 74_1)"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
PEAK = 256 * 4 * 64 / 4 * 2.4e9
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
rng = random.Random(1)
for bits in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    n = rng.getrandbits(bits) | 1 | (1 << (bits - 1))
    m = pa.Modulus(ctx, n)
    nbytes = bits // 8
    e = rng.getrandbits(bits - 1) | (1 << (bits - 2))
    for B in (16384, 32768, 65536, 131072):
        x = torch.from_numpy(np.random.default_rng(B).integers(0, 256, size=(B, nbytes), dtype=np.uint8)).cuda()
        x[:, 0] = 0
        o = torch.zeros_like(x)
        for per_number in (0, 1):
            if per_number:
                ee = torch.from_numpy(np.random.default_rng(B + 1).integers(0, 256, size=(B, nbytes), dtype=np.uint8)).cuda()
                call = lambda: m.exp_raw(B, x.data_ptr(), nbytes, ee.data_ptr(), nbytes, nbytes, o.data_ptr(), nbytes, MEM_DEVICE)
            else:
                call = lambda: m.exp_raw(B, x.data_ptr(), nbytes, e.to_bytes(nbytes, "big"), nbytes, 0, o.data_ptr(), nbytes, MEM_DEVICE)
            call(); call()
            pr = ctx.last_profile()
            print(f"{bits} bits  B {B:6d}  {'per-number' if per_number else 'shared    '} exponent  {pr['kernel']:14s} {pr['vm_ms']:8.3f} ms  "
                  f"{pr['vm_mads'] / (pr['vm_ms'] * 1e-3) / PEAK:.3f} of peak", flush=True)
