import sys, time, json, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
k = json.load(open('/root/repo/tests/golden/keys.json'))["paillier"]["2048"]
p, q = int(k["p"], 16), int(k["q"], 16); n = p*q; lam = (p-1)*(q-1)
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, n, n+1); sk = pa.SecretKey(ctx, pk, lam)
B = 65536
rng = np.random.default_rng(1)
m = torch.from_numpy(rng.integers(0, 255, size=(B, 256), dtype=np.uint8)).to(dev)
r = torch.from_numpy(rng.integers(1, 255, size=(B, 256), dtype=np.uint8)).to(dev)
c = torch.zeros((B, 512), dtype=torch.uint8, device=dev); out = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
pk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    sk.decrypt_raw(B, c.data_ptr(), 512, out.data_ptr(), 256, MEM_DEVICE)
    torch.cuda.synchronize(); print(i, round((time.perf_counter() - t) * 1e3, 2), "ms", ctx.last_profile()["vm_ms"])
