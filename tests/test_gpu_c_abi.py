"""Builds tests/c/test_cabi.c with `gcc -std=c99 -pedantic -Wall -Werror` and runs it on the GPU: the C ABI called the way the
reference's cgo shim (go/gpu.go, go/batch.go) calls it -- plain C, caller-owned flat buffers, int status -- against the
committed fixtures.  The compile step alone (header is valid C99, every call type-checks) also runs in the CPU suite."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def build(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = tmp_path / "test_cabi"
    libdir = os.path.join(ROOT, "paillier_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O1", "-pthread", os.path.join(ROOT, "tests", "c", "test_cabi.c"),
                           "-I" + os.path.join(ROOT, "include"), "-o", str(exe), f"-L{libdir}", "-lpaillier_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c99_boundary_compiles(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_c99_boundary_runs(tmp_path):
    exe = build(tmp_path)
    K = json.load(open(os.path.join(G, "keys.json")))
    k, v = K["paillier"]["2048"], json.load(open(os.path.join(G, "vectors.json")))["2048"]
    P = json.load(open(os.path.join(G, "proofs.json")))
    t, th, z, d = K["threshold"]["2048"], P["threshold"], P["share_zkp"]["proofs"], P["ddleq"]
    n = int(k["n"], 16)
    ins = d["instances"][:8]
    st = d["statements"]
    lines = [f"n {k['n']}", f"g {k['g']}", f"h {k['h']}", f"k {k['k']}", f"lambda {k['lambda']}", f"n2 {n * n:x}",
             "enc_m " + " ".join(v["encrypt"]["m"]), "enc_r " + " ".join(v["encrypt"]["r"]), "enc_c " + " ".join(v["encrypt"]["c"]),
             "dec_c " + " ".join(v["decrypt"]["c"]), "dec_m " + " ".join(v["decrypt"]["m"]),
             "add_a " + " ".join(v["add"]["a"]), "add_b " + " ".join(v["add"]["b"]), "add_out " + " ".join(v["add"]["out"]),
             "cm_k0 " + v["const_mult"]["k"][0], "cm_out " + " ".join(v["const_mult"]["out_shared_k0"]),
             "l2_m " + " ".join(P["level2"]["m"]), "l2_r " + " ".join(P["level2"]["r"]), "l2_c " + " ".join(P["level2"]["c"]),
             f"t_n {t['n']}", f"t_v {t['v']}", "t_vks " + " ".join(t["vks"]), "t_shares " + " ".join(t["shares"]),
             "t_c " + " ".join(th["c"]), "t_m " + " ".join(th["m"])]
    lines += [f"t_part{i + 1} " + " ".join(th["partials"][i]) for i in range(5)]
    for key, name in (("c", "z_c"), ("r", "z_r"), ("dec", "z_dec"), ("e", "z_e"), ("z", "z_z"), ("a", "z_a"), ("b", "z_b"),
                      ("c4", "z_c4"), ("ci2", "z_ci2")):
        lines.append(name + " " + " ".join(rec[key] for rec in z))
    for key, name in (("ct1", "d_ct1"), ("ct2", "d_ct2"), ("a", "d_a"), ("b", "d_b")):
        lines.append(name + " " + " ".join(st[i["s"]][key] for i in ins))
    for key, name in (("x", "d_x"), ("y", "d_y"), ("alpha", "d_alpha"), ("e", "d_e"), ("f", "d_f")):
        lines.append(name + " " + " ".join(i[key] for i in ins))
    vec = tmp_path / "vectors.txt"
    vec.write_text("\n".join(lines) + "\n")
    out = subprocess.run([str(exe), str(vec)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c abi ok" in out.stdout


@pytest.mark.gpu
def test_library_first_then_torch_on_the_gpu():
    """The order that used to fail (`No HIP GPUs are available` once torch initialised after the library): create a
    context and run a batch FIRST, then initialise torch's CUDA state and use both on the same device."""
    import subprocess
    code = (
        "import paillier_amd as pa\n"
        "from paillier_amd import api\n"
        "ctx = pa.Context(0)\n"
        "m = pa.Modulus(ctx, (1 << 127) - 1)\n"
        "assert m.exp_batch([3, 5], 65537) == [pow(3, 65537, (1 << 127) - 1), pow(5, 65537, (1 << 127) - 1)]\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "t = torch.arange(8, device='cuda').sum().item()\n"
        "assert t == 28 and len(api.hip_runtimes_mapped()) == 1\n"
        "assert m.exp_batch([7], 3) == [343]\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])
