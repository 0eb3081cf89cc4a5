"""Builds and runs the C++ host-mirror test (tests/cpp/test_host_mirror.cpp over paillier_amd/host/paillier.hpp and the C
ABI) on the GPU, feeding it the committed golden vectors."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(tmp_path):
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))["paillier"]["2048"]
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors.json")))["2048"]
    lines = [f"n {k['n']}", f"g {k['g']}", f"lambda {k['lambda']}",
             "enc_m " + " ".join(v["encrypt"]["m"]), "enc_r " + " ".join(v["encrypt"]["r"]), "enc_c " + " ".join(v["encrypt"]["c"]),
             "dec_c " + " ".join(v["decrypt"]["c"]), "dec_m " + " ".join(v["decrypt"]["m"]),
             "add_a " + " ".join(v["add"]["a"]), "add_b " + " ".join(v["add"]["b"]), "add_out " + " ".join(v["add"]["out"]),
             "cm_k0 " + v["const_mult"]["k"][0], "cm_out " + " ".join(v["const_mult"]["out_shared_k0"])]
    vec = tmp_path / "vectors.txt"
    vec.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "test_host_mirror"
    libdir = os.path.join(ROOT, "paillier_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-o", str(exe),
                           f"-L{libdir}", "-lpaillier_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), str(vec)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror ok" in out.stdout
