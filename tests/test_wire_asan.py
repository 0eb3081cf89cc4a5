"""pgpu_gob_unpack on untrusted blobs under AddressSanitizer + UBSan (CPU build of wire.cpp only; GPU sanitizers are not
available on the pool).  tests/cpp/fuzz_gob.cpp mutates blobs pgpu_gob_pack wrote -- byte flips, truncations, 64-bit varints
spliced in at every position -- and every call must come back PGPU_OK or PGPU_ERR_INVALID with no sanitizer report.  The harness
finds round 4's out-of-bounds field index (wire.cpp:193) within a few thousand iterations on the old source."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_gob_unpack_fuzz_under_asan(tmp_path):
    exe = tmp_path / "fuzz_gob"
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "paillier_amd", "csrc", "wire.cpp"),
                           os.path.join(ROOT, "tests", "cpp", "fuzz_gob.cpp"), "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", str(exe)])
    out = subprocess.run([str(exe), "150000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fuzz_gob ok" in out.stdout
