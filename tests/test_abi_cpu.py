"""CPU-side checks of the drop-in boundary (no GPU needed): the C-ABI library loads, exports every symbol that
include/paillier_hip.h declares, refuses to run without a device (no CPU fallback), and the product package never
imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    import paillier_amd as pa
    return pa.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "paillier_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/paillier_hip.h but not exported"


def test_python_binding_covers_the_header(lib):
    from paillier_amd.api import SIGNATURES
    assert sorted(SIGNATURES) == declared_symbols()


def test_debug_hooks_are_not_part_of_the_boundary(lib):
    """Test hooks (raw VM programs, the planning predicates) live in include/paillier_hip_debug.h: exported for the tests,
    declared nowhere in the drop-in header, bound by no Go / C / C++ host."""
    from paillier_amd.api import DEBUG_SIGNATURES
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import abi_coverage
    dbg = [s for s in abi_coverage.declared("paillier_hip_debug.h") if s not in declared_symbols()]
    assert sorted(dbg) == sorted(DEBUG_SIGNATURES) and len(dbg) == 3
    for s in dbg:
        assert hasattr(lib, s)
        assert s not in declared_symbols()
        for f in (("go", "gpu.go"), ("go", "batch.go"), ("tests", "c", "test_cabi.c"), ("paillier_amd", "host", "paillier.hpp")):
            assert s not in open(os.path.join(ROOT, *f)).read(), f"{s} leaked into {f[-1]}"


def test_every_entry_point_is_bound_in_go_called_from_c_and_from_python():
    """header <-> Go shim <-> plain-C test <-> ctypes, one line per symbol, no gaps."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import abi_coverage
    rows = abi_coverage.coverage()
    assert len(rows) == len(declared_symbols())
    for r in rows:
        assert r["go"], f"{r['symbol']} has no Go binding in go/*.go"
        assert r["c"], f"{r['symbol']} is not called by tests/c/test_cabi.c"
        assert r["py"], f"{r['symbol']} is not called by paillier_amd/*.py"
    table = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for r in rows:
        assert f"| `{r['symbol']}` | `{r['go']}` |" in table, f"INTEGRATION.md 3b is stale for {r['symbol']} (python tools/abi_coverage.py)"


def test_library_exports_only_the_headers(lib):
    """-fvisibility=hidden: the dynamic symbol table holds the pgpu_* functions of the two headers and no other function."""
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "paillier_amd", "libpaillier_hip.so")], capture_output=True,
                         text=True, check=True).stdout
    funcs = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "Tt")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import abi_coverage
    assert funcs == sorted(set(declared_symbols()) | set(abi_coverage.declared("paillier_hip_debug.h")))


def test_no_cpu_fallback(lib):
    """Without a gfx950 device context creation must fail loudly; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import paillier_amd as pa
    with pytest.raises(pa.PaillierHipError) as ei:
        pa.Context(0)
    assert ei.value.code == -2


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "paillier_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.replace("random_oracle", ""), f"{f} mentions the oracle"
    assert "oracle" not in open(os.path.join(ROOT, "bench.py")).read().split("cpu_baseline = None")[0].replace(
        "oracle spot check", ""), "bench.py may touch the oracle only in its cpu_baseline leg"


def test_asm_generator_model():
    """The assembly generator must emit exactly 2*WL multiplies per row for every shape and stay within 256 VGPRs."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "paillier_amd", "csrc"))
    import gen_vm_asm
    for wl, k in gen_vm_asm.SHAPES:
        g = gen_vm_asm.make_gen(wl, k)
        text = g.generate()
        assert g.n_vgpr <= 256
        # number-major window tables: VM_STORET with VM_MULVT / VM_MULVT5 on the pair kernels that run per-number windows (the
        # one-lane kernel for 37-limb primes and its four-lane twin, the two- and four-lane kernels), VM_MULV7 / VM_MULVT5 on the
        # three-digit kernel
        has = lambda lbl: f"\n{lbl}:" in text
        if (wl, k) in ((37, 16), (74, 32), (55, 32), (37, 32), (37, 64), (37, 1), (10, 4)):
            assert has("L_storet") and has("L_mulvt") and has("L_mulvt5") and not has("L_mulv7")
        elif (wl, k) in gen_vm_asm.TRIPLE:
            assert has("L_storet") and has("L_mulv7") and has("L_mulvt5") and not has("L_mulvt")
        else:
            assert not has("L_storet") and not has("L_mulvt") and not has("L_mulvt5")
        if (wl, k) in gen_vm_asm.PAIR:
            # pair kernel: H(H-1)/2 + H + H^2 multiplies in the unrolled phase 1 of a squaring, 2 H^2 in phase 1 of a product,
            # phase-2 rows of 2H (squaring) and 3H (product) multiplies, one peeled row + a two-row loop body each
            H = wl
            sq = text.split("L_montsq:")[1]
            p1 = sq.split("L_p2s:")[0]
            assert p1.count("v_mad_u64_u32") == H * (H - 1) // 2 + H + H * H + 2 * H        # + the peeled phase-2 row
            assert sq.split("L_p2s:")[1].split("s_cbranch_scc1 L_p2s")[0].count("v_mad_u64_u32") == 2 * 2 * H
            mm = text.split("L_montmul:")[1].split("L_montsq:")[0]
            if isinstance(g, gen_vm_asm.GenP2):
                # the wider one-lane kernel: both phases of a product are loops of five-row bodies (2H / 3H multiplies a
                # row) followed by a five-row tail; LDS for two workgroups per CU
                RB = g.RB
                assert mm.split("L_m1:")[1].split("s_cbranch_scc1 L_m1")[0].count("v_mad_u64_u32") == RB * 2 * H
                assert mm.split("L_m2:")[1].split("s_cbranch_scc1 L_m2")[0].count("v_mad_u64_u32") == RB * 3 * H
                assert mm.count("v_mad_u64_u32") == 2 * RB * 5 * H
                assert mm.count("global_load_dword") == 3 * g.PD + 3 * (RB + 1)     # ring fills + one load per row and stream
                assert g.lds_bytes * 2 <= 160 * 1024
                continue
            assert mm.split("L_p2m:")[0].count("v_mad_u64_u32") == 2 * H * H + 3 * H
            assert mm.split("L_p2m:")[1].split("s_cbranch_scc1 L_p2m")[0].count("v_mad_u64_u32") == 2 * 3 * H
            continue
        if (wl, k) in gen_vm_asm.PAIR8:
            # eight-lane pair kernel: GenQ4's rows with four slices per digit (19-limb slices of a 76-limb digit): four-row bodies of
            # 2 WL (squaring) / 3 WL (one-pass product) multiplies per lane, the link across quads by row_shr:4
            for lbl, where, per_row in (("L_qs", "L_montsq:", 2 * wl), ("L_qm", "L_montmul:", 3 * wl)):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * per_row
                # (10-limb slices: a linked squaring row is too short to hide its chain between multiplies -- one wait state in front
                # of each of its two DPP reads)
                assert body.count("row_shr:4") == 4 and body.count("s_nop") == (8 if (wl == 10 and lbl == "L_qs") else 0)
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 4 * per_row <= 4 * (13 if wl == 10 else 11)
            assert g.H == 4 * wl and g.lds_bytes * 2 <= 160 * 1024
            continue
        if isinstance(g, gen_vm_asm.GenS4):
            # one digit in the four lanes of a quad (the 37-limb primes as 40-limb moduli): four-row bodies of 2 WL multiplies per lane
            # in a squaring and in a product alike, no quotient link, no s_nop, at most 7 other instructions a row
            for lbl, where in (("L_qs", "L_montsq:"), ("L_qm", "L_montmul:")):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * 2 * wl
                assert body.count("v_mad_i64_i32") == 0 and body.count("s_nop") == 0
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 8 * wl <= 4 * 7
            assert g.H == 4 * wl == g.WT and g.lds_bytes * 4 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.PAIR16:
            # sixteen-lane pair kernel: GenQ4's rows with eight slices per digit (a DPP row per number): the link by row_shr:8, the
            # quotient digit's broadcast in two steps, neighbours by row_shl:1 with bound_ctrl (lane 15 reads zero)
            for lbl, where, per_row in (("L_qs", "L_montsq:", 2 * wl), ("L_qm", "L_montmul:", 3 * wl)):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * per_row
                assert body.count("row_shr:8") == 4 and body.count("bank_mask:0xa") == 4 and body.count("bound_ctrl:1") == 4
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 4 * per_row <= 4 * 15
            assert g.H == 8 * wl and g.NPB == 16 and g.lds_bytes * 4 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.PAIR4:
            # four-lane pair kernel: a squaring is one pass of four-row bodies of 2 WL multiplies per lane; a product is ONE pass
            # too, with two multiplier streams (3 WL multiplies a row); one quotient link per row; no s_nop inside a row, and at
            # most 11 instructions a row that are not multiplies (one wave per SIMD: each of them costs a multiply's issue slot)
            for lbl, where, per_row in (("L_qs", "L_montsq:", 2 * wl), ("L_qm", "L_montmul:", 3 * wl)):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * per_row
                assert body.count("quad_perm:[0,1,0,3]") == 4 and body.count("s_nop") == 0
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 4 * per_row <= 4 * 11
            assert g.lds_bytes * 2 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.TRIPLE4:
            # three-digit kernel with four lanes per digit (a quad per digit, a DPP row per number): GenQ6's passes with the links by
            # row_shr:8 / row_shl:4
            for lbl, where, hops in (("L_qs", "L_montsq:", 2), ("L_qm1", "L_montmul:", 2), ("L_qm2", "L_montmul:", 1)):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * 2 * wl
                assert body.count("v_mad_i64_i32") == 4 * hops and body.count("s_nop") == 0
                assert body.count("row_shr:8") == 4 and body.count("row_shl:4") == (4 if hops == 2 else 0)
            assert g.H == 4 * wl and g.NPB == 16 and g.n_vgpr <= 256 and g.lds_bytes * 4 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.TRIPLE2:
            # three-digit kernel with two lanes per digit: four-row bodies of 2 WL multiplies per lane in every pass (squaring: one
            # pass; product: two), both quotient links in the linked passes (row_shr:4, row_shl:2), one in pass two of a product
            for lbl, where, hops in (("L_qs", "L_montsq:", 2), ("L_qm1", "L_montmul:", 2), ("L_qm2", "L_montmul:", 1)):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * 2 * wl
                assert body.count("v_mad_i64_i32") == 4 * hops and body.count("s_nop") == 0
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 8 * wl <= 4 * (7 + 3 * hops)
            assert g.H == 2 * wl and g.n_vgpr <= 256 and g.lds_bytes * 2 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.TRIPLE:
            # three-digit kernel: every pass is a loop of four single-lane rows of 2H multiplies (a squaring: one pass in four
            # lanes; a product: two passes, the second in two lanes); two quotient links per row in the linked passes
            # (H = 37, 55: the registers allow a copy of the digit below, and the product is ONE pass with two multiplier streams)
            loops = ((("L_qs", "L_montsq:", 2, 2), ("L_qm", "L_montmul:", 2, 3)) if g.merged else
                     (("L_qs", "L_montsq:", 2, 2), ("L_qm1", "L_montmul:", 2, 2), ("L_qm2", "L_montmul:", 1, 2)))
            assert g.merged == (wl <= 55)
            for lbl, where, hops, streams in loops:
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * streams * wl
                assert body.count("quad_perm:[0,0,1,2]") == 4 * hops and body.count("s_nop") == 0
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 4 * streams * wl <= 4 * 13
            assert g.lds_bytes * 2 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.PAIR2:
            # two-lane pair kernel: every pass is a loop of four single-lane rows of 2H multiplies; a squaring has one pass,
            # a product two
            for lbl, where in (("L_qs", "L_montsq:"), ("L_qm1", "L_montmul:"), ("L_qm2", "L_montmul:")):
                body = text.split(where)[1].split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert body.count("v_mad_u64_u32") == 4 * 2 * wl
                assert body.count("s_nop") == 0 and body.count("s_load_dword") == 0      # Cadj enters once per pass, not per row
                assert len([l for l in body.splitlines() if l.strip() and not l.strip().startswith(".")]) - 8 * wl <= 4 * 10
            assert g.lds_bytes * 2 <= 160 * 1024
            continue
        if (wl, k) in gen_vm_asm.WAVE_SLICED:
            # slices in different waves: each wave's product loop is two rows of 2*WL multiplies; the squaring rows of both
            # waves are triangular tables of WL-1 / WL entries behind a computed jump, plus WL reduction multiplies
            for lbl in ("L_rowb", "L_rowt"):
                row = text.split(lbl + ":")[1].split("s_cbranch_scc1 " + lbl)[0]
                assert row.count("v_mad_u64_u32") == 2 * wl * 2
            sq = text.split("L_sqb:")[1].split("s_cbranch_scc1 L_sqb")[0]
            assert sq.count("v_mad_u64_u32") == 2 * (wl - 1) + 2 * wl + 1          # two rows + one diagonal
            assert g.lds_bytes * 2 <= 160 * 1024, "two workgroups per CU must fit in LDS"
            continue
        if isinstance(g, gen_vm_asm.GenM):
            # the unrolled one-lane kernel: H(H-1)/2 + H + H^2 multiplies in a squaring, 2 H^2 in a product, no loop, and a squaring
            # touches no LDS
            H = wl
            sq = text.split("L_montsq:")[1].split("s_branch L_next")[0]
            mm = text.split("L_montmul:")[1].split("s_branch L_next")[0]
            assert sq.count("v_mad_u64_u32") == H * (H - 1) // 2 + H + H * H and sq.count("ds_") == 0 and "s_cbranch" not in sq
            assert mm.count("v_mad_u64_u32") == 2 * H * H and mm.count("ds_read_b32") == H and "s_cbranch" not in mm
            continue
        row = text.split("L_row:")[1].split("s_cbranch_scc1 L_row")[0]
        if "L_noflush" in row:
            row = row.split("s_add_u32 s19, s19, 1")[0]
        bodies = 2 if (wl * k) % 2 == 0 else 1       # the row loop is unrolled by two when the limb count is even
        assert row.count("v_mad_u64_u32") == 2 * wl * bodies


def _run_py(code):
    import subprocess
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_one_hip_runtime_whichever_is_loaded_first():
    """libpaillier_hip.so and PyTorch must share ONE libamdhip64 in a process, in either load order (two copies: the second
    runtime to initialise finds no GPU -- VERDICT r2).  No GPU needed: only the mappings are inspected."""
    pytest.importorskip("torch")
    first_lib = _run_py("from paillier_amd import api; api.load_library(); import torch; print(len(api.hip_runtimes_mapped()))")
    first_torch = _run_py("import torch; from paillier_amd import api; api.load_library(); print(len(api.hip_runtimes_mapped()))")
    assert first_lib.strip().splitlines()[-1] == "1" and first_torch.strip().splitlines()[-1] == "1"
