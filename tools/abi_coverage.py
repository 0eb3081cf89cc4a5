#!/usr/bin/env python3
"""Which binding calls which entry point of include/paillier_hip.h: the Go shim (go/*.go: `C.<symbol>(`), the plain-C99 test with
cgo's calling convention (tests/c/test_cabi.c), the C++ mirror (paillier_amd/host/paillier.hpp) and the ctypes mirror
(paillier_amd/*.py).  Prints the table of INTEGRATION.md section 3b; tests/test_abi_cpu.py asserts that the Go, C and ctypes
columns have no gap.  (CPU; reads sources only.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header="paillier_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgpu_[a-z0-9_]+)\s*\(", text)))


def read(*parts):
    return open(os.path.join(ROOT, *parts)).read()


def go_method(sym, go_src):
    """name of the first Go func whose body calls C.<sym>("""
    pos = go_src.find("C." + sym + "(")
    if pos < 0:
        return None
    heads = list(re.finditer(r"^func (?:\([^)]*\) )?([A-Za-z0-9_]+)\(", go_src[:pos], flags=re.M))
    return heads[-1].group(1) if heads else "?"


def coverage():
    go = read("go", "gpu.go") + read("go", "batch.go")
    c = read("tests", "c", "test_cabi.c")
    cpp = read("paillier_amd", "host", "paillier.hpp")
    py = "".join(read("paillier_amd", f) for f in os.listdir(os.path.join(ROOT, "paillier_amd")) if f.endswith(".py"))
    rows = []
    for s in declared():
        rows.append({"symbol": s, "go": go_method(s, go), "c": (s + "(") in c, "cpp": (s + "(") in cpp,
                     "py": ("lib." + s + "(") in py or ("." + s + "(") in py})
    return rows


if __name__ == "__main__":
    print("| entry point | Go shim (`go/*.go`) | `tests/c/test_cabi.c` | C++ mirror | ctypes |")
    print("|---|---|---|---|---|")
    for r in coverage():
        y = lambda b: "yes" if b else "—"
        print(f"| `{r['symbol']}` | {('`' + r['go'] + '`') if r['go'] else '**missing**'} | {y(r['c'])} | {y(r['cpp'])} | {y(r['py'])} |")
