"""CPU tests of the boundary: the C-ABI library loads, exports every symbol the header
declares, the ctypes table covers them, and the product path has no CPU fallback."""
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "taxidispatcher_amd.h")


def declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"^TD_API[^;(]*?\b(td_[a-z_0-9]+)\s*\(", txt, flags=re.M)))


@pytest.fixture(scope="module")
def built():
    entry.build()
    from taxidispatcher_amd import _ffi
    return _ffi


def test_header_symbols_exported(built):
    syms = declared_symbols()
    assert len(syms) >= 16
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH], text=True)
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [s for s in syms if s not in exported]
    assert not missing, missing
    extra = [s for s in exported if s.startswith("td_") and s not in syms]
    assert not extra, extra


def test_ctypes_table_matches_header(built):
    lib = built.load()
    assert sorted(built.SIGNATURES) == declared_symbols()
    for name in built.SIGNATURES:
        assert getattr(lib, name) is not None
    assert lib.td_version() >= 100


def test_gfx950_code_object_present(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + built.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        assert "gfx950" in out.stdout
    else:  # fall back to a string scan of the fat binary
        assert b"gfx950" in open(built.LIB_PATH, "rb").read()


def test_no_cpu_fallback(built):
    """Without a GPU every compute entry point must fail loudly (never route to the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the failure path is exercised on the CPU tier only")
    lib = built.load()
    assert lib.td_init(0) != 0
    assert b"no HIP device" in lib.td_last_error() or b"HIP" in lib.td_last_error()
    import taxidispatcher_amd as td
    with pytest.raises(td.TdError):
        td.assign(np.zeros((2, 2), np.int32))
    with pytest.raises(td.TdError):
        td.calculate_cost(None, [(0, 1, 2)], [(0, 1, 2)])
    # not initialised -> ENOINIT, not a silent result
    import ctypes
    tot = ctypes.c_int64(0)
    r = np.zeros(2, np.int32)
    c = np.zeros((2, 2), np.int32)
    assert lib.td_assign(2, c.ctypes.data, r.ctypes.data, ctypes.byref(tot), None) == -3


def test_product_never_imports_oracle():
    """No module under taxidispatcher_amd/ imports, includes, dlopens or executes anything from
    oracle/ (checked on the syntax tree / include lines, not on prose)."""
    import ast
    pkg = os.path.join(ROOT, "taxidispatcher_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    if isinstance(node, ast.Import):
                        assert not any(a.name.split(".")[0] == "oracle" for a in node.names), f
                    if isinstance(node, ast.ImportFrom):
                        assert (node.module or "").split(".")[0] != "oracle", f
                    if isinstance(node, ast.Constant) and isinstance(node.value, str) and "\n" not in node.value:
                        assert "liboracle" not in node.value and "td_oracle" not in node.value, f
            elif f.endswith((".hip", ".h")):
                for line in open(path):
                    if line.lstrip().startswith("#include"):
                        assert "oracle" not in line, f


def test_solver_file_protocol_host_side(tmp_path):
    """cost.txt / solv_out.txt formats (Simulator.java:512-518, solver.py:30-39) — host I/O only."""
    from taxidispatcher_amd import solver
    cost = np.array([[3, 3, 0, 2], [1, 1, 2, 4], [5, 5, 2, 0], [16, 16, 16, 16]], np.int32)
    p = tmp_path / "cost.txt"
    solver.write_cost(str(p), cost)
    assert p.read_text() == "4\n3 3 0 2 \n1 1 2 4 \n5 5 2 0 \n16 16 16 16 \n"
    n, back = solver.read_cost(str(p))
    assert n == 4 and np.array_equal(back, cost)
    x = np.zeros(16, np.uint8)
    x[[2, 4, 11, 13]] = 1
    q = tmp_path / "solv_out.txt"
    solver.write_solution(str(q), x)
    assert q.read_text().count("\n") == 16 and set(q.read_text().split()) == {"0", "1"}
    assert np.array_equal(solver.read_solution(str(q), 4), x)
    q.write_text("0\n1\n")
    with pytest.raises(ValueError):
        solver.read_solution(str(q), 4)
