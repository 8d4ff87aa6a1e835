"""A compiled C caller of the drop-in boundary (SURVEY.md 8b "Callers": the reference's own C user is example/c_example.c:99-140).
ctypes proves symbols and argument order; this proves that include/cmfrec_hip.h compiles as C99 (bool, size_t, prototypes) for a
C translation unit that links libcmfrec_hip_double.so, and that such a program gets the reference's numbers (fixture g5)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_cases as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_caller", "fit_implicit_caller.c")


def build_caller(tmp_path):
    exe = str(tmp_path / "fit_implicit_caller")
    libdir = os.path.join(ROOT, "cmfrec_amd", "lib")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-lcmfrec_hip_double", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_header_compiles_as_c99_and_links(tmp_path):
    """CPU part: the header is valid C99 for a C compiler and every symbol the caller uses resolves at link time."""
    exe = build_caller(tmp_path)
    assert os.path.exists(exe)
    # both precisions of the header, syntax only
    for flag in ([], ["-DCMFREC_HIP_FLOAT"]):
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                               "-I", os.path.join(ROOT, "include")] + flag + [os.path.join(ROOT, "include", "cmfrec_hip.h")])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cg", "chol", "cgfin"])
def test_c_caller_reproduces_g5(tmp_path, mode):
    exe = build_caller(tmp_path)
    g = gc.load("g5_fit_implicit", np.float64)
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    row = np.ascontiguousarray(g["row"], np.int32); col = np.ascontiguousarray(g["col"], np.int32)
    val = np.ascontiguousarray(g["val"], np.float64); A0 = np.ascontiguousarray(g["A0"], np.float64)
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<6i", m, n, k, int(g["niter"]), int(mode != "chol"), int(mode == "cgfin")))
        f.write(struct.pack("<q", len(val)))
        f.write(struct.pack("<2d", float(g["lam"]), float(g["alpha"])))
        f.write(row.tobytes()); f.write(col.tobytes()); f.write(val.tobytes()); f.write(A0.tobytes())
    subprocess.check_call([exe, inp, out])
    raw = open(out, "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == 0
    AB = np.frombuffer(raw[4:], np.float64)
    A, B = AB[:m * k].reshape(m, k), AB[m * k:].reshape(n, k)
    assert gc.frob(A, g["A_" + mode]) < 1e-6 and gc.frob(B, g["B_" + mode]) < 1e-6      # the whole-fit tolerance of test_gpu_golden.py
