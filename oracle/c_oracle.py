"""ctypes wrapper of oracle/lkg_oracle.c (plain-C restatement; TEST INFRASTRUCTURE ONLY).
``build_c_oracle()`` compiles it with gcc into oracle/_build/ (git-ignored)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "lkg_oracle.c")
LIB = os.path.join(HERE, "_build", "liblkg_oracle.so")
_lib = None


def build_c_oracle(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: cannot build the C oracle")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run([gcc, "-O2", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_c_oracle())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def spmm(rowptr, col, val, x):
    rowptr, col = np.ascontiguousarray(rowptr, np.int64), np.ascontiguousarray(col, np.int64)
    val, x = np.ascontiguousarray(val, np.float32), np.ascontiguousarray(x, np.float32)
    out = np.empty((len(rowptr) - 1, x.shape[1]), np.float32)
    rc = _load().oracle_spmm(C.c_int64(len(rowptr) - 1), C.c_int64(x.shape[1]), _p(rowptr), _p(col), _p(val), _p(x), _p(out))
    assert rc == 0
    return out


def attention(h, t, r, ent, rel):
    h, t, r = (np.ascontiguousarray(a, np.int64) for a in (h, t, r))
    ent, rel = np.ascontiguousarray(ent, np.float32), np.ascontiguousarray(rel, np.float32)
    e = len(h)
    rows, cols, vals, nnz = np.empty(e, np.int64), np.empty(e, np.int64), np.empty(e, np.float32), np.zeros(1, np.int64)
    rc = _load().oracle_attention(C.c_int64(e), C.c_int64(ent.shape[1]), _p(h), _p(t), _p(r), _p(ent), _p(rel), _p(rows),
                                  _p(cols), _p(vals), _p(nnz))
    assert rc == 0
    k = int(nnz[0])
    return rows[:k], cols[:k], vals[:k]
