"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/voltrix_oracle.c (built by
``make -C oracle`` into ``oracle/_build/libvoltrix_oracle.so``).  Same functions as
oracle_np.py, fast enough for the reference's own test sizes (N=8192, density 0.1)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvoltrix_oracle.so")
_lib = None

_ROUNDING = {None: 0, "none": 0, "fp32": 0, "tf32": 1, "fp16": 2}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "voltrix_oracle.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_preprocess.restype = ctypes.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def preprocess(indptr, indices, num_nodes):
    indptr, indices = _i32(indptr), _i32(indices)
    nw = (num_nodes + 15) // 16
    bp = np.zeros(nw, np.int32)
    e2c = np.zeros(indices.shape[0], np.int32)
    e2r = np.zeros(indices.shape[0], np.int32)
    p1 = np.zeros(nw + 1, np.int32)
    lib().oracle_preprocess(_p(indices), _p(indptr), ctypes.c_int32(num_nodes), _p(bp), _p(e2c), _p(e2r), _p(p1))
    return bp, e2c, e2r, p1


def hmat_gen(indptr, indices, block_partition, edge_to_column, edge_to_row, pointer1, num_nodes):
    indptr, indices = _i32(indptr), _i32(indices)
    total = int(pointer1[-1])
    hspa = np.zeros(total * 128, np.float32)
    hind = np.zeros(total * 8, np.int32)
    lib().oracle_hmat_gen(_p(indptr), _p(indices), _p(_i32(block_partition)), _p(_i32(edge_to_column)),
                          _p(_i32(edge_to_row)), _p(_i32(pointer1)), ctypes.c_int32(len(pointer1) - 1),
                          ctypes.c_int32(num_nodes), _p(hspa), _p(hind))
    return hspa, hind


def hmat_packed_swizzle(pointer1, hspa):
    total = int(pointer1[-1])
    packed = np.zeros(total * 4, np.uint32)
    hspa = np.ascontiguousarray(hspa, dtype=np.float32)
    lib().oracle_hmat_packed_swizzle(ctypes.c_int32(len(pointer1) - 1), _p(_i32(pointer1)), _p(hspa), _p(packed))
    return packed


def csr_preprocess(indptr, indices, num_nodes):
    """(pointer1, hspa_packed, hind) exactly as voltrix/spmm/spmm.py:16-89 returns them."""
    bp, e2c, e2r, p1 = preprocess(indptr, indices, num_nodes)
    hspa, hind = hmat_gen(indptr, indices, bp, e2c, e2r, p1, num_nodes)
    return p1, hmat_packed_swizzle(p1, hspa), hind


def spmm_blocked(pointer1, hspa_packed, hind, num_nodes, feat, rounding="tf32"):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    out = np.zeros((num_nodes, feat.shape[1]), np.float32)
    lib().oracle_spmm_blocked(_p(_i32(pointer1)), _p(np.ascontiguousarray(hspa_packed, dtype=np.uint32)),
                              _p(_i32(hind)), ctypes.c_int32(num_nodes), ctypes.c_int32(feat.shape[1]),
                              _p(feat), _p(out), ctypes.c_int32(_ROUNDING[rounding]))
    return out


def spmm_csr(indptr, indices, feat, num_nodes, rounding="none"):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    out = np.zeros((num_nodes, feat.shape[1]), np.float32)
    lib().oracle_spmm_csr(_p(_i32(indptr)), _p(_i32(indices)), ctypes.c_int32(num_nodes),
                          ctypes.c_int32(feat.shape[1]), _p(feat), _p(out), ctypes.c_int32(_ROUNDING[rounding]))
    return out
