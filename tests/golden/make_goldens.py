#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run in the BUILD container only (it imports the reference's pure-Python modules from
/root/reference, which does not exist on the GPU box):

    python tests/golden/make_goldens.py

What it writes
--------------
ref_python_goldens.json
    Outputs of the *reference itself* (imported, not copied): ``voltrix.jit.generate`` /
    ``cpp_format`` / ``hash_to_hex`` / type maps, the flag-name constants, and
    ``voltrix.utils.calc_diff`` / ``relative_error`` on seeded inputs.  Data only.
ref_known_answers.json
    The known answers SURVEY.md section 8c recorded from the reference's own
    ``voltrix::preprocess`` (nnz / W / T / TCb-per-window range for three seeded
    ``scipy.sparse.random`` inputs) plus the numpy/scipy versions they hold for.
csr_*.npz
    Small CSR inputs (the arrays themselves, because ``sp.random`` streams differ across
    numpy versions), the block-format handle the oracle derives from them, and the
    ``torch.sparse.mm`` result -- the reference's own oracle call (tests/test_spmm.py:24-29).
"""
import json
import os
import sys

import numpy as np
import scipy
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import oracle_c, oracle_np, torch_ref  # noqa: E402


def ref_python_goldens():
    sys.path.insert(0, "/root/reference")
    import voltrix.jit as rjit  # the reference
    from voltrix.jit.compiler import hash_to_hex
    from voltrix.jit import template as rtemplate
    from voltrix.project import const as rconst
    from voltrix import utils as rutils
    from voltrix.jit_kernels import spmm as rspmm, preprocess as rpre, hmat_gem as rhmat, bmat_swizzle as rswz

    out = {}
    # tests/test_jit.py:37-47 inputs
    args = (("lhs", torch.float8_e4m3fn), ("rhs", torch.float8_e4m3fn), ("scale", torch.float),
            ("out", torch.bfloat16), ("enable_double_streams", bool), ("stream", torch.cuda.Stream))
    body = "\n"
    for n in ("lhs", "rhs", "scale", "out"):
        body += f"std::cout << reinterpret_cast<uint64_t>({n}) << std::endl;\n"
    body += "std::cout << enable_double_streams << std::endl;\n"
    body += "std::cout << reinterpret_cast<uint64_t>(stream) << std::endl;\n"
    out["generate_test_jit"] = {
        "arg_defs": [[n, rtemplate.typename_map[t]] for n, t in args],
        "body": body,
        "code": rjit.generate((), args, body),
    }
    # the four live kernels' generated wrappers (arg lists are the C-ABI contract)
    spmm_args = (("blk_offsets", torch.int32), ("hspa_packed", torch.uint32), ("hind", torch.int32),
                 ("num_nodes", int), ("num_edges", int), ("embedding_dim", int),
                 ("input", torch.float), ("output", torch.float), ("stream", torch.cuda.Stream))
    out["generate_spmm"] = {
        "arg_defs": [[n, rtemplate.typename_map[t]] for n, t in spmm_args],
        "template": rspmm.template,
        "keys": {"model": 1},
        "body": rjit.cpp_format(rspmm.template, {"model": 1}),
        "includes": list(rspmm.includes),
        "code": rjit.generate(rspmm.includes, spmm_args, rjit.cpp_format(rspmm.template, {"model": 1})),
    }
    pre_args = (("edge_list", torch.int), ("node_pointer", torch.int), ("num_nodes", int),
                ("block_partition", torch.int), ("edge_to_column", torch.int), ("edge_to_row", torch.int),
                ("pointer1", torch.int))
    out["generate_preprocess"] = {
        "arg_defs": [[n, rtemplate.typename_map[t]] for n, t in pre_args],
        "body": rpre.template, "includes": list(rpre.includes),
        "code": rjit.generate(rpre.includes, pre_args, rpre.template),
    }
    hmat_args = (("node_pointer", torch.int), ("edge_list", torch.int), ("block_partition", torch.int),
                 ("edge_to_column", torch.int), ("edge_to_row", torch.int), ("pointer1", torch.int),
                 ("num_row_windows", int), ("num_nodes", int), ("num_edges", int),
                 ("hspa", torch.float), ("hind", torch.int))
    out["generate_hmat_gen"] = {
        "arg_defs": [[n, rtemplate.typename_map[t]] for n, t in hmat_args],
        "body": rhmat.template, "includes": list(rhmat.includes),
        "code": rjit.generate(rhmat.includes, hmat_args, rhmat.template),
    }
    swz_args = (("num_row_windows", int), ("pointer1", torch.int), ("hspa", torch.float),
                ("hspa_packed", torch.uint32))
    out["generate_swizzle"] = {
        "arg_defs": [[n, rtemplate.typename_map[t]] for n, t in swz_args],
        "body": rswz.template, "includes": list(rswz.includes),
        "code": rjit.generate(rswz.includes, swz_args, rswz.template),
    }
    out["cpp_format"] = [
        {"template": "f<{a}, {b}>({a}); {{keep}} {c}", "keys": {"a": 1, "b": "x"},
         "result": rjit.cpp_format("f<{a}, {b}>({a}); {{keep}} {c}", {"a": 1, "b": "x"})},
        {"template": rspmm.template, "keys": {"model": 2}, "result": rjit.cpp_format(rspmm.template, {"model": 2})},
    ]
    out["hash_to_hex"] = {s: hash_to_hex(s) for s in ("", "voltrix", "test_20_8192_0.1", "140234234234")}
    out["typename_map"] = {str(k): v for k, v in rtemplate.typename_map.items()}
    out["genc_map"] = {rtemplate.typename_map[k]: list(v) for k, v in rtemplate.genc_map.items()}
    out["const"] = {k: getattr(rconst, k) for k in dir(rconst) if k.isupper()}
    out["public_names"] = {
        "voltrix.jit": ["build", "cpp_format", "generate", "get_nvcc_compiler", "Runtime"],
        "voltrix.jit_kernels": ["hmat_packed_swizzle_kernel", "hmat_gen_kernel", "spmm_kernel", "preprocess_kernel"],
        "voltrix.spmm": ["BLK_H", "BLK_W", "csr_preprocess", "spmm"],
    }
    # metrics on seeded inputs
    g = torch.Generator().manual_seed(7)
    x = torch.randn(257, 33, generator=g)
    y = x + 1e-3 * torch.randn(257, 33, generator=g)
    out["metrics"] = {
        "x": x.numpy().astype(np.float64).ravel().tolist()[:0],  # inputs are regenerated from the seed
        "seed": 7, "shape": [257, 33], "noise": 1e-3,
        "calc_diff": float(rutils.calc_diff(x, y)),
        "calc_diff_f64": float(rutils.calc_diff(x, y, dtype=torch.float64)),
        "relative_error": float(rutils.relative_error(y, x)),
        "x_sum": float(x.double().sum()), "y_sum": float(y.double().sum()),
    }
    return out


def csr_fixture(name, indptr, indices, num_nodes, feat, note):
    indptr = np.asarray(indptr, dtype=np.int32)
    indices = np.asarray(indices, dtype=np.int32)
    bp, e2c, e2r, p1 = oracle_c.preprocess(indptr, indices, num_nodes)
    bp2, e2c2, e2r2, p12 = oracle_np.preprocess(indptr, indices, num_nodes)
    assert (bp == bp2).all() and (e2c == e2c2).all() and (e2r == e2r2).all() and (p1 == p12).all()
    hspa, hind = oracle_c.hmat_gen(indptr, indices, bp, e2c, e2r, p1, num_nodes)
    hspa2, hind2 = oracle_np.hmat_gen(indptr, indices, bp, e2c, e2r, p1, num_nodes)
    assert (hspa == hspa2).all() and (hind == hind2).all()
    packed = oracle_c.hmat_packed_swizzle(p1, hspa)
    assert (packed == oracle_np.hmat_packed_swizzle(p1, hspa)).all()
    ref = torch_ref.spmm(indptr, indices, feat, num_nodes).numpy()
    np.savez_compressed(
        os.path.join(HERE, f"csr_{name}.npz"),
        indptr=indptr, indices=indices, num_nodes=np.int64(num_nodes), feat=feat.astype(np.float32),
        block_partition=bp, edge_to_column=e2c, edge_to_row=e2r, pointer1=p1,
        hspa_packed=packed, hind=hind, torch_sparse_mm=ref.astype(np.float32), note=np.array(note))
    print(f"csr_{name}.npz: N={num_nodes} nnz={indices.size} T={int(p1[-1])}")


def csr_fixtures():
    rng = np.random.default_rng(1234)
    # (1) toy, 40 nodes, empty middle window (rows 16..31 have no edges), N%16 = 8
    rows = {0: [5, 17, 39], 1: [5], 2: [0, 1, 2, 7, 9, 11, 20, 21, 33], 15: [2, 1, 0], 33: [39], 39: [0, 38]}
    indptr = [0]
    indices = []
    for r in range(40):
        indices += rows.get(r, [])
        indptr.append(len(indices))
    csr_fixture("toy40", indptr, indices, 40, rng.standard_normal((40, 16)), "toy, empty middle window, unsorted row 15")
    # (2) cora-like: N=2708, symmetric, no self loops, F=32 (BASELINE.json configs[0])
    np.random.seed(0)
    a = sp.random(2708, 2708, density=0.00072, format="csr")
    a = ((a + a.T) != 0).astype(np.float32).tocsr()
    a.setdiag(0)
    a.eliminate_zeros()
    a.sort_indices()
    csr_fixture("cora_like", a.indptr, a.indices, 2708, rng.standard_normal((2708, 32)), "cora-like symmetric, N%16=4")
    # (3) sp.random seed 0, N=2708, density 0.0015 -- the survey's third known-answer input
    np.random.seed(0)
    a = sp.random(2708, 2708, density=0.0015, format="csr")
    csr_fixture("sprandom_2708", a.indptr, a.indices, 2708, rng.standard_normal((2708, 48)), "sp.random(2708, d=0.0015, seed 0)")
    # (4) skewed: a few very heavy rows + many empty rows, N%16 = 13, hub column 0
    n = 1005
    deg = np.zeros(n, dtype=np.int64)
    deg[rng.choice(n, 300, replace=False)] = rng.integers(1, 12, 300)
    deg[[3, 500, 1004]] = [900, 640, 77]
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg if d > 0])
    csr_fixture("skewed_1005", indptr, indices, n, rng.standard_normal((n, 64)), "skewed degrees, empty rows, N%16=13")


def known_answers():
    # outputs of the reference's voltrix::preprocess on np.random.seed(s); sp.random(N,N,d,"csr")
    ka = {
        "source": "SURVEY.md section 8c (reference voltrix::preprocess, bmat_kernels.cuh:264-320)",
        "numpy": np.__version__, "scipy": scipy.__version__,
        "cases": [
            {"N": 2708, "density": 0.0015, "seed": 0, "nnz": 11000, "W": 170, "T": 1439, "min": 2, "max": 12},
            {"N": 8192, "density": 0.01, "seed": 20, "nnz": 671089, "W": 512, "T": 78102, "min": 137, "max": 168},
            {"N": 8192, "density": 0.1, "seed": 20, "nnz": 6710886, "W": 512, "T": 427364, "min": 824, "max": 847},
        ],
    }
    for c in ka["cases"]:
        np.random.seed(c["seed"])
        a = sp.random(c["N"], c["N"], density=c["density"], format="csr")
        assert a.nnz == c["nnz"], (a.nnz, c)
        # checksum of the generated CSR so that a test on another numpy can tell "stream differs" from "oracle wrong"
        c["indices_sum"] = int(a.indices.astype(np.int64).sum())
        c["indptr_sum"] = int(a.indptr.astype(np.int64).sum())
    return ka


if __name__ == "__main__":
    with open(os.path.join(HERE, "ref_python_goldens.json"), "w") as f:
        json.dump(ref_python_goldens(), f, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "ref_known_answers.json"), "w") as f:
        json.dump(known_answers(), f, indent=1)
    csr_fixtures()
