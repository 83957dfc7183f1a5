"""CPU: the C-ABI library loads and exports every symbol include/voltrix_capi.h declares; host-only entry points
(tile enumeration, the host `preprocess` launch, argument validation) behave.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_csr_fixture

from voltrix import capi

HEADER = os.path.join(REPO, "include", "voltrix_capi.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(voltrix_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared_functions() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    for name in _declared_functions():
        assert hasattr(lib, name), f"libvoltrix_hip.so does not export {name}"
    assert lib.voltrix_abi_version() == 2
    from voltrix import hybrid
    assert capi.fused_panel_geometry() == (hybrid.FUSED_WAVES, hybrid.FUSED_ROW_BLOCKS) == (4, 8)


def test_tile_space_enumeration_and_defaults():
    f16 = capi.tiles(True)
    f32 = capi.tiles(False)
    assert len(f16) == len(set(f16)) >= 30 and (128, 4, 1) in f16 and (256, 4, 4) not in f16
    assert (128, 2, 1) in f32 and all(fs <= 128 for fs, _, _ in f32)
    for fs, depth, waves in f16 + f32:
        assert fs in (32, 64, 128, 256) and 2 <= depth <= 4 and waves in (1, 2, 4, 8)
    for f16_flag, tiles in ((True, f16), (False, f32)):
        for dim in (8, 32, 33, 64, 100, 128, 512, 1024):
            assert capi.default_tile(dim, f16_flag) in tiles
    assert capi.default_tile(32, True) == (32, 4, 4) and capi.default_tile(64, True) == (64, 3, 4)
    assert capi.default_tile(128, True) == capi.default_tile(512, True) == (128, 3, 4)   # the measured best (profiles/HISTORY.md 5)
    assert capi.default_tile(128, False) == (64, 3, 1)


def test_host_preprocess_entry_point_matches_golden(csr_fixture):
    g = csr_fixture
    n = int(g["num_nodes"])
    indptr = np.ascontiguousarray(g["indptr"], np.int32)
    indices = np.ascontiguousarray(g["indices"], np.int32)
    w = (n + 15) // 16
    bp, e2c = np.zeros(w, np.int32), np.zeros(indices.size, np.int32)
    e2r, p1 = np.zeros(indices.size, np.int32), np.zeros(w + 1, np.int32)
    rc = ctypes.c_int(-1)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    capi.lib().voltrix_launch_preprocess(ptr(indices), ptr(indptr), ctypes.c_int(n), ptr(bp), ptr(e2c), ptr(e2r),
                                         ptr(p1), ctypes.byref(rc))
    assert rc.value == 0
    assert (bp == g["block_partition"]).all() and (p1 == g["pointer1"]).all()
    assert (e2c == g["edge_to_column"]).all() and (e2r == g["edge_to_row"]).all()


def test_host_preprocess_is_deterministic_and_equals_the_oracle(monkeypatch):
    import scipy.sparse as sp

    np.random.seed(3)
    a = sp.random(4000, 4000, density=0.02, format="csr")
    indptr, indices = a.indptr.astype(np.int32), a.indices.astype(np.int32)
    ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    outs = []
    for _ in range(2):     # threads over windows (hardware_concurrency): two runs, same bytes
        bp, e2c = np.zeros(250, np.int32), np.zeros(indices.size, np.int32)
        e2r, p1 = np.zeros(indices.size, np.int32), np.zeros(251, np.int32)
        rc = ctypes.c_int(-1)
        capi.lib().voltrix_launch_preprocess(ptr(indices), ptr(indptr), ctypes.c_int(4000), ptr(bp), ptr(e2c), ptr(e2r),
                                             ptr(p1), ctypes.byref(rc))
        assert rc.value == 0
        outs.append((bp, e2c, e2r, p1))
    for x, y in zip(*outs):
        assert (x == y).all()
    from oracle import oracle_c

    for x, y in zip(outs[0], oracle_c.preprocess(indptr, indices, 4000)):
        assert (x == y).all()


def test_return_codes_on_bad_arguments(monkeypatch):
    lib = capi.lib()
    rc = ctypes.c_int(-1)
    z = ctypes.c_void_p(0)
    lib.voltrix_launch_preprocess(z, z, ctypes.c_int(-1), z, z, z, z, ctypes.byref(rc))
    assert rc.value == 1
    # embedding_dim not a multiple of 8 halves / unknown tile: rejected on the host, before any HIP call
    lib.voltrix_launch_spmm_f16_tile(z, z, z, ctypes.c_int(16), ctypes.c_int(0), ctypes.c_int(12), z, z,
                                     ctypes.c_int(128), ctypes.c_int(4), ctypes.c_int(1), z, z, z, ctypes.byref(rc))
    assert rc.value == 1
    lib.voltrix_launch_spmm_f16_tile(z, z, z, ctypes.c_int(16), ctypes.c_int(0), ctypes.c_int(16), z, z,
                                     ctypes.c_int(48), ctypes.c_int(4), ctypes.c_int(1), z, z, z, ctypes.byref(rc))
    assert rc.value == 3
    lib.voltrix_launch_cast_f32_f16(z, z, ctypes.c_int64(12), z, ctypes.byref(rc))
    assert rc.value == 1
    lib.voltrix_launch_cast_f32_f16_scaled(z, z, ctypes.c_int64(16), z, z, ctypes.byref(rc))  # no scale buffer
    assert rc.value == 1
    lib.voltrix_launch_window_order(z, ctypes.c_int(64), ctypes.c_int(8192), z, z, ctypes.byref(rc))
    assert rc.value == 1  # chunk above the 4096-window LDS sort capacity
    # sort path (column universe too large for the LDS bitmap): one uint32 key per edge; bitmap path: scan scratch only
    assert capi.csr_preprocess_workspace_bytes(2449029, 2449029, 123718280) >= 4 * 123718280
    assert 0 < capi.csr_preprocess_workspace_bytes(232965, 232965, 114615892) < 1 << 20
    assert capi.csr_preprocess_workspace_bytes(232965, 232965, 114615892, "sort") >= 4 * 114615892
    with pytest.raises(capi.VoltrixError, match="return code 3"):
        capi.check(3, "x")
