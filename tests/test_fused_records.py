"""CPU: the stage-record stream of the one-launch two-level kernel (hybrid.build_fused_records_torch; the HIP kernel that
consumes it is checked on the GPU, tests/test_gpu_fused.py) against the plain-loop definition
(oracle_np.fused_records), bit for bit, and the records' consumer-side interpretation against the residual CSR."""
import numpy as np
import pytest
import torch

from oracle import oracle_c, oracle_np
from voltrix import hybrid


def _random_csr(n, max_deg, seed, ncols=None):
    rng = np.random.default_rng(seed)
    ncols = ncols or n
    rows = [np.unique(rng.integers(0, ncols, rng.integers(0, max_deg + 1))) for _ in range(n)]
    indptr = np.zeros(n + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return indptr, np.concatenate(rows + [np.zeros(0, np.int64)]).astype(np.int32)


def _check(indptr, indices, n, tau, ncols=None):
    ncols = ncols or n
    ri, rx, plan = hybrid.build_panel_plan_torch(torch.from_numpy(indptr), torch.from_numpy(indices), n, ncols, 8, 4, tau)
    p1, packed, hind = oracle_c.csr_preprocess(ri.numpy(), rx.numpy(), n)
    wave_ptr, records = oracle_np.fused_records(p1, packed, hind, n)
    fr = hybrid.build_fused_records_torch(torch.from_numpy(p1), torch.from_numpy(packed.view(np.int32)).view(torch.uint32),
                                          torch.from_numpy(hind), n)
    assert np.array_equal(fr.wave_ptr.numpy(), wave_ptr)
    assert np.array_equal(fr.records.view(torch.int32).numpy().view(np.uint32), records)
    assert fr.num_records == len(records) - 1 and not records[-1].any()
    # every wave's records sweep the columns in order; row blocks 0 .. 7 only
    for gw in range(len(wave_ptr) - 1):
        first = records[wave_ptr[gw]:wave_ptr[gw + 1], 0].astype(np.int64)
        assert (np.diff(first) >= 0).all()
    assert (records[:, 48] < 8).all() and len(wave_ptr) - 1 == 4 * ((n + 511) // 512)
    edges = oracle_np.fused_records_to_edges(wave_ptr, records, n)
    resid = sorted((r, int(c)) for r in range(n) for c in rx.numpy()[ri[r]:ri[r + 1]])
    assert edges == resid
    return fr, len(resid)


@pytest.mark.parametrize("n,max_deg,tau,ncols", [(4000, 40, 3, None), (1100, 120, 2, 20000), (530, 7, 2, None),
                                                 (515, 300, 10 ** 6, None)])
def test_records_match_the_definition(n, max_deg, tau, ncols):
    indptr, indices = _random_csr(n, max_deg, seed=n + tau, ncols=ncols)
    fr, num_resid = _check(indptr, indices, n, tau, ncols)
    assert num_resid > 0 and fr.num_records > 0


def test_records_edge_cases():
    # an empty residual (every column shared), empty windows in the middle (the reference's one-zero-block quirk), a tail
    # window, an empty matrix
    indptr, indices = _random_csr(600, 200, seed=1)
    fr, num_resid = _check(indptr, indices, 600, 1)
    assert num_resid == 0 and fr.num_records == 0
    a, b = _random_csr(700, 9, 4)
    lo, hi = a[100], a[400]
    b = np.r_[b[:lo], b[hi:]]
    a = a.copy()
    a[100:401] = lo
    a[401:] -= hi - lo
    _check(a, b, 700, 2)
    _check(np.zeros(41, np.int32), np.zeros(0, np.int32), 40, 2)
