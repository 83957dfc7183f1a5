"""CPU: detection of edge values that factor as r_i * c_j (voltrix/weighted.py::separable_scales, round 6): exact on the common
normalisations, refused for everything the binary operator between two row scalings would compute differently."""
import torch

import synth_graphs
from voltrix import weighted


def _graph():
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.004)
    n = indptr.numel() - 1
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n), deg)
    return indptr, indices, n, deg.double(), rows


def test_common_normalisations_and_general_factors_are_detected():
    indptr, indices, n, deg, rows = _graph()
    indeg = torch.bincount(indices.long(), minlength=n).double().clamp(min=1)
    torch.manual_seed(0)
    cases = {"symmetric": 1 / torch.sqrt(deg.clamp(min=1)[rows] * indeg[indices.long()]), "row mean": 1 / deg.clamp(min=1)[rows],
             "column": 1 / indeg[indices.long()], "ones": torch.ones(indices.numel(), dtype=torch.float64),
             "general": (torch.rand(n, dtype=torch.float64) * 3 + 0.05)[rows] * (torch.rand(n, dtype=torch.float64) + 0.1)[indices.long()]}
    for name, v in cases.items():
        got = weighted.separable_scales(indptr, indices, v.float(), n, n)
        assert got is not None, name
        r, c = got
        assert r.dtype == c.dtype == torch.float32 and r.numel() == c.numel() == n
        err = ((r[rows].double() * c[indices.long()].double()) / v - 1).abs().max()
        assert float(err) <= weighted.SEPARABLE_TOLERANCE, (name, float(err))
        assert 1e-3 < float(r[deg > 0].median() / c.median()) < 1e3      # balanced factors: the scaled B stays inside fp16's range


def test_what_does_not_factor_is_refused():
    indptr, indices, n, deg, rows = _graph()
    torch.manual_seed(1)
    assert weighted.separable_scales(indptr, indices, torch.rand(indices.numel()) + 0.5, n, n) is None        # random positive values
    assert weighted.separable_scales(indptr, indices, torch.randn(indices.numel()), n, n) is None             # signs
    v = torch.ones(indices.numel())
    v[7] = 0.0
    assert weighted.separable_scales(indptr, indices, v, n, n) is None                                         # a zero
    v[7] = float("inf")
    assert weighted.separable_scales(indptr, indices, v, n, n) is None
    # a duplicate (row, col) entry ADDS in the weighted product and counts once in the binary format
    first = int(deg[0])
    if first >= 1:
        dup_indices = torch.cat([indices[:1], indices])
        dup_indptr = indptr.clone()
        dup_indptr[1:] += 1
        assert weighted.separable_scales(dup_indptr, dup_indices, torch.ones(dup_indices.numel()), n, n) is None
    # one edge off by 1 %: not separable at 2^-13
    w = (1 / deg.clamp(min=1)[rows]).float()
    w[11] *= 1.01
    assert weighted.separable_scales(indptr, indices, w, n, n) is None


def test_edge_slots_are_where_value_plane_puts_every_edge():
    """Round 6 (update_values): the edge -> plane map, chunk by chunk, against the plane builder itself -- reading the plane at the
    slots gives the values back (duplicate-free rows), slots are distinct, and everything else in the plane is zero."""
    import numpy as np

    from oracle import oracle_c

    indptr, indices, n, _, _ = _graph()
    p1 = torch.from_numpy(oracle_c.csr_preprocess(indptr.numpy().astype(np.int32), indices.numpy().astype(np.int32), n)[0])
    values = torch.arange(1, indices.numel() + 1, dtype=torch.float32)
    plane = weighted.value_plane(indptr, indices, values, p1, n, n)
    for chunk in (weighted.CHUNK_EDGES, 997):
        old = weighted.CHUNK_EDGES
        weighted.CHUNK_EDGES = chunk
        try:
            slot = weighted.edge_slots(indptr, indices, p1, n, n)
        finally:
            weighted.CHUNK_EDGES = old
        assert slot.dtype == torch.int64 and slot.numel() == indices.numel()
        assert torch.unique(slot).numel() == slot.numel()
        assert torch.equal(plane.view(-1)[slot], values)
        assert int((plane != 0).sum()) == indices.numel()
    # transpose_order: entry k of A^T's CSR is entry order[k] of A's
    order = weighted.transpose_order(indptr, indices, n)
    t_indptr, t_indices, t_values = weighted.transpose_weighted(indptr, indices, values, n, n)
    assert torch.equal(t_values, values[order])
