"""Cluster order (voltrix/cluster_order.py, round 6): connected components + multilevel size-constrained label propagation -- the
reorder for label-shuffled graphs of mean degree 2-12 (the reference reads externally reordered files,
bench/graph_gen.py:42-45, bench_all.py:120-149).  CPU tests of the torch restatement's properties; the GPU tests assert the
relabelled operator against torch.sparse.mm on P A P^T for one graph per family."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.sparse.csgraph import connected_components as scipy_components

import synth_graphs
from oracle import torch_ref
from voltrix import cluster_order, reorder


def _shuffled(name, scale, seed=5):
    indptr, indices, _ = synth_graphs.generate(name, scale=scale)
    n = indptr.numel() - 1
    s_indptr, s_indices, label = synth_graphs.shuffle_labels(indptr, indices, seed)
    return n, indptr, indices, s_indptr, s_indices, label


def test_connected_components_match_scipy():
    n, _, _, s_indptr, s_indices, _ = _shuffled("yeast_like", 0.01)
    u, v = cluster_order.symmetric_edges(s_indptr, s_indices, n)
    got = cluster_order.connected_components(u, v, n)
    a = sp.csr_matrix((np.ones(s_indices.numel()), s_indices.numpy(), s_indptr.numpy()), shape=(n, n))
    count, want = scipy_components(a, directed=False)
    assert int(torch.unique(got).numel()) == count
    # same partition: the smallest node id of a component names it
    first = np.full(count, n, dtype=np.int64)
    np.minimum.at(first, want, np.arange(n))
    assert np.array_equal(got.numpy(), first[want])


def test_union_of_small_graphs_every_component_becomes_contiguous():
    n, indptr, indices, s_indptr, s_indices, _ = _shuffled("yeast_like", 0.02)
    info = {}
    perm = cluster_order.cluster_permutation(s_indptr, s_indices, n, info=info)
    assert torch.equal(torch.sort(perm).values, torch.arange(n))
    u, v = cluster_order.symmetric_edges(s_indptr, s_indices, n)
    comp = cluster_order.connected_components(u, v, n)[perm]
    runs = int((comp[1:] != comp[:-1]).sum()) + 1
    assert runs == info["components"] == int(torch.unique(comp).numel())
    # every edge stays inside a few dozen positions: the generating order's locality is back
    new = torch.empty(n, dtype=torch.int64)
    new[perm] = torch.arange(n)
    assert reorder.local_fraction(s_indptr, s_indices, n, new) == 1.0
    assert reorder.local_fraction(s_indptr, s_indices, n) < 0.6 < reorder.local_fraction(indptr, indices, n)


def test_band_graph_with_random_edges_most_edges_end_up_inside_their_cluster():
    """com-amazon-like (mean degree 5.5, 70 % of the edges in a band of +- 1024, 30 % uniformly random; locally tree-like), labels
    shuffled: a breadth-first search and a spectral order both fail here; the clusters keep half of the edges."""
    n, _, _, s_indptr, s_indices, label = _shuffled("com_amazon_like", 0.12)
    info = {}
    perm = cluster_order.cluster_permutation(s_indptr, s_indices, n, caps=(32, 512, 4096), info=info)
    assert torch.equal(torch.sort(perm).values, torch.arange(n))
    assert info["largest_cluster"] <= 1.05 * 4096 + 1 and info["clusters"] >= n // 4096
    assert info["edges_inside_cluster"] > 0.5 > info["edges_inside_cluster_before_refinement"] > 0.2
    # a random partition into clusters of this size keeps cap / n of the edges
    assert info["edges_inside_cluster"] > 4 * 4096 / n
    again = cluster_order.cluster_permutation(s_indptr, s_indices, n, caps=(32, 512, 4096))
    assert torch.equal(again, perm)                                    # deterministic for a seed
    # the clusters are laid out along the chain they form (Fiedler order of the cluster graph): the generating order comes back
    # up to its direction, which no single eigenvector of the fine graph manages at this degree
    true_pos = torch.empty(n, dtype=torch.int64)
    true_pos[label] = torch.arange(n)
    corr = float(torch.corrcoef(torch.stack([true_pos[perm].double(), torch.arange(n).double()]))[0, 1])
    assert abs(corr) > 0.6, corr            # ten clusters here: one cluster of two stretches already costs 0.2
    unchained = cluster_order.cluster_permutation(s_indptr, s_indices, n, caps=(32, 512, 4096), chain=False)
    corr0 = float(torch.corrcoef(torch.stack([true_pos[unchained].double(), torch.arange(n).double()]))[0, 1])
    assert abs(corr0) < abs(corr)


@pytest.mark.gpu
@pytest.mark.parametrize("graph,scale,method", [("com_amazon_like", 0.5, "auto"), ("yeast_like", 0.1, "auto"),
                                                ("fraud_yelp_rsr_like", 1.0, "clusters"), ("web_berkstan_like", 0.3, "clusters"),
                                                ("protein_like", 0.25, "auto")])
def test_relabelled_operator_on_shuffled_family_graphs_equals_the_oracle(cuda_device, graph, scale, method, monkeypatch):
    """One graph per family of the reference's evaluation set, labels shuffled, reordered by the library (relabel=True): the
    operator on the handle of P A P^T against torch.sparse.mm on the relabelled CSR (integer features: exact) and, unpermuted,
    against the product of the caller's graph."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    import voltrix
    from voltrix.reorder import relabel_csr

    n, _, _, s_indptr, s_indices, _ = _shuffled(graph, scale)
    info = {}
    h = voltrix.csr_preprocess_reordered(s_indptr, s_indices, n, method=method, relabel=True, info=info)
    assert h.row_map is None
    torch.manual_seed(2)
    feat = torch.randint(-3, 4, (n, 64)).half()
    fin = voltrix.permute_features(h, feat.cuda())
    out_new = voltrix.spmm_reordered(h, fin, hash_tag=f"family/{graph}")
    out_old = voltrix.spmm_reordered(h, fin, unpermute=True)
    ref_old = torch_ref.spmm(s_indptr.numpy(), s_indices.numpy(), feat.float(), n)
    assert torch.equal(out_old.cpu(), ref_old)
    perm = h.perm.cpu()
    if h.relabelled:
        r_indptr, r_indices = relabel_csr(s_indptr, s_indices, n, perm)
        ref_new = torch_ref.spmm(r_indptr.numpy(), r_indices.numpy(), feat[perm].float(), n)
        assert torch.equal(out_new.cpu(), ref_new)
    if method == "auto":
        assert "clusters" in info["report"], sorted(info["report"])
        local_before = info["report"]["identity"]["local_fraction"]
        new = torch.empty(n, dtype=torch.int64)
        new[perm] = torch.arange(n)
        assert reorder.local_fraction(s_indptr, s_indices, n, new) >= local_before     # never less local than it came


def test_cluster_order_edge_cases():
    """No edges, isolated nodes, self loops, the empty graph: always a permutation; nodes without edges keep their relative order."""
    ip, ix = torch.zeros(11, dtype=torch.int32), torch.zeros(0, dtype=torch.int32)
    assert cluster_order.cluster_permutation(ip, ix, 10).tolist() == list(range(10))
    ip, ix = torch.tensor([0, 2, 3, 3, 4, 4], dtype=torch.int32), torch.tensor([1, 0, 0, 3], dtype=torch.int32)   # 3 -> 3: a self loop
    info = {}
    perm = cluster_order.cluster_permutation(ip, ix, 5, info=info)
    assert sorted(perm.tolist()) == list(range(5)) and info["components"] == 4
    assert cluster_order.cluster_permutation(torch.zeros(1, dtype=torch.int32), ix[:0], 0).numel() == 0
