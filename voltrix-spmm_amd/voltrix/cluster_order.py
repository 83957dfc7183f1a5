"""Cluster order of a square graph (round 6): connected components + multilevel size-constrained label propagation.

Why.  The reference times Voltrix on REORDERED copies of all twelve evaluation graphs
(bench/bench_all.py:120-149, bench/graph_gen.py:42-45 read ``<name>.reorder.npz``); eight of them have mean degree 2-12.  On
label-shuffled copies of those the two orders this package had do not work: a breadth-first search sees the whole graph after
three levels as soon as a few per cent of the edges are random, and the spectral order needs the leading eigenvectors of an
operator whose band harmonics sit 0.1 % apart below a bulk of sparse-graph noise (com-amazon-like: 0.7681, 0.7672, 0.7670 ...;
the unfolded chain of boxes folds, profiles/r06/experiment_reorder_lowdeg_prototype.log).  What the SpMM needs from an order
is much less than a one-dimensional embedding: B's rows referenced by the windows that run together must share the XCD's 4 MiB
L2 (16 k rows of 256 B).  That is a PARTITION question -- groups of a few thousand nodes that keep most of their edges inside
-- and the order of the groups does not matter.

How.  (1) Connected components by min-label hooking + pointer jumping (TU-style unions of small graphs -- DD, Yeast, YeastH,
ppi -- are solved by this alone: every component becomes contiguous).  (2) Label propagation with a size cap, multilevel:
from singletons to clusters of <= 32 nodes, contract (cluster-cluster edge weights: on a locally tree-like graph a single edge
says nothing, but two clusters of 32 band neighbours are joined by several edges while random edges stay at weight one),
<= 512, contract, <= ``cap`` (8192).  (3) Refinement on the FINE graph: every node moves to the top-level cluster that holds
most of its neighbours (Kernighan-Lin style, capped); this is where most of the quality comes from (com-amazon-like, shuffled:
37 % of the edges inside their cluster after the multilevel pass, 58.5 % after refinement; the generating order has 70.7 %
within +- 4096).  (4) The top-level clusters are laid out along the CHAIN they form -- the Fiedler order of the cluster graph (a few
hundred nodes, background-subtracted weights, dense): what no eigenvector of the fine graph manages at this degree works on clusters of
thousands of nodes, whose mutual edge counts average the randomness out (co-purchase stand-ins, shuffled: correlation with the generating
order 0.99 / -1.00 / 0.89; step / natural-order step 1.06 -> 1.04, 1.02 -> 1.01, 1.01 -> 0.96).  Rows are then sorted by (component, chain
position of the top cluster, middle cluster, first cluster).

Everything is torch tensor ops on the CSR's device (sorts and segmented sums; no host loop over nodes), deterministic for a
given seed.  No reference counterpart: the reference reads externally reordered files."""
from __future__ import annotations

import torch

# cluster sizes (nodes) of the three levels.  The last one is what should fit the XCD's L2 (4 MiB = 16 k rows of 256 B) with room;
# measured on the four label-shuffled band stand-ins, F = 128 fp16, step / natural-order step (profiles/r06/experiment_reorder_cluster_caps.log):
#   4096: 1.108 1.039 1.003 1.035    8192: 1.064 1.015 1.007 1.030    16384: 1.052 1.005 1.001 1.032   (shuffled: 1.31 1.20 1.14 1.26)
CLUSTER_CAPS = (32, 512, 8192)
LEVEL_ITERATIONS = (12, 20, 20)
REFINE_ITERATIONS = 20
MAX_EDGES = 1 << 28                # symmetrised edges above which the candidate is not tried (sort keys: 2 x 8 B per edge)


def symmetric_edges(indptr: torch.Tensor, indices: torch.Tensor, n: int):
    """(u, v) int64 of A + A^T without self loops; an edge present in both directions appears twice per direction (weight 2)."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg)
    cols = indices.long()
    keep = (cols < n) & (cols != rows)
    rows, cols = rows[keep], cols[keep]
    return torch.cat([rows, cols]), torch.cat([cols, rows])


def connected_components(u: torch.Tensor, v: torch.Tensor, n: int, max_rounds: int = 256) -> torch.Tensor:
    """int64 [n]: the smallest node id of every node's connected component.  Hooking (every node takes the smallest label among
    itself and its neighbours) + pointer jumping (label <- label[label]) until nothing changes; rounds ~ log of the diameter."""
    dev = u.device
    label = torch.arange(n, device=dev, dtype=torch.int64)
    for _ in range(max_rounds):
        new = label.clone()
        if u.numel():
            new.scatter_reduce_(0, u, label[v], reduce="amin")
        for _ in range(3):
            new = new[new]
        if torch.equal(new, label):
            break
        label = new
    return label


def _propagate(u, v, w, sizes, label, cap: float, iterations: int, gen, frac: float = 0.5):
    """Size-constrained weighted label propagation on a symmetric edge list: per round every node scores the labels of its
    neighbours (sum of edge weights; a foreign label loses weight as its cluster fills up, own label wins ties) and a random
    ``frac`` of the nodes that want to move do so while the target's size stays <= cap.  Returns COMPACT labels [0, k)."""
    dev = u.device
    n = sizes.numel()
    for _ in range(iterations):
        csize = torch.zeros(n, device=dev, dtype=sizes.dtype).index_add_(0, label, sizes)
        key = u * n + label[v]
        key, order = torch.sort(key)
        uk, inverse = torch.unique_consecutive(key, return_inverse=True)
        votes = torch.zeros(uk.numel(), device=dev, dtype=w.dtype).index_add_(0, inverse, w[order])
        del key, order, inverse
        node, cand = torch.div(uk, n, rounding_mode="floor"), uk % n
        own = cand == label[node]
        room = (1.0 - (csize[cand] + sizes[node]) / cap).clamp(min=0.0)
        score = torch.where(own, votes + 1e-3, votes * (2.0 * room).clamp(max=1.0))
        score = score + 1e-4 * torch.rand(score.numel(), device=dev, generator=gen)       # ties: random, reproducibly
        best = torch.full((n,), -1.0, device=dev, dtype=score.dtype).scatter_reduce_(0, node, score, reduce="amax")
        winner = score == best[node]
        choice = torch.full((n,), -1, device=dev, dtype=torch.int64)
        choice[node[winner]] = cand[winner]
        move = (choice >= 0) & (choice != label) & (best > 0) & (torch.rand(n, device=dev, generator=gen) < frac)
        movers = torch.nonzero(move).flatten()
        if movers.numel() == 0:
            break
        target = choice[movers]
        target, by_target = torch.sort(target, stable=True)
        movers = movers[by_target]
        msize = sizes[movers]
        running = torch.cumsum(msize, 0)
        start = torch.ones(target.numel(), dtype=torch.bool, device=dev)
        start[1:] = target[1:] != target[:-1]
        seg = torch.cumsum(start.long(), 0) - 1
        before = (running - msize)[start][seg]                 # running size at the start of the target's segment
        fits = csize[target] + (running - before) <= cap
        label = label.clone()
        label[movers[fits]] = target[fits]
        if int(fits.sum()) * 10000 < n:
            break
    return torch.unique(label, return_inverse=True)[1]


def _contract(u, v, w, label, sizes):
    k = int(label.max()) + 1 if label.numel() else 0
    cu, cv = label[u], label[v]
    keep = cu != cv
    key, inverse = torch.unique(cu[keep] * k + cv[keep], return_inverse=True)
    weight = torch.zeros(key.numel(), device=u.device, dtype=w.dtype).index_add_(0, inverse, w[keep])
    csize = torch.zeros(k, device=u.device, dtype=sizes.dtype).index_add_(0, label, sizes)
    return torch.div(key, k, rounding_mode="floor"), key % k, weight, csize


def default_caps():
    return CLUSTER_CAPS


CHAIN_MAX_CLUSTERS = 4096          # the cluster graph is handled densely (k x k doubles): above this many clusters their ids stand


def chain_order(top: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """int64 [n]: for every node the RANK of its cluster in the Fiedler order of the cluster graph (edge counts between clusters minus
    what a uniform background of the same degrees would put there, negative entries dropped; symmetric normalised Laplacian, second
    eigenvector, float64, dense).  Clusters that hang on nothing (other components, one-cluster graphs) keep their relative order
    at the end.  ``top`` itself when there are too many clusters for a dense matrix (unions of tens of thousands of small graphs:
    every cluster is a whole component there and has no neighbour to sit next to)."""
    k = int(top.max()) + 1 if top.numel() else 0
    if k < 3 or k > CHAIN_MAX_CLUSTERS or u.numel() == 0:
        return top
    dev = top.device
    cu, cv = top[u], top[v]
    cross = cu != cv
    w = torch.bincount(cu[cross] * k + cv[cross], minlength=k * k).view(k, k).double()
    w = 0.5 * (w + w.T)
    d = w.sum(1)
    w = (w - torch.outer(d, d) / d.sum().clamp(min=1.0)).clamp(min=0.0)
    d = w.sum(1)
    linked = d > 0
    if int(linked.sum()) < 3:
        return top
    idx = torch.nonzero(linked).flatten()
    ws = w[idx][:, idx]
    ds = ws.sum(1)
    inv = ds.rsqrt()
    lap = torch.eye(idx.numel(), dtype=torch.float64, device=dev) - inv[:, None] * ws * inv[None, :]
    evals, evecs = torch.linalg.eigh(lap)
    fiedler = evecs[:, 1] * inv                       # random-walk coordinates
    rank = torch.full((k,), 0, dtype=torch.int64, device=dev)
    rank[idx[torch.argsort(fiedler)]] = torch.arange(idx.numel(), device=dev)
    rest = torch.nonzero(~linked).flatten()
    rank[rest] = idx.numel() + torch.arange(rest.numel(), device=dev)
    return rank[top]


def cluster_permutation(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, caps=None, seed: int = 0,
                        refine_iterations: int = REFINE_ITERATIONS, info: dict = None, chain: bool = True) -> torch.Tensor:
    """Row order (int64 [N], position k holds node ``perm[k]``) that makes every connected component contiguous and, inside a
    component, every cluster of the multilevel label propagation (module docstring).  Square graphs; meant for the symmetric
    relabelling ``P A P^T`` (``csr_preprocess_reordered(..., relabel=True)``): what it restores is where B's rows sit in memory."""
    n = num_nodes
    dev = indptr.device
    caps = default_caps() if caps is None else tuple(caps)
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    u0, v0 = symmetric_edges(indptr, indices, n)
    component = connected_components(u0, v0, n)
    w0 = torch.ones(u0.numel(), device=dev, dtype=torch.float32)
    ones = torch.ones(n, device=dev, dtype=torch.float32)
    u, v, w, sizes = u0, v0, w0, ones
    fine = torch.arange(n, device=dev, dtype=torch.int64)       # fine node -> its cluster at the current level
    levels = []
    for level, cap in enumerate(caps):
        start = torch.arange(sizes.numel(), device=dev, dtype=torch.int64)
        label = _propagate(u, v, w, sizes, start, float(cap), LEVEL_ITERATIONS[min(level, len(LEVEL_ITERATIONS) - 1)], gen)
        fine = label[fine]
        levels.append(fine)
        u, v, w, sizes = _contract(u, v, w, label, sizes)
        if u.numel() == 0:          # nothing left to merge (unions of small graphs: every component is one cluster)
            break
    top = levels[-1]
    inside_before = float((top[u0] == top[v0]).float().mean()) if u0.numel() else 1.0
    if refine_iterations > 0 and u0.numel():
        top = _propagate(u0, v0, w0, ones, top, 1.05 * float(caps[-1]), refine_iterations, gen, frac=0.7)
    # the ORDER of the top-level clusters: along the chain they form (band graphs: a cluster's cut edges go to its two neighbours on
    # the band; side by side they stay within reach of one L2 / one XCD range) -- the Fiedler order of the cluster graph
    # (only the TOP level: a chain needs clusters that keep most of their edges inside, i.e. at least about twice the band wide; the
    # middle level's 512-node clusters are narrower than the co-purchase stand-ins' bands -- chained there, the local share of
    # com-amazon-like fell from 0.67 to 0.38)
    top_key = chain_order(top, u0, v0) if chain else top
    # sort by (component, top, middle, first): stable sorts from the least significant key
    perm = torch.arange(n, device=dev, dtype=torch.int64)
    for key in levels[:-1] + [top_key, component]:
        perm = perm[torch.argsort(key[perm], stable=True)]
    if info is not None:
        csize = torch.bincount(top)
        info.update(components=int(torch.unique(component).numel()), clusters=int(csize.numel()),
                    largest_cluster=int(csize.max()), levels=len(levels),
                    edges_inside_cluster=float((top[u0] == top[v0]).float().mean()) if u0.numel() else 1.0,
                    edges_inside_cluster_before_refinement=inside_before)
    return perm
