"""Seeded synthetic stand-ins for the BASELINE.json workloads (no datasets are reachable offline).

Bench / test infrastructure, not part of the product package.  Generators follow SURVEY.md section 8(d):
every graph is a CSR with sorted, duplicate-free rows, int32 ``indptr`` / ``indices``; they run on any torch
device (GPU for the full sizes, CPU for the scaled-down test cases).

Degree laws: ``lognormal`` (sigma given; reddit / products / papers stand-ins) and ``zipf`` (P(d) ~ d^-alpha on
[d_min, max_deg], d_min solved for the target mean; the power-law stress config).
Every row ends up with EXACTLY its drawn degree: the duplicate edges that the first draw loses (3.4 % on the reddit
stand-in, mostly in the saturated band of the hub rows) are topped up with uniformly drawn new columns, so the edge
count of a config is the sum of its degrees (reddit-like: the quoted 114.6 M edges; round 1 ran 110.7 M).

``rows=(r0, r1)`` generates only that row range (same degrees as the full graph, per-range edge stream): the
multi-GPU bench lets every rank build its own shard instead of the whole graph.
"""
from __future__ import annotations

import math

import torch

CONFIGS = {
    # name: (num_nodes, target_nnz, degree law, max_degree, band fraction, band half-width, feat, seed)
    "cora_like": dict(num_nodes=2708, mean_deg=10556 / 2708, sigma=0.9, max_deg=168, band_frac=0.0, band=0, feat=32, seed=0),
    "reddit_like": dict(num_nodes=232965, mean_deg=114615892 / 232965, sigma=1.2, max_deg=21657, band_frac=0.5,
                        band=4096, feat=128, seed=1),
    "reddit_uniform": dict(num_nodes=232965, mean_deg=114615892 / 232965, sigma=1.2, max_deg=21657, band_frac=0.0,
                           band=0, feat=128, seed=1),
    "products_like": dict(num_nodes=2449029, mean_deg=123718280 / 2449029, sigma=1.4, max_deg=17481, band_frac=0.5,
                          band=8192, feat=512, seed=2),
    "papers_like": dict(num_nodes=111059956, mean_deg=1615685872 / 111059956, sigma=1.0, max_deg=20000, band_frac=0.5,
                        band=32768, feat=128, seed=4),
    # the same graphs with their node LABELS randomly permuted (P A P^T, seed stated): what a dataset looks like before anybody
    # reordered it -- the locality-reorder evidence of profiles/HISTORY.md section 3.4 (reference: bench/graph_gen.py:42-45 reads
    # externally reordered <name>.reorder.npz files, bench_all.py:120-129 times both)
    "reddit_shuffled": dict(base="reddit_like", shuffle_seed=101),
    "products_shuffled": dict(base="products_like", shuffle_seed=102),
    # reddit-size stochastic block model (round 4): same N / edge count / degree law as reddit_like, but the local half of
    # the mixture is COMMUNITY structure instead of a band: 40 communities of unequal size (log-normal, sigma 0.8; nodes of a
    # community are contiguous, i.e. the graph as a community-aware reorder leaves it), 75 % of a row's edges uniform inside its
    # own community (the Reddit dataset's edge homophily is ~ 0.76 over 41 classes), the rest uniform over all nodes.
    # reddit_sbm_shuffled = the same graph before anybody reordered it.
    "reddit_sbm": dict(num_nodes=232965, mean_deg=114615892 / 232965, sigma=1.2, max_deg=21657, band_frac=0.75, band=0,
                       communities=40, community_sigma=0.8, feat=128, seed=5),
    "reddit_sbm_shuffled": dict(base="reddit_sbm", shuffle_seed=105),
    # density 1e-4 of 4 M x 4 M = 1.6e9 edges; Zipf alpha = 2 degrees up to 4e5, uniform columns (load-balance stress)
    "powerlaw_4m": dict(num_nodes=4000000, mean_deg=400.0, law="zipf", alpha=2.0, sigma=0.0, max_deg=400000,
                        band_frac=0.0, band=0, feat=256, seed=3),
    # ---- the reference's OWN evaluation set (bench/plot.py:8 x bench_all.py:21; round 5).  datasets.zip is unreachable here, so
    # these are stand-ins by PUBLIC node / edge count and a degree / locality law chosen per graph family (stated per line; the
    # numbers are the published sizes of the TC-GNN / SNAP / TU / OGB versions of the graphs, directed edge counts).  reddit is
    # reddit_like above.  What is approximate: every column law below (band width, community sizes) is a guess at structure we
    # cannot see; the node count, the edge count and the kind of degree law are not.
    # co-purchase graphs (SNAP amazon0505 / amazon0601 / com-amazon): out-degree capped near 10, in-degree heavy-tailed; ids in
    # crawl order, so neighbours are near in id: 60-70 % of a row's edges in a band around it, the rest uniform
    "amazon0505_like": dict(num_nodes=410236, mean_deg=4878874 / 410236, sigma=0.6, max_deg=2760, band_frac=0.6, band=2048,
                            feat=128, seed=11),
    "amazon0601_like": dict(num_nodes=403394, mean_deg=3387388 / 403394, sigma=0.6, max_deg=2751, band_frac=0.6, band=2048,
                            feat=128, seed=12),
    "com_amazon_like": dict(num_nodes=334863, mean_deg=1851744 / 334863, sigma=0.7, max_deg=549, band_frac=0.7, band=1024,
                            feat=128, seed=13),
    # TU graph-classification sets (DD, Yeast, YeastH): block-diagonal unions of many small graphs, every edge inside its own
    # graph (`graphs` = their number, sizes log-normal, `graph_min` nodes at least); molecule / protein-contact degrees
    "dd_like": dict(num_nodes=334925, mean_deg=1686092 / 334925, sigma=0.3, max_deg=19, band_frac=1.0, band=0, graphs=1178,
                    community_sigma=0.6, graph_min=30, feat=128, seed=14),
    "yeast_like": dict(num_nodes=1710902, mean_deg=3636546 / 1710902, sigma=0.35, max_deg=6, band_frac=1.0, band=0,
                       graphs=79601, community_sigma=0.35, graph_min=8, feat=128, seed=15),
    "yeasth_like": dict(num_nodes=3138114, mean_deg=6487230 / 3138114, sigma=0.35, max_deg=6, band_frac=1.0, band=0,
                        graphs=79601, community_sigma=0.35, graph_min=12, feat=128, seed=16),
    # web graph (SNAP web-BerkStan): power-law degrees, pages of one host are neighbours in id (URL order): 85 % of the edges
    # within +- 512 ids
    "web_berkstan_like": dict(num_nodes=685230, mean_deg=7600595 / 685230, law="zipf", alpha=2.1, sigma=0.0, max_deg=84230,
                              band_frac=0.85, band=512, feat=128, seed=17),
    # GraphSAGE ppi: 24 tissue graphs of ~2.4 k proteins each
    "ppi_like": dict(num_nodes=56944, mean_deg=818716 / 56944, sigma=1.0, max_deg=721, band_frac=1.0, band=0, graphs=24,
                     community_sigma=0.3, graph_min=500, feat=128, seed=18),
    # ogbl-ddi: 4,267 drugs, density 11.7 %, no structure assumed
    "ddi_like": dict(num_nodes=4267, mean_deg=2135822 / 4267, sigma=0.9, max_deg=2234, band_frac=0.0, band=0, feat=128, seed=19),
    # YelpChi, relation R-S-R (reviews of one product with the same rating in one month): near-cliques of reviews; 90 % of a
    # row's edges inside its clique-like community (as a reordered file would have them, bench_all.py:120-129)
    "fraud_yelp_rsr_like": dict(num_nodes=45954, mean_deg=6805486 / 45954, sigma=1.0, max_deg=6000, band_frac=0.9, band=0,
                                communities=300, community_sigma=1.0, feat=128, seed=20),
    # ogbn-proteins: 132,534 proteins of 8 species, 79.1 M directed edges, associations mostly inside a species
    "protein_like": dict(num_nodes=132534, mean_deg=79122504 / 132534, sigma=0.9, max_deg=7750, band_frac=0.85, band=0,
                         communities=8, community_sigma=0.6, feat=128, seed=21),
}

# bench/plot.py:8 order -> stand-in
EVALUATION_SET = {"amazon0505": "amazon0505_like", "DD": "dd_like", "ppi": "ppi_like", "reddit": "reddit_like",
                  "amazon0601": "amazon0601_like", "com-amazon": "com_amazon_like", "ddi": "ddi_like",
                  "FraudYelp-RSR": "fraud_yelp_rsr_like", "web-BerkStan": "web_berkstan_like", "protein": "protein_like",
                  "YeastH": "yeasth_like", "Yeast": "yeast_like"}


def lognormal_degrees(num_nodes, mean_deg, sigma, max_deg, gen, device):
    z = torch.randn(num_nodes, generator=gen, device=device, dtype=torch.float64)
    base = torch.exp(sigma * z)
    scale = mean_deg / base.mean()
    for _ in range(12):  # rescale so that the clipped mean hits the target
        d = (base * scale).clamp(1.0, float(max_deg))
        scale = scale * (mean_deg / d.mean())
    return (base * scale).clamp(1.0, float(max_deg)).round().to(torch.int64)


def zipf_degrees(num_nodes, mean_deg, alpha, max_deg, gen, device):
    """Truncated power law P(d) ~ d^-alpha on [d_min, max_deg] by inverse-CDF sampling; d_min is solved (bisection on
    the analytic mean) so that the mean degree is ``mean_deg``."""
    u = torch.rand(num_nodes, generator=gen, device=device, dtype=torch.float64)
    b = float(max_deg)

    def sample(a):
        if abs(alpha - 1.0) < 1e-9:
            return a * torch.exp(u * math.log(b / a))
        p = 1.0 - alpha
        return (a ** p + u * (b ** p - a ** p)) ** (1.0 / p)

    lo, hi = 1.0, min(b, float(mean_deg))
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if float(sample(mid).mean()) < mean_deg:
            lo = mid
        else:
            hi = mid
    return sample(0.5 * (lo + hi)).clamp(1.0, b).round().to(torch.int64)


def draw_degrees(n, mean_deg, sigma, max_deg, gen, device, law="lognormal", alpha=2.0):
    if law == "zipf":
        return zipf_degrees(n, mean_deg, alpha, max_deg, gen, device)
    return lognormal_degrees(n, mean_deg, sigma, max_deg, gen, device)


def community_bounds(n, communities, sigma, seed, device):
    """int64 [communities + 1]: first node of every community (contiguous node ranges; sizes log-normal with the given
    sigma, at least 16 nodes each; seeded separately from the edge stream, so every shard of a graph sees the same ones)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed * 7919 + 13)
    w = torch.exp(sigma * torch.randn(communities, generator=gen, dtype=torch.float64))
    sizes = torch.clamp((w / w.sum() * n).floor().to(torch.int64), min=min(16, max(1, n // communities)))
    sizes[-1] += n - int(sizes.sum())
    if int(sizes[-1]) < 1:                                  # tiny test graphs: fall back to equal sizes
        sizes = torch.full((communities,), n // communities, dtype=torch.int64)
        sizes[-1] += n - int(sizes.sum())
    bounds = torch.zeros(communities + 1, dtype=torch.int64)
    bounds[1:] = torch.cumsum(sizes, 0)
    return bounds.to(device)


def graph_bounds(n, graphs, sigma, min_nodes, seed, device):
    """int64 [G + 1]: first node of every component of a block-diagonal union of about ``graphs`` small graphs over ``n`` nodes
    (TU-style sets): sizes log-normal around n / graphs, at least ``min_nodes``, laid out one after the other until the nodes
    run out (the last component takes what is left)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed * 7919 + 17)
    mean = n / max(1, graphs)
    draw = int(graphs * 1.5) + 16
    w = torch.exp(sigma * torch.randn(draw, generator=gen, dtype=torch.float64) - 0.5 * sigma * sigma)
    sizes = torch.clamp((w * mean).round().to(torch.int64), min=min(min_nodes, max(1, n)))
    ends = torch.cumsum(sizes, 0)
    k = int(torch.searchsorted(ends, torch.tensor(n, dtype=torch.int64)))
    if k >= draw:   # sizes came out small: equal components for the rest
        extra = torch.arange(int(ends[-1]) + int(mean) + 1, n + int(mean) + 1, max(1, int(mean)), dtype=torch.int64)
        ends = torch.cat([ends, extra])
        k = int(torch.searchsorted(ends, torch.tensor(n, dtype=torch.int64)))
    bounds = torch.zeros(k + 2, dtype=torch.int64)
    bounds[1:k + 1] = ends[:k]
    bounds[k + 1] = n
    if int(bounds[k + 1] - bounds[k]) < min(min_nodes, n) and k >= 1:   # a sliver at the end joins its neighbour
        bounds = torch.cat([bounds[:k], bounds[k + 1:]])
    return bounds.to(device)


def _local_columns(grow, n, gen, device, half=0, bounds=None):
    """The LOCAL half of the column mixture for global rows ``grow``: a band of +- half around the row, or a uniform node of
    the row's own community (``bounds``)."""
    if bounds is not None:
        c = torch.searchsorted(bounds, grow, right=True) - 1
        lo, size = bounds[c], bounds[c + 1] - bounds[c]
        return lo + (torch.rand(grow.numel(), generator=gen, device=device, dtype=torch.float64) * size).long().minimum(size - 1)
    return (grow + torch.randint(-half, half + 1, (grow.numel(),), generator=gen, device=device, dtype=torch.int64)).clamp_(0, n - 1)


def _top_up(keys, deg, r0, n, gen, device, band_frac=0.0, half=0, bounds=None):
    """``keys`` = sorted unique (row - r0) * n + col; adds new columns until every row has deg[row] edges (rows are
    never over-full: the first draw makes at most deg[row] distinct ones).  The new columns follow the config's own
    band / uniform mixture for six rounds, then they are uniform (a hub row can saturate its band)."""
    nrows = deg.numel()
    for round_no in range(14):
        rows = torch.div(keys, n, rounding_mode="floor")
        have = torch.bincount(rows, minlength=nrows)
        del rows
        deficit = deg - have
        total = int(deficit.sum())
        if total == 0:
            break
        # candidates: 1.25 x deficit + 4 per short row (a few collide with existing edges or with each other)
        want = torch.where(deficit > 0, (deficit * 5) // 4 + 4, torch.zeros_like(deficit))
        crow = torch.repeat_interleave(torch.arange(nrows, device=device, dtype=torch.int64), want)
        ccol = torch.randint(0, n, (crow.numel(),), generator=gen, device=device, dtype=torch.int64)
        if band_frac > 0 and (half > 0 or bounds is not None) and round_no < 6:
            local = _local_columns(crow + r0, n, gen, device, half, bounds)
            pick = torch.rand(crow.numel(), generator=gen, device=device) < band_frac
            ccol = torch.where(pick, local, ccol)
            del local, pick
        cand = torch.unique(crow * n + ccol)
        del crow, ccol
        pos = torch.searchsorted(keys, cand).clamp_(max=keys.numel() - 1)
        cand = cand[keys[pos] != cand]                      # new edges only
        # keep a uniformly random subset of deficit[row] candidates per row: random priority, stable by row
        prio = torch.rand(cand.numel(), generator=gen, device=device)
        cand = cand[torch.argsort(prio)]
        crow = torch.div(cand, n, rounding_mode="floor")
        order = torch.argsort(crow, stable=True)
        cand, crow = cand[order], crow[order]
        first = torch.searchsorted(crow, torch.arange(nrows, device=device, dtype=torch.int64))
        rank = torch.arange(cand.numel(), device=device, dtype=torch.int64) - first[crow]
        cand = cand[rank < deficit[crow]]
        keys = torch.sort(torch.cat([keys, cand])).values   # disjoint sets: still duplicate-free
    return keys


def generate_csr(num_nodes, mean_deg, sigma, max_deg, band_frac, band, seed, device="cpu", scale=1.0, law="lognormal",
                 alpha=2.0, rows=None, exact_degrees=True, communities=0, community_sigma=0.8, graphs=0, graph_min=2, **_):
    """Returns ``(indptr int32 [R+1], indices int32 [nnz])`` on ``device`` for the row range ``rows`` (default: all
    ``R = N`` rows; column ids are always global).

    ``scale`` < 1 shrinks the node count (degrees are kept, capped at N/2) for CPU-sized test cases.
    """
    device = torch.device(device)
    n = max(16, int(round(num_nodes * scale)))
    max_deg = int(min(max_deg, max(1, n // 2)))
    mean_deg = min(mean_deg, max_deg / 2)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    deg = draw_degrees(n, mean_deg, sigma, max_deg, gen, device, law, alpha)
    r0, r1 = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    assert 0 <= r0 <= r1 <= n
    if rows is not None:
        deg = deg[r0:r1].contiguous()
        gen.manual_seed(seed * 1000003 + r0 + 1)             # the edge stream of a shard depends on its first row only
    nrows = r1 - r0
    lrow = torch.repeat_interleave(torch.arange(nrows, device=device, dtype=torch.int64), deg)
    e = lrow.numel()
    cols = torch.randint(0, n, (e,), generator=gen, device=device, dtype=torch.int64)
    half = min(band, max(1, n // 4)) if (band_frac > 0 and band > 0) else 0
    bounds = community_bounds(n, communities, community_sigma, seed, device) if (band_frac > 0 and communities > 0) else None
    if band_frac > 0 and graphs > 0:   # block-diagonal union of small graphs: their number shrinks with the node count
        bounds = graph_bounds(n, max(1, int(round(graphs * n / num_nodes))), community_sigma, graph_min, seed, device)
    if half > 0 or bounds is not None:
        local = _local_columns(lrow + r0, n, gen, device, half, bounds)
        pick = torch.rand(e, generator=gen, device=device) < band_frac
        cols = torch.where(pick, local, cols)
        del local, pick
    keys = lrow * n + cols
    del lrow, cols
    keys = torch.unique(keys, sorted=True)  # sorts by (row, col) and drops duplicate edges
    if exact_degrees:
        keys = _top_up(keys, deg, r0, n, gen, device, band_frac, half, bounds)
    lrow = torch.div(keys, n, rounding_mode="floor")
    indices = (keys - lrow * n).to(torch.int32)
    del keys
    counts = torch.bincount(lrow, minlength=nrows)
    indptr = torch.zeros(nrows + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    assert int(indptr[-1]) < 2 ** 31
    return indptr.to(torch.int32), indices


def _resolve(name: str) -> dict:
    """Config of ``name`` with a ``base=`` entry (label-shuffled variants) merged over its base config."""
    cfg = dict(CONFIGS[name])
    if "base" in cfg:
        cfg = dict(CONFIGS[cfg["base"]], shuffle_seed=cfg["shuffle_seed"], base=cfg["base"])
    return cfg


def shuffle_labels(indptr: torch.Tensor, indices: torch.Tensor, seed: int):
    """``P A P^T`` for a seeded random relabelling of the nodes: ``(indptr, indices, label)`` with ``label[old] = new`` (int64);
    rows sorted, int32, on the input's device."""
    device = indptr.device
    n = indptr.numel() - 1
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    label = torch.randperm(n, generator=gen, device=device)
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    keys = torch.sort(label[rows] * n + label[indices.long()]).values
    del rows
    new_rows = torch.div(keys, n, rounding_mode="floor")
    new_indices = (keys - new_rows * n).to(torch.int32)
    new_indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    new_indptr[1:] = torch.cumsum(torch.bincount(new_rows, minlength=n), 0)
    return new_indptr.to(torch.int32), new_indices, label


def target_degrees(name: str, device="cpu", scale: float = 1.0) -> torch.Tensor:
    """int64 [N]: the degree of every row of config ``name`` (what ``generate`` produces with exact degrees) -- lets
    every rank of a sharded run compute the same edge-balanced row partition without building the graph."""
    cfg = _resolve(name)
    if "shuffle_seed" in cfg:   # the base graph's degrees, moved to the rows' new labels
        base = target_degrees(cfg["base"], device=device, scale=scale)
        gen = torch.Generator(device=base.device)
        gen.manual_seed(cfg["shuffle_seed"])
        label = torch.randperm(base.numel(), generator=gen, device=base.device)
        out = torch.empty_like(base)
        out[label] = base
        return out
    device = torch.device(device)
    n = max(16, int(round(cfg["num_nodes"] * scale)))
    max_deg = int(min(cfg["max_deg"], max(1, n // 2)))
    mean_deg = min(cfg["mean_deg"], max_deg / 2)
    gen = torch.Generator(device=device)
    gen.manual_seed(cfg["seed"])
    return draw_degrees(n, mean_deg, cfg["sigma"], max_deg, gen, device, cfg.get("law", "lognormal"), cfg.get("alpha", 2.0))


def generate(name: str, device="cpu", scale: float = 1.0, rows=None):
    cfg = _resolve(name)
    if "shuffle_seed" in cfg:
        assert rows is None, "label-shuffled stand-ins are generated whole"
        indptr, indices, _ = generate(cfg["base"], device=device, scale=scale)
        indptr, indices, _ = shuffle_labels(indptr, indices, cfg["shuffle_seed"])
        return indptr, indices, cfg
    indptr, indices = generate_csr(device=device, scale=scale, rows=rows, **cfg)
    return indptr, indices, cfg


def algorithmic_bytes(num_nodes: int, nnz: int, feat: int, in_bytes: int, out_bytes: int = 4) -> int:
    """int32 CSR once, B once, C once."""
    return 4 * (nnz + num_nodes + 1) + num_nodes * feat * in_bytes + num_nodes * feat * out_bytes


def flops(nnz: int, feat: int) -> int:
    return 2 * nnz * feat
