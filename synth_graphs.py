"""Seeded synthetic stand-ins for the BASELINE.json workloads (no datasets are reachable offline).

Bench / test infrastructure, not part of the product package.  Generators follow SURVEY.md section 8(d):
every graph is a CSR with sorted, duplicate-free rows, int32 ``indptr`` / ``indices``; they run on any torch
device (GPU for the full sizes, CPU for the scaled-down test cases).
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
    "powerlaw_4m": dict(num_nodes=4000000, mean_deg=400.0, sigma=2.0, max_deg=400000, band_frac=0.0, band=0,
                        feat=256, seed=3),
}


def lognormal_degrees(num_nodes, mean_deg, sigma, max_deg, gen, device):
    z = torch.randn(num_nodes, generator=gen, device=device, dtype=torch.float64)
    base = torch.exp(sigma * z)
    scale = mean_deg / base.mean()
    for _ in range(12):  # rescale so that the clipped mean hits the target
        d = (base * scale).clamp(1.0, float(max_deg))
        scale = scale * (mean_deg / d.mean())
    return (base * scale).clamp(1.0, float(max_deg)).round().to(torch.int64)


def generate_csr(num_nodes, mean_deg, sigma, max_deg, band_frac, band, seed, device="cpu", scale=1.0, **_):
    """Returns ``(indptr int32 [N+1], indices int32 [nnz])`` on ``device``.

    ``scale`` < 1 shrinks the node count (degrees are kept, capped at N/2) for CPU-sized test cases.
    """
    device = torch.device(device)
    n = max(16, int(round(num_nodes * scale)))
    max_deg = int(min(max_deg, max(1, n // 2)))
    mean_deg = min(mean_deg, max_deg / 2)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    deg = lognormal_degrees(n, mean_deg, sigma, max_deg, gen, device)
    rows = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    e = rows.numel()
    cols = torch.randint(0, n, (e,), generator=gen, device=device, dtype=torch.int64)
    if band_frac > 0 and band > 0:
        half = min(band, max(1, n // 4))
        local = rows + torch.randint(-half, half + 1, (e,), generator=gen, device=device, dtype=torch.int64)
        local = local.clamp_(0, n - 1)
        pick = torch.rand(e, generator=gen, device=device) < band_frac
        cols = torch.where(pick, local, cols)
        del local, pick
    keys = rows * n + cols
    del rows, cols
    keys = torch.unique(keys, sorted=True)  # sorts by (row, col) and drops duplicate edges
    rows = torch.div(keys, n, rounding_mode="floor")
    indices = (keys - rows * n).to(torch.int32)
    del keys
    counts = torch.bincount(rows, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    assert int(indptr[-1]) < 2 ** 31
    return indptr.to(torch.int32), indices


def generate(name: str, device="cpu", scale: float = 1.0):
    cfg = dict(CONFIGS[name])
    indptr, indices = generate_csr(device=device, scale=scale, **cfg)
    return indptr, indices, cfg


def algorithmic_bytes(num_nodes: int, nnz: int, feat: int, in_bytes: int, out_bytes: int = 4) -> int:
    """int32 CSR once, B once, C once."""
    return 4 * (nnz + num_nodes + 1) + num_nodes * feat * in_bytes + num_nodes * feat * out_bytes


def flops(nnz: int, feat: int) -> int:
    return 2 * nnz * feat
