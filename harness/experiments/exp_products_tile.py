"""Round 6: products-like, label-shuffled and reordered (clusters+votes): same local share as the generating order and 27 % slower -- is it the
order or the TILE the reordered handle gets (a bucket miss -> the first-call sweep on a sample)?  Prints, per handle, the tuner's statistics,
the chosen point and the step; then the reordered handle again with the natural handle's point forced through an exact-key store entry."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402


def ms(fn):
    for _ in range(5):
        fn()
    t = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        e.synchronize()
        t.append(s.elapsed_time(e) / 5)
    return sorted(t)[2]


graph = sys.argv[1] if len(sys.argv) > 1 else "products_like"
indptr, indices, _ = synth_graphs.generate(graph, device="cuda")
n, e = indptr.numel() - 1, indices.numel()
feat = torch.randn(n, 128, device="cuda").half()
nat = voltrix.csr_preprocess_device(indptr, indices, n)
nat[1].hash_tag = f"exp_products_tile/{graph}/natural"
before = dict(jit_tuner.stats)
t_nat = ms(lambda: voltrix.spmm(*nat, num_nodes=n, num_edges=e, feat=feat))
print("natural", round(t_nat, 4), {k: jit_tuner.stats[k] - before.get(k, 0) for k in ("sweeps", "bucket_hits", "stored_hits")},
      [dict(p) for p in jit_tuner.tuned_keys.values()][-1], flush=True)
s_indptr, s_indices, _ = synth_graphs.shuffle_labels(indptr, indices, 1000 + len(graph))
del indptr, indices
h = voltrix.csr_preprocess_reordered(s_indptr, s_indices, n, method="clusters+votes", relabel=True)
h.hspa_packed.hash_tag = f"exp_products_tile/{graph}/reordered"
fin = voltrix.permute_features(h, feat)
before = dict(jit_tuner.stats)
t_re = ms(lambda: voltrix.spmm_reordered(h, fin))
print("reordered", round(t_re, 4), {k: jit_tuner.stats[k] - before.get(k, 0) for k in ("sweeps", "bucket_hits", "stored_hits")},
      [dict(p) for p in jit_tuner.tuned_keys.values()][-1], "tc_blocks natural / reordered", int(nat[0][-1]), int(h.blk_offsets[-1]), flush=True)
