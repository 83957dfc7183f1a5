"""Child process of tests/test_gpu_tuner_bucket.py::test_shipped_buckets_on_held_out_graphs: one held-out graph (NOT one of the
stand-ins / seeds / sizes the shipped tuned_defaults.json was collected on), first call + steady step, either with the shipped
buckets (``shipped``) or with an empty store and the full tuning sweep (``swept``).
    python tuner_heldout_worker.py <shipped|swept> <graph> <feat> <store.json> <out.json>"""
import json
import os
import sys
import time

mode, graph, feat_dim, store, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ["VOLTRIX_TUNED_STORE"] = store
if mode == "swept":
    os.environ.update(VOLTRIX_TUNED_DEFAULTS="0", VOLTRIX_TUNE_SPACE="full")
else:
    os.environ.update(VOLTRIX_TUNED_DEFAULTS="1", VOLTRIX_TUNE_SPACE="default")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402

C = synth_graphs.CONFIGS
HELD_OUT = {
    # another seed and HALF the nodes of the co-purchase stand-in (same degree law)
    "copurchase_half": dict(C["amazon0505_like"], num_nodes=C["amazon0505_like"]["num_nodes"] // 2, seed=911),
    # another seed and TWICE the nodes of the small-graph union (DD-like): longer B, same window statistics
    "union_double": dict(C["dd_like"], num_nodes=2 * C["dd_like"]["num_nodes"], graphs=2 * C["dd_like"]["graphs"], seed=912),
    # a degree law none of the stand-ins has: Zipf alpha 2.4, mean degree 40, 30 % of the edges in a band of +- 3000
    "zipf_mid_degree": dict(num_nodes=300_000, mean_deg=40.0, law="zipf", alpha=2.4, sigma=0.0, max_deg=30_000, band_frac=0.3,
                            band=3000, feat=128, seed=913),
    # the headline family at half size with another seed (two-level side-car decided from its own counts)
    "reddit_half": dict(C["reddit_like"], num_nodes=C["reddit_like"]["num_nodes"] // 2, seed=914),
}

if graph in HELD_OUT:
    indptr, indices = synth_graphs.generate_csr(device="cuda", **HELD_OUT[graph])
else:       # a named stand-in (experiments: what does the full sweep find on the graphs the shipped buckets were collected on?)
    indptr, indices, _ = synth_graphs.generate(graph, device="cuda")
n, e = indptr.numel() - 1, indices.numel()
handle = voltrix.csr_preprocess_device(indptr, indices, n)
handle[1].hash_tag = f"heldout/{graph}/{mode}"
gen = torch.Generator(device="cuda").manual_seed(0)
feat = torch.randn(n, feat_dim, generator=gen, device="cuda").half()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
torch.cuda.synchronize()
first_call_s = time.perf_counter() - t0
del out


def median_step():
    times = []
    for _ in range(7):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
        t.record()
        t.synchronize()
        times.append(s.elapsed_time(t) / 10)
    return sorted(times)[3]


for _ in range(3):
    voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
json.dump({"graph": graph, "mode": mode, "num_nodes": n, "nnz": e, "first_call_s": first_call_s, "step_ms": median_step(),
           "tuner": jit_tuner.stats, "two_level": voltrix.two_level_of(handle[1]) is not None,
           "points": [{k: str(v) for k, v in p.items()} for p in jit_tuner.tuned_keys.values()]}, open(out_path, "w"))
