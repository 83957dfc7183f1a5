"""Child process of tests/test_gpu_tuner_bucket.py: one ``voltrix.spmm`` call sequence on a seeded graph, with the tuner's
store redirected; writes the product and what the tuner / the JIT did.
    python tuner_bucket_worker.py <store.json> <tag or -> <out.pt> <stats.json> [graph]"""
import json
import os
import sys

store, tag, out_path, stats_path = sys.argv[1:5]
graph = sys.argv[5] if len(sys.argv) > 5 else "reddit_like:0.03"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.update(VOLTRIX_TUNED_STORE=store, VOLTRIX_TUNED_DEFAULTS="0", VOLTRIX_TUNE_SPACE="default", VOLTRIX_HYBRID="0")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit import compiler  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402

name, _, scale = graph.partition(":")
indptr, indices, _ = synth_graphs.generate(name, scale=float(scale or 1.0))
n = indptr.numel() - 1
handle = voltrix.csr_preprocess(indptr, indices, n)
if tag != "-":
    handle[1].hash_tag = tag
torch.manual_seed(0)
feat = torch.randn(n, 64).half().cuda()
import warnings  # noqa: E402

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=feat)
    again = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=feat)
torch.cuda.synchronize()
assert torch.equal(out, again)
torch.save(out.cpu(), out_path)
json.dump({"tuner": jit_tuner.stats, "jit": compiler.build_stats,
           "point": {k: str(v) for k, v in list(jit_tuner.tuned_keys.values())[-1].items()}}, open(stats_path, "w"))
