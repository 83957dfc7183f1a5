"""Child process of tests/test_gpu_tuner_bucket.py::test_bucket_miss_first_call_is_bounded: the FIRST ``voltrix.spmm`` call on a
graph whose bucket the store does not know (shipped defaults off, empty store): one bounded sweep.
    python tuner_sweep_worker.py <store.json> <workload> <feat> <out.json>"""
import json
import os
import sys
import time

store, workload, feat_dim, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.update(VOLTRIX_TUNED_STORE=store, VOLTRIX_TUNED_DEFAULTS="0", VOLTRIX_TUNE_SPACE="default")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit import compiler  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402

indptr, indices, _ = synth_graphs.generate(workload, device="cuda")
n, e = indptr.numel() - 1, indices.numel()
handle = voltrix.csr_preprocess_device(indptr, indices, n)
handle[1].hash_tag = f"sweep_worker/{workload}"
gen = torch.Generator(device="cuda").manual_seed(0)
feat = torch.randn(n, feat_dim, generator=gen, device="cuda", dtype=torch.float32).half()
voltrix.spmm(*handle, num_nodes=8, num_edges=0, feat=feat[:8]) if False else None
torch.cuda.synchronize()
t0 = time.perf_counter()
out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
torch.cuda.synchronize()
first_call_s = time.perf_counter() - t0
del out
times = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
json.dump({"first_call_s": first_call_s, "step_s": sorted(times)[2], "tuner": jit_tuner.stats, "jit": compiler.build_stats,
           "two_level": voltrix.two_level_of(handle[1]) is not None,
           "points": [{k: str(v) for k, v in p.items()} for p in jit_tuner.tuned_keys.values()]}, open(out_path, "w"))
