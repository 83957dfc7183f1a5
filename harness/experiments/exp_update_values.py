"""Round 6: new edge values on the same pattern -- voltrix.update_edge_values (one scatter through the cached edge -> plane map) against
rebuilding the weighted handle, at full size.  python exp_update_values.py [workload] [F]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402


def wall(fn, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return sorted(out)[len(out) // 2]


workload = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
num_feats = int(sys.argv[2]) if len(sys.argv) > 2 else 128
indptr, indices, _ = synth_graphs.generate(workload, device="cuda")
n, e = indptr.numel() - 1, indices.numel()
gen = torch.Generator(device="cuda").manual_seed(3)
values = torch.rand(e, device="cuda", generator=gen) + 0.1
feat = torch.randn(n, num_feats, device="cuda", generator=gen).half()
t_build = wall(lambda: voltrix.csr_preprocess_weighted(indptr, indices, values, n, plane_dtype=torch.float16, separable=False), reps=2)
h = voltrix.csr_preprocess_weighted(indptr, indices, values, n, plane_dtype=torch.float16, separable=False)
out = voltrix.spmm_weighted(h, feat, hash_tag=f"exp_update/{workload}")
step = wall(lambda: voltrix.spmm_weighted(h, feat), reps=7)
new = torch.rand(e, device="cuda", generator=gen) + 0.1
t_first = wall(lambda: voltrix.update_edge_values(h, new), reps=1)          # builds the edge -> plane map
t_update = wall(lambda: voltrix.update_edge_values(h, new), reps=5)
fresh = voltrix.csr_preprocess_weighted(indptr, indices, new, n, plane_dtype=torch.float16, separable=False)
same = torch.equal(h.planes[torch.float16], fresh.planes[torch.float16])
print(f"{workload} N={n} nnz={e} F={num_feats}: weighted step {step:.3f} ms; rebuild of the handle {t_build:.1f} ms; update_edge_values first call "
      f"{t_first:.1f} ms (edge -> plane map, {h.edge_slot.element_size()} B per edge), then {t_update:.3f} ms per change of values; same plane as a rebuild: {same}")
