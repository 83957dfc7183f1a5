"""reddit_shuffled, symmetric reorder: does more local refinement of the spectral order close the gap to the natural order
(k-steps 224 k vs 182 k; step 1.49 vs 1.37 ms)?  spectral_permutation(refine, refine_width) -> relabelled handle -> step time.
    python harness/experiments/exp_spectral_refine.py [graph] [refine:width,...]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import reorder  # noqa: E402

from exp_reorder_relabel import time_ms  # noqa: E402

graph = sys.argv[1] if len(sys.argv) > 1 else "reddit_shuffled"
variants = [tuple(int(x) for x in v.split(":")) for v in (sys.argv[2] if len(sys.argv) > 2 else "4:8192,6:8192,8:8192,6:4096,8:4096").split(",")]
dev = torch.device("cuda", 0)
indptr, indices, _ = synth_graphs.generate(graph, device=dev)
n, e = indptr.numel() - 1, indices.numel()
feat = torch.randn(n, 128, device=dev).half()
for refine, width in variants:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    perm = reorder.spectral_permutation(indptr, indices, n, iterations=reorder.AUTO_SPECTRAL_ITERATIONS, refine=refine, refine_width=width)
    torch.cuda.synchronize()
    t_order = (time.perf_counter() - t0) * 1e3
    label = torch.empty(n, dtype=torch.int64, device=dev)
    label[perm] = torch.arange(n, device=dev)
    local = reorder.local_fraction(indptr, indices, n, label)
    st = reorder.order_statistics(*reorder.permute_rows_csr(indptr, indices, n, perm), n)
    h = voltrix.csr_preprocess_reordered(indptr, indices, n, method=perm, relabel=True)
    fin = voltrix.permute_features(h, feat)
    ms = time_ms(lambda: voltrix.spmm_reordered(h, fin, hash_tag=f"exp_refine/{graph}/{refine}/{width}"))
    print(json.dumps({"graph": graph, "refine": refine, "width": width, "order_ms": round(t_order, 1), "local_fraction": round(local, 3),
                      "tc_blocks": st["tc_blocks"], "ksteps": st["ksteps"], "shared": round(st["shared_fraction"], 3),
                      "step_ms": round(ms, 4)}), flush=True)
    del h, fin
