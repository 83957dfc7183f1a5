"""Child process of tests/test_gpu_slab_launches.py: the operator on a wide operand with VOLTRIX_SLAB_LAUNCHES as given by the
parent (the switch is read once per process); writes the products.
    python slab_launch_worker.py <out.pt>"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ["VOLTRIX_TUNE_SPACE"] = "none"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402

indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.04)
n = indptr.numel() - 1
torch.manual_seed(0)
feat = torch.randint(-4, 5, (n, 320)).half().cuda()       # 2.5 slabs of 128 columns; integers: every summation order is exact
out = {}
for hybrid in ("0", "1"):
    os.environ["VOLTRIX_HYBRID"] = hybrid
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = f"slab_launches_{hybrid}"
    assert (voltrix.two_level_of(handle[1]) is not None) == (hybrid == "1")
    out[hybrid] = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=feat).cpu()
torch.save(out, sys.argv[1])
