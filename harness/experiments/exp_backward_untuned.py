"""Round 6: the transposed reddit-like handle under VOLTRIX_TUNE_SPACE=none ran 7.8 ms (default space: 1.27): which kernel?"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ["VOLTRIX_TUNE_SPACE"] = sys.argv[1] if len(sys.argv) > 1 else "none"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.autograd import SpMM  # noqa: E402
from voltrix.utils import KernelTimer  # noqa: E402

indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda")
n = indptr.numel() - 1
op = SpMM(indptr, indices, n, hash_tag="exp_backward_untuned")
feat = torch.randn(n, 128, device="cuda").half()
for name, h in (("forward", op.handle), ("backward", op.handle_t)):
    two = voltrix.two_level_of(h[1])
    for _ in range(3):
        voltrix.spmm(*h, num_nodes=n, num_edges=indices.numel(), feat=feat)
    with KernelTimer() as t:
        for _ in range(5):
            voltrix.spmm(*h, num_nodes=n, num_edges=indices.numel(), feat=feat)
    blk = two.blk_offsets
    nst = ((blk[1:] - blk[:-1]) + 3) // 4
    print(name, {k: (v[0], round(v[1], 4)) for k, v in t.summary().items()}, "ksteps", two.plan.num_ksteps, "resid stages", int(nst.sum()),
          "max stages/window", int(nst.max()), "median", float(nst.float().median()), "parts", None if two.plan.parts is None else two.plan.parts.num_parts,
          "longest panel", int((two.plan.panel_ptr[1:] - two.plan.panel_ptr[:-1]).max()), flush=True)
