"""Work per XCD range of the headline pair: residual stages (window kernel) and k-steps (panel kernel) in the eight equal-count
row ranges both kernels use.    python harness/experiments/exp_xcd_work.py [workload]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ["VOLTRIX_HYBRID"] = "1"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
ip, ix, _ = synth_graphs.generate(name, device="cuda")
n = ip.numel() - 1
handle = voltrix.csr_preprocess_device(ip, ix, n)
two = voltrix.two_level_of(handle[1])
blk = two.blk_offsets.long()
stages = ((blk[1:] - blk[:-1]) + 3) // 4
w = stages.numel()
wpx = (w + 7) // 8
ks = (two.plan.panel_ptr[1:] - two.plan.panel_ptr[:-1]).long()
npan = ks.numel()
ppx = (npan + 7) // 8
s_x = [int(stages[x * wpx:(x + 1) * wpx].sum()) for x in range(8)]
k_x = [int(ks[x * ppx:(x + 1) * ppx].sum()) for x in range(8)]
print("windows per XCD", wpx, "panels per XCD", ppx, "(last range:", w - 7 * wpx, "windows,", npan - 7 * ppx, "panels)")
print("residual stages per XCD", s_x, "spread", round(max(s_x) / (sum(s_x) / 8), 4))
print("k-steps per XCD        ", k_x, "spread", round(max(k_x) / (sum(k_x) / 8), 4))
