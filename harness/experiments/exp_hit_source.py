"""Where do the window kernel's TCC hits come from (round 2)?  The two-level residual of the reddit-like graph (uniform random
columns) through the same unit table at F = 64 / 128 / 256 (one / two / four 128-byte lines per gathered row), plus a copy of
B with 384-byte rows (F = 128 in a padded buffer is not expressible, so: F = 192 = three lines).  Run under
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- python3 harness/experiments/exp_hit_source.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "voltrix-spmm_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(ROOT, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import capi, hybrid  # noqa: E402
from voltrix.schedule import unit_table  # noqa: E402

dev = torch.device("cuda")
indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev)
n = indptr.numel() - 1
r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
rn = r_indices.numel()
resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
tb = unit_table(resid[0], n)
stream = torch.cuda.current_stream().cuda_stream
for F, tile in ((64, (64, 3, 4)), (128, (128, 3, 4)), (192, (64, 3, 4)), (256, (128, 3, 4)), (64, (32, 4, 4))):
    feat = torch.randn(n, F, device=dev).half()
    out = torch.empty(n, F, device=dev)
    buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)
    for _ in range(4):
        assert capi.launch_spmm_sched(resid[0].data_ptr(), resid[1].data_ptr(), resid[2].data_ptr(), n, rn, F, feat.data_ptr(),
                                      out.data_ptr(), tile, stream, 0, 0, False, False, tb, buf.data_ptr()) == 0
    torch.cuda.synchronize()
    print(f"F={F} tile {tile}: row = {F * 2} bytes = {F * 2 / 128:g} lines", flush=True)

# ---- second question: are the hits reuse BETWEEN windows?  Same handle, but every window's column ids shifted by its own
# random offset (mod N): each window still gathers as many distinct rows, in (nearly) sorted order, but no two windows want
# the same rows at the same time any more.  (The product is a different one; only the traffic pattern matters here.)
F, tile = 128, (128, 3, 4)
feat = torch.randn(n, F, device=dev).half()
out = torch.empty(n, F, device=dev)
buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)
nblk = (resid[0][1:] - resid[0][:-1]).long()
win_of_block = torch.repeat_interleave(torch.arange(nblk.numel(), device=dev), nblk)
shift = torch.randint(0, n, (nblk.numel(),), device=dev)
hind2 = ((resid[2].view(-1, 8).long() + shift[win_of_block][:, None]) % n).to(torch.int32).reshape(-1).contiguous()
for label, hind in (("original column ids", resid[2]), ("per-window shifted column ids", hind2)):
    for _ in range(4):
        assert capi.launch_spmm_sched(resid[0].data_ptr(), resid[1].data_ptr(), hind.data_ptr(), n, rn, F, feat.data_ptr(),
                                      out.data_ptr(), tile, stream, 0, 0, False, False, tb, buf.data_ptr()) == 0
    torch.cuda.synchronize()
    print(f"F={F} tile {tile}: {label}", flush=True)
