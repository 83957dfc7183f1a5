"""Where the host time of one voltrix.spmm call goes on a launch-bound graph (cProfile over 2000 calls; ppi-like, F = 128)."""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ppi_like"
f = int(sys.argv[2]) if len(sys.argv) > 2 else 128
indptr, indices, _ = synth_graphs.generate(name, device="cuda")
n, e = indptr.numel() - 1, indices.numel()
h = voltrix.csr_preprocess_device(indptr, indices, n)
h[1].hash_tag = f"host_overhead/{name}"
for dtype in (torch.float16, torch.float32):
    feat = torch.randn(n, f, device="cuda").to(dtype)
    for _ in range(20):
        voltrix.spmm(*h, n, e, feat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        voltrix.spmm(*h, n, e, feat)
    t_issue = (time.perf_counter() - t0) / 2000 * 1e6
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 2000 * 1e6
    print(f"{name} F={f} {dtype}: host issue {t_issue:.1f} us/call, with the device {t_total:.1f} us/call")
    if True:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            voltrix.spmm(*h, n, e, feat)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(28)
