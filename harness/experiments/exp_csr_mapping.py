"""Round 6: the CSR row-gather kernel's workgroup -> rows mapping: consecutive row groups round the XCDs (0) against one contiguous
eighth of the rows per XCD (1).  F = 128 / 512, fp32 and fp16 rows."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
from voltrix import capi  # noqa: E402
from harness import bm_rocsparse  # noqa: E402


def ms(fn):
    for _ in range(5):
        fn()
    t = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        e.synchronize()
        t.append(s.elapsed_time(e) / 10)
    return sorted(t)[2]


for graph in (sys.argv[1] if len(sys.argv) > 1 else "dd_like,com_amazon_like,amazon0601_like,amazon0505_like,ppi_like,yeast_like").split(","):
    indptr, indices, _ = synth_graphs.generate(graph, device="cuda")
    n = indptr.numel() - 1
    line = {"graph": graph}
    for f in (128, 512):
        for dt in (torch.float32, torch.float16):
            feat = torch.randn(n, f, device="cuda").to(dt)
            out = torch.empty(n, f, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            line[f"F{f} {str(dt)[6:]}"] = [round(ms(lambda x=x: capi.launch_spmm_csr_rows(indptr, indices, n, feat, out, stream, x)), 4) for x in (0, 1)]
            cells = bm_rocsparse.baselines(indptr, indices, n, feat, algorithms={}, iters=20, warmup=5)      # the harness's kernel, same process
            line[f"F{f} {str(dt)[6:]}"].append({k: round(v, 4) for k, v in cells.items()})
    print(json.dumps(line), flush=True)
