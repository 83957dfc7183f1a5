"""Round 6: what the CSR row-gather kernel would do on graphs with hub rows if long rows were split -- emulated with the shipped kernel on a
CSR of VIRTUAL rows (segments of at most SEG edges; the per-row sum of segments is left out: a lower bound for a split-row kernel).
python exp_csr_split_rows.py [graph] [SEG ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import capi  # noqa: E402


def ms(fn):
    for _ in range(3):
        fn()
    t = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        e.synchronize()
        t.append(s.elapsed_time(e) / 5)
    return sorted(t)[2]


graph = sys.argv[1] if len(sys.argv) > 1 else "web_berkstan_like"
segs = [int(a) for a in sys.argv[2:]] or [64, 128, 256, 512]
indptr, indices, _ = synth_graphs.generate(graph, device="cuda")
n, e = indptr.numel() - 1, indices.numel()
deg = (indptr[1:] - indptr[:-1]).long()
handle = voltrix.csr_preprocess_device(indptr, indices, n)
handle[1].hash_tag = f"exp_split/{graph}"
stream = torch.cuda.current_stream().cuda_stream
print(f"{graph}: N={n} nnz={e} max degree {int(deg.max())}, rows above 256 edges: {int((deg > 256).sum())} holding {int(deg[deg > 256].sum())} edges", flush=True)
for width in (128, 512):
    for dtype in (torch.float16, torch.float32):
        feat = torch.randn(n, width, device="cuda").to(dtype)
        os.environ["VOLTRIX_CSR_PATH"] = "0"
        block = ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat))
        os.environ.pop("VOLTRIX_CSR_PATH")
        out = torch.empty(n, width, device="cuda")
        whole = ms(lambda: capi.launch_spmm_csr_rows(indptr, indices, n, feat, out, stream, 1))
        line = f"F={width} {str(dtype)[6:]}: block-format path {block:.4f} ms, CSR kernel on whole rows {whole:.4f}"
        for seg in segs:
            nseg = (deg + seg - 1) // seg
            nseg = nseg.clamp(min=1)
            first = torch.cumsum(nseg, 0) - nseg                      # first virtual row of every row
            vrows = int(nseg.sum())
            owner = torch.repeat_interleave(torch.arange(n, device="cuda"), nseg)
            k = torch.arange(vrows, device="cuda") - first[owner]     # segment number inside its row
            vptr = torch.empty(vrows + 1, dtype=torch.int32, device="cuda")
            vptr[:-1] = (indptr[:-1].long()[owner] + k * seg).to(torch.int32)
            vptr[-1] = e
            vout = torch.empty(vrows, width, device="cuda")
            t = ms(lambda: capi.launch_spmm_csr_rows(vptr, indices, vrows, feat, vout, stream, 1))
            line += f", segments of {seg}: {t:.4f} ({vrows - n} extra rows)"
        print(line, flush=True)
