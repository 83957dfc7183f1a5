"""Round 6: general edge values -- the value-plane path (window / stream kernels, 16-bit planes) against the CSR row-gather kernel with
values (no plane), per graph, width and operand type.  python exp_weighted_csr.py [F ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402


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


widths = [int(a) for a in sys.argv[1:]] or [128]
for graph in ("amazon0505_like", "amazon0601_like", "com_amazon_like", "dd_like", "ppi_like", "yeast_like", "web_berkstan_like",
              "fraud_yelp_rsr_like"):
    indptr, indices, _ = synth_graphs.generate(graph, device="cuda")
    n, e = indptr.numel() - 1, indices.numel()
    gen = torch.Generator(device="cuda").manual_seed(5)
    values = torch.rand(e, device="cuda", generator=gen) + 0.1
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n, separable=False)
    h.hspa_packed.hash_tag = f"exp_weighted_csr/{graph}"
    binary = voltrix.csr_preprocess_device(indptr, indices, n)
    binary[1].hash_tag = f"exp_weighted_csr/{graph}/binary"
    for width in widths:
        for dtype in (torch.float16, torch.float32):
            feat = torch.randn(n, width, device="cuda", generator=gen).to(dtype)
            os.environ["VOLTRIX_CSR_PATH"] = "0"
            plane = ms(lambda: voltrix.spmm_weighted(h, feat))
            os.environ["VOLTRIX_CSR_PATH"] = "1"
            csr = ms(lambda: voltrix.spmm_weighted(h, feat))
            os.environ.pop("VOLTRIX_CSR_PATH")
            auto = ms(lambda: voltrix.spmm_weighted(h, feat))
            unweighted = ms(lambda: voltrix.spmm(*binary, num_nodes=n, num_edges=e, feat=feat))
            print(f"{graph} F={width} {str(dtype)[6:]}: value plane {plane:.4f} ms, CSR kernel with values {csr:.4f} ms, auto {auto:.4f} "
                  f"({h.path_choice.get((width, str(dtype)))}); binary operator {unweighted:.4f}", flush=True)
