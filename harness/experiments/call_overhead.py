"""Host-side cost of one voltrix.spmm call on a launch-bound (cora-like) problem, with and without graph capture."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "voltrix-spmm_amd"))
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(os.path.dirname(__file__), "..", "..", "voltrix-spmm_amd", ".jit_cache"))
import torch, synth_graphs, voltrix
indptr, indices, _ = synth_graphs.generate("cora_like", device="cuda")
n = indptr.numel() - 1
h = voltrix.csr_preprocess(indptr.cpu(), indices.cpu(), n)
h[1].hash_tag = "cora_like"
for dtype in (torch.float16, torch.float32):
    feat = torch.randn(n, 32, device="cuda").to(dtype)
    for _ in range(5):
        out = voltrix.spmm(*h, n, indices.numel(), feat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        out = voltrix.spmm(*h, n, indices.numel(), feat)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200 * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        voltrix.spmm(*h, n, indices.numel(), feat)
        with torch.cuda.graph(g, stream=s):
            out_g = voltrix.spmm(*h, n, indices.numel(), feat)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(out_g, out)
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{dtype}: eager {eager:.1f} us/call, graph replay {graph:.1f} us/call, same result {ok}")
