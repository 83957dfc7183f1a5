import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/voltrix-spmm_amd"]
os.environ.setdefault("VOLTRIX_CACHE_DIR", "/root/repo/voltrix-spmm_amd/.jit_cache")
os.environ["VOLTRIX_TUNE_SPACE"] = "none"; os.environ["VOLTRIX_HYBRID"] = "0"
import torch, synth_graphs, voltrix
from voltrix.autograd import csr_transpose_device
dev = torch.device("cuda")
ip, ix, _ = synth_graphs.generate("reddit_shuffled", device=dev, scale=1.0)
n = ip.numel() - 1; nnz = ix.numel()
h = voltrix.csr_preprocess_device(ip, ix, n)
tip, tix = csr_transpose_device(ip, ix, n, n)
ht = voltrix.csr_preprocess_device(tip, tix, n)
h[1].hash_tag = "prof/a"; ht[1].hash_tag = "prof/at"
x = torch.randn(n, 32, device=dev)
def t(fn, it=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
print("spmm A   f32 feat:", t(lambda: voltrix.spmm(*h, num_nodes=n, num_edges=nnz, feat=x)))
print("spmm A^T f32 feat:", t(lambda: voltrix.spmm(*ht, num_nodes=n, num_edges=nnz, feat=x)))
xh = x.half()
print("spmm A   f16 feat:", t(lambda: voltrix.spmm(*h, num_nodes=n, num_edges=nnz, feat=xh)))
print("gram:", t(lambda: x.T @ x))
def chol():
    g = x.T @ x
    r = torch.linalg.inv(torch.linalg.cholesky(g.double().cpu()).T).float().to(dev)
    return x @ r
print("cholqr pass:", t(chol))
