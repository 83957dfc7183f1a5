import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
os.environ.setdefault('VOLTRIX_CACHE_DIR', os.path.join(ROOT, 'voltrix-spmm_amd', '.jit_cache'))
import torch, voltrix, synth_graphs
from voltrix import capi
dev='cuda'
def bench(h, N, E, F, feat, out, tile, iters=5):
    stream = torch.cuda.current_stream().cuda_stream
    p1, packed, hind = h
    for _ in range(2): capi.launch_spmm(p1.data_ptr(), packed.data_ptr(), hind.data_ptr(), N, E, F, feat.data_ptr(), out.data_ptr(), True, tile, stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): capi.launch_spmm(p1.data_ptr(), packed.data_ptr(), hind.data_ptr(), N, E, F, feat.data_ptr(), out.data_ptr(), True, tile, stream)
    e.record(); e.synchronize()
    return s.elapsed_time(e)/iters
F=128
for name, sigma in (('reddit_uniform', 1.2), ('reddit_uniform', 0.05), ('reddit_like', 1.2)):
    cfg = dict(synth_graphs.CONFIGS[name]); cfg['sigma'] = sigma
    indptr, indices = synth_graphs.generate_csr(device=dev, **cfg)
    Nfull = indptr.numel()-1
    feat = torch.randn(Nfull, F, device=dev).half()
    for rows in (8192, 16384, 24576, 32768, 65536, 131072, Nfull):
        ip = indptr[:rows+1].contiguous(); ix = indices[:int(ip[-1])].contiguous()
        p1, packed, hind, bp = voltrix.csr_fused_preprocess_kernel(ip, ix, rows)
        T = int(p1[-1]); out = torch.empty(rows, F, device=dev)
        res = []
        for tile in ((128,3,1),(64,4,4),(128,4,1)):
            ms = bench((p1,packed,hind), rows, ix.numel(), F, feat, out, tile)
            res.append(f"{tile}: {ms:.3f} ms {8*T*F*2/ms/1e9:5.2f} TB/s")
        print(f"{name} sigma={sigma} rows={rows} windows={rows//16} T={T} | " + " | ".join(res), flush=True)
