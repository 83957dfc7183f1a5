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
def permute_windows(indptr, indices, order):
    # order: tensor of window ids (new position j holds old window order[j]); full 16-row windows only
    N = indptr.numel()-1
    rows = (order[:, None]*16 + torch.arange(16, device=dev)[None, :]).reshape(-1)
    rows = torch.cat([rows, torch.arange(order.numel()*16, N, device=dev)])  # tail rows stay in place
    deg = (indptr[1:] - indptr[:-1]).long()
    newdeg = deg[rows]
    newptr = torch.zeros(N+1, dtype=torch.int64, device=dev); newptr[1:] = torch.cumsum(newdeg, 0)
    # gather edges
    starts = indptr[:-1].long()[rows]
    idx = torch.repeat_interleave(starts - newptr[:-1], newdeg) + torch.arange(int(newptr[-1]), device=dev)
    return newptr.to(torch.int32), indices[idx].contiguous()
F=128
for name in ('reddit_like', 'reddit_uniform'):
    indptr, indices, cfg = synth_graphs.generate(name, device=dev)
    N = indptr.numel()-1; W = (N+15)//16
    Wfull = N//16
    feat = torch.randn(N, F, device=dev).half(); out = torch.empty(N, F, device=dev)
    p1, packed, hind, bp = voltrix.csr_fused_preprocess_kernel(indptr, indices, N)
    nblk = bp[:Wfull].clone()
    wpx = (W + 7)//8
    def order_sorted_within(chunk):
        ids = torch.arange(Wfull, device=dev)
        key = (ids // chunk) * (1<<20) + ((1<<20) - 1 - nblk.long())   # descending length inside each chunk
        return ids[torch.argsort(key)]
    variants = {'natural': torch.arange(Wfull, device=dev)}
    for chunk in (64, 256, 1024, wpx):
        variants[f'sorted/{chunk}'] = order_sorted_within(chunk)
    g = torch.Generator(device=dev).manual_seed(0)
    variants['shuffled/xcd'] = torch.cat([ (torch.arange(s, min(s+wpx, Wfull), device=dev))[torch.randperm(min(s+wpx, Wfull)-s, device=dev, generator=g)] for s in range(0, Wfull, wpx)])
    for vname, order in variants.items():
        ip, ix = permute_windows(indptr, indices, order)
        h = voltrix.csr_fused_preprocess_kernel(ip, ix, N)[:3]
        assert int(h[0][-1]) >= int(p1[-1]) - 2
        res = []
        for tile in ((64,4,4),(128,3,1)):
            ms = bench(h, N, ix.numel(), F, feat, out, tile)
            res.append(f"{tile}: {ms:.3f} ms")
        print(f"{name} {vname:14s} " + " | ".join(res), flush=True)
