import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
os.environ.setdefault('VOLTRIX_CACHE_DIR', os.path.join(ROOT, 'voltrix-spmm_amd', '.jit_cache'))
import torch, voltrix, synth_graphs
from voltrix import capi
dev='cuda'
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e)/iters
for name, feats in (('reddit_like', (128, 256, 512)), ('products_like', (128,))):
    indptr, indices, cfg = synth_graphs.generate(name, device=dev)
    N = indptr.numel()-1; E = indices.numel()
    h = voltrix.csr_fused_preprocess_kernel(indptr, indices, N)[:3]
    s = torch.cuda.current_stream().cuda_stream
    order = torch.empty((N+15)//16, dtype=torch.int32, device=dev); capi.launch_window_order(h[0], N, order, s)
    for F in feats:
        for dt in (torch.float32, torch.float16):
            feat = torch.randn(N, F, device=dev).to(dt)
            res = {}
            try:
                A = torch.sparse_csr_tensor(indptr, indices, torch.ones(E, device=dev, dtype=dt), size=(N, N))
                ref = A @ feat
                res['rocsparse_csr'] = timeit(lambda: A @ feat)
            except Exception as ex:
                res['rocsparse_csr'] = 'ERR ' + str(ex)[:80]
                ref = None
            out = torch.empty(N, F, device=dev)
            f16 = feat.half()
            tile = (64,3,4) if F <= 128 else (128,3,4)
            if dt == torch.float16:
                res['voltrix_f16'] = timeit(lambda: capi.launch_spmm(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), N, E, F, f16.data_ptr(), out.data_ptr(), True, tile, s, order.data_ptr()))
            else:
                ws = torch.empty(N, F, device=dev, dtype=torch.float16)
                def f():
                    capi.launch_cast_f32_f16(feat, ws, s)
                    capi.launch_spmm(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), N, E, F, ws.data_ptr(), out.data_ptr(), True, tile, s, order.data_ptr())
                res['voltrix_f32_via_f16'] = timeit(f)
            err = float((out - ref.float()).norm() / ref.float().norm()) if ref is not None else float('nan')
            print(f"{name} F={F} {str(dt)[6:]}: " + " ".join(f"{k}={v if isinstance(v,str) else f'{v:.3f}ms'}" for k, v in res.items()) + f" relerr_vs_rocsparse={err:.2e}", flush=True)
            del feat, out
