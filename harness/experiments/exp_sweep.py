import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
os.environ.setdefault('VOLTRIX_CACHE_DIR', os.path.join(ROOT, 'voltrix-spmm_amd', '.jit_cache'))
import torch, voltrix, synth_graphs
from voltrix import capi
from voltrix.schedule import build_stage_list
dev='cuda'
def timeit(fn, iters=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e)/iters
F=128
names = sys.argv[1:] or ['reddit_like', 'reddit_uniform']
for name in names:
    indptr, indices, cfg = synth_graphs.generate(name, device=dev)
    N = indptr.numel()-1; E = indices.numel()
    feat = torch.randn(N, F, device=dev).half(); out = torch.empty(N, F, device=dev); ref = torch.empty(N, F, device=dev)
    h = voltrix.csr_fused_preprocess_kernel(indptr, indices, N)[:3]
    T = int(h[0][-1]); s = torch.cuda.current_stream().cuda_stream
    order = torch.empty((N+15)//16, dtype=torch.int32, device=dev); capi.launch_window_order(h[0], N, order, s)
    base = timeit(lambda: capi.launch_spmm(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), N, E, F, feat.data_ptr(), ref.data_ptr(), True, (64,3,4), s, order.data_ptr()))
    print(f"== {name}: window kernel (64,3,4)+order {base:.3f} ms  gather {8*T*F*2/base/1e9:.2f} TB/s", flush=True)
    for (fs, depth, groups, nw) in ((128,3,2,1536),(128,3,4,1024),(128,4,4,1024)):
        for mode, panel, near, bal in (('plain',0,0,False),('plain',0,0,True),('sweep',8192,6144,True),('sweep',8192,0,True),('sweep',16384,6144,True),('sweep',4096,6144,True),('sweep',32768,6144,True)):
            t0=time.time()
            sl = build_stage_list(h[0], h[1], h[2], N, num_waves=nw, groups=groups, depth=depth, mode=mode, panel_rows=max(panel,1), near_rows=near, balance=bal)
            torch.cuda.synchronize(); tb=time.time()-t0
            fn = lambda: capi.launch_spmm_list(h[1].data_ptr(), h[2].data_ptr(), N, F, feat.data_ptr(), out.data_ptr(), sl.entries, sl.wave_ptr, sl.num_waves, (fs, depth, groups), s)
            rc = fn(); torch.cuda.synchronize()
            if rc: print('rc', rc); continue
            err = float((out-ref).abs().max()/ref.abs().max())
            ms = timeit(fn)
            print(f"   fs={fs} D={depth} G={groups} waves={nw} {mode:5s} bal={int(bal)} panel={panel:5d} near={near:5d} stages={sl.num_stages} rounds={sl.rounds} build={tb*1e3:.0f}ms -> {ms:.3f} ms gather {8*T*F*2/ms/1e9:5.2f} TB/s relerr={err:.1e}", flush=True)
