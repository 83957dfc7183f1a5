import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
os.environ.setdefault('VOLTRIX_CACHE_DIR', os.path.join(ROOT, 'voltrix-spmm_amd', '.jit_cache'))
import torch, voltrix, synth_graphs
from voltrix import capi
from voltrix.schedule import build_stage_list
dev='cuda'
name, mode, fs, depth, groups, nw, panel, near = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
F=128
indptr, indices, cfg = synth_graphs.generate(name, device=dev)
N = indptr.numel()-1; E = indices.numel()
feat = torch.randn(N, F, device=dev).half(); out = torch.empty(N, F, device=dev)
h = voltrix.csr_fused_preprocess_kernel(indptr, indices, N)[:3]
s = torch.cuda.current_stream().cuda_stream
if mode == 'window':
    order = torch.empty((N+15)//16, dtype=torch.int32, device=dev); capi.launch_window_order(h[0], N, order, s)
    fn = lambda: capi.launch_spmm(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), N, E, F, feat.data_ptr(), out.data_ptr(), True, (fs,depth,groups), s, order.data_ptr())
else:
    sl = build_stage_list(h[0], h[1], h[2], N, num_waves=nw, groups=groups, depth=depth, mode=mode, panel_rows=max(panel,1), near_rows=near)
    fn = lambda: capi.launch_spmm_list(h[1].data_ptr(), h[2].data_ptr(), N, F, feat.data_ptr(), out.data_ptr(), sl.entries, sl.wave_ptr, sl.num_waves, (fs, depth, groups), s)
for _ in range(4): assert fn() == 0
torch.cuda.synchronize()
