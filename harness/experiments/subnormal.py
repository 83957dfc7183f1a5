import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
os.environ.setdefault('VOLTRIX_CACHE_DIR', os.path.join(ROOT, 'voltrix-spmm_amd', '.jit_cache'))
import torch, voltrix
from voltrix import capi
indptr = torch.tensor([0,1,2] + [2]*15, dtype=torch.int32).cuda(); indices = torch.tensor([0,1], dtype=torch.int32).cuda()
n = 17
h = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
vals = torch.tensor([3.0e-5, -3.0436e-5, 5.96e-8, 6.0e-5, 6.2e-5, 1e-7, 2e-6, 1.0])
feat = torch.zeros(n, 8); feat[0] = vals; feat[1] = vals * 2
for is16, f in ((True, feat.half().cuda()), (False, feat.cuda())):
    out = torch.zeros(n, 8, device='cuda')
    rc = capi.launch_spmm(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, 2, 8, f.data_ptr(), out.data_ptr(), is16, (32,4,1), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print('f16' if is16 else 'f32', rc)
    print('  in ', f[0].float().cpu().tolist())
    print('  out', out[0].cpu().tolist())
