import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
os.environ.setdefault('VOLTRIX_CACHE_DIR', os.path.join(ROOT, 'voltrix-spmm_amd', '.jit_cache'))
import torch, voltrix, synth_graphs
from voltrix import jit
from voltrix.jit_kernels import spmm as sp
dev='cuda'
def build(diag, tile, extra_defs=''):
    arg_defs = (("blk_offsets", torch.int32), ("hspa_packed", torch.uint32), ("hind", torch.int32), ("num_nodes", int), ("num_edges", int), ("embedding_dim", int), ("input", torch.float16), ("output", torch.float), ("win_order_a", torch.int32), ("win_order_b", torch.int32), ("win_order_c", torch.int32), ("stream", torch.cuda.Stream))
    body = jit.cpp_format(sp.template, {"FS": tile[0], "DEPTH": tile[1], "WAVES": tile[2], "EB": 2, "SCHED": 2})
    code = jit.generate(sp.includes, arg_defs, body)
    code = f"#define VOLTRIX_DIAG {diag}\n{extra_defs}\n" + code
    return jit.build(f"diag{diag}", arg_defs, code)
if __name__ == '__main__':
    tiles = [(128,3,4),(64,3,4)]
    diags = [0]
    EXTRA = ['', '#define VOLTRIX_EXP_ADDR32 1']
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        for t in tiles:
            for d in diags:
                for x in EXTRA: build(d, t, x)
        sys.exit(0)
    wl = sys.argv[1] if len(sys.argv) > 1 else 'reddit_like'
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    indptr, indices, cfg = synth_graphs.generate(wl, device=dev)
    N = indptr.numel()-1; E = indices.numel()
    p1, packed, hind, bp = voltrix.csr_fused_preprocess_kernel(indptr, indices, N)
    feat = torch.randn(N, F, device=dev).half(); out = torch.empty(N, F, device=dev)
    T = int(p1[-1])
    for t in tiles:
        for d, x in [(d, x) for d in diags for x in EXTRA]:
            rt = build(d, t, x)
            from voltrix.jit_kernels.spmm import window_order
            args = (p1, packed, hind, N, E, F, feat, out, window_order(p1, packed, N, 1), window_order(p1, packed, N, 2), window_order(p1, packed, N, 3), torch.cuda.current_stream())
            for _ in range(2): assert rt(*args) == 0
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): rt(*args)
            e.record(); e.synchronize()
            ms = s.elapsed_time(e)/5
            print(f"tile={t} diag={d} extra={x[8:]!r} ({'noconsume ' if d&1 else ''}{'L2-only' if d&2 else ''}) {ms:.3f} ms gather={8*T*F*2/ms/1e9:.2f} TB/s", flush=True)
