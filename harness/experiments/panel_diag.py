"""Where does the panel kernel's time go?  Builds diagnostic variants (VOLTRIX_PANEL_DIAG bits) and times them.
Build here (no GPU): python harness/experiments/panel_diag.py build ; run on the GPU box: ... run"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
VARIANTS = {"full": 0, "no_rows": 2, "no_bits": 32, "no_dma": 34, "no_dma_no_expand": 42, "no_dma_no_barrier": 38,
            "no_dma_no_expand_no_barrier": 46, "mfma_only": 2 | 4 | 8 | 16 | 32}
CONFIGS = [(8, 4, 4), (4, 4, 4)]  # (waves, rb, depth)


def so(name, extra=""):
    return os.path.join(HERE, "build", f"panel_diag_{name}{extra}.so")


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for waves, rb, depth in CONFIGS:
        for name, bits in VARIANTS.items():
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                   f"-DVOLTRIX_PANEL_DIAG={bits}", f"-DPD_WAVES={waves}", f"-DPD_RB={rb}", f"-DPD_DEPTH={depth}",
                   f"-I{PKG}/voltrix/include", f"-I{REPO}/include",
                   os.path.join(HERE, "panel_diag.hip"), "-o", so(name, f"_{waves}_{rb}_{depth}")]
            procs.append(subprocess.Popen(cmd))
        for p in procs:
            assert p.wait() == 0
        procs = []


def run():
    import torch
    import synth_graphs
    from voltrix import hybrid
    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda")
    n = indptr.numel() - 1
    feat = torch.randn(n, 128, device="cuda").half()
    out = torch.zeros(n, 128, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for waves, rb, depth in CONFIGS:
      _, _, plan = hybrid.build_panel_plan(indptr, indices, n, None, waves, rb, 4)
      print(f"waves {waves} rb {rb} depth {depth}: k-steps {plan.num_ksteps}, panels {plan.num_panels}", flush=True)
      for name in VARIANTS:
        lib = ctypes.CDLL(so(name, f"_{waves}_{rb}_{depth}"))

        def go():
            rc = lib.panel_diag_launch(ctypes.c_void_p(plan.panel_ptr.data_ptr()), ctypes.c_void_p(plan.panel_cols.data_ptr()),
                                       ctypes.c_void_p(plan.panel_bits.data_ptr()), n, 128, ctypes.c_void_p(feat.data_ptr()),
                                       ctypes.c_void_p(out.data_ptr()), 0, ctypes.c_void_p(stream))
            assert rc == 0
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            go()
        b.record()
        torch.cuda.synchronize()
        print(f"  {name:22s} {a.elapsed_time(b) / 10:.3f} ms", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
