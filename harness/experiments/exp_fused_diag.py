"""Where does the one-launch kernel's time go?  Diagnostic builds of spmm_fused_kernel (VOLTRIX_FUSED_DIAG): without the
residual half-steps, without the panel loop, with every residual row folded into L2.

    python harness/experiments/exp_fused_diag.py build     (no GPU needed)
    python harness/experiments/exp_fused_diag.py run [scale]
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG, ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
VARIANTS = {"full": 0, "no_residual": 1, "no_panel": 2, "rows_in_l2": 4, "no_panel_rows_in_l2": 6}
EXTRA = os.environ.get("EXP_FUSED_DEFINES", "").split()


def so(name):
    return os.path.join(HERE, "build", f"fused_diag_{name}.so")


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-DVOLTRIX_EXPERIMENTAL", f"-DVOLTRIX_FUSED_DIAG={bits}", *EXTRA, f"-I{PKG}/voltrix/include", f"-I{REPO}/include",
                               os.path.join(PKG, "csrc", "capi_spmm_fused.hip"), "-o", so(name)])
             for name, bits in VARIANTS.items()]
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch

    import synth_graphs
    import voltrix
    from voltrix import capi

    dev = torch.device("cuda")
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev, scale=scale)
    n, F = indptr.numel() - 1, 128
    os.environ["VOLTRIX_FUSED"] = "1"
    two = voltrix.csr_preprocess_hybrid(indptr.cpu(), indices.cpu(), n)
    plan, fr = two.plan, two.fused
    pace = int(os.environ.get("EXP_FUSED_PACE", "0"))
    print(f"reddit_like x{scale}: N={n} k-steps={plan.num_ksteps} records={fr.num_records}", flush=True)
    feat = torch.randn(n, F, device=dev).half()
    out = torch.empty(n, F, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for name in VARIANTS:
        lib = ctypes.CDLL(so(name))
        fn = lib.voltrix_launch_spmm_fused_f16
        fn.restype = None

        def launch():
            rc = ctypes.c_int(-1)
            fn(capi._ptr(plan.panel_ptr), capi._ptr(plan.panel_cols), capi._ptr(plan.panel_bits),
               ctypes.c_void_p(plan.panel_order.data_ptr()), capi._ptr(fr.wave_ptr), capi._ptr(fr.records), ctypes.c_int(n),
               ctypes.c_int(F), capi._ptr(feat), capi._ptr(out), ctypes.c_int(128), ctypes.c_int(3), ctypes.c_int(pace),
               ctypes.c_void_p(0), ctypes.c_void_p(stream), ctypes.byref(rc))
            assert rc.value == 0, rc.value

        for _ in range(3):
            launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            launch()
        e.record()
        e.synchronize()
        ms = s.elapsed_time(e) / 10
        resid_gb = 8 * int(two.blk_offsets[-1]) * F * 2 / 1e9
        panel_gb = 32 * plan.num_ksteps * F * 2 / 1e9
        gb = (0 if VARIANTS[name] & 1 else resid_gb) + (0 if VARIANTS[name] & 2 else panel_gb)
        print(f"  {name:22s} {ms:.4f} ms   gathered {gb:.2f} GB -> {gb / ms / 256 * 1e3:.1f} GB/s per CU", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
