"""What slows the window kernel while the panel kernel runs beside it?  Co-runs the residual window kernel (unit table,
atomic output) with diagnostic builds of the panel kernel (MFMAs removed / row gathers removed / both) and reports each
kernel's own duration (event pair per launch on its stream) and the pair's.

    python harness/experiments/exp_corun_diag.py build     (no GPU needed)
    python harness/experiments/exp_corun_diag.py run       (GPU box)
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
VARIANTS = {"full": 0, "no_mfma": 1, "no_mfma_no_rows": 3, "only_loop_control": 31}
if os.environ.get("EXP_META"):    # how much do the panel kernel's metadata DMAs (16 of its 24 LDS-DMAs per k-step) cost the pair?
    VARIANTS = {"full": 0, "no_meta_dma": 16}
if os.environ.get("EXP_SLEEP"):   # the panel kernel throttled by s_sleep per k-step (name -> quanta)
    VARIANTS = {"full": 0, "sleep1": 0, "sleep2": 0, "sleep4": 0, "sleep8": 0}


def so(name):
    return os.path.join(HERE, "build", f"corun_diag_{name}.so")


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-DVOLTRIX_PANEL_DIAG={bits}", f"-DVOLTRIX_PANEL_SLEEP={int(name[5:]) if name.startswith('sleep') else 0}",
                               f"-I{PKG}/voltrix/include", f"-I{REPO}/include",
                               os.path.join(HERE, "corun_diag.hip"), "-o", so(name)]) for name, bits in VARIANTS.items()]
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch

    import synth_graphs
    import voltrix
    from voltrix import capi, hybrid
    from voltrix.schedule import unit_table

    dev = torch.device("cuda")
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev, scale=scale)
    n, F = indptr.numel() - 1, 128
    print(f"reddit_like scale {scale}: N={n} ({(n + 511) // 512} panels of 512 rows) nnz={indices.numel()}", flush=True)
    feat = torch.randn(n, F, device=dev).half()
    out = torch.zeros(n, F, device=dev)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
    resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
    rn = r_indices.numel()
    from voltrix.schedule import default_max_stages

    tb = unit_table(resid[0], n, max(8, int(1.25 * default_max_stages(resid[0], n) / 1.5)))   # the shipped residual launch:
    buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)

    flags = [1]   # atomic output (round 2 experiment build: | 2 = s_setprio 3 in the window kernel; shipped: always set)

    def window(stream):
        rc = capi.launch_spmm_sched(resid[0].data_ptr(), resid[1].data_ptr(), resid[2].data_ptr(), n, rn, F, feat.data_ptr(),
                                    out.data_ptr(), (128, 3, 4), stream, 0, 0, flags[0], False, tb, buf.data_ptr(), 0, 2)   # two units per wave
        assert rc == 0

    def timed(fn, stream, iters=10):
        pairs = []
        for _ in range(iters):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            fn()
            e.record(stream)
            pairs.append((s, e))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

    for _ in range(3):
        window(main.cuda_stream)
    print(f"window kernel (residual, unit table) alone: {timed(lambda: window(main.cuda_stream), main):.3f} ms", flush=True)
    for name, prio in [(v, p) for v in VARIANTS for p in (1,)]:
        flags[0] = prio
        lib = ctypes.CDLL(so(name))

        def panel(stream):
            rc = lib.corun_diag_launch(ctypes.c_void_p(plan.panel_ptr.data_ptr()), ctypes.c_void_p(plan.panel_cols.data_ptr()),
                                       ctypes.c_void_p(plan.panel_bits.data_ptr()),
                                       ctypes.c_void_p(plan.panel_order.data_ptr()), n, F, ctypes.c_void_p(feat.data_ptr()),
                                       ctypes.c_void_p(out.data_ptr()), 2, ctypes.c_void_p(stream))
            assert rc == 0

        for _ in range(3):
            panel(main.cuda_stream)
        alone = timed(lambda: panel(main.cuda_stream), main)
        w_pairs, p_pairs, tot = [], [], []
        for _ in range(10):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(main)
            fork = torch.cuda.Event()
            fork.record(main)
            side.wait_event(fork)
            ps, pe = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ps.record(side)
            panel(side.cuda_stream)
            pe.record(side)
            ws, we = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ws.record(main)
            window(main.cuda_stream)
            we.record(main)
            main.wait_event(pe)
            t1.record(main)
            w_pairs.append((ws, we)); p_pairs.append((ps, pe)); tot.append((t0, t1))
        torch.cuda.synchronize()
        avg = lambda pairs: sum(a.elapsed_time(b) for a, b in pairs[2:]) / len(pairs[2:])   # noqa: E731
        print(f"panel [{name:24s}] panel alone {alone:.3f} ms | side by side: panel {avg(p_pairs):.3f}, window {avg(w_pairs):.3f}, "
              f"pair {avg(tot):.3f} ms", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
