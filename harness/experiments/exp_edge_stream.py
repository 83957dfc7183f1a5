"""Experiment (round 2): the two-level residual as a column-ordered edge stream with LDS accumulators
(harness/experiments/edge_stream.hip) against the window kernel on the same residual -- alone, and beside the panel kernel.

    python harness/experiments/exp_edge_stream.py build      (here: hipcc cross-compiles the variants)
    python harness/experiments/exp_edge_stream.py run        (on the GPU box)
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
VARIANTS = [(64, 4, 16), (32, 8, 16), (32, 8, 32), (16, 16, 24), (64, 4, 32)]   # rows per wave, waves per workgroup, batch


def so(v):
    return os.path.join(HERE, "build", "edge_stream_%d_%d_%d.so" % v)


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-DES_ROWS={v[0]}", f"-DES_WAVES={v[1]}", f"-DES_BATCH={v[2]}",
                               os.path.join(HERE, "edge_stream.hip"), "-o", so(v)]) for v in VARIANTS]
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch

    import synth_graphs
    import voltrix
    from voltrix import capi, hybrid
    from voltrix.schedule import unit_table

    dev = torch.device("cuda")
    indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev)
    n, F = indptr.numel() - 1, 128
    feat = torch.randn(n, F, device=dev).half()
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
    rn = r_indices.numel()
    resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
    tb = unit_table(resid[0], n)
    buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)
    out = torch.zeros(n, F, device=dev)
    print(f"reddit_like: N={n} nnz={indices.numel()} residual edges {rn}", flush=True)

    def window(stream, atomic):
        assert capi.launch_spmm_sched(resid[0].data_ptr(), resid[1].data_ptr(), resid[2].data_ptr(), n, rn, F,
                                      feat.data_ptr(), out.data_ptr(), (128, 3, 4), stream, 0, 0, atomic, False, tb,
                                      buf.data_ptr()) == 0

    def panel(stream):
        assert capi.launch_spmm_panel(plan, feat.data_ptr(), out.data_ptr(), F, 2, False, (128, 3, 1), 0, stream) == 0

    def timed(fn, iters=10):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) / iters

    def pair(resid_fn):
        def go():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                panel(side.cuda_stream)
            resid_fn(main.cuda_stream, True)
            main.wait_stream(side)
        return go

    # reference result of the residual alone (window kernel, store mode)
    window(main.cuda_stream, False)
    assert capi.launch_combine_partials(tb, buf.data_ptr(), out.data_ptr(), n, F, False, main.cuda_stream) == 0
    ref = out.clone()
    print(f"window kernel (unit table) alone: {timed(lambda: window(main.cuda_stream, False)):.3f} ms; beside the panel kernel: "
          f"{timed(pair(window)):.3f} ms; panel kernel alone {timed(lambda: panel(main.cuda_stream)):.3f} ms", flush=True)

    deg = (r_indptr[1:] - r_indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    cols = r_indices.long()
    for v in VARIANTS:
        R = v[0]
        lib = ctypes.CDLL(so(v))
        group = rows // R
        key = (group << 40) | (cols << 8) | (rows % R)
        key, _ = torch.sort(key)
        num_groups = (n + R - 1) // R
        stream_ptr = torch.zeros(num_groups + 1, dtype=torch.int64, device=dev)
        stream_ptr[1:] = torch.cumsum(torch.bincount(key >> 40, minlength=num_groups), 0)
        words = ((((key >> 8) & 0xFFFFFFFF) << 6) | (key & 0xFF)).to(torch.int32)
        stream_ptr = stream_ptr.to(torch.int32)
        work = (stream_ptr[1:] - stream_ptr[:-1]).long()
        for label, order in (("natural order", None),
                             ("groups longest first", torch.argsort(work, descending=True, stable=True).to(torch.int32))):
            def es(stream, atomic, order=order):
                rc = lib.edge_stream_launch(ctypes.c_void_p(stream_ptr.data_ptr()), ctypes.c_void_p(words.data_ptr()),
                                            ctypes.c_void_p(order.data_ptr() if order is not None else 0), num_groups, n,
                                            ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(out.data_ptr()), int(atomic),
                                            ctypes.c_void_p(stream))
                assert rc == 0, rc

            out.fill_(float("nan"))
            es(main.cuda_stream, False)
            torch.cuda.synchronize()
            err = float((out - ref).norm() / ref.norm())
            print(f"edge stream rows/wave {v[0]:2d} waves {v[1]:2d} batch {v[2]:2d} [{label:20s}]: rel diff vs window kernel "
                  f"{err:.2e} | alone {timed(lambda: es(main.cuda_stream, False)):.3f} ms | beside the panel kernel "
                  f"{timed(pair(es)):.3f} ms", flush=True)
        del key, words


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
