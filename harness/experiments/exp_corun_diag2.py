"""Round 4 rerun of the co-run diagnostics on today's kernels: the residual's window kernel (shipped pair launch, atomic output)
beside builds of the panel kernel with parts REMOVED (VOLTRIX_PANEL_DIAG bits: 1 MFMAs, 2 row DMAs, 4 barrier, 8 fragment
reads, 16 metadata DMAs of the loop; results wrong by design).  exp_occupant.py showed that occupancy alone is free; which of
the panel kernel's activities is it that costs the window kernel 0.34 ms?
    python harness/experiments/exp_corun_diag2.py build | run"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG, ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
os.environ["VOLTRIX_HYBRID"] = "1"
# bit 16 (no metadata DMAs) ONLY together with bit 2 (no row DMAs): the row gathers take their row ids from the metadata slots,
# and without the metadata DMAs those are whatever LDS held -- a variant with 16 alone faulted (out-of-bounds gathers)
VARIANTS = {"full": 0, "no_mfma": 1, "no_rows": 2, "no_frag_reads": 8, "no_mfma_no_frag_reads": 9,
            "no_rows_no_meta": 18, "no_mfma_no_rows": 3, "no_mfma_no_rows_no_meta": 19, "only_loop_control": 31}
# (The metadata-once-per-workgroup form was A/B'd against the per-wave form through this harness before the latter was deleted:
# profiles/r04/experiment_meta_ab.log.)
if os.environ.get("EXP_ONLY"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["EXP_ONLY"].split(",")}


def so(name):
    return os.path.join(HERE, "build", f"corun2_{name}.so")


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-DVOLTRIX_EXPERIMENTAL", f"-DVOLTRIX_PANEL_DIAG={bits}", "-DVOLTRIX_PANEL_SLEEP=0",
                               f"-I{PKG}/voltrix/include", f"-I{REPO}/include", os.path.join(HERE, "corun_diag.hip"), "-o", so(name)])
             for name, bits in VARIANTS.items()]
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch

    import synth_graphs
    import voltrix
    from voltrix import hybrid
    from voltrix.jit_kernels.spmm import spmm_kernel

    dev = torch.device("cuda", 0)
    indptr, indices, cfg = synth_graphs.generate("reddit_like", device=dev)
    n, e, f = indptr.numel() - 1, indices.numel(), cfg["feat"]
    feat = torch.randn(n, f, device=dev).half()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = "corun2"
    two = voltrix.two_level_of(handle[1])
    plan = two.plan
    out = torch.zeros(n, f, device=dev)
    voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    main, side = torch.cuda.current_stream(), hybrid.side_stream(dev)

    def window():
        spmm_kernel(two.blk_offsets, two.hspa_packed, two.hind, num_nodes=n, num_edges=plan.num_resid_edges, embedding_dim=f,
                    input=feat, output=out, atomic_out=True, beside_panel=True, defer_combine=True, xcd_ptr=two.window_xcd_ptr)

    def timed(other):
        acc = [0.0, 0.0, 0.0]
        for _ in range(12):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            torch.cuda.synchronize()
            ev[4].record(main)
            side.wait_event(ev[4])
            if other is not None:
                ev[2].record(side)
                other(side.cuda_stream)
                ev[3].record(side)
            ev[0].record(main)
            window()
            ev[1].record(main)
            main.wait_stream(side)
            ev[5].record(main)
            torch.cuda.synchronize()
            acc[0] += ev[0].elapsed_time(ev[1])
            acc[1] += ev[2].elapsed_time(ev[3]) if other is not None else 0.0
            acc[2] += ev[4].elapsed_time(ev[5])
        return [round(a / 12, 4) for a in acc]

    print(json.dumps({"panel": "none", "ms [window, panel, both]": timed(None)}), flush=True)
    for name in VARIANTS:
        lib = ctypes.CDLL(so(name))

        def panel(stream, lib=lib):
            rc = lib.corun_diag_launch(ctypes.c_void_p(plan.panel_ptr.data_ptr()), ctypes.c_void_p(plan.panel_cols.data_ptr()),
                                       ctypes.c_void_p(plan.panel_bits.data_ptr()), ctypes.c_void_p(plan.panel_order.data_ptr()), n, f,
                                       ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(out.data_ptr()), 2, ctypes.c_void_p(stream))
            assert rc == 0, rc

        alone = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record(side)
            panel(side.cuda_stream)
            b.record(side)
            torch.cuda.synchronize()
            alone.append(a.elapsed_time(b))
        print(json.dumps({"panel": name, "panel_alone_ms": round(sorted(alone)[2], 4), "ms [window, panel, both]": timed(panel)}), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
