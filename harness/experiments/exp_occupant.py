"""What does co-residency itself cost the window kernel?  The residual's window kernel (the shipped pair launch, atomic
output) beside a kernel that only OCCUPIES CUs the way the panel kernel does -- 512 threads, 176 VGPRs, 44 KiB of LDS, 0.55 ms
per workgroup, 456 workgroups -- and executes nothing (occupant.hip: it sleeps on the wall clock), and beside smaller
footprints of the same thing.  Against the real pair.
    python harness/experiments/exp_occupant.py build | run"""
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
SO = os.path.join(HERE, "build", "occupant.so")


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(HERE, "occupant.hip"), "-o", SO])


def run():
    import torch

    import synth_graphs
    import voltrix
    from voltrix import hybrid
    from voltrix.jit_kernels.spmm import spmm_kernel

    lib = ctypes.CDLL(SO)
    lib.occupant_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    indptr, indices, cfg = synth_graphs.generate("reddit_like", device=dev)
    n, e, f = indptr.numel() - 1, indices.numel(), cfg["feat"]
    feat = torch.randn(n, f, device=dev).half()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = "occupant"
    two = voltrix.two_level_of(handle[1])
    out = torch.zeros(n, f, device=dev)
    voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    main, side = torch.cuda.current_stream(), hybrid.side_stream(dev)

    def window():
        pending = spmm_kernel(two.blk_offsets, two.hspa_packed, two.hind, num_nodes=n, num_edges=two.plan.num_resid_edges,
                              embedding_dim=f, input=feat, output=out, atomic_out=True, beside_panel=True, defer_combine=True,
                              xcd_ptr=two.window_xcd_ptr)
        return pending

    def timed_pair(other):
        """window kernel on main, `other(stream)` on side; -> (window ms, other ms, both ms), means over 10 runs."""
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

    def occupant(grid, threads, vgprs, lds, us):
        def launch(stream):
            rc = lib.occupant_launch(grid, threads, vgprs, lds, us, ctypes.c_void_p(stream))
            assert rc == 0, rc
        return launch

    def panel(stream):
        hybrid.launch_panel(two.plan, feat, out, accumulate=2, stream=stream)

    print(json.dumps({"what": "window kernel alone", "ms": timed_pair(None)}), flush=True)
    print(json.dumps({"what": "beside the real panel kernel (456 workgroups)", "ms": timed_pair(panel)}), flush=True)
    for label, args in (("occupant 456 x 512 threads, 176 VGPRs, 44 KiB LDS, 0.55 ms each", (456, 512, 176, 45056, 550.0)),
                        ("occupant 456 x 512 threads, 176 VGPRs, no LDS", (456, 512, 176, 0, 550.0)),
                        ("occupant 456 x 512 threads, 32 VGPRs, 44 KiB LDS", (456, 512, 32, 45056, 550.0)),
                        ("occupant 456 x 512 threads, 32 VGPRs, no LDS", (456, 512, 32, 0, 550.0)),
                        ("occupant 456 x 256 threads, 32 VGPRs, no LDS", (456, 256, 32, 0, 550.0)),
                        ("occupant 256 x 512 threads, 176 VGPRs, 44 KiB LDS, 1.0 ms each", (256, 512, 176, 45056, 1000.0)),
                        ("occupant 256 x 512 threads, 32 VGPRs, no LDS, 1.0 ms each", (256, 512, 32, 0, 1000.0)),
                        ("occupant 1824 x 512 threads, 176 VGPRs, 44 KiB LDS, 0.14 ms each", (1824, 512, 176, 45056, 140.0))):
        print(json.dumps({"what": label, "ms [window, occupant, both]": timed_pair(occupant(*args))}), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
