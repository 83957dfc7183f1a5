"""A/B of the panel kernel's metadata fetch (once per workgroup = shipped, against the per-wave form of rounds 2-3, built from
commit 39c241b's header into build/panel_meta_old.so) inside the full two-level step at several feature widths; ABAB rounds on
one box.  The step here = zero fill, panel kernel (side stream) beside the window kernel, join, combine pass.
    python harness/experiments/exp_meta_ab_wide.py"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG, ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
os.environ["VOLTRIX_HYBRID"] = "1"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid  # noqa: E402
from voltrix.jit_kernels.spmm import spmm_kernel  # noqa: E402

from exp_panel_parts import time_ms  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    libs = {v: ctypes.CDLL(os.path.join(HERE, "build", f"panel_meta_{v}.so")) for v in ("new", "old")}
    for name in ("reddit_like", "reddit_sbm"):
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"meta_ab_wide/{name}"
        two = voltrix.two_level_of(handle[1])
        plan = two.plan
        plan.parts = None                      # the experiment library launches whole panels in the plan's order
        side = hybrid.side_stream(dev)
        for f in (128, 512, 1024):
            feat = torch.randn(n, f, device=dev).half()
            out = torch.empty(n, f, device=dev)
            voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)

            def step(lib):
                main_stream = torch.cuda.current_stream()
                out.zero_()
                fork = torch.cuda.Event()
                fork.record(main_stream)
                side.wait_event(fork)
                rc = lib.corun_diag_launch(ctypes.c_void_p(plan.panel_ptr.data_ptr()), ctypes.c_void_p(plan.panel_cols.data_ptr()),
                                           ctypes.c_void_p(plan.panel_bits.data_ptr()), ctypes.c_void_p(plan.panel_order.data_ptr()),
                                           n, f, ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(out.data_ptr()), 2,
                                           ctypes.c_void_p(side.cuda_stream))
                assert rc == 0, rc
                join = torch.cuda.Event()
                join.record(side)
                pending = spmm_kernel(two.blk_offsets, two.hspa_packed, two.hind, num_nodes=n, num_edges=plan.num_resid_edges,
                                      embedding_dim=f, input=feat, output=out, atomic_out=True, beside_panel=True,
                                      defer_combine=True, xcd_ptr=two.window_xcd_ptr)
                main_stream.wait_event(join)
                if pending is not None:
                    pending.run()
                return out

            ref = step(libs["new"]).clone()
            same = bool(torch.equal(step(libs["old"]), ref))
            times = {"new": [], "old": []}
            for _ in range(3):
                for v in ("new", "old"):
                    times[v].append(time_ms(lambda: step(libs[v]), reps=5, batch=4))
            print(json.dumps({"graph": name, "F": f, "bits_equal": same, "shared_meta_ms": [round(t, 4) for t in times["new"]],
                              "per_wave_meta_ms": [round(t, 4) for t in times["old"]]}), flush=True)
        del handle, two


if __name__ == "__main__":
    main()
