"""Unit order of the window kernel when F spans several column slabs (round 2): window-major (slabs of a window side by
side) against slab-major (the whole XCD range per slab), fixed tiles, natural window order and unit tables.

    python harness/experiments/exp_slab_order.py [workload ...]

VOLTRIX_SLAB_ORDER is read by the launcher on every call, so both orders run in one process on the same handle.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "voltrix-spmm_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(ROOT, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import capi  # noqa: E402
from voltrix.schedule import unit_table  # noqa: E402

WIDE = {   # EXP_WIDE=1: 256-column slabs (512-byte row pieces) for the wide-feature cases
    "reddit_like": [(512, (128, 3, 4)), (512, (256, 2, 4)), (512, (256, 3, 2)), (512, (256, 3, 1)), (256, (256, 2, 4)), (256, (256, 3, 2))],
    "products_like": [(512, (64, 3, 4)), (512, (128, 3, 4)), (512, (256, 2, 4)), (512, (256, 3, 2)), (512, (256, 3, 1)), (256, (128, 3, 4)),
                      (256, (256, 2, 4)), (256, (256, 3, 2)), (256, (256, 3, 1))],
    "powerlaw_4m": [(256, (128, 3, 4)), (256, (256, 2, 4)), (256, (256, 3, 2)), (256, (256, 3, 1))],
}
CASES = {   # workload -> [(F, tile)]
    "reddit_like": [(64, (32, 4, 4)), (128, (64, 3, 4)), (128, (32, 4, 4)), (256, (128, 3, 4)), (256, (64, 3, 4)),
                    (512, (128, 3, 4)), (512, (128, 4, 4))],
    "products_like": [(64, (32, 4, 4)), (128, (64, 3, 4)), (256, (128, 3, 4)), (512, (128, 3, 4)), (512, (64, 3, 4))],
    "powerlaw_4m": [(256, (128, 3, 4))],
}


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / iters


def main():
    dev = torch.device("cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for name in sys.argv[1:] or ["reddit_like", "products_like"]:
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        h = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
        del indptr, indices
        tb = unit_table(h[0], n)
        print(f"{name}: N={n} nnz={e} units {tb.num_units} cuts {tb.num_cuts}", flush=True)
        for F, tile in (WIDE if os.environ.get("EXP_WIDE") else CASES)[name]:
            feat = torch.randn(n, F, device=dev).half()
            out = torch.empty(n, F, device=dev)
            buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)
            res = {}
            for order in ("minor", "major"):
                os.environ["VOLTRIX_SLAB_ORDER"] = order

                def natural():
                    assert capi.launch_spmm_sched(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, e, F,
                                                  feat.data_ptr(), out.data_ptr(), tile, stream) == 0

                def units():
                    assert capi.launch_spmm_sched(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, e, F,
                                                  feat.data_ptr(), out.data_ptr(), tile, stream, 0, 0, False, False, tb,
                                                  buf.data_ptr()) == 0
                    assert capi.launch_combine_partials(tb, buf.data_ptr(), out.data_ptr(), n, F, False, stream) == 0

                res[order] = (timeit(natural), timeit(units), out.double().sum().item())
            print(f"  F={F:4d} tile {tile} B {n * F * 2 / 1e6:7.0f} MB | natural order: window-major {res['minor'][0]:8.3f} "
                  f"slab-major {res['major'][0]:8.3f} ms | unit table: window-major {res['minor'][1]:8.3f} slab-major "
                  f"{res['major'][1]:8.3f} ms | checksums equal {res['minor'][2] == res['major'][2]}", flush=True)
            del feat, out, buf
        del h, tb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
