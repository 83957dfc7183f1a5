"""Round 6: the column-sliced residual of the two-level step, measured (DESIGN section 7 had it closed by arithmetic only).

The residual half of the headline graph gathers one 256-byte row of B per edge from uniformly random columns: every XCD's 4 MiB L2
sees all 60 MB of B (7 % hits on that half), and the step runs at the fabric's rate in L2-miss bytes.  Slice the residual's COLUMNS
into S ranges and give every range to 8 / S XCDs: an XCD then gathers from 60 / S MB of B only.  The price: an output row takes one
float-atomic contribution per slice instead of one.

No new kernel: the S slices are stacked as one (S x N') x N matrix (rows of slice s = rows of A restricted to the slice's columns),
preprocessed into ONE reference handle, and run by the shipped window kernel with a row map (stacked row -> row of C), atomic
epilogues and XCD ranges that put slice s on its XCDs (unit-table schedules take the ranges as given).

    python exp_column_sliced_residual.py [workload] [F]   -> residual alone and the pair (panel kernel beside it), S = 1 (shipped), 2, 4, 8
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid  # noqa: E402
from voltrix.jit_kernels import csr_fused_preprocess_kernel, jit_tuner, spmm_kernel  # noqa: E402


if os.getenv("EXP_UNIT_TABLES_ONLY"):      # only the schedules that take the XCD ranges as given (unit table, paired units)
    import voltrix.jit_kernels.spmm as _js

    _tile_space = _js.tile_space
    _js.tile_space = lambda *a, **k: [p for p in _tile_space(*a, **k) if p["SCHED"] in (4, 5)]


def ms(fn, reps=5, inner=5):
    for _ in range(3):
        fn()
    t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        e.synchronize()
        t.append(s.elapsed_time(e) / inner)
    return sorted(t)[len(t) // 2]


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    num_feats = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    indptr, indices, _ = synth_graphs.generate(workload, device="cuda")
    n, e = indptr.numel() - 1, indices.numel()
    feat = torch.randn(n, num_feats, device="cuda").half()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = f"exp_sliced/{workload}/whole"
    two = voltrix.two_level_of(handle[1])
    assert two is not None, "the workload has no two-level side-car"
    ref = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    shipped = ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat))
    print(f"{workload} N={n} nnz={e} F={num_feats}: shipped two-level step {shipped:.4f} ms; residual edges {two.plan.num_resid_edges}, "
          f"residual TC blocks {int(two.blk_offsets[-1])}", flush=True)

    resid_indptr, resid_indices, _plan = hybrid.build_panel_plan(indptr, indices, n, n, two.plan.waves, two.plan.row_blocks, two.plan.tau,
                                                                 min_share=0.0)
    del _plan
    deg = (resid_indptr[1:] - resid_indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), deg)
    cols = resid_indices.long()
    e_r = int(cols.numel())
    np16 = (n + 15) // 16 * 16
    wp = np16 // 16
    resid_only = torch.zeros(n, num_feats, device="cuda")

    for slices in (1, 2, 4, 8):
        slice_w = -(-n // slices)
        new_row = (cols // slice_w) * np16 + rows
        order = torch.argsort(new_row * n + cols)
        st_cols = cols[order].to(torch.int32)
        st_indptr = torch.zeros(slices * np16 + 1, dtype=torch.int64, device="cuda")
        st_indptr[1:] = torch.bincount(new_row, minlength=slices * np16).cumsum(0)
        st_indptr = st_indptr.to(torch.int32)
        del order, new_row
        p1, packed, hind, _ = csr_fused_preprocess_kernel(st_indptr, st_cols, slices * np16, num_cols=n)
        packed.hash_tag = f"exp_sliced/{workload}/S{slices}" + ("/units" if os.getenv("EXP_UNIT_TABLES_ONLY") else "")
        row_map = torch.arange(slices * np16, device="cuda", dtype=torch.int64) % np16
        row_map[row_map >= n] = -1
        row_map = row_map.to(torch.int32)
        # XCD x owns windows [x * S * wp / 8, (x + 1) * S * wp / 8): slice s = XCDs [8 s / S, 8 (s + 1) / S)
        xcd_ptr = torch.tensor([x * slices * wp // 8 for x in range(9)], dtype=torch.int32, device="cuda")
        out_big = torch.empty(slices * np16, num_feats, device="cuda")      # only rows [0, n) are ever written (row map)
        out = out_big[:n]

        def run_window(atomic, beside=False):
            return spmm_kernel(p1, packed, hind, num_nodes=slices * np16, num_edges=e_r, embedding_dim=num_feats, input=feat,
                               output=out_big, atomic_out=True, beside_panel=beside, defer_combine=True, row_map=row_map, xcd_ptr=xcd_ptr)

        def residual_alone():
            out.zero_()
            pending = run_window(True)
            if pending is not None:
                pending.run()

        residual_alone()
        if slices == 1:
            resid_only.copy_(out)
        err = float((out - resid_only).abs().max() / resid_only.abs().max())
        t_alone = ms(residual_alone)
        point = [dict(p) for p in jit_tuner.tuned_keys.values()][-1]

        def pair():
            hybrid.run_two_level(two.plan, feat, out, lambda atomic: run_window(atomic, True), concurrent=True)

        pair()
        err_pair = float((out - ref).abs().max() / ref.abs().max())
        t_pair = ms(pair)
        print(f"S={slices}: TC blocks {int(p1[-1])}, residual alone {t_alone:.4f} ms (rel diff to S=1 {err:.2e}), pair {t_pair:.4f} ms "
              f"(rel diff to the shipped step {err_pair:.2e}); tile {point}", flush=True)
        del p1, packed, hind, out_big, out, row_map, st_cols, st_indptr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
