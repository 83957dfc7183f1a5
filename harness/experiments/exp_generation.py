"""Bounded-length units for the window kernel, and the two-level step without a join pass.

    python harness/experiments/exp_generation.py [workload] [F] [quick]

Times, on one graph: (A) the window format alone -- balance schedule (chunk 512 / 2048) vs unit tables (long windows cut
into interleaved units of <= L stages, listed longest first per XCD range or per chunk of consecutive windows; partial
tiles of cut windows summed in unit order by combine_partials); (B) the same on the residual handle of the two-level
format; (C) the two-level step end to end: two streams + add pass (round 1) vs float-atomic epilogues onto a zero-filled
output (no second buffer, no add pass), with the balance schedule and with unit tables.  Every variant is checked against
the classic result (bit-equal where the summation order is kept, 1e-6 norm-wise where windows are cut; cut results are
checked to be run-to-run identical).

(Round 2 also measured persistent launches -- every wave walking a static list of windows, generations of equal length,
long and short alternating per wave: slower than the hardware's dynamic dispatch in every configuration, 1.67 vs 1.41 ms on
the residual; profiles/r02/experiment_generation_persistent.log.)
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
from voltrix import capi, hybrid  # noqa: E402
from voltrix.schedule import unit_table  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
quick = len(sys.argv) > 3
dev = torch.device("cuda")
indptr, indices, cfg = synth_graphs.generate(name, device=dev)
n, e = indptr.numel() - 1, indices.numel()
feat = torch.randn(n, F, device=dev).half()
stream = torch.cuda.current_stream().cuda_stream
W = (n + 15) // 16
print(f"{name}: N={n} nnz={e} F={F}", flush=True)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / iters


def balance_order(handle, chunk):
    o = torch.empty(W, dtype=torch.int32, device=dev)
    capi.launch_window_order(handle[0], n, o, stream, chunk)
    return o


class Partials:
    def __init__(self, table):
        self.table = table
        self.buf = torch.empty(max(1, table.num_slots) * 16 * F, dtype=torch.float32, device=dev)


def launch(handle, nnz, out, tile, order=None, atomic=False, part=None):
    """Window kernel (+ the combine pass of a unit table when `atomic` is False; the two-level step runs it after the join)."""
    rc = capi.launch_spmm_sched(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, nnz, F,
                                feat.data_ptr(), out.data_ptr(), tile, stream, order.data_ptr() if order is not None else 0,
                                0, atomic, False, part.table if part is not None else None,
                                part.buf.data_ptr() if part is not None else 0)
    assert rc == 0, rc
    if part is not None and not atomic and part.table.num_cuts:
        rc = capi.launch_combine_partials(part.table, part.buf.data_ptr(), out.data_ptr(), n, F, False, stream)
        assert rc == 0, rc


TILES = [(128, 3, 4), (64, 4, 4)] if quick else [(128, 3, 4), (128, 2, 4), (128, 4, 4), (64, 3, 4), (64, 4, 4), (64, 3, 8)]


def rel(a, b):
    return float((a - b).norm() / b.norm())


def sweep(tag, handle, nnz):
    ref = torch.empty(n, F, device=dev)
    launch(handle, nnz, ref, (128, 3, 4))
    out = torch.empty(n, F, device=dev)
    orders = {c: balance_order(handle, c) for c in (512, 2048)}
    nst = ((handle[0][1:] - handle[0][:-1]) + 3) // 4
    med, mx = int(nst.float().median()), int(nst.max())
    print(f"  [{tag}] stages per window: median {med}, mean {float(nst.float().mean()):.0f}, max {mx}", flush=True)
    parts = {}
    for L in sorted({med, int(med * 1.25), int(med * 1.5), 2 * med}):
        for chunk in (None, 512):
            parts[(L, chunk)] = Partials(unit_table(handle[0], n, L, chunk))
    for tile in TILES:
        res = []
        for c, o in orders.items():
            res.append(f"balance chunk {c}: {timeit(lambda: launch(handle, nnz, out, tile, o)):.3f}")
        assert torch.equal(out, ref)
        print(f"  [{tag}] tile {tile}: " + " | ".join(res) + " ms", flush=True)
        res = []
        for (L, chunk), part in parts.items():
            out.fill_(float("nan"))
            ms = timeit(lambda: launch(handle, nnz, out, tile, None, False, part))
            err = rel(out, ref)
            again = out.clone()
            launch(handle, nnz, out, tile, None, False, part)
            assert err < 1e-6 and torch.equal(out, again), (err, "not reproducible")
            tb = part.table
            res.append(f"L={L} chunk={chunk} ({tb.num_units} units, {tb.num_cuts} cut): {ms:.3f}")
        print(f"  [{tag}] tile {tile} unit tables + combine: " + " | ".join(res) + " ms", flush=True)


full = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
print(f"(A) window format, T={int(full[0][-1])} TC blocks")
sweep("window", full, e)

r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
rn = r_indices.numel()
print(f"(B) residual of the two-level format: {rn} edges ({rn / e:.1%}), T={int(resid[0][-1])}, panel k-steps {plan.num_ksteps}")
if plan.num_ksteps:
    sweep("residual", resid, rn)

    print("(C) two-level step end to end")
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    out = torch.empty(n, F, device=dev)
    shared = torch.empty(n, F, device=dev)
    ptile = (128, 3, 1) if F >= 128 else ((64, 6, 2) if F > 32 else (32, 6, 2))

    def two_streams_add(tile, order):
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        hybrid.launch_panel(plan, feat, shared, accumulate=False, tile=ptile, stream=side.cuda_stream)
        join = torch.cuda.Event()
        join.record(side)
        launch(resid, rn, out, tile, order)
        main.wait_event(join)
        capi.launch_add_inplace_f32(out, shared, stream)

    def atomic_step(tile, order, part=None):
        out.zero_()
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        rc = capi.launch_spmm_panel(plan, feat.data_ptr(), out.data_ptr(), F, 2, False, ptile, 0, side.cuda_stream)
        assert rc == 0
        join = torch.cuda.Event()
        join.record(side)
        launch(resid, rn, out, tile, order, True, part)
        main.wait_event(join)
        if part is not None and part.table.num_cuts:
            rc = capi.launch_combine_partials(part.table, part.buf.data_ptr(), out.data_ptr(), n, F, True, stream)
            assert rc == 0

    def panel_only():
        hybrid.launch_panel(plan, feat, shared, accumulate=False, tile=ptile, stream=stream)

    print(f"  panel kernel alone: {timeit(panel_only):.3f} ms", flush=True)
    o512 = balance_order(resid, 512)
    nst = ((resid[0][1:] - resid[0][:-1]) + 3) // 4
    med = int(nst.float().median())
    two_streams_add((128, 3, 4), o512)
    torch.cuda.synchronize()
    ref = out.clone()
    print(f"  two streams + add pass, (128,3,4) chunk 512: {timeit(lambda: two_streams_add((128, 3, 4), o512)):.3f} ms", flush=True)
    for tile in ((128, 3, 4), (128, 4, 4), (64, 4, 4)):
        ms = timeit(lambda: atomic_step(tile, o512))
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        res = []
        for L in (med, int(1.25 * med), int(1.5 * med), 2 * med):
            part = Partials(unit_table(resid[0], n, L))
            ms3 = timeit(lambda: atomic_step(tile, None, part))
            torch.cuda.synchronize()
            err = rel(out, ref)
            again = out.clone()
            atomic_step(tile, None, part)
            torch.cuda.synchronize()
            res.append(f"units L={L}: {ms3:.3f} (rel {err:.1e}, reproducible {torch.equal(out, again)})")
        print(f"  atomic epilogues, tile {tile}: balance chunk 512 {ms:.3f} ms (bit-equal {same}) | " + " | ".join(res),
              flush=True)
