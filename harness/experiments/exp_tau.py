"""Two-level step: threshold (tau) x unit length x panel order sweep with the atomic join (round 2).

    python harness/experiments/exp_tau.py [workload] [F]
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
dev = torch.device("cuda")
indptr, indices, cfg = synth_graphs.generate(name, device=dev)
n, e = indptr.numel() - 1, indices.numel()
feat = torch.randn(n, F, device=dev).half()
main = torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
stream = main.cuda_stream
out = torch.empty(n, F, device=dev)
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


ref = None
for waves, rb in ((8, 4), (8, 2)):
    for tau in (3, 4, 5, 6, 8, 12):
        r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, waves, rb, tau)
        if plan.num_ksteps == 0:
            continue
        resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
        rn = r_indices.numel()
        nst = ((resid[0][1:] - resid[0][:-1]) + 3) // 4
        med = max(1, int(nst.float().median()))
        # panel order: longest panel first (LPT) vs natural
        nks = (plan.panel_ptr[1:] - plan.panel_ptr[:-1]).long()
        npan = nks.numel()
        ppx = (npan + 7) // 8
        key = (torch.arange(npan, device=dev) // ppx) * (int(nks.max()) + 1) + (int(nks.max()) - nks)
        lpt = torch.argsort(key, stable=True).to(torch.int32)
        ptile = (128, 3, 1) if rb == 4 else (128, 4, 1)
        res = []
        for order_name, porder in (("natural", None), ("longest-first", lpt)):
            plan.panel_order = porder
            for mult in (1.5, 2.0):
                tb = unit_table(resid[0], n, max(8, int(mult * med)))
                buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)

                def step():
                    out.zero_()
                    fork = torch.cuda.Event()
                    fork.record(main)
                    side.wait_event(fork)
                    rc = capi.launch_spmm_panel(plan, feat.data_ptr(), out.data_ptr(), F, 2, False, ptile, 0, side.cuda_stream)
                    assert rc == 0
                    join = torch.cuda.Event()
                    join.record(side)
                    rc = capi.launch_spmm_sched(resid[0].data_ptr(), resid[1].data_ptr(), resid[2].data_ptr(), n, rn, F,
                                                feat.data_ptr(), out.data_ptr(), (128, 3, 4), stream, 0, 0, True, False, tb,
                                                buf.data_ptr())
                    assert rc == 0
                    main.wait_event(join)
                    if tb.num_cuts:
                        assert capi.launch_combine_partials(tb, buf.data_ptr(), out.data_ptr(), n, F, True, stream) == 0

                ms = timeit(step)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).norm() / ref.norm())
                assert err < 1e-6, err
                res.append(f"{order_name} L={mult}x: {ms:.3f}")
        plan.panel_order = None
        print(f"  panel {waves * rb * 16} rows, tau {tau}: shared {plan.num_shared_edges / e:.1%}, k-steps {plan.num_ksteps}, "
              f"residual median {med} stages | " + " | ".join(res) + " ms", flush=True)
