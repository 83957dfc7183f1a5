#!/usr/bin/env python3
"""The reference's own evaluation set in ONE process: 12 graphs (bench/plot.py:8) x feature widths (bench_all.py:21: 256, 512,
1024; plus north_star's 32 and 128) x methods {hipSPARSE, Voltrix (fp32 inputs, as the reference feeds them), Voltrix-fp16}.

    python harness/eval_set.py [--datasets amazon0505,DD,...] [--feature_dims 32,128,256,512,1024]
                               [--output_file results.csv] [--jsonl eval.jsonl] [--scale 1.0] [--reorder]

``harness/bench_all.py`` is the file-based form of the same sweep (one process per cell, as bench/bench_all.py:62-172); this
driver generates every stand-in once (``synth_graphs.EVALUATION_SET``: datasets.zip is unreachable offline), preprocesses it
once and loops over widths and methods in-process -- 12 x 5 x 3 cells in minutes instead of an hour of process start-ups.
Two timings per cell:
  * ``Time (ms)`` of results.csv = the reference's protocol (bm_voltrix.py:36: ``GPU_bench(spmm, iters=10, warmup=10,
    kernel_name="spmm")`` -- every call after a cache flush, its own event pair), in the reference's CSV schema
    (bench_all.py:75);
  * the jsonl line adds the steady-state time (median of batches of back-to-back calls: bench.py's protocol), the HBM-roofline
    fraction of the algorithmic bytes 4 (nnz + N + 1) + N F (s_in + 4), launches per call, the first call's
    wall time, the handle's TC blocks / stages (= gathered bytes) and the tile the tuner picked.
``--reorder`` adds Reorder=S and Reorder=Y rows (round 6): S = the graph with its labels SHUFFLED (seeded random P A P^T), Y = the
library's symmetric relabelling of THAT (csr_preprocess_reordered(shuffled, method="auto", relabel=True)), features permuted
outside the timed call, as bench_all.py:120-129 runs Voltrix on ``<name>.reorder.npz``; N = the stand-in as generated.
Bench infrastructure, not part of the product.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "voltrix-spmm_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402
from voltrix.jit_kernels.spmm import feature_hash  # noqa: E402
from voltrix.utils import GPU_bench, KernelTimer, calc_diff  # noqa: E402

HBM_PEAK = 8.0e12


def steady_ms(fn, iters=7, warm=3, batch=10):
    for _ in range(warm):
        fn()
    times = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(batch):
            fn()
        e.record()
        e.synchronize()
        times.append(s.elapsed_time(e) / batch)
    return sorted(times)[len(times) // 2]


def tuned_point(hspa_packed, f, two_level, dev):
    keys = {"feature_hash": feature_hash(hspa_packed), "embedding_dim": f, "dtype": str(torch.float16),
            "device": torch.cuda.get_device_name(dev), "two_level": bool(two_level), "weighted": False}
    return dict(jit_tuner.tuned_point("spmm_kernel", keys))


def launches_per_call(fn):
    with KernelTimer() as timer:
        fn()
    got = timer.summary()
    return {k: v[0] for k, v in got.items()}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", default=",".join(synth_graphs.EVALUATION_SET))
    ap.add_argument("--feature_dims", default="32,128,256,512,1024")
    ap.add_argument("--methods", default="hipSPARSE,rocSPARSE-best,rocSPARSE-best-fp16,Voltrix,Voltrix-fp16")
    ap.add_argument("--output_file", default="results.csv")
    ap.add_argument("--jsonl", default="eval_set.jsonl")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--reorder", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare every Voltrix cell's output with hipSPARSE's (calc_diff)")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    dims = [int(d) for d in args.feature_dims.split(",")]
    methods = args.methods.split(",")
    with open(args.output_file, "w") as f:
        f.write("Method,Dataset,FeatDim,Reorder,Time (ms)\n")
    jl = open(args.jsonl, "w")

    def record(method, name, dim, mark, ms, extra):
        with open(args.output_file, "a") as f:
            f.write(f"{method},{name},{dim},{mark},{ms:.4f}\n")
        line = dict(method=method, dataset=name, feat=dim, reorder=mark, ref_protocol_ms=ms, **extra)
        jl.write(json.dumps(line) + "\n")
        jl.flush()
        rf = extra.get("hbm_roofline_frac")
        print(f"{method:13s} {name:14s} F={dim:<5d} reorder={mark} flush {ms:9.4f} ms  steady {extra.get('steady_ms', float('nan')):9.4f} ms"
              + (f"  roofline {100 * rf:5.1f} %" if rf is not None else ""), flush=True)

    for name in args.datasets.split(","):
        standin = synth_graphs.EVALUATION_SET.get(name, name)
        indptr, indices, _ = synth_graphs.generate(standin, device=dev, scale=args.scale)
        n, nnz = indptr.numel() - 1, indices.numel()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - t0) * 1e3
        handle[1].hash_tag = f"eval/{standin}/{args.scale}"
        blocks = int(handle[0][-1])
        nblk = (handle[0][1:] - handle[0][:-1])
        stages = int(((nblk + 3) // 4).sum())
        two = voltrix.two_level_of(handle[1])
        variants = [("N", handle, None)]
        if args.reorder:
            # round 6: the Reorder rows start from SHUFFLED labels -- P A P^T for a seeded random P, what a
            # dataset looks like before anybody reordered it.  "S" = that graph as it is, "Y" = the library's own symmetric
            # reorder of it (bench_all.py:120-149 times Voltrix on <name>.reorder.npz of every dataset); "N" = the stand-in in its
            # generating order (what a well-reordered file looks like).
            s_indptr, s_indices, _ = synth_graphs.shuffle_labels(indptr, indices, 4242)
            sh = voltrix.csr_preprocess_device(s_indptr, s_indices, n)
            sh[1].hash_tag = f"eval/{standin}/{args.scale}/shuffled"
            variants.append(("S", sh, None))
            reorder_info = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rh = voltrix.csr_preprocess_reordered(s_indptr, s_indices, n, method="auto", relabel=True, info=reorder_info)
            torch.cuda.synchronize()
            reorder_ms = (time.perf_counter() - t0) * 1e3
            rh.hspa_packed.hash_tag = f"eval/{standin}/{args.scale}/reordered"
            variants.append(("Y", rh, reorder_ms))
            del s_indptr, s_indices
        csr = None
        if "hipSPARSE" in methods or args.check:
            csr = torch.sparse_csr_tensor(indptr, indices, torch.ones(nnz, device=dev), size=(n, n))
        for dim in dims:
            torch.manual_seed(20)
            feat32 = torch.randn(n, dim, device=dev)
            base = None
            if csr is not None:
                run = lambda: csr @ feat32  # noqa: E731
                base = run()
                if "hipSPARSE" in methods:
                    ms = GPU_bench(run, iters=args.iters, warmup=3, kernel_name="spmm")
                    record("hipSPARSE", name, dim, "N", ms, dict(num_nodes=n, nnz=nnz, steady_ms=steady_ms(run, iters=3, batch=3)))
            # round 6: rocSPARSE's generic SpMM, the best of its four CSR algorithms per cell (buffer size + preprocess stages
            # outside the timed loop), fp32 and fp16-in / fp32-compute, and a plain CSR row-gather kernel beside them
            for method in [m for m in methods if m.startswith("rocSPARSE-best")]:
                from harness import bm_rocsparse

                operand = feat32.half() if method.endswith("fp16") else feat32
                suffix = "-fp16" if method.endswith("fp16") else ""
                got = torch.empty(n, dim, device=dev)
                detail = {}
                flushed = bm_rocsparse.baselines(indptr, indices, n, operand, flush=True, iters=args.iters, out=got, details=detail,
                                                  reference=base if args.check else None)
                steady = bm_rocsparse.baselines(indptr, indices, n, operand, flush=False, iters=args.iters)
                name_f, ms_f = bm_rocsparse.best(flushed)
                name_s, ms_s = bm_rocsparse.best(steady)
                extra = dict(num_nodes=n, nnz=nnz, steady_ms=ms_s, best_algorithm=name_s, best_algorithm_flushed=name_f,
                             flushed=flushed, steady=steady, preprocess={k: v for k, v in detail.items()})
                if args.check and base is not None:
                    extra["calc_diff_vs_hipsparse"] = max((v.get("calc_diff", 0.0) for v in detail.values()), default=None)
                if ms_f is not None:
                    record(method, name, dim, "N", ms_f, extra)
                name_g, ms_g = bm_rocsparse.best(flushed, prefix="csr_row_gather")
                if ms_g is not None:
                    record("CSR-gather" + suffix, name, dim, "N", ms_g,
                           dict(num_nodes=n, nnz=nnz, steady_ms=bm_rocsparse.best(steady, prefix="csr_row_gather")[1]))
                del got, operand
            for method in [m for m in methods if m.startswith("Voltrix")]:
                feat = feat32.half() if method == "Voltrix-fp16" else feat32
                for mark, h, reorder_ms in variants:
                    if mark == "Y":
                        fin = voltrix.permute_features(h, feat)
                        call = lambda: voltrix.spmm_reordered(h, fin)  # noqa: E731  (B and C in the new order)
                    elif mark == "S":
                        call = lambda: voltrix.spmm(*h, num_nodes=n, num_edges=nnz, feat=feat)  # noqa: E731
                    else:
                        call = lambda: voltrix.spmm(*h, num_nodes=n, num_edges=nnz, feat=feat)  # noqa: E731
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = call()
                    torch.cuda.synchronize()
                    first_ms = (time.perf_counter() - t0) * 1e3
                    ms = GPU_bench(call, iters=args.iters, warmup=10, kernel_name="spmm")
                    st = steady_ms(call)
                    s_in = feat.element_size()
                    alg = synth_graphs.algorithmic_bytes(n, nnz, dim, s_in)
                    if mark != "N":      # the variant's own handle: TC blocks / stages of the shuffled or reordered graph
                        offs = h.blk_offsets if mark == "Y" else h[0]
                        v_blocks, v_stages = int(offs[-1]), int((((offs[1:] - offs[:-1]) + 3) // 4).sum())
                    else:
                        v_blocks, v_stages = blocks, stages
                    extra = dict(num_nodes=n, nnz=nnz, steady_ms=st, first_call_ms=first_ms, preprocess_ms=prep_ms,
                                 algorithmic_bytes=alg, hbm_roofline_frac=alg / (st * 1e-3) / HBM_PEAK,
                                 hbm_roofline_frac_flushed=alg / (ms * 1e-3) / HBM_PEAK, gflops=2.0 * nnz * dim / st / 1e6,
                                 tc_blocks=v_blocks, stages=v_stages, gathered_bytes=v_stages * 32 * dim * 2,
                                 two_level=two is not None and mark == "N", launches=launches_per_call(call),
                                 tile=tuned_point(two.hspa_packed if (two is not None and mark == "N") else h[1],
                                                  (dim + 7) // 8 * 8, two is not None and mark == "N", dev) if mark == "N" else None,
                                 tuner=dict(jit_tuner.stats))
                    from voltrix import sidecar as _sidecar

                    csr_sc = _sidecar.lookup_csr(h.hspa_packed if mark == "Y" else h[1])
                    extra["path"] = None if csr_sc is None else csr_sc.choice.get((int(feat.shape[1]), str(feat.dtype)))   # "csr" | "block"
                    if reorder_ms is not None:
                        extra["reorder_ms"] = reorder_ms
                        extra["reorder_picked"] = h.method
                        extra["reorder_estimates"] = {k: round(v.get("estimated_ms", 0.0), 4) for k, v in (reorder_info.get("report") or {}).items()}
                    if args.check and base is not None and mark == "N":
                        extra["calc_diff_vs_hipsparse"] = float(calc_diff(out, base))
                    record(method, name, dim, mark, ms, extra)
                    del out
            del feat32, base
        del handle, csr, indptr, indices, variants
        torch.cuda.empty_cache()
    jl.close()
    print(f"results -> {os.path.abspath(args.output_file)}, {os.path.abspath(args.jsonl)}")


if __name__ == "__main__":
    main()
