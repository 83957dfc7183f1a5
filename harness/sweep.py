#!/usr/bin/env python3
"""1-GPU sweep over the BASELINE.json workloads x feature widths (north_star: "GFLOP/s on ... graphs at
feat_dim in {32,128,512} reported at 1 GPU ... alongside the CPU torch.sparse.mm baseline").

    python harness/sweep.py [--workloads reddit_like,products_like] [--feats 32,128,512] [--cpu] [--vendor] > out.jsonl

One JSON line per (workload, F), all through the OPERATOR (``voltrix.spmm`` on a ``csr_preprocess_device`` handle; the
JIT tuner picks tile and schedule on the first call): the window format (VOLTRIX_HYBRID=0), the two-level side-car where
``csr_preprocess`` would attach one (VOLTRIX_HYBRID=1 with the default share threshold), and for reference the round-1
launch (tile (128|64|32, 3, 4), balance schedule chunk 512, no unit table).  ms = median over 7 batches of 5 back-to-back
calls after 3 warm-ups (HIP events); GFLOP/s = 2 nnz F / t; algorithmic GB/s and its fraction of 8 TB/s; gathered-row
TB/s; optional CPU (torch.sparse.mm, all host threads, fp32) and GPU-vendor (hipSPARSE through torch.sparse.mm, fp32)
baselines.  Bench infrastructure; the headline line the driver consumes is bench.py's.
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
from voltrix import capi  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402
from voltrix.jit_kernels.spmm import feature_hash  # noqa: E402


def median_ms(fn, iters=7, warm=3, batch=5):
    """Median over `iters` batches of `batch` back-to-back launches (one HIP event pair per batch): the bench.py
    protocol.  Synchronising the host after every single launch lets the clocks drop between launches and reads
    10-15 % high on MI355X."""
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


def tuned_point(hspa_packed, f, beside_panel, dev):
    keys = {"feature_hash": feature_hash(hspa_packed), "embedding_dim": f, "dtype": str(torch.float16),
            "device": torch.cuda.get_device_name(dev), "two_level": bool(beside_panel), "weighted": False}
    return dict(jit_tuner.tuned_point("spmm_kernel", keys))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="reddit_like,reddit_uniform,reddit_sbm,products_like")
    ap.add_argument("--feats", default="32,128,512")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--vendor", action="store_true")
    ap.add_argument("--tune", default="default", choices=["default", "full", "none"])
    args = ap.parse_args()
    os.environ["VOLTRIX_TUNE_SPACE"] = args.tune
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for name in args.workloads.split(","):
        indptr, indices, cfg = synth_graphs.generate(name, device=dev, scale=args.scale)
        n, nnz = indptr.numel() - 1, indices.numel()
        os.environ["VOLTRIX_HYBRID"] = "1"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - t0) * 1e3
        handle[1].hash_tag = f"sweep/{name}/{args.scale}"
        two = voltrix.two_level_of(handle[1])
        total_blocks = int(handle[0][-1])
        order = torch.empty((n + 15) // 16, dtype=torch.int32, device=dev)
        capi.launch_window_order(handle[0], n, order, stream, 512)
        for f in [int(x) for x in args.feats.split(",")]:
            feat = torch.randn(n, f, device=dev).half()
            out = torch.empty(n, f, device=dev)
            os.environ["VOLTRIX_HYBRID"] = "0"
            ms = median_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat))
            alg = synth_graphs.algorithmic_bytes(n, nnz, f, 2)
            line = {"workload": name, "scale": args.scale, "num_nodes": n, "nnz": nnz, "feat": f, "dtype": "f16",
                    "tc_blocks": total_blocks, "preprocess_ms_with_side_car_attempt": prep_ms,
                    "window_format": {"ms": ms, "choice": tuned_point(handle[1], f, False, dev),
                                      "gflops": 2.0 * nnz * f / ms / 1e6, "algorithmic_gbs": alg / ms / 1e6,
                                      "hbm_roofline_frac": alg / ms / 1e6 / 8000.0,
                                      "gather_tbs": 8.0 * total_blocks * f * 2 / ms / 1e9}}
            fs = 32 if f <= 32 else (64 if f <= 64 else 128)

            def round1():
                rc = capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, nnz, f,
                                      feat.data_ptr(), out.data_ptr(), True, (fs, 4 if fs == 32 else 3, 4), stream,
                                      order.data_ptr())
                assert rc == 0

            line["round1_launch_ms"] = median_ms(round1)
            if two is not None:
                os.environ["VOLTRIX_HYBRID"] = "1"
                ms2 = median_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat))
                line["two_level"] = {"ms": ms2, "choice": tuned_point(two.hspa_packed, f, True, dev),
                                     "gflops": 2.0 * nnz * f / ms2 / 1e6, "algorithmic_gbs": alg / ms2 / 1e6,
                                     "hbm_roofline_frac": alg / ms2 / 1e6 / 8000.0,
                                     "shared_edge_fraction": two.plan.num_shared_edges / max(1, nnz),
                                     "panel_ksteps": two.plan.num_ksteps, "tau": two.plan.tau}
            else:
                line["two_level"] = None   # fewer than VOLTRIX_HYBRID_MIN_SHARE of the edges in shared columns
            best = min(ms, line["two_level"]["ms"]) if line["two_level"] else ms
            line["best_ms"] = best
            line["best_hbm_roofline_frac"] = alg / best / 1e6 / 8000.0
            if args.vendor:
                try:
                    a = torch.sparse_csr_tensor(indptr, indices, torch.ones(nnz, device=dev), size=(n, n))
                    f32 = feat.float()
                    line["vendor_hipsparse_f32_ms"] = median_ms(lambda: a @ f32, iters=3, warm=1, batch=1)
                    del a, f32
                except Exception as exc:
                    line["vendor_hipsparse_error"] = str(exc)[:120]
            if args.cpu:
                from oracle import torch_ref  # baseline leg only

                torch.set_num_threads(os.cpu_count())
                a = torch_ref.csr_ones(indptr.cpu(), indices.cpu(), n, n)
                fc = feat.float().cpu()
                a @ fc
                t0 = time.perf_counter()
                a @ fc
                a @ fc
                cpu_ms = (time.perf_counter() - t0) * 500
                line["cpu_torch_sparse_mm_f32_ms"] = cpu_ms
                line["cpu_gflops"] = 2.0 * nnz * f / cpu_ms / 1e6
                line["cpu_threads"] = os.cpu_count()
            print(json.dumps(line), flush=True)
            del feat, out
        del handle, two, indptr, indices
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
