"""The stream kernel (spmm_stream_kernels.hpp, tuner schedule 6) against the handle's tuned window kernel on the low-degree
stand-ins of the reference's evaluation set: exactness on integer operands (vs hipSPARSE fp32) and kernel time over ring depth,
waves per workgroup and run cost.
    python harness/experiments/exp_stream.py [--graphs a,b] [--feat 128] [--points D:W,...] [--costs 0,8,16,32] [--build-only]"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit import build, cpp_format, generate  # noqa: E402
from voltrix.jit_kernels import spmm as S  # noqa: E402
from voltrix.schedule import stream_tables  # noqa: E402


def runtime_for(point, dtype=torch.float16):
    defs = S.arg_defs_for(dtype)
    return build("spmm_kernel", defs, generate(S.includes, defs, cpp_format(S.template, point)))


def graph_ms(make_fn, batch=10, iters=5):
    """Device time per launch with the host out of the way: ``batch`` launches captured in one HIP graph, replayed."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn = make_fn()          # argument tuples carry the current (= capture) stream
        fn()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(batch):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        graph.replay()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) / batch)
    return sorted(ts)[len(ts) // 2]


def time_ms(fn, iters=5, warm=2, batch=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(batch):
            fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) / batch)
    return sorted(ts)[len(ts) // 2]


def stream_args(handle, n, nnz, feat, out, table):
    bo, hp, hi = handle
    dummy = bo
    scale = S.unit_scale(feat.device)
    partials = torch.empty(max(1, table.num_slots) * 16 * feat.shape[1], dtype=torch.float32, device=feat.device)
    return (bo, hp, hi, n, nnz, feat.shape[1], feat, out, dummy, dummy, dummy, scale, 0, dummy, dummy, 0, dummy, 0, scale,
            dummy, dummy, 0, dummy, 0, scale, 1, dummy, 0, feat, int(feat.shape[0]), int(S.SLAB_POLICY),
            table.units, table.runs, table.run_ptr, table.max_runs_per_xcd, table.cuts, table.num_cuts, partials,
            torch.cuda.current_stream())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="yeast_like,yeasth_like,dd_like,com_amazon_like,amazon0601_like,amazon0505_like,"
                                        "web_berkstan_like,ppi_like")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--points", default="2:1,3:1,4:1,2:2,3:2")
    ap.add_argument("--costs", default="0,8,16,32")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--no-base", action="store_true")
    args = ap.parse_args()
    f = args.feat
    fs = 32 if f <= 32 else (64 if f <= 64 else 128)
    points = [dict(FS=fs, DEPTH=int(p.split(":")[0]), WAVES=int(p.split(":")[1]), EB=2, BF16=0, WEIGHTED=0, SCHED=6)
              for p in args.points.split(",")]
    with ThreadPoolExecutor(max_workers=8) as pool:
        runtimes = list(pool.map(runtime_for, points))
    if args.build_only:
        print("built", len(runtimes))
        return
    dev = torch.device("cuda", 0)
    for name in args.graphs.split(","):
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, nnz = indptr.numel() - 1, indices.numel()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"exp_stream/{name}"
        torch.manual_seed(0)
        feat = torch.randint(-3, 4, (n, f), device=dev).half()      # integers: every sum is exact in fp32
        ref = torch.sparse_csr_tensor(indptr, indices, torch.ones(nnz, device=dev), size=(n, n)) @ feat.float()
        os.environ["VOLTRIX_HYBRID"] = "0"
        alg = synth_graphs.algorithmic_bytes(n, nnz, f, 2)
        line = {"graph": name, "feat": f, "nnz": nnz, "num_nodes": n}
        base = None
        if not args.no_base:
            base = voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)
            line.update({"base_exact": bool(torch.equal(base, ref)),
                         "base_ms": round(time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)), 4)})
            line["base_roofline"] = round(alg / line["base_ms"] / 8e9, 4)
            line["base_graph_ms"] = round(graph_ms(lambda: (lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat))), 4)
        out = torch.empty(n, f, device=dev)
        best = None
        for cost in [int(c) for c in args.costs.split(",")]:
            table = stream_tables(*handle, n, run_cost=cost or None)
            for point, rt in zip(points, runtimes):
                a = stream_args(handle, n, nnz, feat, out, table)
                out.fill_(float("nan"))
                rc = rt(*a)
                assert rc == 0, rc
                torch.cuda.synchronize()
                exact = bool(torch.equal(out, ref))
                def make():
                    b = stream_args(handle, n, nnz, feat, out, table)
                    return lambda: rt(*b)

                ms = graph_ms(make)
                key = f"D{point['DEPTH']}W{point['WAVES']}c{table.run_cost}"
                line[key] = round(ms, 4)
                if not exact:
                    line[key + "_WRONG"] = int((out != ref).sum())
                if exact and (best is None or ms < best[0]):
                    best = (ms, key, table.num_runs, table.num_cuts)
        if best:
            line["best"] = best[1]
            line["best_ms"] = round(best[0], 4)
            line["best_roofline"] = round(alg / best[0] / 8e9, 4)
            line["runs"], line["cuts"] = best[2], best[3]
        print(json.dumps(line), flush=True)
        del handle, ref, base, feat, out


if __name__ == "__main__":
    main()
