"""Diagnostic builds of the stream kernel (VOLTRIX_STREAM_DIAG: 1 = no B fragment reads / MFMA, 2 = no stores, 4 = gathered rows
folded into the first 1024 rows of B, 8 = no row gathers): which part of the step bounds the kernel on which graph.
    python harness/experiments/exp_stream_diag.py [--graphs a,b] [--diags 0,1,2,3,4,6,8,11] [--point 2:1] [--build-only]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd"), os.path.dirname(os.path.abspath(__file__))]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit import build, cpp_format, generate  # noqa: E402
from voltrix.jit_kernels import spmm as S  # noqa: E402
from voltrix.schedule import stream_tables  # noqa: E402

from exp_stream import graph_ms, stream_args  # noqa: E402


def runtime_for(point, diag):
    defs = S.arg_defs_for(torch.float16)
    code = generate(S.includes, defs, cpp_format(S.template, point))
    if diag:
        code = f"#define VOLTRIX_EXPERIMENTAL 1\n#define VOLTRIX_STREAM_DIAG {diag}\n" + code
    return build(f"stream_diag{diag}", defs, code)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="dd_like,amazon0505_like,yeasth_like")
    ap.add_argument("--diags", default="0,1,2,3,4,6,8,11")
    ap.add_argument("--point", default="2:1")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    d, w = (int(x) for x in args.point.split(":"))
    f = args.feat
    fs = 32 if f <= 32 else (64 if f <= 64 else 128)
    point = dict(FS=fs, DEPTH=d, WAVES=w, EB=2, BF16=0, WEIGHTED=0, SCHED=6)
    diags = [int(x) for x in args.diags.split(",")]
    rts = {x: runtime_for(point, x) for x in diags}
    if args.build_only:
        return
    dev = torch.device("cuda", 0)
    for name in args.graphs.split(","):
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, nnz = indptr.numel() - 1, indices.numel()
        os.environ["VOLTRIX_HYBRID"] = "0"
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        feat = torch.randn(n, f, device=dev).half()
        out = torch.empty(n, f, device=dev)
        table = stream_tables(*handle, n)
        line = {"graph": name, "point": args.point, "runs": table.num_runs, "stages": int(table.runs[:, 2].sum())}
        if 16 in diags and table.num_slots > 0:
            # the stamps go to the `partials` argument: a handle with cut windows stores its partial tiles there (the run on
            # web_berkstan_like wrote them past the stamp buffer and the process was aborted by the memory fault)
            print(json.dumps({"graph": name, "skipped": "diag 16 needs a handle without cut windows", "slots": table.num_slots}), flush=True)
            diags = [x for x in diags if x != 16]
        if 16 in diags:   # in-kernel cycle sums per phase (diag bit 4): partials is the debug buffer
            grid_waves = table.max_runs_per_xcd * 8 + 64
            dbg = torch.zeros(grid_waves * 8, dtype=torch.float32, device=dev)
            b = list(stream_args(handle, n, nnz, feat, out, table))
            b[37] = dbg
            for _ in range(3):
                assert rts[16](*b) == 0
            torch.cuda.synchronize()
            v = dbg.view(-1, 8)
            v = v[v[:, 7] > 0]
            names = ["prologue", "wait_vm", "lds", "refill_issue", "mfma_issue", "epilogue", "control", "total"]
            line["waves"] = int(v.shape[0])
            line["cycles_per_wave"] = {k: round(float(v[:, i].mean()), 0) for i, k in enumerate(names)}
            line["stages_per_wave"] = round(line["stages"] / max(1, v.shape[0]), 2)
            tot = v[:, 7]
            line["wave_total_cycles"] = {"mean": round(float(tot.mean())), "p50": round(float(tot.median())),
                                         "p99": round(float(tot.quantile(0.99))), "max": round(float(tot.max()))}
        for x in [y for y in diags if y != 16]:
            def make():
                b = stream_args(handle, n, nnz, feat, out, table)
                return lambda: rts[x](*b)
            line[f"diag{x}"] = round(graph_ms(make), 4)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
