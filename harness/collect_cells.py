#!/usr/bin/env python3
"""Round 6: refresh the shipped tuner buckets CELL BY CELL, A/B on one box.  For one graph, every (width, feature dtype) cell is
timed (steady state) with the choice the shipped buckets give (mode ``shipped``) and with the choice of a fresh first-call sweep
on an empty store (mode ``fresh``), each mode in its own process; ``merge`` keeps a fresh bucket entry only where the cells that
produced it ran faster than with the shipped choice by MERGE_GAIN (cross-run comparisons on different boxes are +- 10 %).
    python harness/collect_cells.py shipped|fresh <graph> <out.json>
    python harness/collect_cells.py merge <dir with *_shipped.json / *_fresh.json> <merged_store.json>"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MERGE_GAIN = 0.04
WIDTHS = (32, 128, 256, 512, 1024)


def run(mode, graph, out_path):
    store = out_path + ".store"
    if os.path.exists(store):
        os.remove(store)
    os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
    os.environ.update(VOLTRIX_TUNED_STORE=store, VOLTRIX_TUNED_DEFAULTS="1" if mode == "shipped" else "0", VOLTRIX_TUNE_SPACE="default")
    sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
    import torch

    import synth_graphs
    import voltrix
    from voltrix.jit_kernels import jit_tuner

    indptr, indices, _ = synth_graphs.generate(graph, device="cuda")
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = f"collect_cells/{graph}"
    cells = {}

    def load():
        try:
            return json.load(open(store))
        except (OSError, ValueError):
            return {}

    for width in WIDTHS:
        gen = torch.Generator(device="cuda").manual_seed(width)
        feat32 = torch.randn(n, width, generator=gen, device="cuda")
        for dtype in ("float16", "float32"):
            feat = feat32.half() if dtype == "float16" else feat32
            before = load()
            call = lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)  # noqa: E731
            call()
            torch.cuda.synchronize()
            after = load()
            for _ in range(5):
                call()
            times = []
            for _ in range(7):
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    call()
                t.record()
                t.synchronize()
                times.append(s.elapsed_time(t) / 10)
            cells[f"{width}|{dtype}"] = {"steady_ms": sorted(times)[3],
                                         "new_entries": {k: v for k, v in after.items() if k not in before and "@bucket" in k}}
            del feat
        del feat32
    json.dump({"graph": graph, "mode": mode, "cells": cells, "tuner": jit_tuner.stats}, open(out_path, "w"), indent=1)


def merge(directory, out_path):
    merged, report = {}, []
    for name in sorted(os.listdir(directory)):
        if not name.endswith("_fresh.json"):
            continue
        fresh = json.load(open(os.path.join(directory, name)))
        shipped = json.load(open(os.path.join(directory, name.replace("_fresh.json", "_shipped.json"))))
        by_key = {}
        for cell, got in fresh["cells"].items():
            for key, point in got["new_entries"].items():
                by_key.setdefault(key, {"point": point, "cells": []})["cells"].append(cell)
        # a key is written by the FIRST cell that needs it; later cells of the same key (fp32 features cast to fp16, "wide" widths)
        # reuse it: attribute them by dtype / width class
        for key, info in by_key.items():
            wide = "'embedding_dim': 'wide'" in key
            f16 = "'dtype': 'torch.float16'" in key
            for cell in fresh["cells"]:
                width, dtype = cell.split("|")
                same_width = (wide and int(width) > 128) or f"'embedding_dim': {width}," in key
                if same_width and f16 and cell not in info["cells"] and not fresh["cells"][cell]["new_entries"]:
                    info["cells"].append(cell)
        for key, info in by_key.items():
            a = sum(shipped["cells"][c]["steady_ms"] for c in info["cells"])
            b = sum(fresh["cells"][c]["steady_ms"] for c in info["cells"])
            keep = b < (1.0 - MERGE_GAIN) * a
            report.append((fresh["graph"], sorted(info["cells"]), round(a, 4), round(b, 4), round(b / a, 3), "KEEP" if keep else "drop",
                           {k: info["point"][k] for k in ("FS", "DEPTH", "WAVES", "SCHED")}))
            if keep:
                merged[key] = info["point"]
    for line in report:
        print(*line)
    json.dump(merged, open(out_path, "w"), indent=1, sort_keys=True)
    print(f"{len(merged)} bucket entries kept of {len(report)} -> {out_path}")


if __name__ == "__main__":
    if sys.argv[1] == "merge":
        merge(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1], sys.argv[2], sys.argv[3])
