"""Two-level step over the panel SHAPE (waves x row blocks per workgroup = 512- or 256-row panels), the piece bound and the
panel kernel's loop (classic / software-pipelined), through voltrix.spmm with the side-car swapped.
    python harness/experiments/exp_panel_shapes.py [--graphs protein_like,reddit_like] [--feat 128]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ["VOLTRIX_HYBRID"] = "1"
os.environ["VOLTRIX_HYBRID_MIN_SHARE"] = "0"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid, sidecar  # noqa: E402
import importlib  # noqa: E402

spmm_mod = importlib.import_module("voltrix.spmm.spmm")
from voltrix.utils import KernelTimer  # noqa: E402

from exp_panel_pipe import time_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="protein_like,reddit_like")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--shapes", default="8:4:3,8:2:3,8:2:2")          # waves:row_blocks:tau
    ap.add_argument("--tiles", default="3:1,4:1,4:17,6:1,6:17")        # depth:ksteps (17 = pipelined loop)
    ap.add_argument("--factors", default="1.0,0.5,0.25")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    f = args.feat
    for name in args.graphs.split(","):
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, f, device=dev).half()
        ints = torch.randint(-3, 4, (n, f), device=dev).half()
        ref = None
        for shape in args.shapes.split(","):
            waves, rb, tau = (int(x) for x in shape.split(":"))
            handle = voltrix.csr_preprocess_device(indptr, indices, n)
            two = spmm_mod._build_two_level(indptr, indices, n, n, waves, rb, tau, min_share=0.0)
            two.hash_tag = handle[1].hash_tag = f"panel_shapes/{name}/{shape}"
            sidecar.register(handle[1], two)
            plan = two.plan
            default_parts = plan.parts
            print(json.dumps({"graph": name, "shape": shape, "panel_rows": plan.panel_rows, "ksteps": plan.num_ksteps,
                              "shared": round(plan.num_shared_edges / e, 4), "resid_blocks": int(two.blk_offsets[-1])}), flush=True)
            run = lambda x=feat: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)  # noqa: E731
            classic = hybrid.default_panel_tile
            for factor in [float(x) for x in args.factors.split(",")]:
                plan.parts = hybrid.panel_parts(plan.panel_ptr, max(8, int(factor * plan.num_ksteps / hybrid.NUM_CUS)), plan.xcd_ptr)
                for t in args.tiles.split(","):
                    tile = (128 if f >= 128 else f, int(t.split(":")[0]), int(t.split(":")[1]))
                    hybrid.default_panel_tile = (lambda tt: (lambda *a, **k: tt))(tile)
                    try:
                        got = run(ints).clone()
                    except Exception as ex:   # tile not instantiated for this shape
                        print(json.dumps({"graph": name, "shape": shape, "factor": factor, "tile": tile, "error": str(ex)[:80]}), flush=True)
                        hybrid.default_panel_tile = classic
                        continue
                    if ref is None:
                        ref = got
                    ms = time_ms(run)
                    with KernelTimer() as timer:
                        for _ in range(5):
                            run()
                    kernels = {k: round(v[1], 4) for k, v in timer.summary().items()}
                    hybrid.default_panel_tile = classic
                    print(json.dumps({"graph": name, "shape": shape, "factor": factor, "pieces": plan.parts.num_parts, "tile": tile,
                                      "ms": round(ms, 4), "exact": bool(torch.equal(got, ref)), "kernels_ms": kernels}), flush=True)
            plan.parts = default_parts
            del handle, two, plan


if __name__ == "__main__":
    main()
