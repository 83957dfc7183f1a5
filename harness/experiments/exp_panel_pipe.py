"""The software-pipelined k-step loop of the panel kernel (PanelTile<..., PIPE = true>; C-ABI ksteps = 17) against the classic
loop: same bits on integer operands, the panel kernel alone (whole panels / pieces of bounded length), and the two-level step
through voltrix.spmm with the panel tile switched.
    python harness/experiments/exp_panel_pipe.py [--graphs reddit_like,protein_like] [--feat 128]"""
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
from voltrix import hybrid  # noqa: E402
from voltrix.utils import KernelTimer  # noqa: E402

PIPE = 17


def time_ms(fn, reps=7, batch=5):
    for _ in range(3):
        fn()
    times = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(batch):
            fn()
        e.record()
        e.synchronize()
        times.append(s.elapsed_time(e) / batch)
    return sorted(times)[len(times) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="reddit_like,protein_like")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--tiles", default="3:1,3:17,4:1,4:17")
    ap.add_argument("--factors", default="0,1.0,0.5,0.25")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    f = args.feat
    tiles = [(128 if f >= 128 else f, int(t.split(":")[0]), int(t.split(":")[1])) for t in args.tiles.split(",")]
    for name in args.graphs.split(","):
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, f, device=dev).half()
        ints = torch.randint(-3, 4, (n, f), device=dev).half()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"panel_pipe/{name}"
        two = voltrix.two_level_of(handle[1])
        plan = two.plan
        nks = torch.diff(plan.panel_ptr)
        print(json.dumps({"graph": name, "F": f, "ksteps": plan.num_ksteps, "panels": plan.num_panels,
                          "longest_panel": int(nks.max()), "fair_share_per_cu": plan.num_ksteps / hybrid.NUM_CUS}), flush=True)
        # ---- the panel kernel alone: bits and time, whole panels and pieces ----
        default_parts = plan.parts
        ref = None
        for factor in [float(x) for x in args.factors.split(",")]:
            plan.parts = None if factor == 0 else hybrid.panel_parts(plan.panel_ptr, max(8, int(factor * plan.num_ksteps / hybrid.NUM_CUS)),
                                                                     plan.xcd_ptr)
            for tile in tiles:
                out = torch.zeros(n, f, device=dev)
                hybrid.launch_panel(plan, ints, out, 0, tile=tile)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out, ref))
                ms = time_ms(lambda: hybrid.launch_panel(plan, feat, out, 0, tile=tile))
                print(json.dumps({"graph": name, "panel_alone": True, "factor": factor,
                                  "pieces": plan.parts.num_parts if plan.parts else plan.num_panels, "tile": tile,
                                  "ms": round(ms, 4), "same_bits_as_first": same}), flush=True)
        # ---- the two-level step through the operator, panel tile switched ----
        run = lambda x=feat: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)  # noqa: E731
        classic = hybrid.default_panel_tile
        for factor in (None, 0.5, 0.25):
            plan.parts = default_parts if factor is None else hybrid.panel_parts(plan.panel_ptr, max(8, int(factor * plan.num_ksteps / hybrid.NUM_CUS)), plan.xcd_ptr)
            for tile in tiles:
                hybrid.default_panel_tile = (lambda t: (lambda *a, **k: t))(tile)
                try:
                    got = run(ints).clone()
                    if tile == tiles[0] and factor is None:
                        ref_step = got
                    ms = time_ms(run)
                    with KernelTimer() as timer:
                        for _ in range(5):
                            run()
                    kernels = {k: round(v[1], 4) for k, v in timer.summary().items()}
                    print(json.dumps({"graph": name, "step": True, "factor": factor, "tile": tile, "ms": round(ms, 4),
                                      "integers_bit_equal": bool(torch.equal(got, ref_step)), "kernels_ms": kernels}), flush=True)
                finally:
                    hybrid.default_panel_tile = classic
        plan.parts = default_parts
        del handle, two, plan


if __name__ == "__main__":
    main()
