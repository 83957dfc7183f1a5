"""Does the two-level format pay on graphs with fewer 512-row panels than CUs now that long panels run in pieces?  The auto
rule asks for >= 131 k rows (one panel workgroup per CU); with pieces of at most S / 256 k-steps a plan with 100 panels still
launches ~256 workgroups.  reddit-like and block-model graphs at 1/8 .. 1/2 of the full size: window format against the
forced side-car, and the side-car without pieces.
    python harness/experiments/exp_small_graphs.py [feat]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
os.environ["VOLTRIX_HYBRID_MIN_SHARE"] = "0"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid  # noqa: E402

from exp_panel_parts import time_ms  # noqa: E402


def main():
    feat_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    for name in ("reddit_like", "reddit_sbm"):
        for scale in (0.125, 0.25, 0.5, 1.0):
            indptr, indices, _ = synth_graphs.generate(name, device=dev, scale=scale)
            n, e = indptr.numel() - 1, indices.numel()
            feat = torch.randn(n, feat_dim, device=dev).half()
            line = {"graph": name, "scale": scale, "N": n, "nnz": e, "F": feat_dim, "panels": (n + 511) // 512}
            for mode in ("0", "1"):
                os.environ["VOLTRIX_HYBRID"] = mode
                handle = voltrix.csr_preprocess_device(indptr, indices, n)
                handle[1].hash_tag = f"small/{name}/{scale}/{mode}"
                run = lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)  # noqa: E731
                if mode == "0":
                    line["window_ms"] = round(time_ms(run), 4)
                    continue
                two = voltrix.two_level_of(handle[1])
                if two is None:
                    line["two_level_ms"] = None
                    continue
                plan = two.plan
                line.update(shared_fraction=round(plan.num_shared_edges / e, 3), ksteps=plan.num_ksteps,
                            longest_panel=int(torch.diff(plan.panel_ptr).max()), cap=hybrid.default_part_cap(plan.num_ksteps),
                            cut_panels=plan.parts.num_cuts if plan.parts else 0, pieces=plan.parts.num_parts if plan.parts else plan.num_panels)
                line["two_level_ms"] = round(time_ms(run), 4)
                saved = plan.parts
                plan.parts = None
                line["two_level_whole_panels_ms"] = round(time_ms(run), 4)
                for factor in (0.5, 0.25):
                    plan.parts = hybrid.panel_parts(plan.panel_ptr, max(8, int(factor * plan.num_ksteps / hybrid.NUM_CUS)), plan.xcd_ptr)
                    line[f"two_level_pieces_{factor}_ms"] = round(time_ms(run), 4)
                plan.parts = saved
            print(json.dumps(line), flush=True)
            del handle


if __name__ == "__main__":
    main()
