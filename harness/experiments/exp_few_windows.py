"""Graphs of FEW, LONG windows (ddi-like: 267 windows of ~130 stages): a window is one wave's serial stream, so the launch has
fewer waves than the chip has SIMDs.  Unit tables with a shorter length bound (every window cut into interleaved units + the
combine pass) against the default (1.5 x the median: nothing cut when all windows are alike).
    python harness/experiments/exp_few_windows.py [--graphs a,b] [--feats 32,128,512] [--bounds 0,64,32,16,8]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd"), os.path.join(REPO, "harness")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402
from voltrix.schedule import balanced_xcd_windows, unit_table  # noqa: E402

from eval_set import launches_per_call, steady_ms, tuned_point  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="ddi_like,fraud_yelp_rsr_like,ppi_like")
    ap.add_argument("--feats", default="32,128,512")
    ap.add_argument("--bounds", default="0,64,32,16,8")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    os.environ["VOLTRIX_HYBRID"] = "0"
    os.environ["VOLTRIX_TUNED_DEFAULTS"] = "0"   # every (graph, width, bound) sweeps the tile space on ITS unit table
    for name in args.graphs.split(","):
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n, nnz = indptr.numel() - 1, indices.numel()
        for f in [int(x) for x in args.feats.split(",")]:
            feat = torch.randint(-3, 4, (n, f), device=dev).half()
            ref = torch.sparse_csr_tensor(indptr, indices, torch.ones(nnz, device=dev), size=(n, n)) @ feat.float()
            for bound in [int(x) for x in args.bounds.split(",")]:
                os.environ["VOLTRIX_TUNED_STORE"] = f"/tmp/few_windows_{name}_{f}_{bound}.json"
                jit_tuner.tuned.clear()
                jit_tuner.tuned_keys.clear()
                jit_tuner.generation += 1
                handle = voltrix.csr_preprocess_device(indptr, indices, n)
                handle[1].hash_tag = f"few_windows/{name}/{bound}"
                bo, hp = handle[0], handle[1]
                stages = int((((bo[1:] - bo[:-1]) + 3) // 4).sum())
                if bound:
                    table = unit_table(bo, n, bound, xcd_ptr=balanced_xcd_windows(bo, n))
                    hp._voltrix_unit_table = ((bo.data_ptr(), n, 0), table)
                    hp._voltrix_unit_table_pairs = ((bo.data_ptr(), n, 0), table)
                call = lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)  # noqa: E731
                out = call()
                exact = bool(torch.equal(out, ref))
                ms = steady_ms(call)
                tbl = getattr(hp, "_voltrix_unit_table", (None, None))[1]
                print(json.dumps({"graph": name, "feat": f, "bound": bound, "windows": (n + 15) // 16, "stages": stages,
                                  "units": tbl.num_units if tbl is not None else None,
                                  "slots": tbl.num_slots if tbl is not None else None, "ms": round(ms, 4), "exact": exact,
                                  "launches": launches_per_call(call), "tile": tuned_point(hp, f, False, dev)}), flush=True)


if __name__ == "__main__":
    main()
