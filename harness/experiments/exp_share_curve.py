"""Two-level side-car against the window format across the SHARE of edges in shared columns (the auto
threshold VOLTRIX_HYBRID_MIN_SHARE = 0.4 rested on two measured points, 29 % and 55 %).  reddit-size graphs (same N, degree
law and edge count as reddit_like) whose local half of the column mixture goes from 0 to 65 % of the edges: for each, the
share the plan builder's count phase reports, the operator's step in the window format and with the side-car forced.
    python harness/experiments/exp_share_curve.py [feat]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402


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
    feat_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    base = dict(synth_graphs.CONFIGS["reddit_like"])
    for band_frac in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.65):
        cfg = dict(base, band_frac=band_frac, band=base["band"] if band_frac > 0 else 0)
        indptr, indices = synth_graphs.generate_csr(device=dev, **cfg)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, feat_dim, device=dev).half()
        line = {"band_fraction": band_frac, "N": n, "nnz": e, "F": feat_dim}
        for mode in ("0", "1"):
            os.environ["VOLTRIX_HYBRID"] = mode
            os.environ["VOLTRIX_HYBRID_MIN_SHARE"] = "0"
            handle = voltrix.csr_preprocess_device(indptr, indices, n)
            handle[1].hash_tag = f"share_curve/{band_frac}/{mode}"
            two = voltrix.two_level_of(handle[1])
            ms = time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat))
            if mode == "0":
                line["window_ms"] = ms
                line["tc_blocks"] = int(handle[0][-1])
            else:
                line["two_level_ms"] = ms
                line["shared_fraction"] = two.plan.num_shared_edges / e if two is not None else 0.0
                line["ksteps"] = two.plan.num_ksteps if two is not None else 0
            del handle, two
        line["two_level_over_window"] = line["two_level_ms"] / line["window_ms"]
        print(json.dumps(line), flush=True)
        del indptr, indices, feat
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
