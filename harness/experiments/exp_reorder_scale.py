"""Locality reorder at BASELINE scale (profiles/HISTORY.md section 3.4): the stand-in in its natural order, with its node labels
shuffled (synth_graphs.*_shuffled), and the shuffled one after the row reorders of voltrix/reorder.py (bfs, spectral):
TC blocks of the window format, share of edges the two-level plan takes, time of the operator call, reorder time.

    python harness/experiments/exp_reorder_scale.py [reddit|products] [scale] [feat] [methods]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import reorder  # noqa: E402


def time_ms(fn, iters=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def wall_ms(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def describe(handle, n, nnz, feat):
    two = voltrix.two_level_of(handle[1])
    blocks = int(handle[0][-1])
    line = {"tc_blocks": blocks, "gather_GB_window": 8 * blocks * feat.shape[1] * 2 / 1e9}
    if two is not None:
        line.update(two_level=True, shared_fraction=two.plan.num_shared_edges / nnz,
                    gather_GB_two_level=(8 * int(two.blk_offsets[-1]) + 32 * two.plan.num_ksteps) * feat.shape[1] * 2 / 1e9)
    else:
        line["two_level"] = False
    return line


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "reddit"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    feat_dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    methods = (sys.argv[4] if len(sys.argv) > 4 else "bfs,spectral").split(",")
    dev = torch.device("cuda", 0)
    base, shuffled = f"{which}_like", f"{which}_shuffled"
    torch.manual_seed(0)
    results = {}

    indptr, indices, _ = synth_graphs.generate(base, device=dev, scale=scale)
    n, nnz = indptr.numel() - 1, indices.numel()
    feat = torch.randn(n, feat_dim, device=dev).half()
    voltrix.csr_preprocess_device(indptr, indices, n)                       # library load, allocator
    handle, pre_ms = wall_ms(lambda: voltrix.csr_preprocess_device(indptr, indices, n))
    handle[1].hash_tag = f"exp_reorder/{base}/{scale}"
    ms = time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat))
    two = voltrix.two_level_of(handle[1])
    results["natural"] = dict(describe(handle, n, nnz, feat), ms=ms, preprocess_ms=pre_ms,
                              ksteps=two.plan.num_ksteps if two else 0, residual_blocks=int(two.blk_offsets[-1]) if two else 0)
    print(json.dumps({"graph": base, "scale": scale, "N": n, "nnz": nnz, "F": feat_dim, "natural": results["natural"]}), flush=True)
    del handle

    s_indptr, s_indices, label = synth_graphs.shuffle_labels(indptr, indices, synth_graphs.CONFIGS[shuffled]["shuffle_seed"])
    del indptr, indices
    feat_s = torch.empty_like(feat)
    feat_s[label] = feat                                                      # B in the shuffled labelling
    handle, pre_ms = wall_ms(lambda: voltrix.csr_preprocess_device(s_indptr, s_indices, n))
    handle[1].hash_tag = f"exp_reorder/{shuffled}/{scale}"
    ms = time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat_s))
    ref = voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat_s)
    results["shuffled"] = dict(describe(handle, n, nnz, feat), ms=ms, preprocess_ms=pre_ms)
    print(json.dumps({"shuffled": results["shuffled"]}), flush=True)
    del handle

    for method in methods:
        fn = {"bfs": reorder.bfs_permutation, "spectral": reorder.spectral_permutation}[method]
        if method == "spectral" and os.environ.get("EXP_SPECTRAL"):      # "vectors,iterations" (experiments)
            import functools

            v, it = (int(t) for t in os.environ["EXP_SPECTRAL"].split(","))
            fn = functools.partial(reorder.spectral_permutation, vectors=v, iterations=it)
        perm, order_ms = wall_ms(lambda: fn(s_indptr, s_indices, n))
        extra = {}
        if method == "spectral":   # once more with phase timings (kernels loaded, allocator warm)
            (perm, info), order_ms = wall_ms(lambda: fn(s_indptr, s_indices, n, return_info=True))
            extra = {"spectral_info": info}
        rh, build_ms = wall_ms(lambda: reorder.csr_preprocess_reordered(s_indptr, s_indices, n, method=perm))
        rh.hspa_packed.hash_tag = f"exp_reorder/{shuffled}/{scale}/{method}"
        ms = time_ms(lambda: reorder.spmm_reordered(rh, feat_s))
        out = reorder.spmm_reordered(rh, feat_s)
        err = float((out - ref).norm() / ref.norm())
        # how local the order is: the rows of a window, mapped back to the UNSHUFFLED labels, should be neighbours
        inv_label = torch.empty_like(label)
        inv_label[label] = torch.arange(n, device=dev)
        orig = inv_label[rh.perm]
        spread = (orig[1:] - orig[:-1]).abs().float().median().item()
        handle = (rh.blk_offsets, rh.hspa_packed, rh.hind)
        two = voltrix.two_level_of(rh.hspa_packed)
        if two is not None:
            extra.update(ksteps=two.plan.num_ksteps, residual_blocks=int(two.blk_offsets[-1]),
                         ms_without_unpermute=time_ms(lambda: voltrix.spmm(rh.blk_offsets, rh.hspa_packed, rh.hind, num_nodes=n,
                                                                          num_edges=nnz, feat=feat_s)))
        results[method] = dict(describe(handle, n, nnz, feat), **extra, ms=ms, order_ms=order_ms, handle_ms=build_ms,
                               rel_err_vs_unreordered=err, median_neighbour_distance_in_natural_labels=spread)
        print(json.dumps({method: results[method]}), flush=True)
        del rh, handle, out
    print(json.dumps({"summary": {k: {"ms": round(v["ms"], 4), "tc_blocks": v["tc_blocks"], "two_level": v["two_level"]}
                                  for k, v in results.items()}}), flush=True)


if __name__ == "__main__":
    main()
