"""Bounded sweep against the exhaustive one: every point of the default tile x schedule space timed on the
FULL handle (3 launches each), the two-stage sweep on the 1/16 sample as the operator runs it, and the full-size time of
the point each picks.
    python harness/experiments/exp_tuner_sample.py powerlaw_4m 256"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.update(VOLTRIX_TUNED_STORE="/tmp/exp_tuner_sample_store.json", VOLTRIX_TUNED_DEFAULTS="0", VOLTRIX_HYBRID="0")
if os.path.exists("/tmp/exp_tuner_sample_store.json"):
    os.remove("/tmp/exp_tuner_sample_store.json")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.jit_kernels import jit_tuner  # noqa: E402
from voltrix.jit_kernels import spmm as wrapper  # noqa: E402


def time_ms(fn, iters=3):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    workload, feat_dim = sys.argv[1], int(sys.argv[2])
    indptr, indices, _ = synth_graphs.generate(workload, device="cuda")
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    feat = torch.randn(n, feat_dim, device="cuda").half()
    # 1. the operator's bounded sweep
    handle[1].hash_tag = f"exp_tuner/{workload}/bounded"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    picked = dict(list(jit_tuner.tuned_keys.values())[-1])
    t_picked = time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat))
    print(json.dumps({"bounded_sweep_first_call_s": first, "stats": jit_tuner.stats, "picked": picked, "picked_full_ms": t_picked}),
          flush=True)
    # 2. every point of the space on the full handle
    space = wrapper.tile_space(feat_dim, 2)
    results = []
    for i, point in enumerate(space):
        os.environ["VOLTRIX_TUNED_STORE"] = f"/tmp/exp_tuner_sample_point_{i}.json"
        json.dump({}, open(os.environ["VOLTRIX_TUNED_STORE"], "w"))
        handle[1].hash_tag = f"exp_tuner/{workload}/point{i}"
        saved = wrapper.tile_space
        wrapper.tile_space = lambda *a, _p=point, **k: (_p,)
        try:
            ms = time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat))
        finally:
            wrapper.tile_space = saved
        results.append((ms, point))
        print(f"{ms:9.3f} ms  {point}", flush=True)
    best = min(results, key=lambda r: r[0])
    print(json.dumps({"exhaustive_best_ms": best[0], "exhaustive_best": best[1], "bounded_pick_ms": t_picked,
                      "bounded_over_best": t_picked / best[0]}), flush=True)


if __name__ == "__main__":
    main()
