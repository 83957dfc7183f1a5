"""One-launch two-level kernel (spmm_fused_kernel) against the round-2 pair (panel || window, atomic join): same bits on
integer operands, relative error against torch on random ones, kernel time of both forms.
    python harness/experiments/exp_fused.py [workload] [scale] [feat]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
os.environ["VOLTRIX_FUSED"] = "1"      # csr_preprocess_hybrid builds the stage records only when the one-launch form is asked for

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.spmm.spmm import _run_two_level  # noqa: E402


def time_ms(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    feat_dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    dev = torch.device("cuda", 0)
    indptr, indices, _ = synth_graphs.generate(workload, device=dev, scale=scale)
    n = indptr.numel() - 1
    two = voltrix.csr_preprocess_hybrid(indptr.cpu(), indices.cpu(), n)
    two.hash_tag = f"exp_fused/{workload}/{scale}"
    fr = two.fused
    assert fr is not None
    print(f"{workload} x{scale}: N={n} nnz={indices.numel()} k-steps={two.plan.num_ksteps} shared={two.plan.num_shared_edges} "
          f"resid={two.plan.num_resid_edges} records={fr.num_records if fr else None}", flush=True)
    torch.manual_seed(0)
    ints = torch.randint(-3, 4, (n, feat_dim), device=dev).half()
    out_f = torch.empty(n, feat_dim, device=dev)
    out_p = torch.empty(n, feat_dim, device=dev)
    os.environ["VOLTRIX_FUSED"] = "1"
    _run_two_level(two, ints, out_f, None)
    os.environ["VOLTRIX_FUSED"] = "0"
    _run_two_level(two, ints, out_p, None)
    torch.cuda.synchronize()
    bad = int((out_f != out_p).sum())
    print(f"integer operand: {bad} elements differ between the one-launch kernel and the pair", flush=True)
    if bad:
        rows = torch.nonzero((out_f != out_p).any(1)).flatten()
        print("  first differing rows:", rows[:16].tolist(), "count", rows.numel())
        r = int(rows[0])
        print("  fused:", out_f[r, :8].tolist(), "\n  pair: ", out_p[r, :8].tolist())
    feat = torch.randn(n, feat_dim, device=dev).half()
    os.environ["VOLTRIX_FUSED"] = "1"
    _run_two_level(two, feat, out_f, None)
    ref = torch.sparse_csr_tensor(indptr.long(), indices.long(), torch.ones(indices.numel(), device=dev), size=(n, n)) @ feat.float()
    print(f"random operand: rel err vs torch.sparse.mm (GPU, fp32) {float((out_f - ref).norm() / ref.norm()):.3e}", flush=True)
    os.environ["VOLTRIX_FUSED"] = "1"
    from voltrix import hybrid

    t_f = time_ms(lambda: _run_two_level(two, feat, out_f, None))
    paced = {}
    for blocks in (8, 16, 32, 48):
        hybrid.FUSED_PACE_BLOCKS = blocks
        _run_two_level(two, ints, out_f, None)
        torch.cuda.synchronize()
        same = bool((out_f == out_p).all())                      # pacing is advisory: same bits
        paced[blocks] = (time_ms(lambda: _run_two_level(two, feat, out_f, None)), same)
    hybrid.FUSED_PACE_BLOCKS = 0
    os.environ["VOLTRIX_FUSED"] = "0"
    t_p = time_ms(lambda: _run_two_level(two, feat, out_p, None))
    print(f"one launch {t_f:.4f} ms   pair {t_p:.4f} ms   one launch paced (sync points: ms, bits equal): "
          + ", ".join(f"{b}: {t:.4f} {ok}" for b, (t, ok) in paced.items()), flush=True)


if __name__ == "__main__":
    main()
