"""Where does one launch per column slab stop paying?  reddit-like at 1x / 2x / 4x the node count (60 / 119 / 239 MB of B
per 128-column slab), F = 512, window format and two-level, VOLTRIX_SLAB_LAUNCHES from the environment (0 / 1).

    VOLTRIX_SLAB_LAUNCHES=0|1 python harness/experiments/exp_slab_threshold.py [scale] [F]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
from voltrix.spmm.spmm import csr_preprocess_device, spmm, two_level_of  # noqa: E402


def time_ms(fn, iters=8):
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
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    name = sys.argv[3] if len(sys.argv) > 3 else "reddit_like"
    ip, ix, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n, nnz = ip.numel() - 1, ix.numel()
    feat = torch.randn(n, feat_dim, device="cuda").half()
    out = []
    for hybrid in ("0", "1"):
        os.environ["VOLTRIX_HYBRID"] = hybrid
        handle = csr_preprocess_device(ip, ix, n)
        handle[1].hash_tag = f"slab_threshold_{scale}_{hybrid}"
        two = two_level_of(handle[1]) is not None
        out.append(f"{'two-level' if two else 'window'} {time_ms(lambda: spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)):.3f} ms")
        del handle
    print(f"{name} x{scale}: n={n} nnz={nnz} F={feat_dim} slab={n * 256 / 2**20:.0f} MiB  SLAB_LAUNCHES={os.getenv('VOLTRIX_SLAB_LAUNCHES')}  " + "  ".join(out))


if __name__ == "__main__":
    main()
