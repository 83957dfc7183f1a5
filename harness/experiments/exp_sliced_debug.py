"""Debug: stacked (S x N') x N handle through the window kernel -- which of {schedule, row_map, xcd_ptr, atomic_out} gives wrong sums."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix.jit_kernels.spmm as js  # noqa: E402
from voltrix.jit_kernels import csr_fused_preprocess_kernel, jit_tuner, spmm_kernel  # noqa: E402

SCHED = [None]
_tile_space = js.tile_space
js.tile_space = lambda *a, **k: [p for p in _tile_space(*a, **k) if SCHED[0] is None or p["SCHED"] == SCHED[0]]

workload, scale, slices = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
num_feats = 128
indptr, indices, _ = synth_graphs.generate(workload, device="cuda", scale=scale)
n, e = indptr.numel() - 1, indices.numel()
feat = torch.randn(n, num_feats, device="cuda").half()
deg = (indptr[1:] - indptr[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), deg)
cols = indices.long()
ref = torch.sparse.mm(torch.sparse_csr_tensor(indptr.long(), cols, torch.ones(e, device="cuda"), (n, n)), feat.float())
np16 = (n + 15) // 16 * 16
wp = np16 // 16
slice_w = -(-n // slices)
new_row = (cols // slice_w) * np16 + rows
order = torch.argsort(new_row * n + cols)
st_cols = cols[order].to(torch.int32)
st_indptr = torch.zeros(slices * np16 + 1, dtype=torch.int64, device="cuda")
st_indptr[1:] = torch.bincount(new_row, minlength=slices * np16).cumsum(0)
st_indptr = st_indptr.to(torch.int32)
p1, packed, hind, _ = csr_fused_preprocess_kernel(st_indptr, st_cols, slices * np16, num_cols=n)
row_map = torch.arange(slices * np16, device="cuda", dtype=torch.int64) % np16
row_map[row_map >= n] = -1
row_map = row_map.to(torch.int32)
xcd_ptr = torch.tensor([x * slices * wp // 8 for x in range(9)], dtype=torch.int32, device="cuda")
print(f"{workload} x{scale}: N={n} nnz={e} S={slices} stacked windows {slices * wp} TC blocks {int(p1[-1])}", flush=True)
case = 0
for sched in (0, 2, 4, 5):
    for use_map, use_xcd, atomic in ((False, False, False), (False, True, False), (True, False, True), (True, True, True)):
        SCHED[0] = sched
        case += 1
        packed.hash_tag = f"dbg/{workload}/{scale}/{slices}/{case}"
        out_big = torch.zeros(slices * np16, num_feats, device="cuda")
        pending = spmm_kernel(p1, packed, hind, num_nodes=slices * np16, num_edges=e, embedding_dim=num_feats, input=feat, output=out_big,
                              atomic_out=atomic, defer_combine=True, row_map=row_map if use_map else None,
                              xcd_ptr=xcd_ptr if use_xcd else None)
        if pending is not None:
            pending.run()
        torch.cuda.synchronize()
        got = out_big[:n] if use_map else out_big.view(slices, np16, num_feats)[:, :n].sum(0)
        err = float((got - ref).abs().max() / ref.abs().max())
        bad_rows = int(((got - ref).abs().amax(1) > 1e-3 * ref.abs().max()).sum())
        point = [dict(p) for p in jit_tuner.tuned_keys.values()][-1]
        print(f"sched {sched} row_map {use_map} xcd_ptr {use_xcd} atomic {atomic}: rel err {err:.2e}, bad rows {bad_rows}, cut windows pending {pending is not None}; "
              f"tile ({point['FS']},{point['DEPTH']},{point['WAVES']},{point['SCHED']})", flush=True)
