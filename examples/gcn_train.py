"""A two-layer GCN trained with the operator of this package (example; no reference counterpart -- the reference is forward-only).

    H1 = relu(Â X W1),  Y = Â H1 W2,  Â = D^-1/2 (A + I) D^-1/2   (Kipf & Welling)

Â's values factor as r_i c_j, so both aggregations -- and both of their gradients -- run on the BINARY operator between two row
scalings (`voltrix.autograd.SpMM(..., values=)`: the two-level format, tuned tiles and launch plans included); the dense half is
torch (`nn.Linear`, fp16 autocast off: weights fp32, features cast to fp16 for the aggregation).

    python examples/gcn_train.py [workload] [hidden] [epochs]      # synthetic stand-in graph, random features and labels
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402


def normalised_adjacency(indptr, indices, n):
    """CSR of A + I (self loops added where missing) and the values of D^-1/2 (A + I) D^-1/2, all on ``indptr``'s device."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg)
    cols = indices.long()
    key = torch.unique(torch.cat([rows * n + cols, torch.arange(n, device=dev, dtype=torch.int64) * (n + 1)]))
    rows, cols = key // n, key % n
    new_indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    new_indptr[1:] = torch.bincount(rows, minlength=n).cumsum(0)
    d_out = torch.bincount(rows, minlength=n).double()
    d_in = torch.bincount(cols, minlength=n).double().clamp(min=1)
    values = (d_out.rsqrt()[rows] * d_in.rsqrt()[cols]).float()
    return new_indptr.to(torch.int32), cols.to(torch.int32), values


class GCN(torch.nn.Module):
    def __init__(self, aggregate, in_feats, hidden, classes, dtype=torch.float16):
        super().__init__()
        self.aggregate, self.dtype = aggregate, dtype
        self.w1 = torch.nn.Linear(in_feats, hidden)
        self.w2 = torch.nn.Linear(hidden, classes)

    def forward(self, x):
        # aggregate first where it is the narrower side: Â (X W) == (Â X) W
        h = torch.relu(self.aggregate(self.w1(x).to(self.dtype)))
        return self.aggregate(self.w2(h).to(self.dtype))


def main():
    import synth_graphs
    from voltrix.autograd import SpMM

    workload = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    indptr, indices, _ = synth_graphs.generate(workload, device="cuda")
    n = indptr.numel() - 1
    indptr, indices, values = normalised_adjacency(indptr, indices, n)
    t0 = time.perf_counter()
    op = SpMM(indptr, indices, n, values=values, hash_tag=f"example_gcn/{workload}")
    torch.cuda.synchronize()
    print(f"{workload}: N={n} nnz={indices.numel()} (self loops added); operator for A and A^T built in {time.perf_counter() - t0:.2f} s; "
          f"values separable: {op.weighted.separable}")
    torch.manual_seed(0)
    in_feats, classes = 128, 48          # classes: a multiple of 8 keeps the aggregated rows 16-byte aligned without padding
    x = torch.randn(n, in_feats, device="cuda")
    y = torch.randint(0, classes, (n,), device="cuda")
    model = GCN(op, in_feats, hidden, classes).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    times = []
    for epoch in range(epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        if epoch in (0, 1, epochs - 1):
            print(f"epoch {epoch}: loss {float(loss):.4f}, {times[-1]:.2f} ms")
    steady = sorted(times[2:])[len(times[2:]) // 2] if len(times) > 2 else times[-1]
    print(f"steady epoch (forward + backward + Adam, full graph): {steady:.2f} ms -- four aggregations of width {hidden} / {classes} per epoch")


if __name__ == "__main__":
    main()
