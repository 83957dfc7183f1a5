"""Time-boxed parity fuzz of the operator (round 6): mid-size graphs of every family of synth_graphs at random scales, random
widths and dtypes, through every path a caller can reach -- window / stream / two-level / CSR row-gather kernel, default and
full tuning spaces (the sweep itself runs), weighted (separable and general values), relabelled reorder, backward -- each
result against ``torch.sparse.mm`` in float64 on the GPU with the element-wise bound the tests state:

    |C - ref| <= (u + (deg + 2) * 2^-23) (|A| |B|) + deg * 2^-25,      u = 2^-11 (16-bit operands; 2^-8 bfloat16), 0 exact fp32

The small-graph sweeps of tests/test_gpu_random.py pin one tile; this one lets the tuner choose among all of them on graphs big
enough for unit tables, cut windows, several XCD ranges and panel pieces.  Usage: fuzz_parity.py SECONDS [SEED] -> one line per
case, "FAIL" lines with everything needed to replay; exit code 1 on any failure."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402

FAMILIES = ["amazon0505_like", "dd_like", "ppi_like", "reddit_like", "amazon0601_like", "com_amazon_like", "ddi_like",
            "fraud_yelp_rsr_like", "web_berkstan_like", "protein_like", "yeast_like", "products_like", "cora_like"]
WIDTHS = [1, 3, 8, 16, 17, 24, 32, 48, 50, 64, 96, 100, 128, 136, 192, 256, 320, 512, 1024]


def reference(indptr, indices, values, feat64, n, m):
    v = torch.ones(indices.numel(), dtype=torch.float64, device=feat64.device) if values is None else values.double()
    a = torch.sparse_csr_tensor(indptr.long(), indices.long(), v, (n, m))
    return torch.sparse.mm(a, feat64), torch.sparse.mm(torch.sparse_csr_tensor(indptr.long(), indices.long(), v.abs(), (n, m)),
                                                       feat64.abs())


def check(out, ref, aabs, deg, u, what, out_round=0.0):
    """``out_round``: relative rounding of the RESULT itself (autograd hands the gradient back in the input's dtype)."""
    bound = (u * 1.0001 + (deg[:, None] + 2) * 2.0 ** -23) * aabs + deg[:, None] * 2.0 ** -25 + out_round * 1.0001 * ref.abs() + 1e-30
    err = (out.double() - ref).abs()
    bad = err > bound
    if bool(torch.isnan(out).any()) or bool(bad.any()):
        worst = int(torch.argmax((err / bound).flatten()))
        return f"{what}: {int(bad.sum())} elements outside the bound, worst ratio {float((err / bound).flatten()[worst]):.3f} at row {worst // out.shape[1]}"
    return None


def one_case(rng, case_no):
    fam = FAMILIES[int(rng.integers(0, len(FAMILIES)))]
    full_edges = {"reddit_like": 114e6, "products_like": 124e6, "protein_like": 26e6}.get(fam, 8e6)
    scale = float(min(1.0, rng.uniform(0.02, 1.0) * min(1.0, 6e6 / full_edges) * 4))
    num_feats = WIDTHS[int(rng.integers(0, len(WIDTHS)))]
    dtype = [torch.float16, torch.float16, torch.bfloat16, torch.float32][int(rng.integers(0, 4))]
    mode = ["plain", "plain", "full_space", "stream_space", "no_hybrid", "csr_on", "csr_off", "weighted_sep", "weighted_general",
            "reordered", "backward", "exact32", "hybrid_on", "hybrid_on", "update_values", "reordered_scaled"][int(rng.integers(0, 16))]
    if mode == "hybrid_on":     # the two-level side-car whenever enough edges sit in shared columns (auto wants >= 110 k rows of degree >= 64)
        fam = ["reddit_like", "protein_like", "fraud_yelp_rsr_like", "ddi_like", "products_like"][int(rng.integers(0, 5))]
        scale = float(rng.uniform(0.05, 0.6)) if fam in ("reddit_like", "protein_like") else (float(rng.uniform(0.02, 0.2)) if fam == "products_like" else 1.0)
        dtype = torch.float16 if dtype == torch.float32 else dtype
    if mode in ("weighted_sep", "update_values", "reordered_scaled") and dtype == torch.float32:
        dtype = torch.float16          # (weighted_general keeps fp32 features: the value plane after a scaled cast, or the CSR kernel with values)
    if mode == "exact32":
        dtype = torch.float32
    env = {"VOLTRIX_TUNE_SPACE": {"full_space": "full", "stream_space": "stream"}.get(mode, ""),
           "VOLTRIX_HYBRID": {"no_hybrid": "0", "hybrid_on": "1"}.get(mode, ""),
           "VOLTRIX_CSR_PATH": {"csr_on": "1", "csr_off": "0"}.get(mode, ""),
           "VOLTRIX_FP32_MODE": "exact" if mode == "exact32" else ""}
    for k, v in env.items():
        if v:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    desc = {"case": case_no, "family": fam, "scale": round(scale, 4), "F": num_feats, "dtype": str(dtype).replace("torch.", ""), "mode": mode}
    indptr, indices, _ = synth_graphs.generate(fam, device="cuda", scale=scale)
    n, e = indptr.numel() - 1, indices.numel()
    desc.update(N=n, nnz=e)
    if n == 0 or e == 0 or e * num_feats > 3e9:
        return desc, None, "skipped"
    gen = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    feat = torch.randn(n, num_feats, device="cuda", generator=gen).to(dtype)
    feat64 = feat.double()
    deg = (indptr[1:] - indptr[:-1]).double()
    # fp32 features: exact products where asked for (VOLTRIX_FP32_MODE=exact); elsewhere auto may run exact tiles, the CSR kernel or the
    # scaled 16-bit operand -- the 16-bit bound holds for all of them
    u = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8, torch.float32: 0.0 if mode == "exact32" else 2.0 ** -11}[dtype]
    tag = f"fuzz/{case_no}"
    if mode in ("weighted_sep", "weighted_general"):
        d = deg.clamp(min=1.0)
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (indptr[1:] - indptr[:-1]).long())
        if mode == "weighted_sep":
            values = (d[rows] * d[indices.long()]).rsqrt().float()
        else:
            values = (0.25 + torch.rand(e, device="cuda", generator=gen)).float()
        h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
        desc["separable"] = bool(h.separable)
        out = voltrix.spmm_weighted(h, feat, hash_tag=tag)
        ref, aabs = reference(indptr, indices, values, feat64, n, n)
        return desc, check(out, ref, aabs, deg, 2.0 * u + 2.0 ** -20, mode), "ran"      # values and operands both round to 16 bits
    if mode == "update_values":      # general values, then new ones on the same pattern through the edge -> plane map (twice: the map is cached)
        values = (0.25 + torch.rand(e, device="cuda", generator=gen)).float()
        h = voltrix.csr_preprocess_weighted(indptr, indices, values, n, separable=False)
        voltrix.spmm_weighted(h, feat, hash_tag=tag)
        msg = None
        for _ in range(2):
            values = (torch.randn(e, device="cuda", generator=gen)).float()
            voltrix.update_edge_values(h, values)
            out = voltrix.spmm_weighted(h, feat)
            ref, aabs = reference(indptr, indices, values, feat64, n, n)
            msg = msg or check(out, ref, aabs, deg, 2.0 * u + 2.0 ** -20, mode)
        return desc, msg, "ran"
    if mode == "reordered_scaled":   # shuffled labels, the library's reorder, and the normalised adjacency's factors on the same handle
        s_indptr, s_indices, _ = synth_graphs.shuffle_labels(indptr, indices, 77 + case_no)
        sdeg = (s_indptr[1:] - s_indptr[:-1]).double()
        r = sdeg.clamp(min=1.0).rsqrt().float()
        c = torch.bincount(s_indices.long(), minlength=n).double().clamp(min=1.0).rsqrt().float()
        h = voltrix.csr_preprocess_reordered(s_indptr, s_indices, n, method="auto", relabel=bool(case_no % 2), row_scale=r, col_scale=c)
        h.hspa_packed.hash_tag = tag
        out = voltrix.spmm_reordered(h, voltrix.permute_features(h, feat), unpermute=True)
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (s_indptr[1:] - s_indptr[:-1]).long())
        ref, aabs = reference(s_indptr, s_indices, r[rows] * c[s_indices.long()], feat64, n, n)
        return desc, check(out, ref, aabs, sdeg, 2.0 * u + 2.0 ** -20, mode), "ran"
    if mode == "reordered":
        s_indptr, s_indices, _ = synth_graphs.shuffle_labels(indptr, indices, 77 + case_no)
        h = voltrix.csr_preprocess_reordered(s_indptr, s_indices, n, method="auto", relabel=True)
        h.hspa_packed.hash_tag = tag
        desc["picked"] = getattr(h, "method", None) or getattr(h, "picked", None)
        out = voltrix.unpermute_output(h, voltrix.spmm_reordered(h, voltrix.permute_features(h, feat)))
        ref, aabs = reference(s_indptr, s_indices, None, feat64, n, n)
        sdeg = (s_indptr[1:] - s_indptr[:-1]).double()
        return desc, check(out, ref, aabs, sdeg, u, mode), "ran"
    if mode == "backward":
        from voltrix.autograd import SpMM

        op = SpMM(indptr, indices, n, hash_tag=tag)
        b = feat.float().requires_grad_(True)
        c = op(b.to(dtype) if dtype != torch.float32 else b)
        g = torch.randn(c.shape, device="cuda", generator=gen)
        c.backward(g)
        from voltrix import capi

        t_indptr, t_indices = capi.csr_transpose(indptr.contiguous(), indices.contiguous(), n, n)
        ref, aabs = reference(t_indptr, t_indices, None, g.double(), n, n)
        tdeg = (t_indptr[1:] - t_indptr[:-1]).double()
        out_round = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8, torch.float32: 0.0}[dtype]     # grad.to(input dtype)
        return desc, check(b.grad, ref, aabs, tdeg, 2.0 ** -11, mode, out_round), "ran"
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = tag
    desc["two_level"] = voltrix.two_level_of(handle[1]) is not None
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    again = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)          # the launch plan of the repeated call
    ref, aabs = reference(indptr, indices, None, feat64, n, n)
    msg = check(out, ref, aabs, deg, u, mode) or check(again, ref, aabs, deg, u, mode + " (repeated call)")
    return desc, msg, "ran"


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    ran = failed = 0
    case_no = 0
    while time.time() - t0 < seconds:
        case_no += 1
        state = rng.bit_generator.state
        try:
            desc, msg, status = one_case(rng, case_no)
        except Exception as exc:   # noqa: BLE001 -- a fuzz loop reports and goes on
            desc, msg, status = {"case": case_no}, f"exception {type(exc).__name__}: {str(exc)[:300]}", "ran"
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if status == "skipped":
            continue
        ran += 1
        if msg:
            failed += 1
            print("FAIL", json.dumps(desc), msg, "rng state", json.dumps(state["state"]), flush=True)
        else:
            print("ok  ", json.dumps(desc), flush=True)
    print(f"# {ran} cases in {time.time() - t0:.0f} s, {failed} failures (seed {seed})", flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
