"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's hot path.

This module is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``voltrix-spmm_amd/voltrix``) never imports anything from ``oracle/``.

Every function restates one reference function and cites the lines it follows
(paths relative to the reference checkout):

=====================  ==========================================================
``preprocess``         voltrix/include/voltrix/bmat_kernels.cuh:264-320 (+248-262)
``hmat_gen``           voltrix/include/voltrix/bmat_kernels.cuh:21-111
``hmat_packed_swizzle``  voltrix/include/voltrix/bmat_kernels.cuh:151-193
``spmm_blocked``       voltrix/include/voltrix/spmm_kernels.cuh:1632-1716
``calc_diff``          voltrix/utils.py:38-42
``relative_error``     voltrix/utils.py:21-35
=====================  ==========================================================

Pinning (see DESIGN.md "Oracle"): the reference's native code cannot be built in
this image (nvcc / CUDA runtime headers / PTX), so the restatement is pinned by
(i) the reference's own test criterion -- ``calc_diff`` against
``torch.sparse_csr_tensor(indptr, indices, ones) @ feat`` on the inputs of
tests/test_spmm.py and tests/test_spmm_kernel.py, (ii) the known answers the
survey recorded from the reference's ``preprocess``,
(iii) golden vectors produced by importing the reference's pure-Python modules
(tests/golden/make_goldens.py) and (iv) an independent unpack -> CSR round trip.
"""
from __future__ import annotations

import numpy as np

BLK_H = 16  # voltrix/spmm/spmm.py:12, traits.h:6
BLK_W = 8   # voltrix/spmm/spmm.py:13, traits.h:7


# --------------------------------------------------------------------------- a2
def preprocess(indptr, indices, num_nodes, blk_h=BLK_H, blk_w=BLK_W):
    """CPU row-window condensing -- bmat_kernels.cuh:264-320.

    Returns ``(block_partition[W], edge_to_column[E], edge_to_row[E], pointer1[W+1])``,
    all int32.  Quirk kept: a window without edges still owns one (all-zero) TC
    block because ``inplace_deduplication`` unconditionally inserts ``array[0]``
    (bmat_kernels.cuh:252) so the map has size 1 (-> ``:298-299``).
    """
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    num_edges = int(indices.shape[0])
    num_windows = (num_nodes + blk_h - 1) // blk_h
    edge_to_row = np.zeros(num_edges, dtype=np.int32)
    edge_to_column = np.zeros(num_edges, dtype=np.int32)
    block_partition = np.zeros(num_windows, dtype=np.int32)

    # :273-276  edgeToRow[eid] = nid
    deg = np.diff(indptr[: num_nodes + 1])
    edge_to_row[:] = np.repeat(np.arange(num_nodes, dtype=np.int32), deg)

    for w in range(num_windows):
        lo = int(indptr[w * blk_h])
        hi = int(indptr[min((w + 1) * blk_h, num_nodes)])
        if hi == lo:
            block_partition[w] = 1  # quirk, see docstring
            continue
        # :288-293 sort + de-duplicate the window's neighbour ids (as unsigned)
        uniq = np.unique(indices[lo:hi].astype(np.uint32))
        block_partition[w] = (uniq.size + blk_w - 1) // blk_w  # :298-299
        # :304-307 edge -> rank of its column among the window's distinct columns
        edge_to_column[lo:hi] = np.searchsorted(uniq, indices[lo:hi].astype(np.uint32))

    pointer1 = np.zeros(num_windows + 1, dtype=np.int32)  # :312-319
    pointer1[1:] = np.cumsum(block_partition, dtype=np.int64).astype(np.int32)
    return block_partition, edge_to_column, edge_to_row, pointer1


# --------------------------------------------------------------------------- a3
def hmat_gen(indptr, indices, block_partition, edge_to_column, edge_to_row, pointer1,
             num_nodes, blk_h=BLK_H, blk_w=BLK_W):
    """Dense 0/1 fp32 tiles + per-tile column map -- bmat_kernels.cuh:21-111.

    ``hspa`` f32 [T*128] row-major ``[r*8+c]`` (``:100-103``); ``hind`` i32 [T*8],
    unused slots 0 (``:71-73``).
    """
    indices = np.asarray(indices)
    total = int(pointer1[-1])
    hspa = np.zeros(total * blk_h * blk_w, dtype=np.float32)
    hind = np.zeros(total * blk_w, dtype=np.int32)
    num_windows = block_partition.shape[0]
    for w in range(num_windows):
        lo = int(indptr[w * blk_h])
        hi = int(indptr[min((w + 1) * blk_h, num_nodes)])
        if hi == lo:
            continue
        e = np.arange(lo, hi)
        col = edge_to_column[e].astype(np.int64)
        blk = int(pointer1[w]) + col // blk_w                       # :94-96
        row_local = edge_to_row[e].astype(np.int64) % blk_h          # :98
        col_local = col % blk_w                                      # :99
        hspa[blk * (blk_h * blk_w) + row_local * blk_w + col_local] = 1.0   # :100-102
        hind[blk * blk_w + col_local] = indices[e]                   # :103-105
    return hspa, hind


# --------------------------------------------------------------------------- a4
def hmat_packed_swizzle(pointer1, hspa, blk_h=BLK_H, blk_w=BLK_W):
    """128 floats -> 4 x uint32 in mma.m16n8k8 A-fragment order -- bmat_kernels.cuh:151-193.

    word ``t`` bit ``b`` <= ``hspa[(b>>2) + 8*(t&1)][(b&3) + 4*(t>>1)] != 0`` (``:180-188``).
    """
    total = int(pointer1[-1])
    tiles = np.asarray(hspa, dtype=np.float32).reshape(total, blk_h, blk_w)
    nz = np.abs(tiles) > 1e-5                                        # :186
    packed = np.zeros((total, 4), dtype=np.uint32)
    for t in range(4):
        for bit in range(32):
            row = (bit >> 2) + 8 * (t % 2)                           # :180-183
            col = (bit % 4) + 4 * (t // 2)
            packed[:, t] |= nz[:, row, col].astype(np.uint32) << np.uint32(bit)
    return packed.reshape(-1)


def unpack_swizzled(hspa_packed, blk_h=BLK_H, blk_w=BLK_W):
    """Inverse of :func:`hmat_packed_swizzle` -> bool [T,16,8] (independent decode
    written from the consumer side, spmm_kernels.cuh:1632-1644: lane ``l`` tests
    ``word_t & (1<<l)`` and owns A[row=l>>2 (+8 for odd t)][col=l&3 (+4 for t>=2)])."""
    words = np.asarray(hspa_packed, dtype=np.uint32).reshape(-1, 4)
    total = words.shape[0]
    tiles = np.zeros((total, blk_h, blk_w), dtype=bool)
    for lane in range(32):
        for t in range(4):
            r = (lane >> 2) + (8 if (t & 1) else 0)
            c = (lane & 3) + (4 if t >= 2 else 0)
            tiles[:, r, c] = (words[:, t] >> np.uint32(lane)) & np.uint32(1)
    return tiles


def blocked_to_csr(pointer1, hspa_packed, hind, num_nodes, blk_h=BLK_H, blk_w=BLK_W):
    """Round trip: (pointer1, hspa_packed, hind) -> de-duplicated, column-sorted CSR."""
    tiles = unpack_swizzled(hspa_packed, blk_h, blk_w)
    hind = np.asarray(hind, dtype=np.int64).reshape(-1, blk_w)
    rows_out = [[] for _ in range(num_nodes)]
    num_windows = len(pointer1) - 1
    for w in range(num_windows):
        for b in range(int(pointer1[w]), int(pointer1[w + 1])):
            rr, cc = np.nonzero(tiles[b])
            for r, c in zip(rr, cc):
                row = w * blk_h + int(r)
                assert row < num_nodes, "bit set in a row beyond num_nodes"
                rows_out[row].append(int(hind[b, c]))
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    out = []
    for i, r in enumerate(rows_out):
        r.sort()
        out.extend(r)
        indptr[i + 1] = len(out)
    return indptr, np.asarray(out, dtype=np.int64)


# ------------------------------------------------------------------- a8 numerics
def round_tf32_rna(x):
    """``cvt.rna.tf32.f32`` (spmm_kernels.cuh:1642,1671): round-to-nearest, ties away
    from zero, to a 10-bit mantissa, result kept in an fp32 container."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).copy()
    finite = (u & np.uint32(0x7F800000)) != np.uint32(0x7F800000)
    u[finite] = (u[finite] + np.uint32(0x1000)) & np.uint32(0xFFFFE000)
    return u.view(np.float32)


def round_operand(x, mode):
    x = np.asarray(x, dtype=np.float32)
    if mode in (None, "none", "fp32"):
        return x
    if mode == "tf32":
        return round_tf32_rna(x)
    if mode == "fp16":
        return x.astype(np.float16).astype(np.float32)
    if mode == "fp16-scaled":  # voltrix.spmm's fp32 operand: one power-of-two scale per tensor (voltrix_capi.h)
        amax = float(np.abs(x).max()) if x.size else 0.0
        e = 0 if (amax == 0.0 or not np.isfinite(amax)) else max(int(np.frexp(np.float32(amax))[1]) - 1 - 14, -100)
        with np.errstate(over="ignore", under="ignore"):
            return np.ldexp(np.ldexp(x, -e).astype(np.float16).astype(np.float32), e).astype(np.float32)
    if mode == "bf16":
        import torch
        return torch.from_numpy(x.copy()).to(torch.bfloat16).to(torch.float32).numpy()
    raise ValueError(mode)


def spmm_blocked(pointer1, hspa_packed, hind, num_nodes, feat, rounding="tf32",
                 blk_h=BLK_H, blk_w=BLK_W):
    """Reference SpMM semantics on the block format -- spmm_kernels.cuh:1632-1716:

    ``out[16w+r, f] = sum_{TCb i of w} sum_{c<8} bit(i,r,c) * round(feat[hind[8b+c], f])``
    with fp32 accumulation in TC-block order.  ``rounding`` is ``"tf32"`` for the
    reference's ``cvt.rna`` (``:1642,1671``), ``"fp16"`` for the gfx950 build.
    Rows >= ``num_nodes`` of the last window are dropped (the reference leaves the
    N%16 tail uncomputed, ``:1514``; the oracle -- like torch.sparse.mm -- computes it).
    """
    feat_r = round_operand(feat, rounding)
    num_feats = feat_r.shape[1]
    tiles = unpack_swizzled(hspa_packed, blk_h, blk_w).astype(np.float32)
    hind = np.asarray(hind, dtype=np.int64).reshape(-1, blk_w)
    num_windows = len(pointer1) - 1
    out = np.zeros((num_windows * blk_h, num_feats), dtype=np.float32)
    for w in range(num_windows):
        acc = np.zeros((blk_h, num_feats), dtype=np.float32)
        for b in range(int(pointer1[w]), int(pointer1[w + 1])):
            acc += tiles[b] @ feat_r[hind[b]]  # fp32, one TC block (k=8) at a time
        out[w * blk_h:(w + 1) * blk_h] = acc
    return out[:num_nodes]


def spmm_csr(indptr, indices, feat, num_nodes, dtype=np.float64):
    """Plain ``csr(ones) @ feat`` (duplicates summed, like torch.sparse.mm) in ``dtype``."""
    import scipy.sparse as sp
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    a = sp.csr_matrix((np.ones(indices.shape[0], dtype=dtype), indices, indptr),
                      shape=(num_nodes, int(feat.shape[0])))
    return np.asarray(a @ np.asarray(feat, dtype=dtype))


# ----------------------------------------------------------------------- metrics
def calc_diff(x, y):
    """``1 - 2<x,y>/(<x,x>+<y,y>)`` -- voltrix/utils.py:38-42 (fp32 there; fp64 here
    when the inputs are fp64)."""
    x = np.asarray(x)
    y = np.asarray(y)
    den = (x * x + y * y).sum()
    return 1 - 2 * (x * y).sum() / den


def relative_error(value, real):
    """Mean |value-real|/|real| over entries with real != 0 -- voltrix/utils.py:21-35."""
    value = np.asarray(value, dtype=np.float64).ravel()
    real = np.asarray(real, dtype=np.float64).ravel()
    mask = (np.abs(real) == 0) | np.isinf(real) | np.isinf(value)
    return float((np.abs(value - real)[~mask] / np.abs(real[~mask])).mean())


def forward_error_bound(indptr, indices, feat, num_nodes, rounding="fp16"):
    """Element-wise bound of SURVEY.md section 8c (iii):
    ``|out-ref|_ij <= (u + deg_i * 2^-24) * (A |B|)_ij`` with u the unit round-off of
    the operand rounding (2^-11 for fp16 and tf32-rna)."""
    u = {"fp16": 2.0 ** -11, "tf32": 2.0 ** -11, "bf16": 2.0 ** -8, "none": 0.0}[rounding]
    deg = np.diff(np.asarray(indptr, dtype=np.int64))[:num_nodes].astype(np.float64)
    aabs = spmm_csr(indptr, indices, np.abs(np.asarray(feat, dtype=np.float64)), num_nodes)
    return (u + deg[:, None] * 2.0 ** -24) * aabs + 1e-30


# --------------------------------------------------------------------------- two-level format (no reference counterpart)
# The panel plan is this build's own extension (voltrix/hybrid.py, spmm_panel_kernels.hpp); these functions state its
# DEFINITION with plain loops, so that the device builder is checked bit for bit and the panel kernel is checked
# against an interpreter written from the consumer side of the layout.  The numerical oracle stays the reference's:
# csr(ones) @ feat on the FULL matrix (torch_ref.spmm).
def panel_split(indptr, indices, num_nodes, panel_rows, tau):
    """-> ``(resid_indptr int32, resid_indices int32, shared)`` with ``shared[p]`` = sorted list of
    ``(col, [rows in panel, ascending])`` for the columns referenced by >= tau distinct rows of panel p.
    Duplicate (row, col) entries count once (reference quirk 5)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    num_panels = (num_nodes + panel_rows - 1) // panel_rows
    shared = []
    r_rows = [[] for _ in range(num_nodes)]
    for p in range(num_panels):
        r0, r1 = p * panel_rows, min((p + 1) * panel_rows, num_nodes)
        users = {}
        for r in range(r0, r1):
            for c in sorted(set(indices[indptr[r]:indptr[r + 1]].tolist())):
                users.setdefault(c, []).append(r - r0)
        sh = []
        for c in sorted(users):
            if len(users[c]) >= tau:
                sh.append((c, users[c]))
            else:
                for rp in users[c]:
                    r_rows[r0 + rp].append(c)
        shared.append(sh)
    resid_indptr = np.zeros(num_nodes + 1, dtype=np.int32)
    resid_indices = []
    for r in range(num_nodes):
        cs = sorted(r_rows[r])
        resid_indices.extend(cs)
        resid_indptr[r + 1] = len(resid_indices)
    return resid_indptr, np.asarray(resid_indices, dtype=np.int32), shared


def panel_plan(indptr, indices, num_nodes, waves, row_blocks, tau):
    """-> ``(resid_indptr, resid_indices, panel_ptr int32 [NP+1], panel_cols int32 [32 (S+2)], panel_bits uint32
    [(S+1) waves 64])`` -- the layout documented in spmm_panel_kernels.hpp / include/voltrix_capi.h."""
    panel_rows = waves * row_blocks * 16
    resid_indptr, resid_indices, shared = panel_split(indptr, indices, num_nodes, panel_rows, tau)
    panel_ptr = np.zeros(len(shared) + 1, dtype=np.int32)
    for p, sh in enumerate(shared):
        panel_ptr[p + 1] = panel_ptr[p] + (len(sh) + 31) // 32
    s_total = int(panel_ptr[-1])
    panel_cols = np.zeros(32 * (s_total + 2), dtype=np.int32)
    panel_bits = np.zeros((s_total + 1) * waves * 64, dtype=np.uint32)
    for p, sh in enumerate(shared):
        base = int(panel_ptr[p])
        if sh:
            panel_cols[32 * base:32 * int(panel_ptr[p + 1])] = sh[0][0]  # unused slots repeat the first shared column
        for rank, (c, rows) in enumerate(sh):
            ks, k = base + rank // 32, rank % 32
            panel_cols[32 * ks + k] = c
            for rp in rows:
                v, j, r16 = rp // (16 * row_blocks), (rp % (16 * row_blocks)) // 16, rp % 16
                panel_bits[(ks * waves + v) * 64 + (k // 8) * 16 + r16] |= np.uint32(1 << (16 * (k % 2) + 4 * j + (k % 8) // 2))
    return resid_indptr, resid_indices, panel_ptr, panel_cols, panel_bits


def panel_to_edges(panel_ptr, panel_cols, panel_bits, num_nodes, waves, row_blocks):
    """Consumer-side interpreter of the plan (what spmm_panel_kernel multiplies): sorted list of (row, col)."""
    panel_rows = waves * row_blocks * 16
    edges = []
    for p in range(len(panel_ptr) - 1):
        for ks in range(int(panel_ptr[p]), int(panel_ptr[p + 1])):
            for v in range(waves):
                for lane in range(64):
                    word = int(panel_bits[(ks * waves + v) * 64 + lane])
                    g, r16 = lane >> 4, lane & 15
                    for j in range(row_blocks):
                        for c in range(8):
                            if (word >> (16 * (c & 1) + 4 * j + (c >> 1))) & 1:
                                row = p * panel_rows + 16 * (row_blocks * v + j) + r16
                                assert row < num_nodes
                                edges.append((row, int(panel_cols[32 * ks + 8 * g + c])))
    return sorted(edges)


# ------------------------------------------------------- residual stage records of the one-launch two-level format
def fused_records(pointer1, hspa_packed, hind, num_nodes, waves=4, row_blocks=8):
    """Definition (plain loops) of the per-wave stage-record stream that ``spmm_fused_kernel`` consumes
    (spmm_fused_kernels.hpp; no reference counterpart): the block-format handle of the RESIDUAL matrix, re-packed.

    A *stage* is four consecutive TC blocks of one window (32 condensed columns); wave v of panel p owns windows
    ``row_blocks * (waves * p + v) + j``, j < row_blocks, and its records are the stages of those windows MERGED by their first
    column (ties: lower j first).  A window whose only block is all zero (the reference's empty-window quirk,
    bmat_kernels.cuh:252) has no stage.  Record = 64 uint32: words 0..31 the rows of B of the 32 columns (columns
    nobody references, and blocks past the window's end, repeat the window's first column: finite data, zero bits), words
    32..47 the 16 bitmap words (zero past the window's end), word 48 = j, the rest 0.
    -> ``(wave_ptr int32 [waves * NP + 1], records uint32 [R + 1, 64])`` (one zero record of padding)."""
    pointer1 = np.asarray(pointer1, dtype=np.int64)
    packed = np.asarray(hspa_packed, dtype=np.uint32)
    hind = np.asarray(hind, dtype=np.int64)
    num_windows = (num_nodes + BLK_H - 1) // BLK_H
    panel_rows = waves * row_blocks * BLK_H
    num_panels = (num_nodes + panel_rows - 1) // panel_rows
    wave_ptr = np.zeros(waves * num_panels + 1, dtype=np.int32)
    records = []
    for gw in range(waves * num_panels):
        stages = []
        for j in range(row_blocks):
            w = row_blocks * gw + j
            if w >= num_windows:
                continue
            kb0, kb1 = int(pointer1[w]), int(pointer1[w + 1])
            if kb1 - kb0 == 1 and not packed[4 * kb0:4 * kb0 + 4].any():
                continue
            for sb in range(kb0, kb1, 4):
                stages.append((int(hind[8 * sb]), j, sb, kb0, kb1))
        stages.sort(key=lambda s: (s[0], s[1]))
        for _, j, sb, kb0, kb1 in stages:
            rec = np.zeros(64, dtype=np.uint32)
            safe = int(hind[8 * kb0])
            for k in range(32):
                blk, c = sb + k // 8, k % 8
                used = False
                if blk < kb1:
                    mask = np.uint32((0x11111111 << (c & 3)) & 0xFFFFFFFF)
                    used = bool((packed[4 * blk + 2 * (c >> 2)] | packed[4 * blk + 2 * (c >> 2) + 1]) & mask)
                rec[k] = int(hind[8 * blk + c]) if used else safe
            for t in range(16):
                blk = sb + t // 4
                rec[32 + t] = packed[4 * blk + t % 4] if blk < kb1 else 0
            rec[48] = j
            records.append(rec)
        wave_ptr[gw + 1] = len(records)
    records.append(np.zeros(64, dtype=np.uint32))
    return wave_ptr, np.stack(records)


def fused_records_to_edges(wave_ptr, records, num_nodes, waves=4, row_blocks=8):
    """Consumer-side interpreter of the record stream (what one stage of spmm_fused_kernel multiplies: lane (g, R) of the
    MFMA A operand holds row R of TC block g -- nibble R & 7 of bitmap words 4 g + (R >> 3) (columns 0-3) and 4 g + 2 + (R >> 3)
    (columns 4-7) -- against the rows of B in words 8 g .. 8 g + 7): sorted (row, col)."""
    edges = []
    for gw in range(len(wave_ptr) - 1):
        for r in range(int(wave_ptr[gw]), int(wave_ptr[gw + 1])):
            rec = records[r]
            j = int(rec[48]) & (row_blocks - 1)
            for lane in range(64):
                g, row16 = lane >> 4, lane & 15
                for hi in range(2):
                    word = int(rec[32 + 4 * g + 2 * hi + (row16 >> 3)])
                    nib = (word >> (4 * (row16 & 7))) & 0xF
                    for i in range(4):
                        if (nib >> i) & 1:
                            row = BLK_H * (row_blocks * gw + j) + row16
                            assert row < num_nodes
                            edges.append((row, int(rec[8 * g + 4 * hi + i])))
    return sorted(edges)


# ---------------------------------------------------------------------------------------------------------------------
# Cuthill-McKee row order.  NOT a restatement of the reference (it has no reorder code: it reads externally reordered
# <name>.reorder.npz files, bench/graph_gen.py:42-45, bench/bench_all.py:120-129) but of this repository's own
# specification, voltrix-spmm_amd/voltrix/include/voltrix/reorder_kernels.hpp -- plain loops, the checker of the HIP search
# and of the torch host form (tests/test_reorder.py, tests/test_gpu_reorder_search.py).
def cm_order(indptr, indices, num_nodes, num_cols=None, max_components=64):
    n = int(num_nodes)
    m = n if num_cols is None else int(num_cols)
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    nbrs = [[] for _ in range(n)]
    deg = np.zeros(n, dtype=np.int64)
    for u in range(n):
        for c in indices[indptr[u]:indptr[u + 1]]:
            deg[u] += 1                       # an entry of row u of A
            if c < m and c < n:
                deg[c] += 1                   # ... is an entry of row c of A^T
            if c < n:
                nbrs[u].append(int(c))
                nbrs[int(c)].append(u)
    tie = np.empty(n, dtype=np.int64)
    tie[np.argsort(deg, kind="stable")] = np.arange(n)
    level = np.full(n, -1, dtype=np.int64)
    rank = np.full(n, -1, dtype=np.int64)

    def search(start, level, rank, base):
        level[start], rank[start] = 0, base
        order, frontier, d = [start], [start], 0
        while True:
            fresh = {}
            for u in frontier:
                for v in nbrs[u]:
                    if level[v] < 0 or (level[v] == d + 1 and v in fresh):
                        level[v] = d + 1
                        fresh[v] = min(fresh.get(v, 1 << 62), rank[u])
            if not fresh:
                return order, frontier
            d += 1
            frontier = sorted(fresh, key=lambda v: (fresh[v], tie[v]))
            for v in frontier:
                rank[v] = base + len(order)
                order.append(v)

    perm = []
    for _ in range(max_components):
        cand = [u for u in range(n) if level[u] < 0 and deg[u] > 0]
        if not cand:
            break
        start = min(cand, key=lambda u: tie[u])
        probe_level, probe_rank = level.copy(), rank.copy()
        _, last = search(start, probe_level, probe_rank, len(perm))
        start = min(last, key=lambda u: tie[u])
        order, _ = search(start, level, rank, len(perm))
        perm.extend(order)
    rest = [u for u in range(n) if rank[u] < 0]
    rest.sort(key=lambda u: -deg[u])          # stable: ties by id
    perm.extend(rest)
    return np.asarray(perm, dtype=np.int64)
