"""Fused GPU preprocess (no reference counterpart): CSR on the device ->
``(pointer1, hspa_packed, hind)`` bit-identical to preprocess + hmat_gen + hmat_packed_swizzle.
Goes through the ahead-of-time C-ABI (include/voltrix_capi.h: voltrix_launch_csr_window_count / _csr_fill)."""
import torch

from .. import capi


def csr_fused_preprocess_kernel(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                                path: str = None):
    """``num_cols``: the column universe (every id in ``indices`` is in ``[0, num_cols)``); default ``num_nodes`` (square
    adjacency, the reference's contract).  It only selects the rank algorithm (LDS bitmap vs per-window sort); ids
    outside it are detected on the device and the preprocess is redone with the universe-free sort path.  ``path``: None /
    "auto" = the library's rule, "sort" | "bitmap" | "mixed" force a rank algorithm where it applies (same bytes on every
    path; the operator passes what ``VOLTRIX_PREPROCESS=fused:<path>`` says)."""
    assert indptr.is_cuda and indptr.dtype == torch.int32 and indptr.is_contiguous()
    assert indices.is_cuda and indices.dtype == torch.int32 and indices.is_contiguous()
    assert indptr.numel() == num_nodes + 1
    device = indptr.device
    num_edges = indices.numel()
    num_row_windows = (num_nodes + 15) // 16
    stream = torch.cuda.current_stream().cuda_stream
    num_cols = num_nodes if num_cols is None else int(num_cols)

    block_partition = torch.empty(num_row_windows, dtype=torch.int32, device=device)
    pointer1 = torch.empty(num_row_windows + 1, dtype=torch.int32, device=device)
    status = torch.empty(1, dtype=torch.int32, device=device)
    while True:
        workspace = torch.empty(capi.csr_preprocess_workspace_bytes(num_nodes, num_cols, num_edges, path), dtype=torch.uint8,
                                device=device)
        capi.launch_csr_window_count(indptr, indices, num_nodes, num_cols, workspace, block_partition, pointer1, status,
                                     stream, path)
        # host sync point, as in the reference (spmm.py:44): T and the out-of-universe count in one copy
        total_blocks, outside = torch.cat([pointer1[-1:], status]).tolist()
        if outside == 0:
            break
        if num_cols <= 0:
            raise ValueError(f"csr_preprocess: {outside} column ids outside [0, 2^28)")
        num_cols = 0  # ids beyond the declared universe (e.g. a non-square operand): universe-free sort path
    hspa_packed = torch.empty(total_blocks * 4, dtype=torch.uint32, device=device)
    hind = torch.empty(total_blocks * 8, dtype=torch.int32, device=device)
    capi.launch_csr_fill(indptr, indices, num_nodes, num_cols, workspace, pointer1, hspa_packed, hind, stream, path)
    return pointer1, hspa_packed, hind, block_partition
