"""``preprocess_kernel``: host row-window condensing (reference voltrix/jit_kernels/preprocess.py:23-70)."""
import torch

from .tuner import jit_tuner

includes = ('"voltrix/bmat_kernels.hpp"',)
template = """
__return_code = voltrix::preprocess(
    edge_list, node_pointer, num_nodes, VOLTRIX_BLK_H, VOLTRIX_BLK_W,
    block_partition, edge_to_column, edge_to_row, pointer1);
"""

arg_defs = (
    ("edge_list", torch.int),
    ("node_pointer", torch.int),
    ("num_nodes", int),
    ("block_partition", torch.int),
    ("edge_to_column", torch.int),
    ("edge_to_row", torch.int),
    ("pointer1", torch.int),
)


def preprocess_kernel(edge_list, node_pointer, block_partition, edge_to_column, edge_to_row, pointer1):
    for t in (edge_list, node_pointer, block_partition, edge_to_column, edge_to_row, pointer1):
        assert t.is_cpu and t.dtype == torch.int32 and t.is_contiguous()
    num_nodes = node_pointer.shape[0] - 1
    assert block_partition.numel() == (num_nodes + 15) // 16 and pointer1.numel() == block_partition.numel() + 1
    assert edge_to_column.numel() == edge_list.numel() == edge_to_row.numel()

    args = (edge_list, node_pointer, num_nodes, block_partition, edge_to_column, edge_to_row, pointer1)
    runtime = jit_tuner.compile_and_tune(name="preprocess_kernel", keys={}, space=tuple(), includes=includes,
                                         arg_defs=arg_defs, template=template, args=args)
    rc = runtime(*args)
    assert rc == 0, f"preprocess_kernel failed with return code {rc}"
