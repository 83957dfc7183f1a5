"""``hmat_gen_kernel``: dense 0/1 tiles + column map on the GPU (reference voltrix/jit_kernels/hmat_gem.py:13-73)."""
import torch

from .tuner import jit_tuner

includes = ('"voltrix/bmat_kernels.hpp"',)
template = """
__return_code = voltrix::hmat_hip(node_pointer, edge_list, block_partition, edge_to_column, edge_to_row, pointer1, num_row_windows, num_nodes, num_edges, hspa, hind, nullptr);
"""

arg_defs = (
    ("node_pointer", torch.int),
    ("edge_list", torch.int),
    ("block_partition", torch.int),
    ("edge_to_column", torch.int),
    ("edge_to_row", torch.int),
    ("pointer1", torch.int),
    ("num_row_windows", int),
    ("num_nodes", int),
    ("num_edges", int),
    ("hspa", torch.float),
    ("hind", torch.int),
)


def hmat_gen_kernel(node_pointer, edge_list, block_partition, edge_to_column, edge_to_row, pointer1, hspa, hind):
    for t in (node_pointer, edge_list, block_partition, edge_to_column, edge_to_row, pointer1, hind):
        assert t.is_cuda and t.dtype == torch.int32
    assert hspa.is_cuda and hspa.dtype == torch.float
    num_row_windows = block_partition.shape[0]
    num_nodes = node_pointer.shape[0] - 1
    num_edges = edge_list.shape[0]

    args = (node_pointer, edge_list, block_partition, edge_to_column, edge_to_row, pointer1, num_row_windows, num_nodes,
            num_edges, hspa, hind)
    runtime = jit_tuner.compile_and_tune(name="hmat_gen_kernel", keys={}, space=tuple(), includes=includes,
                                         arg_defs=arg_defs, template=template, args=args)
    rc = runtime(*args)
    assert rc == 0, f"hmat_gen_kernel failed with return code {rc}"
