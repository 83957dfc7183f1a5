"""``hmat_packed_swizzle_kernel``: 128 floats -> 4 swizzled bitmap words per TC block
(reference voltrix/jit_kernels/bmat_swizzle.py:14-48)."""
import torch

from .tuner import jit_tuner

includes = ('"voltrix/bmat_kernels.hpp"',)
template = """
__return_code = voltrix::hmat_packed_swizzle_hip(num_row_windows, pointer1, hspa, hspa_packed, nullptr);
"""

arg_defs = (
    ("num_row_windows", int),
    ("pointer1", torch.int),
    ("hspa", torch.float),
    ("hspa_packed", torch.uint32),
)


def hmat_packed_swizzle_kernel(block_partition, pointer1, hspa, hspa_packed):
    assert block_partition.is_cuda and block_partition.dtype == torch.int32
    assert pointer1.is_cuda and pointer1.dtype == torch.int32
    assert hspa.is_cuda and hspa.dtype == torch.float
    assert hspa_packed.is_cuda and hspa_packed.dtype == torch.uint32
    num_row_windows = block_partition.shape[0]

    args = (num_row_windows, pointer1, hspa, hspa_packed)
    runtime = jit_tuner.compile_and_tune(name="hmat_packed_swizzle_kernel", keys={}, space=tuple(), includes=includes,
                                         arg_defs=arg_defs, template=template, args=args)
    rc = runtime(*args)
    assert rc == 0, f"hmat_packed_swizzle_kernel failed with return code {rc}"
