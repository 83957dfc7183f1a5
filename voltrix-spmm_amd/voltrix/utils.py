"""Accuracy metrics and GPU timing helpers.

Only the helpers on the hot path's test/tune surface are provided:
``calc_diff`` / ``relative_error`` (reference voltrix/utils.py:21-42), ``GPU_bench`` / ``CPU_bench``
(:324-364).  Timing uses HIP events on the stream the kernels are launched on (torch's current stream);
the reference's kineto table parsing (:232-321) is replaced by per-launch event pairs, with the same
optional 256 MB cache flush between launches (:277-281).
"""
from __future__ import annotations

import time

import torch


def relative_error(value: torch.Tensor, real: torch.Tensor, exclude_zeros: bool = True) -> float:
    value = value.double().flatten()
    real = real.double().flatten()
    if not exclude_zeros:
        return ((value - real).abs() / (real.abs() + 1e-9)).mean().item()
    mask = (real.abs() == 0) | real.isinf() | value.isinf()
    return ((value[~mask] - real[~mask]).abs() / real[~mask].abs()).mean().item()


def calc_diff(x: torch.Tensor, y: torch.Tensor, dtype=torch.float):
    """1 - 2<x,y> / (<x,x> + <y,y>): 0 for identical tensors (the reference's "difference rate")."""
    x, y = x.to(dtype), y.to(dtype)
    return 1 - 2 * (x * y).sum() / (x * x + y * y).sum()


def _flush_cache():
    torch.empty(int(256e6 // 4), dtype=torch.int, device="cuda").zero_()


def GPU_bench(func, iters: int = 100, warmup: int = 30, kernel_name=None, flush_l2=None) -> float:
    """Milliseconds per call of ``func``.

    ``kernel_name is None``: one event pair around ``iters`` back-to-back calls (reference :331-337).
    ``kernel_name`` given: the reference isolates that kernel with kineto and flushes L2 before every call
    (:339-349); here every call is bracketed by its own event pair after a flush, and the mean is returned.
    """
    if flush_l2 is None:
        flush_l2 = kernel_name is not None
    for _ in range(warmup):
        func()
    if not flush_l2:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            func()
        end.record()
        end.synchronize()
        return start.elapsed_time(end) / iters
    total = 0.0
    for _ in range(iters):
        _flush_cache()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        func()
        end.record()
        end.synchronize()
        total += start.elapsed_time(end)
    return total / iters


def CPU_bench(func, iters: int = 100, warmup: int = 30) -> float:
    for _ in range(warmup):
        func()
    t0 = time.perf_counter()
    for _ in range(iters):
        func()
    return (time.perf_counter() - t0) * 1000 / iters
