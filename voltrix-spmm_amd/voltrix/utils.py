"""Accuracy metrics and GPU timing helpers.

Only the helpers on the hot path's test/tune surface are provided:
``calc_diff`` / ``relative_error`` (reference voltrix/utils.py:21-42), ``GPU_bench`` / ``CPU_bench``
(:324-364) and ``bench_kineto`` (:232-321).  Timing uses HIP events on the stream the kernels are launched on.
The reference isolates ONE kernel of a multi-kernel call by parsing the kineto table for its name; this build of
torch has no GPU activity in its profiler, so the isolation is done at the two choke points every launch of this
package goes through -- ``jit.Runtime.__call__`` (JIT kernels) and the ``capi.launch_*`` wrappers (ahead-of-time
library): inside a ``KernelTimer`` each launch is bracketed by its own event pair on ITS stream and booked under its
kernel name, so the window kernel, the panel kernel on its side stream, the cast and the combine pass of one
``voltrix.spmm`` call are timed separately.  Same optional 256 MB cache flush between calls (:277-281).
"""
from __future__ import annotations

import collections
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
    # 512 MiB: more than L2 (32 MiB) + the 256 MiB Infinity Cache, the same amount bench.py's cold-cache timing writes (the
    # reference writes 256 MB for a 50 MB L2, utils.py:277-281; 256e6 bytes here left part of the Infinity Cache warm)
    torch.empty(512 << 18, dtype=torch.int, device="cuda").zero_()


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


class KernelTimer:
    """Context manager: while active, every launch that goes through ``jit.Runtime`` or ``capi.launch_*`` records a
    HIP event pair around itself on the stream it is launched on.  ``summary()`` -> {kernel name: (calls, mean ms)}.
    Names: the JIT kernel name (``spmm_kernel``, ...) or the C-ABI symbol without its prefix (``spmm_panel``, ...)."""

    active = None

    def __init__(self):
        self.records = collections.defaultdict(list)

    def __enter__(self):
        self._outer = KernelTimer.active
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = self._outer
        return False

    class _Bracket:
        def __init__(self, timer, name, stream):
            self.timer, self.name = timer, name
            if isinstance(stream, torch.cuda.Stream):
                self.stream = stream
            elif stream:
                self.stream = torch.cuda.ExternalStream(int(stream))
            else:
                self.stream = torch.cuda.current_stream()

        def __enter__(self):
            self.start, self.end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.start.record(self.stream)
            return self

        def __exit__(self, *exc):
            self.end.record(self.stream)
            self.timer.records[self.name].append((self.start, self.end))
            return False

    def bracket(self, name, stream):
        return KernelTimer._Bracket(self, name, stream)

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(s.elapsed_time(e) for s, e in v) / len(v)) for k, v in self.records.items()}


def timed_launch(name, stream):
    """``with timed_launch(name, stream):`` around a launch -- a no-op unless a ``KernelTimer`` is active."""
    timer = KernelTimer.active
    return timer.bracket(name, stream) if timer is not None else _NULL


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _Null()


def bench_kineto(fn, kernel_names, num_tests: int = 30, suppress_kineto_output: bool = False, trace_path=None,
                 barrier_comm_profiling: bool = False, flush_l2: bool = False):
    """Average seconds per call of the launches named by ``kernel_names`` (a string or a tuple of strings; a name matches
    when it is contained in the launch's name) inside ``fn`` -- the reference's contract (utils.py:232-321): seconds,
    one value per name, exactly one launch name may match each.  ``flush_l2`` (default False, as in the reference's bench_kineto, utils.py:232) writes 512 MiB before every call."""
    assert trace_path is None and not barrier_comm_profiling, "not supported in this build"
    names = (kernel_names,) if isinstance(kernel_names, str) else tuple(kernel_names)
    fn()  # warm-up (JIT, tuner)
    with KernelTimer() as timer:
        for _ in range(num_tests):
            if flush_l2:
                _flush_cache()
            fn()
    got = timer.summary()
    out = []
    for name in names:
        hits = [k for k in got if name in k]
        assert len(hits) == 1, f"kernel name `{name}` matches {hits or 'no launch'} (launches seen: {sorted(got)})"
        out.append(got[hits[0]][1] / 1e3)
    return out[0] if isinstance(kernel_names, str) else tuple(out)
