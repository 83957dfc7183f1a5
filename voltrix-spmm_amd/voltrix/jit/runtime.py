"""ctypes runtime of one JIT-built kernel directory (kernel.hip, kernel.args, kernel.so).

Counterpart of the reference's voltrix/jit/runtime.py:9-72: same call contract -- positional args checked
against kernel.args (count, tensor dtype / python type), marshalled with ``map_ctype``, ``launch`` called
with a trailing ``byref(c_int)`` and the integer return code handed back (0 = success).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from .template import args_from_text, map_ctype

KERNEL_FILES = ("kernel.hip", "kernel.args", "kernel.so")


class Runtime:
    def __init__(self, path: str) -> None:
        self.path = path
        self.lib = None
        self.args = None
        self._launch = None
        assert self.is_path_valid(self.path), f"not a built kernel directory: {path}"

    @staticmethod
    def is_path_valid(path: str) -> bool:
        return os.path.isdir(path) and all(os.path.exists(os.path.join(path, f)) for f in KERNEL_FILES)

    def _load(self) -> None:
        self.lib = ctypes.CDLL(os.path.join(self.path, "kernel.so"))
        self._launch = self.lib.launch
        self._launch.restype = None
        with open(os.path.join(self.path, "kernel.args"), "r") as f:
            self.args = args_from_text(f.read())

    def __call__(self, *args) -> int:
        if self.lib is None:
            self._load()
        assert len(args) == len(self.args), f"Expected {len(self.args)} arguments, got {len(args)}"
        cargs = []
        for arg, (name, dtype) in zip(args, self.args):
            if isinstance(arg, torch.Tensor):
                assert arg.dtype == dtype, f"Expected tensor dtype `{dtype}` for `{name}`, got `{arg.dtype}`"
            else:
                assert isinstance(arg, dtype), f"Expected built-in type `{dtype}` for `{name}`, got `{type(arg)}`"
            cargs.append(map_ctype(arg))
        return_code = ctypes.c_int(-1)
        from ..utils import KernelTimer   # kernel-isolated timing hook (utils.bench_kineto)

        if KernelTimer.active is None:
            self._launch(*cargs, ctypes.byref(return_code))
        else:
            stream = next((a for a in args if isinstance(a, torch.cuda.Stream)), None)
            with KernelTimer.active.bracket(self.kernel_name, stream):
                self._launch(*cargs, ctypes.byref(return_code))
        return return_code.value

    def launcher(self):
        """The loaded ``launch`` entry point (ctypes function) and the parsed ``kernel.args``: for callers that marshal a fixed
        argument list once and patch only the pointers that change from call to call (jit_kernels/spmm.py::_LaunchPlan)."""
        if self.lib is None:
            self._load()
        return self._launch, self.args

    @property
    def kernel_name(self) -> str:
        """``spmm_kernel`` for .../kernel.spmm_kernel.<hash>."""
        base = os.path.basename(os.path.normpath(self.path))
        parts = base.split(".")
        return parts[1] if len(parts) >= 3 else base


class RuntimeCache:
    """path -> Runtime; falls back to the file system (reference runtime.py:55-72)."""

    def __init__(self) -> None:
        self.cache = {}

    def __getitem__(self, path: str) -> Optional[Runtime]:
        if path in self.cache:
            return self.cache[path]
        if Runtime.is_path_valid(path):
            self.cache[path] = Runtime(path)
            return self.cache[path]
        return None

    def __setitem__(self, path: str, runtime: Runtime) -> None:
        self.cache[path] = runtime
