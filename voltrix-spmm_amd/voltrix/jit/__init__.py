"""JIT layer: source generation -> hipcc -> on-disk cache -> ctypes Runtime.

Public names follow the reference's voltrix/jit/__init__.py:1-3; ``get_nvcc_compiler`` is kept as an
alias of ``get_hipcc_compiler`` so that reference-side callers (tests/test_jit.py:33) keep working.
"""
from .compiler import get_hipcc_compiler, get_nvcc_compiler, build, hash_to_hex
from .template import cpp_format, generate
from .runtime import Runtime
