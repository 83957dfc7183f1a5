"""HIP source generation for JIT kernels.

Counterpart of the reference's voltrix/jit/template.py: the generated translation unit has the same
shape (``extern "C" void launch(<raw args>, int& __return_code)``, ``__raw_`` pointer arguments cast to
typed pointers, the body spliced in with 4-space indentation) so that the C-ABI stays the one
``Runtime`` marshals (runtime.py) -- only the include block and the device type names are HIP's.
tests/test_jit_layer.py pins this against golden output of the reference's ``generate``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Any, Dict, Iterable, Tuple

import torch

from ..project import DEBUG_FLAG, PROJECT_NAME_FULL

_TENSOR_TYPES = (torch.int32, torch.uint32, torch.float32, torch.float16, torch.bfloat16, torch.float8_e4m3fn,
                 torch.int64, torch.uint8)

# python/torch type -> name stored in kernel.args (reference template.py:12-20 uses these for `eval`;
# here the file is JSON and is parsed with `args_from_text`)
typename_map: Dict[Any, str] = {
    bool: "bool",
    int: "int",
    torch.int32: "torch.int",
    torch.uint32: "torch.uint32",
    torch.float32: "torch.float",
    torch.float16: "torch.float16",
    torch.bfloat16: "torch.bfloat16",
    torch.float8_e4m3fn: "torch.float8_e4m3fn",
    torch.int64: "torch.int64",
    torch.uint8: "torch.uint8",
    torch.cuda.Stream: "torch.cuda.Stream",
}
_type_by_name = {v: k for k, v in typename_map.items()}

ctype_map: Dict[Any, Any] = {bool: ctypes.c_bool, int: ctypes.c_int, torch.cuda.Stream: ctypes.c_void_p}
ctype_map.update({t: ctypes.c_void_p for t in _TENSOR_TYPES})

# (type in the launch() signature, type the body sees) -- reference template.py:39-50 with HIP device types
genc_map: Dict[Any, Tuple[str, str]] = {
    bool: ("bool", "bool"),
    int: ("int", "int"),
    torch.uint32: ("void*", "uint32_t*"),
    torch.int32: ("void*", "int*"),
    torch.int64: ("void*", "int64_t*"),
    torch.uint8: ("void*", "uint8_t*"),
    torch.float32: ("void*", "float*"),
    torch.float16: ("void*", "_Float16*"),
    torch.bfloat16: ("void*", "__hip_bfloat16*"),
    torch.float8_e4m3fn: ("void*", "__hip_fp8_e4m3*"),
    torch.cuda.Stream: ("void*", "hipStream_t"),
}

# headers pulled in only when a signature needs them (keeps the common compile fast)
_lazy_includes = {torch.bfloat16: "<hip/hip_bf16.h>", torch.float8_e4m3fn: "<hip/hip_fp8.h>"}


def map_ctype(value: Any) -> Any:
    """Python value -> ctypes argument (reference template.py:53-59)."""
    if isinstance(value, torch.Tensor):
        return ctype_map[value.dtype](value.data_ptr())
    if isinstance(value, torch.cuda.Stream):
        return ctypes.c_void_p(value.cuda_stream)
    return ctype_map[type(value)](value)


def cpp_format(template: str, keys: Dict[str, Any]) -> str:
    """Replace ``{key}`` occurrences only; C++ braces survive (reference template.py:62-67)."""
    out = template
    for key, value in keys.items():
        out = out.replace("{" + key + "}", str(value))
    return out


def args_to_text(arg_defs: Iterable[Tuple[str, Any]]) -> str:
    """kernel.args content: same textual form as the reference writes (compiler.py:156-160)."""
    return ", ".join(f"('{name}', {typename_map[t]})" for name, t in arg_defs)


def args_from_text(text: str):
    """Parse kernel.args without ``eval`` (the reference evals it, runtime.py:32)."""
    out = []
    text = text.strip()
    if not text:
        return out
    for item in text.strip("()").split("), ("):
        name, tname = item.split(", ", 1)
        out.append((name.strip("'\""), _type_by_name[tname.strip()]))
    return out


def generate(includes: Iterable[str], arg_defs: Iterable[Tuple], body: str) -> str:
    assert isinstance(includes, (list, tuple))
    arg_defs = tuple(arg_defs)
    sys_includes = {"<hip/hip_runtime.h>", "<cstdint>", "<iostream>"}
    sys_includes.update(_lazy_includes[t] for _, t in arg_defs if t in _lazy_includes)
    sys_includes.update(i for i in includes if i.startswith("<"))
    pkg_includes = {i for i in includes if i.startswith('"')}

    code = f"// {PROJECT_NAME_FULL} auto-generated JIT HIP source file\n\n"
    code += "\n".join(f"#include {i}" for i in sorted(sys_includes)) + "\n\n"
    code += "\n".join(f"#include {i}" for i in sorted(pkg_includes)) + "\n\n"

    raw = "__raw_"
    params = []
    for name, t in arg_defs:
        sig_t, body_t = genc_map[t]
        params.append(f"{sig_t} {raw if sig_t != body_t else ''}{name}")
    params.append("int& __return_code")
    code += 'extern "C" void launch(' + ", ".join(params) + ") {\n"
    code += "    // Cast raw types (if needed)\n"
    for name, t in arg_defs:
        sig_t, body_t = genc_map[t]
        if sig_t != body_t:
            code += f"    auto {name} = reinterpret_cast<{body_t}>({raw}{name});\n"
    code += "\n".join((("    " if line else "") + line) for line in body.split("\n"))
    code += "}\n\n"

    if os.getenv(DEBUG_FLAG, None):
        print(f"Generated code:\n{code}")
    return code
